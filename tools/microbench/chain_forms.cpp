// Host microbenchmark of the batch-weight chains (chain_host.h): lock-step bundles of 8 / 4 chains, the single low-latency
// chain, merlin.h's generic sponge, and the permutations underneath.  Host only, no GPU:
//   /opt/rocm/lib/llvm/bin/clang++ -O3 -std=c++17 -I bulletproofs-plus_amd/csrc tools/microbench/chain_forms.cpp -o /tmp/chain_forms && /tmp/chain_forms
// (the engine's host code is compiled by the same clang at -O3).
#include <chrono>
#include <cstdio>
#include <vector>

#include "chain_host.h"
using namespace bpp;

static double us(std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
  return std::chrono::duration<double, std::micro>(b - a).count();
}
__attribute__((target("avx512f,avx512vl,avx512bw,avx512dq"))) static double perm8(int iters) {
  typename VecOps<8>::vec a[25];
  for (int i = 0; i < 25; i++)
    for (int w = 0; w < 8; w++) a[i][w] = i * 8 + w;
  auto t0 = std::chrono::steady_clock::now();
  for (int k = 0; k < iters; k++) keccak_f1600_vec<8>(a);
  auto t1 = std::chrono::steady_clock::now();
  volatile uint64_t sink = a[0][0];
  (void)sink;
  return us(t0, t1) / iters;
}
template <class F>
static double perm1(F f, int iters) {
  uint64_t a[25];
  for (int i = 0; i < 25; i++) a[i] = 0x9e3779b97f4a7c15ULL * (i + 1);
  auto t0 = std::chrono::steady_clock::now();
  for (int k = 0; k < iters; k++) f(a);
  auto t1 = std::chrono::steady_clock::now();
  volatile uint64_t sink = a[0];
  (void)sink;
  return us(t0, t1) / iters;
}
int main() {
  const size_t n = 1024;
  std::vector<uint8_t> rng(8 * n * 32), out(8 * n * 32);
  for (size_t i = 0; i < rng.size(); i++) rng[i] = (uint8_t)(i * 131 + (i >> 8) * 17 + 7);
  const uint8_t *in[8];
  uint8_t *o[8];
  for (int k = 0; k < 8; k++) in[k] = rng.data() + k * n * 32, o[k] = out.data() + k * n * 32;
  const bool v512 = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl"), v256 = __builtin_cpu_supports("avx2");
  const bool bmi = __builtin_cpu_supports("bmi") && __builtin_cpu_supports("bmi2");
  for (int rep = 0; rep < 3; rep++) {
    double x8 = 0, x4 = 0;
    auto t0 = std::chrono::steady_clock::now();
    if (v512) {
      for (int k = 0; k < 20; k++) weights_chain_x8(in, n, o);
      x8 = us(t0, std::chrono::steady_clock::now()) / 20 / n;
    }
    t0 = std::chrono::steady_clock::now();
    if (v256) {
      for (int k = 0; k < 20; k++) weights_chain_x4(in, n, o);
      x4 = us(t0, std::chrono::steady_clock::now()) / 20 / n;
    }
    t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < 20; k++) weights_chain_single(in[0], n, o[0]);
    const double single = us(t0, std::chrono::steady_clock::now()) / 20 / n;
    t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < 20; k++) weights_chain_generic(in[0], n, o[0]);
    const double generic = us(t0, std::chrono::steady_clock::now()) / 20 / n;
    printf("us per proof and chain: x8 %.3f  x4 %.3f  single %.3f  generic (merlin.h) %.3f | permutations: 8-way vector %.3f, 64-bit %s %.3f, merlin.h %.3f\n",
           x8 / 8, x4 / 4, single, generic, v512 ? perm8(20000) : 0.0, bmi ? "bmi" : "plain",
           bmi ? perm1(keccak_f1600_host_bmi, 100000) : perm1(keccak_f1600_host_plain, 100000), perm1(keccak_f1600, 100000));
  }
  return 0;
}
