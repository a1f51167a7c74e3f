// Microbenchmark (gfx950): k_msm_accumulate itself (csrc/msm.h) on synthetic sorted lists, to split its distance from
// the arithmetic's issue bound (tools/microbench/fe_rates.hip) into causes.  Shape of the bench step: 64 groups, 23
// windows x 1024 buckets each, 16 514 points per group, 16 514 x 23 terms per group.
//   mode 0  bucket sizes as in the real thing (multinomial), random table indices, buckets in size order
//   mode 1  as 0 but every term points at the SAME table entry (no gather misses)
//   mode 2  every bucket exactly 16 terms (no divergence inside a wavefront), random indices
//   mode 3  16 terms everywhere and one table entry
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I bulletproofs-plus_amd/csrc -o tools/microbench/acc_probe tools/microbench/acc_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <random>
#include <vector>
#include "msm.h"
using namespace bpp;
__global__ void k_fill(niels *t, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  niels q;
  for (int k = 0; k < 10; k++) { q.yplusx.v[k] = (i * 2654435761u + k * 40503u) & 0x1ffffff; q.yminusx.v[k] = (i * 40503u + k * 2654435761u) & 0x1ffffff; q.xy2d.v[k] = (i + k * 977u) & 0x1ffffff; }
  t[i] = q;
}
// shader clock while another kernel runs: one wavefront naps and reads both counters (s_memrealtime ticks at 100 MHz)
__global__ void k_clock(uint64_t *out, int naps) {
  const uint64_t c0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < naps; i++) __builtin_amdgcn_s_sleep(127);
  out[0] = __builtin_amdgcn_s_memtime() - c0;
  out[1] = __builtin_amdgcn_s_memrealtime() - w0;
}
int main() {
  hipStream_t side;
  (void)hipStreamCreateWithFlags(&side, hipStreamNonBlocking);
  uint64_t *d_clk, h_clk[2];
  (void)hipMalloc(&d_clk, 16);
  const uint32_t G = 64, K = 23, NB = 1024, per_group = K * NB, pts = 16514, terms = pts;
  std::mt19937 rng(12345);
  for (int mode = 0; mode < 4; mode++) {
    const bool same = mode & 1, uniform = mode & 2;
    std::vector<uint32_t> counts((size_t)G * per_group), starts((size_t)G * per_group), order((size_t)G * per_group), sorted;
    sorted.reserve((size_t)G * K * terms);
    for (uint32_t g = 0; g < G; g++) {
      for (uint32_t k = 0; k < K; k++) {
        std::vector<uint32_t> cnt(NB, 0);
        if (uniform) { for (auto &c : cnt) c = 16; }
        else for (uint32_t i = 0; i < terms; i++) cnt[rng() % NB]++;
        for (uint32_t b = 0; b < NB; b++) {
          const size_t id = ((size_t)g * K + k) * NB + b;
          counts[id] = cnt[b];
          starts[id] = (uint32_t)sorted.size();
          for (uint32_t j = 0; j < cnt[b]; j++) sorted.push_back((same ? 0u : (g * pts + rng() % pts)) | ((rng() & 1u) << 31));
        }
      }
      // buckets of the group by descending size (ids are global bucket ids)
      std::vector<uint32_t> ids(per_group);
      for (uint32_t i = 0; i < per_group; i++) ids[i] = g * per_group + i;
      std::stable_sort(ids.begin(), ids.end(), [&](uint32_t a, uint32_t b) { return counts[a] > counts[b]; });
      for (uint32_t i = 0; i < per_group; i++) order[(size_t)g * per_group + i] = ids[i];
    }
    uint32_t *d_sorted, *d_starts, *d_counts, *d_order; niels *d_tab; ge *d_b;
    (void)hipMalloc(&d_sorted, sorted.size() * 4); (void)hipMalloc(&d_starts, starts.size() * 4); (void)hipMalloc(&d_counts, counts.size() * 4);
    (void)hipMalloc(&d_order, order.size() * 4); (void)hipMalloc(&d_tab, (size_t)G * pts * sizeof(niels)); (void)hipMalloc(&d_b, counts.size() * sizeof(ge));
    (void)hipMemcpy(d_sorted, sorted.data(), sorted.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(d_starts, starts.data(), starts.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_counts, counts.data(), counts.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(d_order, order.data(), order.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_fill, dim3((G * pts + 255) / 256), dim3(256), 0, 0, d_tab, G * pts);
    PointTables tabs{d_tab, d_tab, 0xffffffffu};
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const dim3 grid(8 * ((G + 7) / 8) * ((per_group + 63) / 64));
    float best = 1e9f;
    for (int r = 0; r < 4; r++) {
      if (r == 3) hipLaunchKernelGGL(k_clock, dim3(1), dim3(64), 0, side, d_clk, 150);  // ~0.6 ms of naps next to the launch
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k_msm_accumulate, grid, dim3(64), 0, 0, d_sorted, d_starts, d_counts, d_order, tabs, per_group, G, d_b);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      if (r) best = std::min(best, ms);
    }
    (void)hipStreamSynchronize(side);
    // steady state: 150 launches back to back, the clock sampled over ~60 ms in the middle of them
    (void)hipEventRecord(e0);
    for (int r = 0; r < 150; r++) {
      if (r == 30) hipLaunchKernelGGL(k_clock, dim3(1), dim3(64), 0, side, d_clk, 15000);
      hipLaunchKernelGGL(k_msm_accumulate, grid, dim3(64), 0, 0, d_sorted, d_starts, d_counts, d_order, tabs, per_group, G, d_b);
    }
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms150; (void)hipEventElapsedTime(&ms150, e0, e1);
    best = ms150 / 150.0f;
    (void)hipStreamSynchronize(side);
    (void)hipMemcpy(h_clk, d_clk, 16, hipMemcpyDeviceToHost);
    const double ghz = (double)h_clk[0] / (double)h_clk[1] * 0.1;
    const double madds = (double)sorted.size() - (double)counts.size();  // first terms cost one product
    // arithmetic bound: madds / 64 lanes / 1024 SIMDs x 4387 cycles (fe_rates, three wavefronts per SIMD) at 2.37 GHz
    const double bound_ms = madds / 64.0 / 1024.0 * 4387.0 / (ghz * 1e6);
    printf("mode %d (%s sizes, %s)  %.3f ms  shader clock %.2f GHz (over %.2f ms)  arithmetic bound %.3f ms  -> %.0f %%\n", mode, uniform ? "equal" : "real",
           same ? "one table entry" : "random gather", best, ghz, (double)h_clk[1] * 1e-5, bound_ms, 100.0 * bound_ms / best);
    (void)hipFree(d_sorted); (void)hipFree(d_starts); (void)hipFree(d_counts); (void)hipFree(d_order); (void)hipFree(d_tab); (void)hipFree(d_b);
  }
  return 0;
}
