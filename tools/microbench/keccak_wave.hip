// Latency of ONE Keccak-f[1600] on one wavefront: wstrobe.h's form (a 64-bit word per lane, two LDS exchanges per round)
// against wkeccak.h's (an interleaved half per lane, one exchange + DPP per round); both held to merlin.h's one-lane form.
// build: hipcc --offload-arch=gfx950 -O3 -I bulletproofs-plus_amd/csrc -o tools/microbench/keccak_wave tools/microbench/keccak_wave.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <vector>

#include "wkeccak.h"
#include "wstrobe.h"

using namespace bpp;

__global__ void __launch_bounds__(64) k_ref(uint64_t *st, int n) {
  if (threadIdx.x != 0) return;
  uint64_t a[25];
  for (int i = 0; i < 25; i++) a[i] = st[blockIdx.x * 25 + i];
  for (int k = 0; k < n; k++) keccak_f1600(a);
  for (int i = 0; i < 25; i++) st[blockIdx.x * 25 + i] = a[i];
}
template <int FORM>  // 0: the 25-lane form wstrobe.h had until round 5; 1: the sponge's permutation now (words in LDS -> halves -> words)
__global__ void __launch_bounds__(64) k_wave64(uint64_t *st, int n) {
  __shared__ uint64_t L[25];
  __shared__ uint32_t img[WK_LDS_DWORDS_RC];
  if (threadIdx.x < 25) L[threadIdx.x] = st[blockIdx.x * 25 + threadIdx.x];
  const KeccakLanes K = keccak_lanes(img);
  ws_sync();
  for (int k = 0; k < n; k++) {
    if (FORM == 0) keccak_f1600_wave25(L, K);
    else keccak_f1600_wave(L, K);
    ws_sync();
  }
  if (threadIdx.x < 25) st[blockIdx.x * 25 + threadIdx.x] = L[threadIdx.x];
}
__global__ void __launch_bounds__(64) k_wave50(uint64_t *st, int n) {
  __shared__ uint64_t S[25];
  __shared__ uint32_t L[WK_LDS_DWORDS];
  if (threadIdx.x < 25) S[threadIdx.x] = st[blockIdx.x * 25 + threadIdx.x];
  const WkLanes W = wk_lanes(L);
  const WkRc R = wk_rc(W);
  ws_sync();
  uint32_t a = wk_load(S, W);
  for (int k = 0; k < n; k++) a = wk_keccak_f1600(a, W, R);
  wk_store(S, L, a, W);
  if (threadIdx.x < 25) st[blockIdx.x * 25 + threadIdx.x] = S[threadIdx.x];
}

#define CK(x)                                                                    \
  do {                                                                           \
    hipError_t e = (x);                                                          \
    if (e != hipSuccess) {                                                       \
      printf("%s -> %s\n", #x, hipGetErrorString(e));                            \
      return 1;                                                                  \
    }                                                                            \
  } while (0)

int main() {
  const int blocks = 64;
  std::vector<uint64_t> h0(25 * blocks), h(25 * blocks), want(25 * blocks);
  for (size_t i = 0; i < h0.size(); i++) h0[i] = 0x9E3779B97F4A7C15ULL * (i + 1) ^ (i << 40);
  uint64_t *d;
  CK(hipMalloc(&d, h0.size() * 8));
  // host-side check of the interleaving helpers
  for (uint64_t w : {0x0123456789abcdefULL, 0x8000000000000001ULL, 0xffffffff00000000ULL})
    if (wk_word(wk_half(w, 0), wk_half(w, 1)) != w) {
      printf("interleave round trip FAILED\n");
      return 1;
    }
  const int n_check = 7;
  CK(hipMemcpy(d, h0.data(), h0.size() * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_ref, dim3(blocks), dim3(64), 0, 0, d, n_check);
  CK(hipMemcpy(want.data(), d, h0.size() * 8, hipMemcpyDeviceToHost));
  const char *names[3] = {"25 lanes, a word each, two exchanges per round (wstrobe.h until round 5)",
                          "wstrobe.h now: LDS words -> halves, one exchange per round, -> words", "wkeccak.h, state kept in registers (the weight chain)"};
  auto launch = [&](int v, int nb, int n) {
    if (v == 0) hipLaunchKernelGGL(k_wave64<0>, dim3(nb), dim3(64), 0, 0, d, n);
    else if (v == 1) hipLaunchKernelGGL(k_wave64<1>, dim3(nb), dim3(64), 0, 0, d, n);
    else hipLaunchKernelGGL(k_wave50, dim3(nb), dim3(64), 0, 0, d, n);
  };
  for (int v = 0; v < 3; v++) {
    CK(hipMemcpy(d, h0.data(), h0.size() * 8, hipMemcpyHostToDevice));
    launch(v, blocks, n_check);
    CK(hipMemcpy(h.data(), d, h0.size() * 8, hipMemcpyDeviceToHost));
    printf("%s: %s\n", names[v], memcmp(h.data(), want.data(), h0.size() * 8) == 0 ? "equal to the one-lane form" : "DIFFERS");
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int nb : {1, 64}) {
    for (int v = 0; v < 3; v++) {
      const int n = 4000;
      for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0, 0));
        launch(v, nb > blocks ? blocks : nb, n);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep == 2) printf("form %d, wavefronts %4d: %.3f us per permutation\n", v, nb > blocks ? blocks : nb, 1e3 * ms / n);
      }
    }
  }
  return 0;
}
