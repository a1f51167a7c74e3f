// Microbenchmark (gfx950): v_mad_u64_u32 throughput as a function of independent chains per wavefront and wavefronts per
// SIMD -- how much instruction-level parallelism does a multiply-bound kernel need at a given occupancy?
// build: hipcc --offload-arch=gfx950 -O3 -o mad_chains mad_chains.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define ITER 16384
template <int CH>
__global__ void __launch_bounds__(64) k(uint32_t *out, uint64_t *clk, uint32_t seed) {
  uint32_t a = threadIdx.x + seed, b = blockIdx.x * 7 + 3;
  uint64_t acc[8];
  for (int i = 0; i < 8; i++) acc[i] = a + i;
  const uint64_t c0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int r = 0; r < 8 / CH; r++)
#pragma unroll
      for (int i = 0; i < CH; i++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
  }
  const uint64_t c1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
  uint32_t s = a;
  for (int i = 0; i < 8; i++) s += (uint32_t)acc[i] + (uint32_t)(acc[i] >> 32);
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}
template <int CH>
void run(int waves_per_simd) {
  const int blocks = 1024 * waves_per_simd;  // one wavefront per block
  uint32_t *out; uint64_t *clk, *hclk = (uint64_t *)malloc((size_t)blocks * 16);
  (void)hipMalloc(&out, (size_t)blocks * 64 * 4); (void)hipMalloc(&clk, (size_t)blocks * 16);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<CH>, dim3(blocks), dim3(64), 0, 0, out, clk, 1u);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<CH>, dim3(blocks), dim3(64), 0, 0, out, clk, 2u);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipMemcpy(hclk, clk, (size_t)blocks * 16, hipMemcpyDeviceToHost);
  double cyc = 0, wall = 0;
  for (int i = 0; i < blocks; i++) { cyc += (double)hclk[2 * i]; wall += (double)hclk[2 * i + 1]; }
  const double ghz = cyc / wall * 0.1, ins = (double)blocks * 64 * ITER * 8;
  printf("chains/wave %d  waves/SIMD %d  %7.3f ms  clock %.2f GHz  %.1f lanes/clk/CU  (%.1f cycles per mad per wave)\n", CH, waves_per_simd, ms, ghz,
         ins / (ms * 1e-3) / 256 / (ghz * 1e9), cyc / blocks / ((double)ITER * 8));
  (void)hipFree(out); (void)hipFree(clk); free(hclk);
}
int main() {
  for (int w : {1, 2, 3, 4, 5, 6, 8}) { run<1>(w); run<2>(w); run<4>(w); run<8>(w); }
  return 0;
}
