// Microbenchmark (gfx950): what does a per-lane select cost?  v_cndmask_b32 with the mask in VCC / in an SGPR pair,
// v_bfi_b32 with a lane mask in a VGPR, and the and / andn2 / or form.  8 independent chains per lane, long kernels.
// build: hipcc --offload-arch=gfx950 -O3 -o select_rates select_rates.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define ITER 32768
template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t *out, uint64_t *clk, uint32_t seed) {
  uint32_t a = threadIdx.x + seed, b = blockIdx.x * 7 + 3, m = 0u - (threadIdx.x & 1u);
  uint32_t r[8];
  for (int i = 0; i < 8; i++) r[i] = a ^ i;
  uint64_t mask64;
  asm volatile("v_cmp_ne_u32 %0, 0, %1" : "=s"(mask64) : "v"(threadIdx.x & 1u));
  asm volatile("v_cmp_ne_u32 vcc, 0, %0" : : "v"(threadIdx.x & 1u) : "vcc");
  const uint64_t c0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (OP == 0) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(b));
      if (OP == 1) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(b), "s"(mask64));
      if (OP == 2) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(r[i]) : "v"(m), "v"(b));
      if (OP == 3) asm volatile("v_and_b32 %0, %0, %1\n\tv_or_b32 %0, %0, %2" : "+v"(r[i]) : "v"(m), "v"(b));
      if (OP == 4) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(b));
      if (OP == 5) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(r[i]) : "v"(a), "v"(b));
    }
  }
  const uint64_t c1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
  uint32_t s = a;
  for (int i = 0; i < 8; i++) s += r[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}
template <int OP>
void run(const char *name, int blocks, double instr_per_iter) {
  uint32_t *out; uint64_t *clk, *hclk = (uint64_t *)malloc((size_t)blocks * 16);
  (void)hipMalloc(&out, (size_t)blocks * 256 * 4); (void)hipMalloc(&clk, (size_t)blocks * 16);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, clk, 1u);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, clk, 2u);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipMemcpy(hclk, clk, (size_t)blocks * 16, hipMemcpyDeviceToHost);
  double cyc = 0, wall = 0;
  for (int i = 0; i < blocks; i++) { cyc += (double)hclk[2 * i]; wall += (double)hclk[2 * i + 1]; }
  const double ghz = cyc / wall * 0.1, ins = (double)blocks * 256 * ITER * 8 * instr_per_iter;
  printf("%-40s waves/SIMD=%d %7.3f ms  clock %.2f GHz  %.1f lanes/clk/CU\n", name, blocks / 256, ms, ghz, ins / (ms * 1e-3) / 256 / (ghz * 1e9));
  (void)hipFree(out); (void)hipFree(clk); free(hclk);
}
int main() {
  for (int blocks : {256 * 3, 256 * 8}) {
    run<4>("v_add_u32 (reference)", blocks, 1);
    run<0>("v_cndmask_b32 dst=src0, vcc", blocks, 1);
    run<5>("v_cndmask_b32 dst fresh, vcc", blocks, 1);
    run<1>("v_cndmask_b32 e64, sgpr pair", blocks, 1);
    run<2>("v_bfi_b32 (vgpr mask)", blocks, 1);
    run<3>("v_and + v_or (2 instr)", blocks, 2);
  }
  return 0;
}
