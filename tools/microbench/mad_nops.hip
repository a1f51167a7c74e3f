// Microbenchmark (gfx950): what do the `s_nop 0` cost that the compiler puts between two inline-asm v_mad_u64_u32 when the
// second reads the first's result (it must assume a forwarding hazard for instructions it cannot see), and does the
// carry-out register matter?  One wavefront per block; modes:
//   0  dependent chain, one asm statement per multiply, carry-out in vcc          (compiler inserts s_nop 0 between them)
//   1  dependent chain, one asm statement per multiply, carry-out in its own SGPR pair
//   2  dependent chain, EIGHT multiplies in ONE asm statement (no s_nop), carry-outs in s[20:35] (clobbered)
//   3  eight independent chains, one statement per multiply, own SGPR pairs
//   4  eight independent chains in one asm statement
// build: hipcc --offload-arch=gfx950 -O3 -o mad_nops mad_nops.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define ITER 16384
template <int MODE>
__global__ void __launch_bounds__(64) k(uint32_t *out, uint64_t *clk, uint32_t seed) {
  uint32_t a = threadIdx.x + seed, b = blockIdx.x * 7 + 3;
  uint64_t acc[8];
  for (int i = 0; i < 8; i++) acc[i] = a + i;
  const uint64_t c0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < ITER; it++) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 8; i++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[0]) : "v"(a), "v"(b) : "vcc");
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        uint64_t co;
        asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc[0]), "=s"(co) : "v"(a), "v"(b));
      }
    } else if (MODE == 2) {
      asm volatile(
          "v_mad_u64_u32 %0, s[20:21], %1, %2, %0\n\tv_mad_u64_u32 %0, s[22:23], %1, %2, %0\n\t"
          "v_mad_u64_u32 %0, s[24:25], %1, %2, %0\n\tv_mad_u64_u32 %0, s[26:27], %1, %2, %0\n\t"
          "v_mad_u64_u32 %0, s[28:29], %1, %2, %0\n\tv_mad_u64_u32 %0, s[30:31], %1, %2, %0\n\t"
          "v_mad_u64_u32 %0, s[32:33], %1, %2, %0\n\tv_mad_u64_u32 %0, s[34:35], %1, %2, %0"
          : "+v"(acc[0])
          : "v"(a), "v"(b)
          : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35");
    } else if (MODE == 3) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        uint64_t co;
        asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc[i]), "=s"(co) : "v"(a), "v"(b));
      }
    } else {
      asm volatile(
          "v_mad_u64_u32 %0, s[20:21], %8, %9, %0\n\tv_mad_u64_u32 %1, s[22:23], %8, %9, %1\n\t"
          "v_mad_u64_u32 %2, s[24:25], %8, %9, %2\n\tv_mad_u64_u32 %3, s[26:27], %8, %9, %3\n\t"
          "v_mad_u64_u32 %4, s[28:29], %8, %9, %4\n\tv_mad_u64_u32 %5, s[30:31], %8, %9, %5\n\t"
          "v_mad_u64_u32 %6, s[32:33], %8, %9, %6\n\tv_mad_u64_u32 %7, s[34:35], %8, %9, %7"
          : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
          : "v"(a), "v"(b)
          : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35");
    }
  }
  const uint64_t c1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
  uint32_t s = a;
  for (int i = 0; i < 8; i++) s += (uint32_t)acc[i] + (uint32_t)(acc[i] >> 32);
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}
template <int MODE>
void run(int waves_per_simd) {
  const int blocks = 1024 * waves_per_simd;
  uint32_t *out; uint64_t *clk, *hclk = (uint64_t *)malloc((size_t)blocks * 16);
  (void)hipMalloc(&out, (size_t)blocks * 64 * 4); (void)hipMalloc(&clk, (size_t)blocks * 16);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, clk, 1u);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, clk, 2u);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipMemcpy(hclk, clk, (size_t)blocks * 16, hipMemcpyDeviceToHost);
  double cyc = 0, wall = 0;
  for (int i = 0; i < blocks; i++) { cyc += (double)hclk[2 * i]; wall += (double)hclk[2 * i + 1]; }
  const double ghz = cyc / wall * 0.1, ins = (double)blocks * 64 * ITER * 8;
  printf("mode %d  waves/SIMD %d  %7.3f ms  clock %.2f GHz  %.1f lanes/clk/CU  (%.1f cycles per multiply per wavefront)\n", MODE, waves_per_simd, ms, ghz,
         ins / (ms * 1e-3) / 256 / (ghz * 1e9), cyc / blocks / ((double)ITER * 8));
  (void)hipFree(out); (void)hipFree(clk); free(hclk);
}
int main() {
  for (int w : {1, 2, 3, 4}) { run<0>(w); run<1>(w); run<2>(w); run<3>(w); run<4>(w); }
  return 0;
}
