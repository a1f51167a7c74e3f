// Microbenchmark: issue rate of the integer multiply forms the field/scalar arithmetic is built from (gfx950).
// Instructions are pinned with inline asm (8 independent chains per lane).
// Build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/int_rates tools/microbench/int_rates.hip && /tmp/int_rates
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define ITER 2048
template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t *out, uint32_t seed) {
  uint32_t a = threadIdx.x + seed, b = blockIdx.x * 7 + 3;
  uint64_t acc[8];
  uint32_t r[8];
  double d[8];
  for (int i = 0; i < 8; i++) { acc[i] = a + i; r[i] = a ^ i; d[i] = (double)(a + i); }
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (OP == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(r[i]), "v"(b) : "vcc");
      if (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r[i]) : "v"(b));
      if (OP == 2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(r[i]) : "v"(b));
      if (OP == 3) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(r[i]) : "v"(b), "v"(a));
      if (OP == 4) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(b));
      if (OP == 5) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(d[(i + 1) & 7]), "v"(d[(i + 2) & 7]));
      if (OP == 6) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(r[i]) : "v"(b));
      if (OP == 7) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %2, vcc, %2, %3, vcc" : "+v"(r[i]), "+v"(a) : "v"(b), "v"(b) : "vcc");
    }
  }
  uint32_t s = a;
  for (int i = 0; i < 8; i++) s += (uint32_t)acc[i] + (uint32_t)(acc[i] >> 32) + r[i] + (uint32_t)d[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP>
void run(const char *name, int blocks, double instr_per_iter) {
  uint32_t *out;
  (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1u);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 2u);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double ins = (double)blocks * 256 * ITER * 8 * instr_per_iter;
  printf("%-22s waves/SIMD=%d  %.3f ms  %.2f T lane-instr/s = %.1f lanes/clk/CU @2.4GHz\n", name, blocks / 256, ms, ins / ms / 1e9,
         ins / (ms * 1e-3) / 256 / 2.4e9);
  (void)hipFree(out);
}
int main() {
  for (int blocks : {256 * 1, 256 * 2, 256 * 8}) {
    run<0>("v_mad_u64_u32", blocks, 1);
    run<1>("v_mul_lo_u32", blocks, 1);
    run<2>("v_mul_hi_u32", blocks, 1);
    run<3>("v_mad_u32_u24", blocks, 1);
    run<4>("v_add_u32", blocks, 1);
    run<6>("v_lshl_add_u32", blocks, 1);
    run<7>("v_add_co+v_addc_co", blocks, 2);
    run<5>("v_fma_f64", blocks, 1);
  }
  return 0;
}
