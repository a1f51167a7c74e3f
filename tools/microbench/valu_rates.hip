// Microbenchmark (gfx950): issue rate of every VALU form the field / scalar arithmetic uses, long kernels (>= 5 ms each) and
// the shader clock actually held while they run, so that "lanes per clock per CU" is a measured quotient, not an assumption.
// Instructions are pinned with inline asm, 8 independent chains per lane.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rates tools/microbench/valu_rates.hip && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define ITER 65536
template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t *out, uint64_t *clk, uint32_t seed) {
  uint32_t a = threadIdx.x + seed, b = blockIdx.x * 7 + 3;
  uint64_t acc[8];
  uint32_t r[8];
  double d[8];
  for (int i = 0; i < 8; i++) { acc[i] = a + i; r[i] = a ^ i; d[i] = (double)(a + i); }
  const uint64_t c0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
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
      if (OP == 8) asm volatile("v_lshrrev_b64 %0, 26, %0" : "+v"(acc[i]));
      if (OP == 9) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[i]) : "v"(acc[(i + 1) & 7]));
      if (OP == 10) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r[i]) : "v"(b));
      if (OP == 11) asm volatile("v_alignbit_b32 %0, %0, %1, 26" : "+v"(r[i]) : "v"(b));
      if (OP == 12) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(b), "v"(a));
      if (OP == 13) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(b));
      if (OP == 14) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r[i]) : "v"(b));
      if (OP == 15) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(r[i]), "v"(b) : "vcc");
      if (OP == 16) asm volatile("v_lshrrev_b32 %0, 26, %0" : "+v"(r[i]));
      if (OP == 17) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(b), "v"(a));
      if (OP == 18) asm volatile("v_bfe_u32 %0, %0, 3, 26" : "+v"(r[i]));
      // mixes: does a cheap instruction ride along with a multiply?
      if (OP == 20) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_add_u32 %1, %1, %2" : "+v"(acc[i]), "+v"(r[i]) : "v"(b) : "vcc");
      if (OP == 21) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_add_u32 %1, %1, %2\n\tv_and_b32 %1, %1, %3" : "+v"(acc[i]), "+v"(r[i]) : "v"(b), "v"(a) : "vcc");
      if (OP == 22) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_lshrrev_b64 %3, 26, %3" : "+v"(acc[i]), "+v"(r[i]), "+v"(b), "+v"(acc[(i + 4) & 7]) : : "vcc");
      if (OP == 23) asm volatile("v_dot2_u32_u16 %0, %0, %1, %2" : "+v"(r[i]) : "v"(b), "v"(a));
      if (OP == 24) asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(r[i]) : "v"(b), "v"(a));
      if (OP == 25) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(r[i]) : "v"(b));
      if (OP == 26) asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(r[i]) : "v"(b), "v"(a));
      if (OP == 27) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(b), "v"(a));
      if (OP == 28) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(acc[(i + 1) & 7]), "v"(acc[(i + 2) & 7]));
    }
  }
  const uint64_t c1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
  uint32_t s = a;
  for (int i = 0; i < 8; i++) s += (uint32_t)acc[i] + (uint32_t)(acc[i] >> 32) + r[i] + (uint32_t)d[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) {
    clk[2 * blockIdx.x] = c1 - c0;
    clk[2 * blockIdx.x + 1] = w1 - w0;
  }
}
template <int OP>
void run(const char *name, int blocks, double instr_per_iter) {
  uint32_t *out;
  uint64_t *clk, *hclk = (uint64_t *)malloc((size_t)blocks * 16);
  (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
  (void)hipMalloc(&clk, (size_t)blocks * 16);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, clk, 1u);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, clk, 2u);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipMemcpy(hclk, clk, (size_t)blocks * 16, hipMemcpyDeviceToHost);
  double cyc = 0, wall = 0;
  for (int i = 0; i < blocks; i++) { cyc += (double)hclk[2 * i]; wall += (double)hclk[2 * i + 1]; }
  const double ghz = cyc / wall * 0.1;  // s_memrealtime ticks at 100 MHz
  double ins = (double)blocks * 256 * ITER * 8 * instr_per_iter;
  printf("%-34s waves/SIMD=%d  %7.3f ms  %6.2f T lane-instr/s  clock %.2f GHz  %.1f lanes/clk/CU\n", name, blocks / 256, ms, ins / ms / 1e9,
         ghz, ins / (ms * 1e-3) / 256 / (ghz * 1e9));
  (void)hipFree(out); (void)hipFree(clk); free(hclk);
}
int main() {
  for (int blocks : {256 * 1, 256 * 2, 256 * 3, 256 * 8}) {
    run<0>("v_mad_u64_u32", blocks, 1);
    run<15>("v_mad_i64_i32", blocks, 1);
    run<1>("v_mul_lo_u32", blocks, 1);
    run<2>("v_mul_hi_u32", blocks, 1);
    run<3>("v_mad_u32_u24", blocks, 1);
    run<14>("v_mul_u32_u24", blocks, 1);
    run<4>("v_add_u32", blocks, 1);
    run<10>("v_and_b32", blocks, 1);
    run<16>("v_lshrrev_b32", blocks, 1);
    run<6>("v_lshl_add_u32", blocks, 1);
    run<12>("v_add3_u32", blocks, 1);
    run<17>("v_and_or_b32", blocks, 1);
    run<18>("v_bfe_u32", blocks, 1);
    run<11>("v_alignbit_b32", blocks, 1);
    run<13>("v_cndmask_b32", blocks, 1);
    run<8>("v_lshrrev_b64", blocks, 1);
    run<9>("v_lshl_add_u64", blocks, 1);
    run<7>("v_add_co+v_addc_co", blocks, 2);
    run<5>("v_fma_f64", blocks, 1);
    run<27>("v_fma_f32", blocks, 1);
    run<28>("v_pk_fma_f32", blocks, 1);
    run<23>("v_dot2_u32_u16", blocks, 1);
    run<24>("v_dot4_u32_u8", blocks, 1);
    run<25>("v_pk_mul_lo_u16", blocks, 1);
    run<26>("v_pk_mad_u16", blocks, 1);
    run<20>("mad_u64 + add_u32 (2 instr)", blocks, 2);
    run<21>("mad_u64 + add + and (3 instr)", blocks, 3);
    run<22>("mad_u64 + lshrrev_b64 (2 instr)", blocks, 2);
  }
  return 0;
}
