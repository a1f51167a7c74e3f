// Microbenchmark (gfx950): what the field / point primitives of csrc/ really cost on the chip, away from memory: cycles of
// SIMD time per wavefront-level fe_mul / fe_sq / mixed point addition at 1..5 wavefronts per SIMD, against their issue-slot
// count (half-rate instructions = 2 slots of 2.25 cycles).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I bulletproofs-plus_amd/csrc -o tools/microbench/fe_rates tools/microbench/fe_rates.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "point.h"
using namespace bpp;
#define ITER 2048
template <int OP>
__global__ void __launch_bounds__(64) k(const uint32_t *in, uint32_t *out, uint64_t *clk) {
  fe x, y;
  for (int i = 0; i < 10; i++) { x.v[i] = in[threadIdx.x * 10 + i] & 0x1ffffff; y.v[i] = in[640 + threadIdx.x * 10 + i] & 0x1ffffff; }
  ge p;
  p.X = x; p.Y = y; p.Z = x; p.T = y;
  niels q;
  q.yplusx = y; q.yminusx = x; q.xy2d = y;
  const uint64_t c0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < ITER; it++) {
    if (OP == 0) fe_mul(x, x, y);
    if (OP == 1) fe_sq(x, x);
    if (OP == 2) ge_madd_swapped(p, p, q, (it & 1) != 0);
    if (OP == 3) ge_dbl(p, p);
  }
  const uint64_t c1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
  uint32_t s = 0;
  for (int i = 0; i < 10; i++) s += x.v[i] + p.X.v[i] + p.Y.v[i] + p.Z.v[i] + p.T.v[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}
template <int OP>
void run(const char *name, int waves, double slots) {
  const int blocks = 1024 * waves;
  uint32_t *in, *out; uint64_t *clk, *hclk = (uint64_t *)malloc((size_t)blocks * 16);
  (void)hipMalloc(&in, 1280 * 4); (void)hipMemset(in, 0x5a, 1280 * 4);
  (void)hipMalloc(&out, (size_t)blocks * 64 * 4); (void)hipMalloc(&clk, (size_t)blocks * 16);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, in, out, clk);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, in, out, clk);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipMemcpy(hclk, clk, (size_t)blocks * 16, hipMemcpyDeviceToHost);
  double cyc = 0, wall = 0;
  for (int i = 0; i < blocks; i++) { cyc += (double)hclk[2 * i]; wall += (double)hclk[2 * i + 1]; }
  const double ghz = cyc / wall * 0.1;
  const double simd_cycles_per_op = ms * 1e-3 * ghz * 1e9 / ((double)ITER * waves);  // SIMD time per wavefront-level operation
  printf("%-18s waves/SIMD %d  %7.3f ms  %.2f GHz  %7.1f SIMD cycles per op (issue slots %.0f x 2.25 = %.0f: %.0f %%)\n", name, waves, ms, ghz,
         simd_cycles_per_op, slots, slots * 2.25, 100.0 * slots * 2.25 / simd_cycles_per_op);
  (void)hipFree(in); (void)hipFree(out); (void)hipFree(clk); free(hclk);
}
int main() {
  for (int w : {1, 2, 3, 4, 5}) {
    run<0>("fe_mul", w, 272);
    run<1>("fe_sq", w, 175);
    run<2>("ge_madd_swapped", w, 1949);
    run<3>("ge_dbl", w, 1775);
  }
  return 0;
}
