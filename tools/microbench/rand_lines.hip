// Random 128-byte line reads over a table of a given size: what does HBM / Infinity Cache / L2 deliver when every lane
// of a wavefront wants a different aligned line (the access pattern of the prover's fixed-base table lookups)?
// build: hipcc -O3 --offload-arch=gfx950 rand_lines.hip -o rand_lines ; run: ./rand_lines
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

__global__ void __launch_bounds__(256) k_rand(const uint4 *__restrict__ tbl, uint32_t n_lines, uint32_t iters, int bytes_per_line,
                                              uint32_t *__restrict__ sink) {
  uint32_t x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
  uint4 acc = {0, 0, 0, 0};
  for (uint32_t i = 0; i < iters; i++) {
    x = x * 1664525u + 1013904223u;
    const uint32_t line = (uint32_t)(((uint64_t)(x ^ (x >> 15)) * n_lines) >> 32);
    const uint4 *p = tbl + (size_t)line * 8;
    for (int k = 0; k < bytes_per_line / 16; k++) {
      const uint4 v = p[k];
      acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w;
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

int main() {
  const size_t sizes_mb[] = {2, 16, 64, 256, 512, 2048, 4096, 8192, 16384, 32768, 65536};
  uint32_t *sink;
  hipMalloc((void **)&sink, 4);
  for (size_t mb : sizes_mb) {
    const size_t bytes = mb << 20;
    uint4 *tbl;
    if (hipMalloc((void **)&tbl, bytes) != hipSuccess) return 1;
    hipMemset(tbl, 1, bytes);
    const uint32_t n_lines = (uint32_t)(bytes / 128);  // < 2^32 up to 512 GB
    for (int bpl : {128, 64}) {
      const uint32_t blocks = 256 * 12, iters = 256;
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      hipLaunchKernelGGL(k_rand, dim3(blocks), dim3(256), 0, 0, tbl, n_lines, iters, bpl, sink);
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k_rand, dim3(blocks), dim3(256), 0, 0, tbl, n_lines, iters, bpl, sink);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double lines = (double)blocks * 256 * iters;
      printf("table %5zu MB  %3d B/line : %7.2f G lines/s  %7.2f TB/s\n", mb, bpl, lines / ms / 1e6, lines * bpl / ms / 1e9);
    }
    hipFree(tbl);
  }
  return 0;
}
