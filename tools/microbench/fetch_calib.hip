// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of k_msm_accumulate
// (MI355X_MICROARCH.md, HBM: "calibrate on a known byte count in your own access pattern before trusting an absolute").
// Every kernel touches each byte of a 4 GiB buffer (16x the Infinity Cache) exactly once, so the true HBM byte count is
// known: run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` (and WRITE_SIZE in a second pass) and divide.
//   k_stream16   coalesced 16 B / lane reads                       (the pattern the guide's x2 correction is quoted for)
//   k_gather128  one aligned 128-byte entry per lane, random order (a niels table entry: 7 x 16 B + 8 B used of 128)
//   k_gather120  one UNALIGNED 120-byte entry per lane, random order (round 1's table layout)
//   k_store160   one 160-byte extended point per lane, random order (a bucket written by its lane)
// build: hipcc -O3 --offload-arch=gfx950 fetch_calib.hip -o fetch_calib
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void __launch_bounds__(256) k_stream16(const uint4 *__restrict__ src, size_t n16, uint32_t *sink) {
  uint4 acc = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 v = src[i];
    acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w;
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}
template <int STRIDE, int WORDS>
__global__ void __launch_bounds__(64) k_gather(const uint8_t *__restrict__ src, uint32_t n_entries, uint32_t mult, uint32_t *sink) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_entries) return;
  const uint32_t e = (uint32_t)(((uint64_t)i * mult) % n_entries);  // mult odd and coprime to n_entries: a permutation
  const uint32_t *p = (const uint32_t *)(src + (size_t)e * STRIDE);
  uint32_t acc = 0;
#pragma unroll
  for (int k = 0; k < WORDS; k++) acc ^= p[k];
  if (acc == 0x12345678u) sink[0] = 1;
}
__global__ void __launch_bounds__(64) k_store160(uint8_t *__restrict__ dst, uint32_t n_entries, uint32_t mult) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_entries) return;
  const uint32_t e = (uint32_t)(((uint64_t)i * mult) % n_entries);
  uint32_t *p = (uint32_t *)(dst + (size_t)e * 160);
#pragma unroll
  for (int k = 0; k < 40; k++) p[k] = i + k;
}
int main() {
  const size_t bytes = 4ull << 30;
  uint8_t *buf;
  uint32_t *sink;
  CHECK(hipMalloc((void **)&buf, bytes + 256));
  CHECK(hipMalloc((void **)&sink, 4));
  CHECK(hipMemset(buf, 1, bytes));
  CHECK(hipDeviceSynchronize());
  hipLaunchKernelGGL(k_stream16, dim3(256 * 32), dim3(256), 0, 0, (const uint4 *)buf, bytes / 16, sink);
  const uint32_t n128 = (uint32_t)(bytes / 128), n120 = (uint32_t)(bytes / 120), n160 = (uint32_t)(bytes / 160);
  hipLaunchKernelGGL((k_gather<128, 30>), dim3((n128 + 63) / 64), dim3(64), 0, 0, buf, n128, 2654435761u, sink);
  hipLaunchKernelGGL((k_gather<120, 30>), dim3((n120 + 63) / 64), dim3(64), 0, 0, buf, n120, 2654435761u, sink);
  hipLaunchKernelGGL(k_store160, dim3((n160 + 63) / 64), dim3(64), 0, 0, buf, n160, 2654435761u);
  CHECK(hipDeviceSynchronize());
  printf("true bytes: k_stream16 %zu  k_gather<128> %zu (120 used of each 128-byte line)  k_gather<120> %zu  k_store160 %zu\n", bytes,
         (size_t)n128 * 128, (size_t)n120 * 120, (size_t)n160 * 160);
  return 0;
}
