// Does hipExtStreamCreateWithCUMask restrict a stream's kernels to the masked compute units on this part / runtime?
// A chip-filling arithmetic kernel on an unmasked stream, then on streams with every 2nd / 4th / 8th CU, blocks of the first 64 / 32 CUs.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
__global__ void spin(uint32_t *out, uint32_t iters) {
  uint32_t x = threadIdx.x + blockIdx.x, y = 0x9e3779b9u;
  for (uint32_t i = 0; i < iters; i++) {
    x = x * 1664525u + 1013904223u;
    y ^= x + (y << 6) + (y >> 2);
  }
  if (y == 0x12345678u) out[0] = x;
}
static float run(hipStream_t s, uint32_t *d) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  hipLaunchKernelGGL(spin, dim3(8192), dim3(256), 0, s, d, 20000u);
  (void)hipStreamSynchronize(s);
  (void)hipEventRecord(a, s);
  hipLaunchKernelGGL(spin, dim3(8192), dim3(256), 0, s, d, 20000u);
  (void)hipEventRecord(b, s);
  (void)hipStreamSynchronize(s);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, a, b);
  return ms;
}
int main() {
  hipDeviceProp_t prop;
  (void)hipGetDeviceProperties(&prop, 0);
  const uint32_t n_cu = prop.multiProcessorCount;
  uint32_t *d;
  (void)hipMalloc(&d, 4);
  hipStream_t s0;
  (void)hipStreamCreateWithFlags(&s0, hipStreamNonBlocking);
  printf("compute units %u; unmasked %.3f ms\n", n_cu, run(s0, d));
  for (uint32_t step : {2u, 4u, 8u}) {
    std::vector<uint32_t> mask((n_cu + 31) / 32, 0u);
    for (uint32_t cu = 0; cu < n_cu; cu += step) mask[cu / 32] |= 1u << (cu % 32);
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
    printf("every %u-th CU: create rc %d, %.3f ms\n", step, (int)e, e == hipSuccess ? run(s, d) : -1.f);
  }
  const uint32_t ranges[][2] = {{0, 64}, {0, 32}, {64, 128}, {128, 192}, {192, 256}, {0, 128}, {128, 256}, {32, 64}, {85, 170}};
  for (auto &rg : ranges) {
    std::vector<uint32_t> mask((n_cu + 31) / 32, 0u);
    for (uint32_t cu = rg[0]; cu < rg[1]; cu++) mask[cu / 32] |= 1u << (cu % 32);
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
    printf("CUs [%u, %u): create rc %d, %.3f ms\n", rg[0], rg[1], (int)e, e == hipSuccess ? run(s, d) : -1.f);
  }
  {  // two masked streams side by side: disjoint halves
    std::vector<uint32_t> m0((n_cu + 31) / 32, 0u), m1((n_cu + 31) / 32, 0u);
    for (uint32_t cu = 0; cu < 128; cu++) m0[cu / 32] |= 1u << (cu % 32);
    for (uint32_t cu = 128; cu < 256; cu++) m1[cu / 32] |= 1u << (cu % 32);
    hipStream_t a, b;
    (void)hipExtStreamCreateWithCUMask(&a, (uint32_t)m0.size(), m0.data());
    (void)hipExtStreamCreateWithCUMask(&b, (uint32_t)m1.size(), m1.data());
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, a);
    hipLaunchKernelGGL(spin, dim3(8192), dim3(256), 0, a, d, 20000u);
    hipLaunchKernelGGL(spin, dim3(8192), dim3(256), 0, b, d, 20000u);
    (void)hipStreamSynchronize(b);
    (void)hipEventRecord(e1, a);
    (void)hipStreamSynchronize(a);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("two launches on disjoint halves, side by side: %.3f ms for both\n", ms);
  }
  return 0;
}
