// How many kernel launches / small copies per second does the HIP runtime take from S host threads, each on its own stream?
// (the ceiling of independent small calls: a 256-proof verify call is about two dozen such operations)
// build: hipcc --offload-arch=gfx950 -O2 -o tools/microbench/launch_rate tools/microbench/launch_rate.hip -lpthread
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
__global__ void k_nop(uint32_t *p) {
  if (p && threadIdx.x == 4096) *p = 1;
}
int main() {
  for (int mode = 0; mode < 3; mode++)  // 0: launches only, 1: launch + 256-byte D2H copy, 2: 8 launches then a stream sync
    for (int S : {1, 4, 8, 16, 32}) {
      std::vector<hipStream_t> st(S);
      std::vector<uint32_t *> dev(S), host(S);
      for (int k = 0; k < S; k++) {
        (void)hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking);
        (void)hipMalloc(&dev[k], 256);
        (void)hipHostMalloc((void **)&host[k], 256, hipHostMallocDefault);
      }
      std::atomic<uint64_t> ops{0};
      const auto t0 = std::chrono::steady_clock::now();
      const auto stop = t0 + std::chrono::milliseconds(700);
      std::vector<std::thread> th;
      for (int k = 0; k < S; k++)
        th.emplace_back([&, k] {
          uint64_t c = 0;
          while (std::chrono::steady_clock::now() < stop) {
            for (int i = 0; i < 8; i++) {
              hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, st[k], dev[k]);
              c++;
              if (mode == 1) {
                (void)hipMemcpyAsync(host[k], dev[k], 256, hipMemcpyDeviceToHost, st[k]);
                c++;
              }
            }
            if (mode != 0 || (c & 1023) == 0) (void)hipStreamSynchronize(st[k]);
          }
          (void)hipStreamSynchronize(st[k]);
          ops += c;
        });
      for (auto &t : th) t.join();
      const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      printf("mode %d (%s), %2d threads: %.0f k operations/s\n", mode, mode == 0 ? "launches" : mode == 1 ? "launch + small D2H copy, sync every 8" : "8 launches, then a sync", S,
             ops / el / 1e3);
      for (int k = 0; k < S; k++) {
        (void)hipStreamDestroy(st[k]);
        (void)hipFree(dev[k]);
        (void)hipHostFree(host[k]);
      }
    }
  return 0;
}
