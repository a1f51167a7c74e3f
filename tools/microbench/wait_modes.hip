// What does a host thread cost while it waits for the GPU?  CPU time (thread clock) and wake-up delay of one thread waiting
// for a kernel of ~T ms through (a) hipStreamSynchronize, (b) hipEventSynchronize on a default event, (c) hipEventSynchronize on
// an event created with hipEventBlockingSync, (d)/(e) (a)/(b) after hipSetDeviceFlags(hipDeviceScheduleBlockingSync),
// (f) polling hipEventQuery with 50 us naps.
// build: hipcc --offload-arch=gfx950 -O2 -o tools/microbench/wait_modes tools/microbench/wait_modes.hip
#include <hip/hip_runtime.h>
#include <time.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>

__global__ void k_spin(uint64_t ticks, uint32_t *out) {
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
  if (out && threadIdx.x == 0) *out = 1;
}
static double thread_cpu_ms() {
  timespec ts;
  clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec / 1e6;
}
static double process_cpu_ms() {  // every thread of the process: the runtime's own helpers included
  timespec ts;
  clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec / 1e6;
}
#define CK(x)                                                         \
  do {                                                                \
    hipError_t e = (x);                                               \
    if (e != hipSuccess) {                                            \
      printf("%s -> %s\n", #x, hipGetErrorString(e));                 \
      return 1;                                                       \
    }                                                                 \
  } while (0)

int main() {
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  uint32_t *d;
  CK(hipMalloc(&d, 4));
  hipEvent_t ev_plain, ev_block;
  CK(hipEventCreateWithFlags(&ev_plain, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&ev_block, hipEventDisableTiming | hipEventBlockingSync));
  const double kernel_ms = 5.0;
  for (int pass = 0; pass < 2; pass++) {
    if (pass == 1) {
      hipError_t e = hipSetDeviceFlags(hipDeviceScheduleBlockingSync);
      printf("hipSetDeviceFlags(hipDeviceScheduleBlockingSync) -> %s\n", hipGetErrorString(e));
    }
    for (int mode = 0; mode < 6; mode++) {
      double cpu = 0, wall = 0, pcpu = 0;
      const int reps = 20;
      for (int r = 0; r < reps; r++) {
        const auto t0 = std::chrono::steady_clock::now();
        const double c0 = thread_cpu_ms(), p0 = process_cpu_ms();
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, (uint64_t)(kernel_ms * 1e5), d);
        if (mode == 0) CK(hipStreamSynchronize(s));
        else if (mode == 1) {
          CK(hipEventRecord(ev_plain, s));
          CK(hipEventSynchronize(ev_plain));
        } else if (mode == 2) {
          CK(hipEventRecord(ev_block, s));
          CK(hipEventSynchronize(ev_block));
        } else if (mode == 3) {
          CK(hipEventRecord(ev_plain, s));
          while (hipEventQuery(ev_plain) == hipErrorNotReady) usleep(50);
        } else if (mode == 4) {
          while (hipStreamQuery(s) == hipErrorNotReady) usleep(50);
        } else {  // one hipStreamQuery first (as a "nothing left?" look), then the event
          (void)hipStreamQuery(s);
          CK(hipEventRecord(ev_plain, s));
          while (hipEventQuery(ev_plain) == hipErrorNotReady) usleep(50);
        }
        cpu += thread_cpu_ms() - c0;
        pcpu += process_cpu_ms() - p0;
        wall += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      }
      const char *names[6] = {"hipStreamSynchronize", "hipEventSynchronize (plain event)", "hipEventSynchronize (hipEventBlockingSync)",
                              "hipEventQuery + usleep(50)", "hipStreamQuery + usleep(50)", "hipStreamQuery once, then hipEventQuery + usleep(50)"};
      printf("%-54s kernel %.1f ms: wall %.3f ms, thread CPU %.3f ms per wait (%.0f %% of a core), PROCESS CPU %.3f ms (%.0f %%)\n", names[mode],
             kernel_ms, wall / reps, cpu / reps, 100.0 * cpu / wall, pcpu / reps, 100.0 * pcpu / wall);
    }
  }
  return 0;
}
