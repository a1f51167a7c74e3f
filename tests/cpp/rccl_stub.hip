// A stand-in for librccl (tests only; loaded through BPP_RCCL_LIB by tests/test_gpu_round5.py): just enough of the RCCL API for the
// engine's deadline handling to be driven on a box with one GPU.  An all_gather here never completes by itself -- its kernel
// spins on a host-mapped flag, as RCCL's does when a peer is missing -- until ncclCommAbort (or stub_release, "the owner aborts")
// raises the flag; every spin also ends on its own after a few seconds, so nothing here can hold the GPU.  The stub counts the
// aborts and destroys it sees: the test asserts that an ADOPTED communicator receives neither.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <string.h>

static volatile int *g_flag = nullptr;  // host-mapped
static int *g_flag_dev = nullptr;
static int g_aborts = 0, g_destroys = 0, g_gathers = 0;

__global__ void k_stub_spin(volatile int *flag, long long max_ticks) {
  const long long t0 = wall_clock64();
  while (*flag == 0 && wall_clock64() - t0 < max_ticks) __builtin_amdgcn_s_sleep(64);
}

static void ensure_flag() {
  if (g_flag) return;
  int *h = nullptr;
  (void)hipHostMalloc((void **)&h, sizeof(int), hipHostMallocMapped);
  *h = 0;
  (void)hipHostGetDevicePointer((void **)&g_flag_dev, h, 0);
  g_flag = h;
}

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
  memset(id, 0x5a, sizeof(*id));
  return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int, ncclUniqueId, int) {
  *comm = (ncclComm_t)(uintptr_t)0x5157;
  return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t) {
  g_destroys++;
  return ncclSuccess;
}
ncclResult_t ncclCommAbort(ncclComm_t) {
  ensure_flag();
  g_aborts++;
  *g_flag = 1;
  return ncclSuccess;
}
ncclResult_t ncclAllGather(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t stream) {
  ensure_flag();
  g_gathers++;
  hipLaunchKernelGGL(k_stub_spin, dim3(1), dim3(1), 0, stream, (volatile int *)g_flag_dev, 300000000LL);  // <= 3 s at 100 MHz
  return ncclSuccess;
}
const char *ncclGetErrorString(ncclResult_t) { return "stub"; }

// the test's own view
void stub_counts(int *aborts, int *destroys, int *gathers) {
  *aborts = g_aborts;
  *destroys = g_destroys;
  *gathers = g_gathers;
}
void stub_release(int raised) {  // what the OWNER of an adopted communicator does when he aborts it; 0 re-arms the flag
  ensure_flag();
  *g_flag = raised;
}
}
