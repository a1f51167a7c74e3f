// Instruction-count probe for the scalar (mod l) primitives; usage: tools/isa/count.sh sc_probe.hip
#include <hip/hip_runtime.h>
#include "../../bulletproofs-plus_amd/csrc/scalar.h"
using namespace bpp;
extern "C" {
__global__ void probe_sc_montmul(const sc *a, const sc *b, sc *o) { sc x = a[threadIdx.x], y = b[threadIdx.x], z; sc_montmul(z, x, y); o[threadIdx.x] = z; }
__global__ void probe_sc9_montmul(const sc9 *a, const sc9 *b, sc *o) { sc9 x = a[threadIdx.x], y = b[threadIdx.x]; sc z; sc9_montmul(z, x, y); o[threadIdx.x] = z; }
__global__ void probe_sc9_lazy(const sc9 *a, const sc9 *b, sc9 *o) { sc9 x = a[threadIdx.x], y = b[threadIdx.x], z; sc9_montmul_lazy(z, x, y); o[threadIdx.x] = z; }
__global__ void probe_sc9_from(const sc *a, sc9 *o) { sc x = a[threadIdx.x]; sc9 z; sc9_from(z, x); o[threadIdx.x] = z; }
__global__ void probe_sc_add(const sc *a, const sc *b, sc *o) { sc x = a[threadIdx.x], y = b[threadIdx.x], z; sc_add(z, x, y); o[threadIdx.x] = z; }
__global__ void probe_sc_from_mont(const sc *a, sc *o) { sc x = a[threadIdx.x], z; sc_from_mont(z, x); o[threadIdx.x] = z; }
}
