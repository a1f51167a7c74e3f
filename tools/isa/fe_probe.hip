// Instruction-count probe: one kernel per field / point primitive, so that `hipcc -S` shows what each costs on gfx950.
// Usage: tools/isa/count.sh   (prints VALU instruction counts per kernel; no GPU needed)
#include <hip/hip_runtime.h>
#include "../../bulletproofs-plus_amd/csrc/point.h"
using namespace bpp;
extern "C" {
__global__ void probe_fe_mul(const fe *a, const fe *b, fe *o) { fe x = a[threadIdx.x], y = b[threadIdx.x], z; fe_mul(z, x, y); o[threadIdx.x] = z; }
__global__ void probe_fe_sq(const fe *a, fe *o) { fe x = a[threadIdx.x], z; fe_sq(z, x); o[threadIdx.x] = z; }
__global__ void probe_fe_sq2(const fe *a, fe *o) { fe x = a[threadIdx.x], z; fe_sq(z, x); fe_sq(z, z); o[threadIdx.x] = z; }
__global__ void probe_fe_sub(const fe *a, const fe *b, fe *o) { fe x = a[threadIdx.x], y = b[threadIdx.x], z; fe_sub(z, x, y); o[threadIdx.x] = z; }
__global__ void probe_fe_carry(const fe *a, fe *o) { fe x = a[threadIdx.x]; fe_carry(x); o[threadIdx.x] = x; }
__global__ void probe_ge_madd(const ge *a, const niels *b, ge *o) { ge x = a[threadIdx.x]; niels y = b[threadIdx.x]; ge_madd(x, x, y); o[threadIdx.x] = x; }
__global__ void probe_cneg_madd(const ge *a, const niels *b, const uint32_t *s, ge *o) { ge x = a[threadIdx.x]; niels y = b[threadIdx.x]; niels_cneg(y, s[threadIdx.x] != 0); ge_madd(x, x, y); o[threadIdx.x] = x; }
__global__ void probe_ge_add(const ge *a, const ge *b, ge *o) { ge x = a[threadIdx.x], y = b[threadIdx.x]; ge_add(x, x, y); o[threadIdx.x] = x; }
__global__ void probe_ge_dbl(const ge *a, ge *o) { ge x = a[threadIdx.x]; ge_dbl(x, x); o[threadIdx.x] = x; }
}
extern "C" __global__ void probe_swapped_madd(const ge *a, const niels *b, const uint32_t *s, ge *o) { ge x = a[threadIdx.x]; niels y; const bool neg = s[threadIdx.x] != 0; niels_load_swapped(y, b + threadIdx.x, neg); ge_madd_swapped(x, x, y, neg); o[threadIdx.x] = x; }
