// Batched variable-time multiscalar multiplication on gfx950: signed-window Pippenger.
//
// Replaces (reference boundary): VartimePrecomputedMultiscalarMul::vartime_mixed_multiscalar_mul
// (src/range_proof.rs:1050-1057, :339-345) and VartimeMultiscalarMul::vartime_multiscalar_mul
// (:482-495, :512-521).  dalek uses Straus there; any algorithm yields the same group element, and the
// result only leaves the engine as a canonical encoding or an identity test.
//
// One launch sequence serves G independent MSMs ("groups").  A term is (scalar index, point index); group g owns
// terms [group_off[g], group_off[g+1]).  Pipeline:
//   k_msm_digits      scalar -> K signed c-bit digits; per-bucket histogram (LDS-aggregated atomics)
//   k_scan_exclusive  bucket start offsets
//   k_msm_scatter     counting sort: term ids (sign in bit 31) grouped by (group, window, |digit|)
//   k_msm_accumulate  one lane per bucket: sum of its points (mixed additions, 7M each)
//   k_msm_bitsum      Q[g][k][b] = sum of buckets whose digit has bit b set   (wave tree reduction in LDS)
//   k_msm_window      W[g][k]    = sum_b 2^b Q[g][k][b]
//   k_msm_final       R[g]       = sum_k 2^(c k) W[g][k]; canonical encoding + identity flag
#pragma once
#include "point.h"
#include "scalar.h"

namespace bpp {

struct MsmPlan {
  uint32_t c;        // window bits
  uint32_t K;        // windows
  uint32_t nb;       // buckets per window = 2^(c-1)
  uint32_t G;        // groups
  uint32_t n_terms;  // total terms
};

// point fetch: index < n_tab -> table A (generators), else table B (dynamic points of the batch)
struct PointTables {
  const niels *tab_a;
  const niels *tab_b;
  uint32_t n_a;
};
__device__ __forceinline__ const niels *point_ptr(const PointTables &t, uint32_t idx) {
  return idx < t.n_a ? (t.tab_a + idx) : (t.tab_b + (idx - t.n_a));
}

// ---- digits + histogram.  grid = (ceil(maxGroupTerms/256), G), block 256 ----
__global__ void __launch_bounds__(256) k_msm_digits(const sc *__restrict__ scalars, const uint32_t *__restrict__ term_sidx,
                                                    const uint32_t *__restrict__ group_off, MsmPlan plan,
                                                    int16_t *__restrict__ digits /* [n_terms][K] */,
                                                    uint32_t *__restrict__ counts /* [G][K][nb] */) {
  const uint32_t g = blockIdx.y;
  const uint32_t t0 = group_off[g], t1 = group_off[g + 1];
  const uint32_t term = t0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (term >= t1) return;
  const sc s = scalars[term_sidx[term]];
  const uint32_t c = plan.c, K = plan.K, nb = plan.nb;
  uint32_t carry = 0;
  uint32_t *cnt = counts + (size_t)g * K * nb;
  for (uint32_t k = 0; k < K; k++) {
    const uint32_t bit = k * c;
    const uint32_t wi = bit >> 5, sh = bit & 31;
    uint32_t raw = 0;
    if (wi < 8) {
      uint64_t two = (uint64_t)s.v[wi] | ((wi + 1 < 8) ? ((uint64_t)s.v[wi + 1] << 32) : 0ULL);
      raw = (uint32_t)(two >> sh) & ((1u << c) - 1u);
    }
    uint32_t v = raw + carry;
    int32_t dgt;
    if (v > nb) {  // nb = 2^(c-1): digits in (-2^(c-1), 2^(c-1)]
      dgt = (int32_t)v - (int32_t)(1u << c);
      carry = 1;
    } else {
      dgt = (int32_t)v;
      carry = 0;
    }
    digits[(size_t)term * K + k] = (int16_t)dgt;
    if (dgt != 0) {
      uint32_t mag = (uint32_t)(dgt < 0 ? -dgt : dgt);
      atomicAdd(&cnt[(size_t)k * nb + (mag - 1)], 1u);
    }
  }
}

// ---- exclusive scan of each group's bucket counts (one 1024-thread block per group); starts[] are absolute
// positions in sorted[]: group g owns sorted[group_off[g]*K, group_off[g+1]*K) ----
__global__ void __launch_bounds__(1024) k_scan_exclusive(const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
                                                         const uint32_t *__restrict__ group_off, MsmPlan plan) {
  __shared__ uint32_t part[1024];
  const uint32_t g = blockIdx.x, tid = threadIdx.x;
  const uint32_t n = plan.K * plan.nb;
  in += (size_t)g * n;
  out += (size_t)g * n;
  const uint32_t per = (n + 1023u) / 1024u;
  const uint32_t a = tid * per < n ? tid * per : n, b = (a + per < n) ? a + per : n;
  uint32_t s = 0;
  for (uint32_t i = a; i < b; i++) s += in[i];
  part[tid] = s;
  __syncthreads();
  for (uint32_t off = 1; off < 1024; off <<= 1) {
    uint32_t v = (tid >= off) ? part[tid - off] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  uint32_t run = group_off[g] * plan.K + ((tid == 0) ? 0 : part[tid - 1]);
  for (uint32_t i = a; i < b; i++) {
    uint32_t v = in[i];
    out[i] = run;
    run += v;
  }
}

// ---- counting-sort scatter.  grid = (ceil(maxGroupTerms/256), G) ----
__global__ void __launch_bounds__(256) k_msm_scatter(const int16_t *__restrict__ digits,
                                                     const uint32_t *__restrict__ group_off, MsmPlan plan,
                                                     const uint32_t *__restrict__ starts, uint32_t *__restrict__ cursor,
                                                     uint32_t *__restrict__ sorted) {
  const uint32_t g = blockIdx.y;
  const uint32_t t0 = group_off[g], t1 = group_off[g + 1];
  const uint32_t term = t0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (term >= t1) return;
  const uint32_t K = plan.K, nb = plan.nb;
  for (uint32_t k = 0; k < K; k++) {
    const int32_t dgt = digits[(size_t)term * K + k];
    if (dgt == 0) continue;
    const uint32_t mag = (uint32_t)(dgt < 0 ? -dgt : dgt);
    const size_t bucket = ((size_t)g * K + k) * nb + (mag - 1);
    const uint32_t pos = atomicAdd(&cursor[bucket], 1u);
    sorted[starts[bucket] + pos] = term | (dgt < 0 ? 0x80000000u : 0u);
  }
}

// ---- bucket sums: one lane per bucket ----
__global__ void __launch_bounds__(64) k_msm_accumulate(const uint32_t *__restrict__ sorted,
                                                       const uint32_t *__restrict__ starts,
                                                       const uint32_t *__restrict__ counts,
                                                       const uint32_t *__restrict__ term_pidx, PointTables tabs,
                                                       uint32_t n_buckets, ge *__restrict__ buckets) {
  const uint32_t bkt = blockIdx.x * blockDim.x + threadIdx.x;
  if (bkt >= n_buckets) return;
  const uint32_t a = starts[bkt], n = counts[bkt];
  ge acc;
  ge_identity(acc);
  for (uint32_t i = 0; i < n; i++) {
    const uint32_t e = sorted[a + i];
    niels q = *point_ptr(tabs, term_pidx[e & 0x7fffffffu]);
    niels_cneg(q, (e >> 31) != 0);  // branch-free: lanes of one wave mix additions and subtractions
    ge_madd(acc, acc, q);
  }
  buckets[bkt] = acc;
}

// ---- Q[g][k][b] = sum over buckets j (digit j+1) with bit b of (j+1) set.  grid = (c, K, G), block 64 ----
__global__ void __launch_bounds__(64) k_msm_bitsum(const ge *__restrict__ buckets, const uint32_t *__restrict__ counts,
                                                   MsmPlan plan, ge *__restrict__ Q /* [G][K][c] */) {
  const uint32_t b = blockIdx.x, k = blockIdx.y, g = blockIdx.z, lane = threadIdx.x;
  const uint32_t nb = plan.nb;
  const size_t base = ((size_t)g * plan.K + k) * nb;
  __shared__ ge red[64];
  ge acc;
  ge_identity(acc);
  for (uint32_t j = lane; j < nb; j += 64) {
    if ((((j + 1) >> b) & 1u) && counts[base + j] != 0) {
      const ge q = buckets[base + j];
      ge_add(acc, acc, q);
    }
  }
  red[lane] = acc;
  __syncthreads();
  for (uint32_t off = 32; off >= 1; off >>= 1) {
    if (lane < off) {
      ge x = red[lane], y2 = red[lane + off];
      ge_add(x, x, y2);
      red[lane] = x;
    }
    __syncthreads();
  }
  if (lane == 0) Q[((size_t)g * plan.K + k) * plan.c + b] = red[0];
}

// ---- W[g][k] = sum_b 2^b Q[g][k][b].  one lane per (g,k) ----
__global__ void __launch_bounds__(64) k_msm_window(const ge *__restrict__ Q, MsmPlan plan, ge *__restrict__ W) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= plan.G * plan.K) return;
  const ge *q = Q + (size_t)i * plan.c;
  ge acc = q[plan.c - 1];
  for (int b = (int)plan.c - 2; b >= 0; b--) {
    ge_dbl(acc, acc);
    const ge x = q[b];
    ge_add(acc, acc, x);
  }
  W[i] = acc;
}

// ---- R[g] = sum_k 2^(ck) W[g][k]; outputs: extended point, canonical encoding, identity flag ----
__global__ void __launch_bounds__(64) k_msm_final(const ge *__restrict__ W, MsmPlan plan, ge *__restrict__ R,
                                                  uint8_t *__restrict__ comp32, uint32_t *__restrict__ is_identity) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= plan.G) return;
  const ge *w = W + (size_t)g * plan.K;
  ge acc = w[plan.K - 1];
  for (int k = (int)plan.K - 2; k >= 0; k--) {
    ge_dbl_n(acc, acc, (int)plan.c);
    const ge x = w[k];
    ge_add(acc, acc, x);
  }
  R[g] = acc;
  uint8_t c32[32];
  ristretto_compress(c32, acc);
  for (int i = 0; i < 32; i++) comp32[(size_t)g * 32 + i] = c32[i];
  is_identity[g] = ge_is_ristretto_identity(acc) ? 1u : 0u;
}

// extended point -> 128 canonical bytes (X,Y,Z,T) and back, for the cross-GPU accumulator exchange
__global__ void k_ge_to_bytes(const ge *__restrict__ R, uint32_t n, uint8_t *__restrict__ out128) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const ge p = R[i];
  uint8_t b[32];
  fe_tobytes(b, p.X);
  for (int k = 0; k < 32; k++) out128[(size_t)i * 128 + k] = b[k];
  fe_tobytes(b, p.Y);
  for (int k = 0; k < 32; k++) out128[(size_t)i * 128 + 32 + k] = b[k];
  fe_tobytes(b, p.Z);
  for (int k = 0; k < 32; k++) out128[(size_t)i * 128 + 64 + k] = b[k];
  fe_tobytes(b, p.T);
  for (int k = 0; k < 32; k++) out128[(size_t)i * 128 + 96 + k] = b[k];
}

// sum of n accumulators (one lane; n = number of ranks) -> identity flag + encoding
__global__ void k_sum_accumulators(const uint8_t *__restrict__ in128, uint32_t n, uint8_t *__restrict__ comp32,
                                   uint32_t *__restrict__ is_identity) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  ge acc;
  ge_identity(acc);
  for (uint32_t i = 0; i < n; i++) {
    ge p;
    uint8_t b[32];
    for (int k = 0; k < 32; k++) b[k] = in128[(size_t)i * 128 + k];
    fe_frombytes(p.X, b);
    for (int k = 0; k < 32; k++) b[k] = in128[(size_t)i * 128 + 32 + k];
    fe_frombytes(p.Y, b);
    for (int k = 0; k < 32; k++) b[k] = in128[(size_t)i * 128 + 64 + k];
    fe_frombytes(p.Z, b);
    for (int k = 0; k < 32; k++) b[k] = in128[(size_t)i * 128 + 96 + k];
    fe_frombytes(p.T, b);
    ge_add(acc, acc, p);
  }
  uint8_t c32[32];
  ristretto_compress(c32, acc);
  for (int i = 0; i < 32; i++) comp32[i] = c32[i];
  is_identity[0] = ge_is_ristretto_identity(acc) ? 1u : 0u;
}

}  // namespace bpp
