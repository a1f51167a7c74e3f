// Batched variable-time multiscalar multiplication on gfx950: signed-window Pippenger.
//
// Replaces (reference boundary): VartimePrecomputedMultiscalarMul::vartime_mixed_multiscalar_mul
// (src/range_proof.rs:1050-1057, :339-345) and VartimeMultiscalarMul::vartime_multiscalar_mul
// (:482-495, :512-521).  dalek uses Straus there; any algorithm yields the same group element, and the
// result only leaves the engine as a canonical encoding or an identity test.
//
// One launch sequence serves G independent MSMs ("groups").  A term is (scalar index, point index); group g owns
// terms [group_off[g], group_off[g+1]).  Pipeline:
//   k_msm_prelude     one workgroup per (group, window): digits of that window, LDS-staged counting sort (bucket histogram +
//                     offsets in LDS, term ids with the sign in bit 31 grouped by |digit|; no global atomics), buckets
//                     ordered by size
//   k_msm_accumulate  one lane per bucket: sum of its points (mixed additions, 7M each)
//   k_msm_bitsum      Q[g][k][b] = sum of buckets whose digit has bit b set   (wave tree reduction in LDS)
//   k_msm_window      W[g][k]    = sum_b 2^b Q[g][k][b]
//   k_msm_final       R[g]       = sum_k 2^(c k) W[g][k]; canonical encoding + identity flag
#pragma once
#include "point.h"
#include "recode.h"
#include "scalar.h"

namespace bpp {

// point fetch: index < n_tab -> table A (generators), else table B (dynamic points of the batch)
struct PointTables {
  const niels *tab_a;
  const niels *tab_b;
  uint32_t n_a;
  // half-scalar plans only (MsmPlan::split): the 2^126 multiples -- of the generators as affine-Niels entries (built once per
  // parameter set), of the batch's dynamic points as PROJECTIVE Niels entries (k_split_shift_quad, every verification)
  const niels *tab_a_hi;
  const struct pniels *tab_b_hi;
};
// (Y+X, Y-X, 2dT, 2Z): a point that is not normalised to Z = 1, in the form the mixed addition consumes.  An affine-Niels
// entry is the special case 2Z = 2, so ONE addition routine (and one instruction stream: no divergence inside a
// wavefront) serves table lines and the 2^126 multiples of a batch's own points, which exist in projective form only
// (normalising them would cost an inversion each).
struct pniels {
  fe yplusx, yminusx, xy2d, z2;
};
__device__ const uint32_t BPP_FE_TWO[10] = {2, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // the 2Z of an affine table line
__device__ __forceinline__ const niels *point_ptr(const PointTables &t, uint32_t idx) {
  return idx < t.n_a ? (t.tab_a + idx) : (t.tab_b + (idx - t.n_a));
}

// (Non-temporal loads of the sorted lists and non-temporal 16-byte stores of the buckets were measured: FETCH_SIZE of
// this kernel went from 389 to 1039 MB per launch, WRITE_SIZE from 240 to 595 MB, its duration from 0.84 to 0.98 ms.)

// ---- The MSM's prelude in ONE launch: digits, counting sort and size ordering of one window of one group per workgroup.
// (Until round 3 these were four dependent launches -- k_msm_digits, k_msm_sort, k_order_hist, k_order_scatter -- plus a
// memset, each queueing behind the other steps' 15 000-wavefront kernels: 1.3 ms of summed in-flight time for 0.3 ms of
// work.)  Workgroup (g, k):
//   pass 1  every term's digit of window k straight from its scalar (msm_digit_at: the two words around the window, the
//           carry decided by the window below), kept in LDS as int16 for pass 2 (as many as fit), histogram by |digit| with
//           LDS atomics
//   scan    counts -> offsets (in LDS); counts[] / starts[] of the window's 2^(c-1) buckets
//   pass 2  term ids (sign in bit 31) scattered into sorted[]: region of (g, k) = [goff[g]*K + k*ng, +ng)
//   order   this window's buckets by descending size (counting sort over the counts clamped to 255) into order_win[], with
//           the class histogram in cls_hist[]; k_msm_order merges the K window orders into the group's
// XCD-aware mapping (blockIdx % 8 = XCD, as in k_msm_accumulate): all windows of group g run on XCD g % 8, so the group's
// scalars are fetched into ONE L2 once and the lists written here are in the L2 that k_msm_accumulate reads them from.
// grid = 8 * ceil(G/8) * K workgroups of T lanes.  Dynamic LDS: 2 * nb u32 + dig_cap int16.
// T: 1024 lanes per (group, window); 256 when the groups are small (the reference's own 256-proof batches: 4 k terms per
// group, 256 such groups x 29 windows per 65 536 proofs).  Sixteen-wavefront workgroups with four terms per lane mostly wait
// for a CU with sixteen free wavefront slots: with four-wavefront ones the stage shrinks from 5.6 to 1.8 ms of a step in
// flight and the step gains 4 % (128-proof batches 7 %, 512: 2 %; at 1024 proofs per batch the wide form is 1-2 % ahead).
#ifndef BPP_SORT_THREADS
#define BPP_SORT_THREADS 1024
#endif
#define BPP_SORT_THREADS_SMALL 256
#define BPP_SORT_SMALL_GROUP_TERMS 9000u
// digit of window k of the term whose term_sidx entry is `si`: the whole scalar, or -- half-scalar plan -- the half the
// entry's top bit selects
__device__ __forceinline__ int32_t msm_term_digit(const sc *__restrict__ scalars, uint32_t si, const MsmPlan &plan, uint32_t k) {
  if (!plan.split) return msm_digit_at(scalars[si].v, plan, k);
  const sc s = scalars[si & ~BPP_TERM_HI];
  uint32_t h[8];
  msm_half_words(h, s.v, (si & BPP_TERM_HI) != 0);
  return msm_digit_at(h, plan, k);
}
template <uint32_t T>
__global__ void __launch_bounds__(T) k_msm_prelude(const sc *__restrict__ scalars, const uint32_t *__restrict__ term_sidx,
                                                   const uint32_t *__restrict__ term_pidx, const uint32_t *__restrict__ group_off,
                                                   MsmPlan plan, uint32_t dig_cap, uint32_t *__restrict__ counts,
                                                   uint32_t *__restrict__ starts, uint32_t *__restrict__ sorted,
                                                   uint32_t *__restrict__ order_win, uint32_t *__restrict__ cls_hist) {
  extern __shared__ uint32_t lds[];
  const uint32_t tid = threadIdx.x, nb = plan.nb, K = plan.K;
  const uint32_t xcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
  const uint32_t g = xcd + 8u * (j / K), k = j % K;
  if (g >= plan.G) return;
  uint32_t *hist = lds, *cur = lds + nb;
  int16_t *dcache = (int16_t *)(lds + 2 * nb);
  __shared__ uint32_t part[T];
  __shared__ uint32_t cls_n[256], cls_start[256], cls_cur[256];
  const uint32_t t0 = group_off[g], ng = group_off[g + 1] - t0;
  const uint32_t region = t0 * K + k * ng;
  for (uint32_t q = tid; q < nb; q += T) hist[q] = 0;
  if (tid < 256) cls_n[tid] = cls_cur[tid] = 0;
  __syncthreads();
  // pass 1: eight scalars in flight per lane before the first atomic
  for (uint32_t i0 = tid; i0 < ng; i0 += 8 * T) {
    uint32_t si[8];
    int32_t d[8];
#pragma unroll
    for (int u = 0; u < 8; u++) si[u] = (i0 + u * T < ng) ? term_sidx[t0 + i0 + u * T] : 0xffffffffu;
#pragma unroll
    for (int u = 0; u < 8; u++) d[u] = si[u] != 0xffffffffu ? msm_term_digit(scalars, si[u], plan, k) : 0;
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const uint32_t i = i0 + u * T;
      if (i < ng) {
        if (i < dig_cap) dcache[i] = (int16_t)d[u];
        if (d[u]) atomicAdd(&hist[(uint32_t)(d[u] < 0 ? -d[u] : d[u]) - 1], 1u);
      }
    }
  }
  __syncthreads();
  // exclusive scan of hist[0..nb): thread t owns a contiguous run of `per` bins
  const uint32_t per = (nb + T - 1u) / T;
  const uint32_t a = tid * per < nb ? tid * per : nb, b = (a + per < nb) ? a + per : nb;
  uint32_t sum = 0;
  for (uint32_t q = a; q < b; q++) sum += hist[q];
  part[tid] = sum;
  __syncthreads();
  for (uint32_t off = 1; off < T; off <<= 1) {
    uint32_t v = (tid >= off) ? part[tid - off] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  uint32_t run = (tid == 0) ? 0 : part[tid - 1];
  const size_t bbase = ((size_t)g * K + k) * nb;
  for (uint32_t q = a; q < b; q++) {
    const uint32_t cnt = hist[q];
    counts[bbase + q] = cnt;
    starts[bbase + q] = region + run;
    cur[q] = run;
    run += cnt;
    atomicAdd(&cls_n[cnt < 255u ? cnt : 255u], 1u);  // size classes of this window's buckets
  }
  __syncthreads();
  // pass 2: scatter
  for (uint32_t i0 = tid; i0 < ng; i0 += 8 * T) {
    int32_t d[8];
    uint32_t pi[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const uint32_t i = i0 + u * T;
      const bool in = i < ng;
      pi[u] = in ? term_pidx[t0 + i] : 0u;
      d[u] = !in ? 0 : (i < dig_cap ? (int32_t)dcache[i] : msm_term_digit(scalars, term_sidx[t0 + i], plan, k));
    }
#pragma unroll
    for (int u = 0; u < 8; u++)
      if (d[u]) {
        const uint32_t pos = atomicAdd(&cur[(uint32_t)(d[u] < 0 ? -d[u] : d[u]) - 1], 1u);
        sorted[region + pos] = pi[u] | (d[u] < 0 ? 0x80000000u : 0u);  // point index, sign in bit 31
      }
  }
  // order, step 1: this window's buckets by descending size class into order_win[] (positions local to the window);
  // cls_hist[g][k] = [how many buckets per class | where each class starts in this window's order]
  uint32_t *gh = cls_hist + (size_t)g * K * 768;  // per window: 256 counts, 256 class starts, 256 offsets (step 2)
  if (tid < 256) part[tid] = cls_n[tid];
  __syncthreads();
  for (uint32_t off = 1; off < 256; off <<= 1) {  // suffix sums over the classes: part[c] = sum_{c2 >= c} n[c2]
    uint32_t v = (tid < 256 && tid + off < 256) ? part[tid + off] : 0;
    __syncthreads();
    if (tid < 256) part[tid] += v;
    __syncthreads();
  }
  if (tid < 256) {
    cls_start[tid] = part[tid] - cls_n[tid];
    gh[k * 768 + tid] = cls_n[tid];
    gh[k * 768 + 256 + tid] = cls_start[tid];
  }
  __syncthreads();
  for (uint32_t q = a; q < b; q++) {
    const uint32_t cnt = hist[q], cls = cnt < 255u ? cnt : 255u;
    const uint32_t pos = cls_start[cls] + atomicAdd(&cls_cur[cls], 1u);
    order_win[bbase + pos] = (uint32_t)(bbase + q);
  }
}

// ---- order, step 2: one workgroup per group merges the K per-window orders into the group's order: all buckets of the
// group by descending size class, so that the 64 lanes of an accumulation wavefront get equally long lists and the kernel's
// last wavefronts the shortest ones (windows of c - 1 bits hold twice as many terms per bucket as the wide ones, so equal
// positions of different windows are NOT equal sizes; ordering per window only was measured: the outliers of every window
// spread over the grid, k_msm_accumulate 0.82 -> 0.96 ms alone, + 5 % instructions).  A bucket keeps its position inside
// its (window, class) run: group slot = local position + offs[window][class] -- a pure gather, no atomics.
// (A launch of its own on purpose.  Done by the last workgroup of k_msm_prelude to finish, it needs device-scope fences
// between workgroups, and on this chip those write back and invalidate a whole XCD's L2 each time: every kernel in flight
// lost its cached lines, the step went from 2.5 to 3.8 ms.)  grid = G workgroups of 1024.
template <uint32_t T>
__global__ void __launch_bounds__(T) k_msm_order(const uint32_t *__restrict__ counts, const uint32_t *__restrict__ order_win,
                                                 uint32_t *cls_hist, MsmPlan plan, uint32_t *__restrict__ order) {
  const uint32_t tid = threadIdx.x, g = blockIdx.x, nb = plan.nb, K = plan.K;
  __shared__ uint32_t part[256];
  uint32_t *gh = cls_hist + (size_t)g * K * 768;
  const size_t gbase = (size_t)g * K * nb;
  uint32_t tot = 0;
  if (tid < 256) {
    for (uint32_t kk = 0; kk < K; kk++) tot += gh[kk * 768 + tid];
    part[tid] = tot;
  }
  __syncthreads();
  for (uint32_t off = 1; off < 256; off <<= 1) {  // suffix sums over the classes
    uint32_t v = (tid < 256 && tid + off < 256) ? part[tid + off] : 0;
    __syncthreads();
    if (tid < 256) part[tid] += v;
    __syncthreads();
  }
  if (tid < 256) {
    uint32_t running = part[tid] - tot;  // first group-level slot of class tid
    for (uint32_t kk = 0; kk < K; kk++) {
      gh[kk * 768 + 512 + tid] = running - gh[kk * 768 + 256 + tid];  // (may wrap: unsigned arithmetic, undone by + position)
      running += gh[kk * 768 + tid];
    }
  }
  __syncthreads();  // (the offsets are read back by the same workgroup: same CU, same L1, written through)
  const uint32_t n_all = K * nb;
  for (uint32_t i0 = tid; i0 < n_all; i0 += 4 * T) {
    uint32_t bk[4], cn[4];
#pragma unroll
    for (int u = 0; u < 4; u++) bk[u] = i0 + u * T < n_all ? order_win[gbase + i0 + u * T] : 0xffffffffu;
#pragma unroll
    for (int u = 0; u < 4; u++) cn[u] = bk[u] != 0xffffffffu ? counts[bk[u]] : 0u;
#pragma unroll
    for (int u = 0; u < 4; u++)
      if (bk[u] != 0xffffffffu) {
        const uint32_t i = i0 + u * T, kk = i >> (plan.c - 1u), q = i & (nb - 1u);
        order[gbase + (uint32_t)(q + gh[kk * 768 + 512 + (cn[u] < 255u ? cn[u] : 255u)])] = bk[u];
      }
  }
}
// dynamic LDS of k_msm_prelude: the two bucket tables + as many cached digits as keep the workgroup under 64 KB
inline uint32_t msm_prelude_dig_cap(const MsmPlan &plan, uint32_t max_group_terms, uint32_t threads = BPP_SORT_THREADS) {
  const uint32_t fixed = 2u * plan.nb * 4u + (threads + 3u * 256u) * 4u;
  const uint32_t room = fixed + 1024u < 65536u ? (65536u - 1024u - fixed) / 2u : 0u;
  return max_group_terms < room ? max_group_terms : room;
}
inline size_t msm_prelude_lds(const MsmPlan &plan, uint32_t dig_cap) { return (size_t)2 * plan.nb * 4 + (size_t)((dig_cap + 1u) & ~1u) * 2; }

// ---- bucket sums: one lane per bucket.  XCD-aware work mapping: workgroups are dealt round-robin over the 8 XCDs
// (blockIdx % 8), so XCD x takes the groups x, x+8, x+16, ... one after the other; a group's buckets (in size order)
// are consecutive for that XCD and its point set stays resident in that XCD's 4 MB L2 instead of being re-fetched
// through the fabric by every wavefront.  grid = 8 * ceil(G/8) * ceil(per_group/64) blocks of 64. ----
// Measured alternatives that lose: a 4-waves-per-SIMD bound (128 VGPRs) spills; turning the first point into the
// accumulator with ge_from_niels (saves one of ~17 additions) costs 40-50 more VGPRs and a wavefront of occupancy.
#ifndef BPP_ACC_WAVES
#define BPP_ACC_WAVES 3  // wavefronts per SIMD the register allocation is bounded for
#endif
__global__ void __launch_bounds__(64, BPP_ACC_WAVES) k_msm_accumulate(const uint32_t *__restrict__ sorted,
                                                       const uint32_t *__restrict__ starts,
                                                       const uint32_t *__restrict__ counts,
                                                       const uint32_t *__restrict__ order, PointTables tabs,
                                                       uint32_t per_group, uint32_t G, ge *__restrict__ buckets) {
  const uint32_t xcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
  const uint32_t bpg = (per_group + 63u) / 64u;  // blocks per group
  const uint32_t g = xcd + 8u * (j / bpg);
  const uint32_t slot = (j % bpg) * 64u + threadIdx.x;
  if (g >= G || slot >= per_group) return;
  const uint32_t bkt = order[(size_t)g * per_group + slot];
  const uint32_t a = starts[bkt], n = counts[bkt];
  if (n == 0) return;  // empty buckets are skipped by the reduction (counts[] == 0)
  // software pipeline: the next entry's index and point are in flight while the current addition runs.  The sign of a
  // term is applied while loading (y+x / y-x exchanged by address) and inside ge_madd_swapped (d - c / d + c exchanged):
  // branch-free, lanes of one wavefront mix additions and subtractions.  The first term becomes the accumulator with
  // one product (ge_from_niels_first) instead of being added to the identity with seven.
  uint32_t e = sorted[a];
  niels q;
  niels_load_swapped(q, point_ptr(tabs, e & 0x7fffffffu), (e >> 31) != 0);
  ge acc;
  ge_from_niels_first(acc, q);
  fe_fence(acc.X);  // the prologue is kept apart from the loop: interleaved with it, it costs 30 more live registers
  fe_fence(acc.Y);  // (a wavefront of occupancy) for one product per bucket
  fe_fence(acc.T);
  if (n > 1) {
    e = sorted[a + 1];
    niels_load_swapped(q, point_ptr(tabs, e & 0x7fffffffu), (e >> 31) != 0);
  }
  // two additions per iteration, the entries alternating between two register sets: with one set the prefetched entry has to
  // be copied into the "current" one every time (30 moves per addition)
  uint32_t i = 1;
  while (i < n) {
    uint32_t e2 = e;
    niels q2;
    if (i + 1 < n) {
      e2 = sorted[a + i + 1];
      niels_load_swapped(q2, point_ptr(tabs, e2 & 0x7fffffffu), (e2 >> 31) != 0);
    }
    ge_madd_swapped(acc, acc, q, (e >> 31) != 0);
    if (++i >= n) break;
    if (i + 1 < n) {
      e = sorted[a + i + 1];
      niels_load_swapped(q, point_ptr(tabs, e & 0x7fffffffu), (e >> 31) != 0);
    }
    ge_madd_swapped(acc, acc, q2, (e2 >> 31) != 0);
    ++i;
  }
  buckets[bkt] = acc;
}

// ---- W[g][k] = sum_j j * B_j over the nb = A * Bc buckets of one window, one wavefront per (g, k).
// With j = Bc*a + b (a in [0,A), b in [1,Bc]):  W = Bc * sum_a a*R_a + sum_b b*C_b  where R_a / C_b are the row / column
// sums of the A x Bc bucket grid.  Lanes 0..A-1 build row sums, lanes 32..32+Bc-1 column sums (<= 32 sequential adds),
// both weighted sums come from a parallel suffix scan + tree sum through LDS (10 add latencies).  ~2.1 adds per bucket
// instead of (c/2) for the bit-plane method.  Requires A, Bc <= 32 (c <= 11). ----
__global__ void __launch_bounds__(64) k_msm_window_rc(const ge *__restrict__ buckets, const uint32_t *__restrict__ counts,
                                                      MsmPlan plan, ge *__restrict__ W) {
  const uint32_t gk = blockIdx.x, lane = threadIdx.x;
  const uint32_t nb = plan.nb, lb = plan.c - 1;
  const uint32_t lBc = lb / 2, Bc = 1u << lBc, A = nb >> lBc;  // A >= Bc
  const size_t base = (size_t)gk * nb;
  __shared__ ge red[64];
  ge acc;
  ge_identity(acc);
  const bool is_row = lane < 32;
  const uint32_t idx = is_row ? lane : lane - 32;
  if (is_row ? (idx < A) : (idx < Bc)) {
    const uint32_t cnt = is_row ? Bc : A;
    for (uint32_t q = 0; q < cnt; q++) {
      // bucket index j-1 with j = Bc*a + b, b in [1, Bc]
      const uint32_t j0 = is_row ? (Bc * idx + q) : (Bc * q + idx);
      if (counts[base + j0]) {
        const ge x = buckets[base + j0];
        ge_add(acc, acc, x);
      }
    }
  }
  // suffix scan within each half: x[i] = sum_{i' >= i} x[i']
  red[lane] = acc;
  __syncthreads();
  const uint32_t half_n = is_row ? A : Bc;
  for (uint32_t off = 1; off < 32; off <<= 1) {
    ge y2;
    const bool act = idx + off < half_n;
    if (act) y2 = red[lane + off];
    __syncthreads();
    if (act) {
      ge_add(acc, acc, y2);
      red[lane] = acc;
    }
    __syncthreads();
  }
  // rows: sum_a a*R_a = sum_{i>=1} suffix_i ; columns (b = idx+1): sum_b b*C_b = sum_{i>=0} suffix_i
  if (is_row ? (idx == 0 || idx >= A) : (idx >= Bc)) ge_identity(acc);
  red[lane] = acc;
  __syncthreads();
  for (uint32_t off = 16; off >= 1; off >>= 1) {
    if (idx < off) {
      ge x = red[lane], y2 = red[lane + off];
      ge_add(x, x, y2);
      red[lane] = x;
    }
    __syncthreads();
  }
  if (lane == 0) {
    ge rows = red[0];
    const ge cols = red[32];
    if (lBc) ge_dbl_n(rows, rows, (int)lBc);
    ge_add(rows, rows, cols);
    W[gk] = rows;
  }
}

// The same for windows of at most 256 buckets (A, Bc <= 16: the many small groups of a throughput call, e.g. the reference's
// own 256-proof batches at 9-bit windows): TWO windows per wavefront, 16 lanes per row / column set instead of 32 with half of
// them idle -- half the wavefronts and half the issued instructions of the stage (a fifth of such a step's VALU work).
// lanes 0-15 rows of window 2 b, 16-31 rows of window 2 b + 1, 32-47 / 48-63 their columns.  grid = ceil(G K / 2) blocks of 64.
__global__ void __launch_bounds__(64) k_msm_window_rc2(const ge *__restrict__ buckets, const uint32_t *__restrict__ counts, MsmPlan plan,
                                                       uint32_t n_windows, ge *__restrict__ W) {
  const uint32_t lane = threadIdx.x;
  const uint32_t nb = plan.nb, lb = plan.c - 1;
  const uint32_t lBc = lb / 2, Bc = 1u << lBc, A = nb >> lBc;  // Bc <= A <= 16
  const bool is_row = lane < 32;
  const uint32_t sub = (lane >> 4) & 1u, idx = lane & 15u;
  const uint32_t gk = blockIdx.x * 2u + sub;
  const bool live = gk < n_windows;
  const size_t base = (size_t)gk * nb;
  __shared__ ge red[64];
  ge acc;
  ge_identity(acc);
  if (live && (is_row ? (idx < A) : (idx < Bc))) {
    const uint32_t cnt = is_row ? Bc : A;
    for (uint32_t q = 0; q < cnt; q++) {
      const uint32_t j0 = is_row ? (Bc * idx + q) : (Bc * q + idx);  // bucket index j-1 with j = Bc*a + b, b in [1, Bc]
      if (counts[base + j0]) {
        const ge x = buckets[base + j0];
        ge_add(acc, acc, x);
      }
    }
  }
  red[lane] = acc;
  __syncthreads();
  const uint32_t half_n = is_row ? A : Bc;
  for (uint32_t off = 1; off < 16; off <<= 1) {  // suffix scan inside each 16-lane set
    ge y2;
    const bool act = idx + off < half_n;
    if (act) y2 = red[lane + off];
    __syncthreads();
    if (act) {
      ge_add(acc, acc, y2);
      red[lane] = acc;
    }
    __syncthreads();
  }
  if (is_row ? (idx == 0 || idx >= A) : (idx >= Bc)) ge_identity(acc);
  red[lane] = acc;
  __syncthreads();
  for (uint32_t off = 8; off >= 1; off >>= 1) {
    if (idx < off) {
      ge x = red[lane], y2 = red[lane + off];
      ge_add(x, x, y2);
      red[lane] = x;
    }
    __syncthreads();
  }
  if (is_row && idx == 0 && live) {
    ge rows = red[lane];
    const ge cols = red[lane + 32];
    if (lBc) ge_dbl_n(rows, rows, (int)lBc);
    ge_add(rows, rows, cols);
    W[gk] = rows;
  }
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
                                                  uint32_t *__restrict__ is_identity) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= plan.G) return;
  const ge *w = W + (size_t)g * plan.K;
  ge acc = w[plan.K - 1];
  for (int k = (int)plan.K - 2; k >= 0; k--) {
    ge_dbl_n(acc, acc, (uint32_t)k < plan.K_wide ? (int)plan.c : (int)plan.c - 1);  // width of window k
    const ge x = w[k];
    ge_add(acc, acc, x);
  }
  R[g] = acc;  // the 32-byte encoding is produced on demand (k_compress_ge): verification only needs the identity test
  is_identity[g] = ge_is_ristretto_identity(acc) ? 1u : 0u;
}

// ---- The same final step with FOUR lanes per group (one quad).  The Horner recurrence is 253 sequential doublings: on
// one lane that is ~1000 dependent instructions per doubling and 0.7 ms whatever the batch size.  A doubling is four
// independent squarings (X^2, Y^2, Z^2, (X+Y)^2) followed by four independent products (E*F, G*H, F*G, E*H), and so is
// an addition (two rounds of four products): each lane of the quad does one of them and the operands travel by
// quad-permute DPP moves (no LDS).  Lane q keeps coordinate q of the accumulator (X, Y, Z, T).  Same field operations in
// the same order as ge_dbl_n / ge_add, so the result is bit-identical.  grid = ceil(G / 16) blocks of 64 lanes. ----
// The broadcast results are fenced (an empty asm "defines" them): otherwise the compiler's DPP combine may fold a broadcast
// into the consumer, and for `x - broadcast(y)` it emits v_subrev_u32_dpp, which on this toolchain / chip computes
// broadcast(x) - y: `v_subrev_u32_dpp d, A, B quad_perm:[3,3,3,3]` returns B[lane 3] - A[lane] instead of B[lane] - A[lane 3]
// (v_sub_u32_dpp and v_add_u32_dpp behave as documented; tools/isa/dpp_hazard_check.py refuses the opcode in any build).
// The ten moves and ONE trailing wait state are a single asm statement: the compiler can neither fold a broadcast into its
// consumer (the subrev problem above) nor put a store of the last result right behind it (the stale-store-data problem:
// a DPP result read as store DATA by the very next instruction; every earlier result already has the following move as its
// wait state).  Correctness then does not depend on what the scheduler or the register allocator happen to do; the
// build-time disassembly check (tools/isa/dpp_hazard_check.py) stays as the net under compiler-generated DPP.
template <int S>
__device__ __forceinline__ void quad_bcast(fe &r, const fe &v) {
  static_assert(S >= 0 && S < 4, "quad position");
#define BPP_QB(i) "v_mov_b32_dpp %" #i ", %1" #i " quad_perm:[%20,%20,%20,%20] row_mask:0xf bank_mask:0xf\n\t"
  asm volatile(BPP_QB(0) BPP_QB(1) BPP_QB(2) BPP_QB(3) BPP_QB(4) BPP_QB(5) BPP_QB(6) BPP_QB(7) BPP_QB(8) BPP_QB(9) "s_nop 0"
               : "=&v"(r.v[0]), "=&v"(r.v[1]), "=&v"(r.v[2]), "=&v"(r.v[3]), "=&v"(r.v[4]), "=&v"(r.v[5]), "=&v"(r.v[6]), "=&v"(r.v[7]),
                 "=&v"(r.v[8]), "=&v"(r.v[9])
               : "v"(v.v[0]), "v"(v.v[1]), "v"(v.v[2]), "v"(v.v[3]), "v"(v.v[4]), "v"(v.v[5]), "v"(v.v[6]), "v"(v.v[7]), "v"(v.v[8]),
                 "v"(v.v[9]), "n"(S));
#undef BPP_QB
}
// per-lane choice among four field elements: three v_cndmask per limb under wavefront-wide lane masks held in scalar
// registers (ternaries get turned into divergent branches with the multiplication duplicated in every arm, which serialises
// the four lanes again; and/or under per-lane masks cost seven instructions per limb)
struct QuadMask {
  uint64_t b1, b2, b3;  // lanes whose quad position is 1, 2, 3
};
__device__ __forceinline__ QuadMask quad_mask(uint32_t q) {
  (void)q;  // position in the quad = lane & 3 for every caller (64-lane blocks, quads of adjacent lanes)
  QuadMask k;
  k.b1 = 0x2222222222222222ull;
  k.b2 = 0x4444444444444444ull;
  k.b3 = 0x8888888888888888ull;
  asm volatile("" : "+s"(k.b1), "+s"(k.b2), "+s"(k.b3));  // opaque: keep them in scalar registers, no per-use materialising
  return k;
}
__device__ __forceinline__ void fe_sel4(fe &r, const QuadMask &k, const fe &a0, const fe &a1, const fe &a2, const fe &a3) {
#pragma unroll
  for (int i = 0; i < 10; i++) {
    uint32_t t;
    asm("v_cndmask_b32 %0, %1, %2, %5\n\tv_cndmask_b32 %0, %0, %3, %6\n\tv_cndmask_b32 %0, %0, %4, %7"
        : "=&v"(t)
        : "v"(a0.v[i]), "v"(a1.v[i]), "v"(a2.v[i]), "v"(a3.v[i]), "s"(k.b1), "s"(k.b2), "s"(k.b3));
    r.v[i] = t;
  }
}
// r = quad position 3 ? b : a
__device__ __forceinline__ void fe_sel_lane3(fe &r, const QuadMask &k, const fe &a, const fe &b) {
#pragma unroll
  for (int i = 0; i < 10; i++) {
    uint32_t t;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(t) : "v"(a.v[i]), "v"(b.v[i]), "s"(k.b3));
    r.v[i] = t;
  }
}
// second round shared by doubling and addition: lane q returns (E*F, G*H, F*G, E*H)[q]
__device__ __forceinline__ void quad_efgh(fe &m, const QuadMask &q, const fe &E, const fe &F, const fe &G, const fe &H) {
  fe l, r;
  fe_sel4(l, q, E, G, F, E);
  fe_sel4(r, q, F, H, G, H);
  fe_mul(m, l, r);
}

// acc += other, acc spread over the quad (lane q holds coordinate q in m): the nine products of ge_add in three rounds
__device__ __forceinline__ void quad_ge_add(fe &m, const QuadMask &q, const ge &other, const fe &d2, const fe &one) {
  fe X, Y, Z, T;
  quad_bcast<0>(X, m);
  quad_bcast<1>(Y, m);
  quad_bcast<2>(Z, m);
  quad_bcast<3>(T, m);
  fe l0, r0, l1, r1, l, r, s2;
  fe_sub(l0, Y, X);
  fe_sub(r0, other.Y, other.X);
  fe_add(l1, Y, X);
  fe_add(r1, other.Y, other.X);
  fe_sel4(l, q, l0, l1, T, Z);
  fe_sel4(r, q, r0, r1, other.T, other.Z);
  fe_mul(s2, l, r);  // a, b, T*T', Z*Z'
  fe_sel4(r, q, one, one, d2, one);
  fe_mul(s2, s2, r);  // lane 2: c = T*T'*2d; the others multiply by one
  fe a, b, c, d, e, f, gg, h;
  quad_bcast<0>(a, s2);
  quad_bcast<1>(b, s2);
  quad_bcast<2>(c, s2);
  quad_bcast<3>(d, s2);
  fe_add(d, d, d);
  fe_sub(e, b, a);
  fe_sub(f, d, c);
  fe_add(gg, d, c);
  fe_add(h, b, a);
  fe_carry(gg);
  quad_efgh(m, q, e, f, gg, h);
}
// acc += pt (affine Niels, already sign-adjusted): the seven products of ge_madd in two rounds
// (zz = 2 Z' of the entry: the constant 2 for an affine table line, pniels::z2 for a projective one)
__device__ __forceinline__ void quad_ge_madd(fe &m, const QuadMask &q, const niels &pt, const fe &zz) {
  fe X, Y, Z, T;
  quad_bcast<0>(X, m);
  quad_bcast<1>(Y, m);
  quad_bcast<2>(Z, m);
  quad_bcast<3>(T, m);
  fe l0, l1, l, r, s2;
  fe_add(l0, Y, X);
  fe_sub(l1, Y, X);
  fe_sel4(l, q, l0, l1, T, Z);
  fe_sel4(r, q, pt.yplusx, pt.yminusx, pt.xy2d, zz);  // lane 3: d = Z * 2Z'
  fe_mul(s2, l, r);  // a, b, c, d
  fe a, b, c, d, e, f, gg, h;
  quad_bcast<0>(a, s2);
  quad_bcast<1>(b, s2);
  quad_bcast<2>(c, s2);
  quad_bcast<3>(d, s2);
  fe_sub(e, a, b);
  fe_add(h, a, b);
  fe_add(gg, d, c);
  fe_sub(f, d, c);
  fe_carry(gg);
  quad_efgh(m, q, e, f, gg, h);  // X3 = e*f, Y3 = g*h, Z3 = f*g, T3 = e*h
}
// acc = 2 acc, acc spread over the quad: four squarings (X^2, Y^2, Z^2, (X+Y)^2), then the four products of quad_efgh
__device__ __forceinline__ void quad_ge_dbl(fe &m, const QuadMask &q) {
  fe x, y, v, s2;
  quad_bcast<0>(x, m);
  quad_bcast<1>(y, m);
  fe_add(v, x, y);
  fe_sel_lane3(v, q, m, v);  // X, Y, Z, X + Y
  fe_sq(s2, v);
  fe a, b, c, t, e, f, gg, h;
  quad_bcast<0>(a, s2);
  quad_bcast<1>(b, s2);
  quad_bcast<2>(c, s2);
  quad_bcast<3>(t, s2);
  // limb classes as in ge_dbl_efgh (point.h): e wide (left operand only), g and h loose, f carried: one carry pass
  fe_add(h, a, b);
  fe_sub_lazy(e, h, t);
  fe_sub_lazy(gg, a, b);
  fe_dbl_add(f, c, gg);
  fe_carry(f);
  quad_efgh(m, q, e, f, gg, h);
}
__device__ __forceinline__ void quad_load(fe &m, const QuadMask &q, const ge &p) { fe_sel4(m, q, p.X, p.Y, p.Z, p.T); }
__device__ __forceinline__ void quad_gather(ge &p, const fe &m) {
  quad_bcast<0>(p.X, m);
  quad_bcast<1>(p.Y, m);
  quad_bcast<2>(p.Z, m);
  quad_bcast<3>(p.T, m);
}

__global__ void __launch_bounds__(64) k_msm_final_quad(const ge *__restrict__ W, MsmPlan plan, ge *__restrict__ R,
                                                       uint32_t *__restrict__ is_identity) {
  const uint32_t lane = threadIdx.x, qi = lane & 3u;
  const QuadMask q = quad_mask(qi);
  const uint32_t g = blockIdx.x * 16u + (lane >> 2);
  const bool active = g < plan.G;
  const ge *w = W + (size_t)(active ? g : plan.G - 1) * plan.K;  // idle quads shadow the last group and write nothing
  fe m;
  {
    const ge top = w[plan.K - 1];
    fe_sel4(m, q, top.X, top.Y, top.Z, top.T);
  }
  fe d2, one;
  fe_const(d2, FE_D2);
  fe_1(one);
  for (int k = (int)plan.K - 2; k >= 0; k--) {
    const int n = (uint32_t)k < plan.K_wide ? (int)plan.c : (int)plan.c - 1;  // width of window k
#pragma unroll 1
    for (int i = 0; i < n; i++) quad_ge_dbl(m, q);
    // acc += W_k
    {
      const ge wk = w[k];
      quad_ge_add(m, q, wk, d2, one);
    }
  }
  ge acc;
  quad_bcast<0>(acc.X, m);
  quad_bcast<1>(acc.Y, m);
  quad_bcast<2>(acc.Z, m);
  quad_bcast<3>(acc.T, m);
  if (active && qi == 0) {
    R[g] = acc;
    is_identity[g] = ge_is_ristretto_identity(acc) ? 1u : 0u;
  }
}

// ---- Latency forms of the bucket accumulation and of the row/column bucket reduction for SMALL inputs (a single
// batch: a few thousand buckets on an otherwise idle chip): four lanes per bucket / per row-column accumulator, every
// addition split over the quad (quad_ge_madd / quad_ge_add).  Same arithmetic, ~3x shorter dependency chains. ----
__global__ void __launch_bounds__(64) k_msm_accumulate_quad(const uint32_t *__restrict__ sorted, const uint32_t *__restrict__ starts,
                                                            const uint32_t *__restrict__ counts, const uint32_t *__restrict__ order,
                                                            PointTables tabs, uint32_t n_buckets, ge *__restrict__ buckets) {
  const uint32_t lane = threadIdx.x, qi = lane & 3u;
  const QuadMask q = quad_mask(qi);
  const uint32_t slot = blockIdx.x * 16u + (lane >> 2);
  if (slot >= n_buckets) return;  // whole quads leave together
  const uint32_t bkt = order[slot];
  const uint32_t a = starts[bkt], n = counts[bkt];
  if (n == 0) return;
  fe m;
  {
    ge id;
    ge_identity(id);
    quad_load(m, q, id);
  }
  // An entry is (sign | high-multiple flag | point index).  Low entries and the generators' 2^126 multiples are affine-Niels
  // table lines (2Z = 2); the 2^126 multiples of the batch's own points are projective Niels entries with their own 2Z.
  // Both run through the same mixed addition.
  // Branch-free fetch: the three Niels fields sit at the same offsets in a table line and in a projective entry, and 2Z is
  // read either from the entry or from a constant in memory -- every load of the four entries of a trip is unconditional, so
  // all of them are in flight together (with a branch per entry kind the loads of one entry were waited for at the end of
  // its branch before the next entry's were issued).
  auto fetch = [&](uint32_t ent, niels &dn, fe &dz) {
    const uint32_t pi = ent & 0x3fffffffu;
    const bool hi = (ent & BPP_POINT_HI) != 0, proj = hi && pi >= tabs.n_a;
    const pniels *pp = tabs.tab_b_hi + (proj ? pi - tabs.n_a : 0u);
    const niels *np = hi ? (pi < tabs.n_a ? tabs.tab_a_hi + pi : (const niels *)pp) : point_ptr(tabs, pi);
    const fe *zp = proj ? &pp->z2 : (const fe *)BPP_FE_TWO;
    dn.yplusx = np->yplusx;
    dn.yminusx = np->yminusx;
    dn.xy2d = np->xy2d;
    dz = *zp;
  };
  // Four entries per trip: their list words are already in registers (loaded during the previous trip), their four points are
  // requested together and only then added one after the other.  With one entry in flight per trip (the first form) every
  // addition waited for two dependent memory round trips of a nearly idle chip (~3 us against ~1.2 us of arithmetic): the
  // kernel took 0.1 ms for the 8-deep lists of a 256-proof half-scalar call.
  uint32_t ew[4];
#pragma unroll
  for (int u = 0; u < 4; u++) ew[u] = (uint32_t)u < n ? sorted[a + u] : 0u;
  for (uint32_t i0 = 0; i0 < n; i0 += 4) {
    uint32_t ec[4];
    niels pt[4];
    fe pzz[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      ec[u] = ew[u];
      if (i0 + u < n) fetch(ec[u], pt[u], pzz[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; u++) ew[u] = i0 + 4 + u < n ? sorted[a + i0 + 4 + u] : 0u;
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (i0 + u < n) {
        niels_cneg(pt[u], (ec[u] >> 31) != 0);
        quad_ge_madd(m, q, pt[u], pzz[u]);
      }
    }
  }
  // lane q writes coordinate q
  fe *dst = (fe *)(buckets + bkt);
  dst[qi] = m;
}

// ---- 2^126 multiples for the half-scalar plan of small calls.  The final Horner step is 253 DEPENDENT doublings, 0.3 ms of
// a 0.65 ms call whatever the batch size; with s = s_lo + 2^126 s_hi the MSM runs over twice the terms, (s_lo, P) and
// (s_hi, 2^126 P), on 127-bit windows: the same number of additions, half the doublings.  2^126 P of a batch's own points is
// 126 doublings per point, but those depend on the decompression only and run beside PASS 1 and the scalar stage.
// One quad per point (the doubling's four squarings / products side by side, as in k_msm_final_quad): (a, b, .) = (y+x, y-x)
// gives the point as (a - b : a + b : 2); T is not an input of a doubling.  The result leaves as a projective Niels entry.
__global__ void __launch_bounds__(64) k_split_shift_quad(const niels *__restrict__ pts, uint32_t n, pniels *__restrict__ out) {
  const uint32_t lane = threadIdx.x, qi = lane & 3u;
  const QuadMask q = quad_mask(qi);
  const uint32_t i = blockIdx.x * 16u + (lane >> 2);
  if (i >= n) return;  // whole quads leave together
  const niels p = pts[i];
  ge g;
  fe_sub(g.X, p.yplusx, p.yminusx);
  fe_carry(g.X);
  fe_add(g.Y, p.yplusx, p.yminusx);
  fe_carry(g.Y);
  fe_1(g.Z);
  g.Z.v[0] = 2;
  fe_0(g.T);
  fe m;
  quad_load(m, q, g);
#pragma unroll 1
  for (uint32_t k = 0; k < BPP_MSM_SPLIT_BIT; k++) quad_ge_dbl(m, q);
  // (X, Y, Z, T) over the quad (lane q holds coordinate q) -> (Y + X, Y - X, 2d T, 2 Z): every lane stores one reduced
  // field element of the entry
  fe X, Y, Z, T, s0, s1, md, m2, d2, o;
  quad_bcast<0>(X, m);
  quad_bcast<1>(Y, m);
  quad_bcast<2>(Z, m);
  quad_bcast<3>(T, m);
  fe_add(s0, Y, X);
  fe_sub(s1, Y, X);
  fe_const(d2, FE_D2);
  fe_mul(md, T, d2);
  fe_add(m2, Z, Z);
  fe_sel4(o, q, s0, s1, md, m2);
  fe_carry(o);
  fe *dst = (fe *)(out + i);
  dst[qi] = o;
}
// the generators' 2^126 multiples as table entries: one lane per generator, once per parameter set
__global__ void __launch_bounds__(64) k_split_shift_table(const niels *__restrict__ tab, uint32_t n, niels *__restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  ge g;
  ge_identity(g);
  const niels p = tab[i];
  ge_madd(g, g, p);
  ge_dbl_n(g, g, (int)BPP_MSM_SPLIT_BIT);
  fe zi, x, y;
  fe_invert(zi, g.Z);
  fe_mul(x, g.X, zi);
  fe_mul(y, g.Y, zi);
  niels e;
  niels_from_affine(e, x, y);
  fe_carry(e.yminusx);
  out[i] = e;
}

// one workgroup of 1024 lanes per (group, window): 256 quads.  Quad (seg, i) adds a quarter of the buckets of row / column i
// of k_msm_window_rc's grid (8 dependent additions instead of 32 at 11 bits; this form only runs when the chip is nearly
// idle, so lanes are free and the chain is what counts), the four partial sums meet in a two-step tree, then quads
// (0, i) go on as lane i of k_msm_window_rc does; the closing doublings and addition are split over quad (0, 0).
#define BPP_RC_QUAD_SEGS 4u
__global__ void __launch_bounds__(1024) k_msm_window_rc_quad(const ge *__restrict__ buckets, const uint32_t *__restrict__ counts,
                                                             MsmPlan plan, ge *__restrict__ W) {
  const uint32_t gk = blockIdx.x, tid = threadIdx.x, qi = tid & 3u, quad = tid >> 2, lane = quad & 63u, seg = quad >> 6;
  const QuadMask q = quad_mask(qi);
  const uint32_t nb = plan.nb, lb = plan.c - 1;
  const uint32_t lBc = lb / 2, Bc = 1u << lBc, A = nb >> lBc;  // A >= Bc
  const size_t base = (size_t)gk * nb;
  __shared__ ge red[64 * BPP_RC_QUAD_SEGS];
  fe d2, one, m;
  fe_const(d2, FE_D2);
  fe_1(one);
  ge id;
  ge_identity(id);
  quad_load(m, q, id);
  const bool is_row = lane < 32;
  const uint32_t idx = is_row ? lane : lane - 32;
  if (is_row ? (idx < A) : (idx < Bc)) {
    const uint32_t cnt = is_row ? Bc : A;
    const uint32_t per = (cnt + BPP_RC_QUAD_SEGS - 1u) / BPP_RC_QUAD_SEGS;
    const uint32_t k0 = seg * per < cnt ? seg * per : cnt, k1 = k0 + per < cnt ? k0 + per : cnt;
    for (uint32_t k = k0; k < k1; k++) {
      const uint32_t j0 = is_row ? (Bc * idx + k) : (Bc * k + idx);
      if (counts[base + j0]) {
        const ge x = buckets[base + j0];
        quad_ge_add(m, q, x, d2, one);
      }
    }
  }
  fe *slot = (fe *)&red[quad];
  slot[qi] = m;
  __syncthreads();
  // partial sums of the segments: (0 += 1, 2 += 3), then 0 += 2
  for (uint32_t step = 1; step < BPP_RC_QUAD_SEGS; step <<= 1) {
    const bool act = (seg % (2u * step)) == 0 && seg + step < BPP_RC_QUAD_SEGS;
    if (act) {
      const ge y2 = red[quad + 64u * step];
      quad_ge_add(m, q, y2, d2, one);
    }
    __syncthreads();
    if (act) slot[qi] = m;
    __syncthreads();
  }
  const bool lead = seg == 0;  // from here on quads (0, i) only; the others keep the barriers company
  // suffix scan within each half: x[i] = sum_{i' >= i} x[i']
  const uint32_t half_n = is_row ? A : Bc;
  for (uint32_t off = 1; off < 32; off <<= 1) {
    ge y2;
    const bool act = lead && idx + off < half_n;
    if (act) y2 = red[lane + off];
    __syncthreads();
    if (act) {
      quad_ge_add(m, q, y2, d2, one);
      slot[qi] = m;
    }
    __syncthreads();
  }
  // rows: sum_a a*R_a = sum_{i>=1} suffix_i ; columns (b = idx+1): sum_b b*C_b = sum_{i>=0} suffix_i
  if (is_row ? (idx == 0 || idx >= A) : (idx >= Bc)) quad_load(m, q, id);
  if (lead) slot[qi] = m;
  __syncthreads();
  for (uint32_t off = 16; off >= 1; off >>= 1) {
    if (lead && idx < off) {
      const ge y2 = red[lane + off];
      quad_ge_add(m, q, y2, d2, one);
      slot[qi] = m;
    }
    __syncthreads();
  }
  if (quad == 0) {  // W = Bc * rows + cols: m holds the rows' sum (red[0]), spread over the quad
    for (uint32_t i = 0; i < lBc; i++) quad_ge_dbl(m, q);
    const ge cols = red[32];
    quad_ge_add(m, q, cols, d2, one);
    ge w;
    quad_gather(w, m);
    if (qi == 0) W[gk] = w;
  }
}

// one lane per point: extended coordinates -> 32-byte Ristretto encoding
__global__ void __launch_bounds__(64) k_compress_ge(const ge *__restrict__ in, uint32_t count, uint8_t *__restrict__ out32) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  uint8_t c32[32];
  ristretto_compress(c32, in[i]);
  uint32_t *o = (uint32_t *)(out32 + (size_t)i * 32);
#pragma unroll
  for (int k = 0; k < 8; k++)
    o[k] = (uint32_t)c32[4 * k] | ((uint32_t)c32[4 * k + 1] << 8) | ((uint32_t)c32[4 * k + 2] << 16) | ((uint32_t)c32[4 * k + 3] << 24);
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
__global__ void __launch_bounds__(64) k_sum_accumulators(const uint8_t *__restrict__ in128, uint32_t n, uint8_t *__restrict__ comp32,
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

// diagnostics (bpp_shader_clock): one wavefront naps and reads the shader-clock counter and the constant 100 MHz one
__global__ void k_shader_clock(uint64_t *out, uint32_t naps) {
  const uint64_t c0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
  for (uint32_t i = 0; i < naps; i++) __builtin_amdgcn_s_sleep(127);
  if (threadIdx.x == 0) {
    out[0] = __builtin_amdgcn_s_memtime() - c0;
    out[1] = __builtin_amdgcn_s_memrealtime() - w0;
  }
}

}  // namespace bpp
