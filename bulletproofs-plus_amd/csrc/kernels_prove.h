// Device kernels of the batch prover: RangeProof::prove_with_rng (src/range_proof.rs:232-608) for B proofs at once.
//
// Schedule (canonical encodings make every algebraically equal schedule byte-identical):
//   * no generator folding.  Round j's L_j / R_j are computed directly over the ORIGINAL generators:
//       Gf[i] = sum_{u = i mod len} cG[u] G_u,  Hf[i] = sum cH[u] H_u   (len = mn >> j)
//     with per-proof coefficient vectors cG/cH that absorb e^-1, e*y^-n, e, e^-1 each round (:511-521);
//     every G_u / H_u lands in exactly one of L_j, R_j, so a round is two MSMs of mn + t + 1 terms over FIXED bases.
//   * fixed-base MSM: signed windows over a precomputed table in HBM (per generator ceil(254/w) slots x 2^(w-1)
//     multiples, affine Niels padded to one 128-byte line; w = 8..11 chosen by fb_geometry) -> 23 additions per term
//     at w = 11 (the full top window stays unsigned and uses the spare slot, recode.h), no doublings, no buckets.
//   * per proof and round: one lane for the Fiat-Shamir / RNG / inversion step (kp_lane), one wavefront for the
//     scalar-vector work (kp_wave), one wavefront per output point for the MSM (k_fb_msm).
// All scalars are Montgomery form in HBM; canonical only in MSM inputs, transcript bytes and the proof.
#pragma once
#include <stdlib.h>

#include "ct.h"
#include "kernels_verify.h"
#include "recode.h"
#include "wstrobe.h"

namespace bpp {

// The LDS of a prover kernel holds witness-keyed generator states, raw draws and partial sums: nothing of it is left for the next
// workgroup on the CU (the reference wraps the same values in Zeroizing<>, src/range_proof.rs:300-301,438-464,542-570).  Called by
// every thread of the workgroup as the kernel's last statement.
template <class T>
__device__ __forceinline__ void lds_wipe(T &obj) {
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < sizeof(T) / 4; k += blockDim.x) ((uint32_t *)&obj)[k] = 0;
}

// Phase clocks of the prover's round kernel (a measurement build only: -DBPP_KP_PHASES, tools/gpu_kp_phases.sh): lane 0 of every
// workgroup adds the shader-clock cycles of each phase of kp_round to a device-global table that bpp_debug_kp_phases() reads.
// The product build compiles none of it.
#ifdef BPP_KP_PHASES
__device__ unsigned long long g_kp_phase[32];
#define KP_T0() unsigned long long kp_t_ = __builtin_readcyclecounter()
#define KP_MARK(id)                                                      \
  do {                                                                   \
    const unsigned long long kp_n_ = __builtin_readcyclecounter();       \
    if (threadIdx.x == 0) atomicAdd(&g_kp_phase[id], kp_n_ - kp_t_);     \
    kp_t_ = __builtin_readcyclecounter();                                \
  } while (0)
#else
#define KP_T0()
#define KP_MARK(id)
#endif

// ---------------------------------------------------------------- fixed-base tables
// One table entry = one affine Niels point padded to a full 128-byte line: a lookup is exactly one aligned line
// (a 160-byte projective entry straddles two), and the mixed addition costs 7 multiplications instead of 8.
struct fbent {
  niels q;  // niels is itself padded and aligned to 128 bytes (point.h)
};
static_assert(sizeof(fbent) == 128, "one fixed-base table entry = one 128-byte line");
struct cached {  // projective niels (table construction only)
  fe yplusx, yminusx, z, t2d;
};
BPP_HD void ge_to_cached(cached &r, const ge &p) {
  fe d2;
  fe_add(r.yplusx, p.Y, p.X);
  fe_carry(r.yplusx);
  fe_sub(r.yminusx, p.Y, p.X);
  fe_copy(r.z, p.Z);
  fe_carry(r.z);
  fe_const(d2, FE_D2);
  fe_mul(r.t2d, p.T, d2);
}

// r = p + q, q projective niels: 8 mul
BPP_HD void ge_add_cached(ge &r, const ge &p, const cached &q) {
  fe a, b, c, d, e, f, g, h;
  fe_add(a, p.Y, p.X);
  fe_sub(b, p.Y, p.X);
  fe_mul(a, a, q.yplusx);
  fe_mul(b, b, q.yminusx);
  fe_mul(c, q.t2d, p.T);
  fe_mul(d, p.Z, q.z);
  fe_add(d, d, d);
  fe_sub(e, a, b);
  fe_add(h, a, b);
  fe_add(g, d, c);
  fe_sub(f, d, c);
  fe_mul(r.X, e, f);
  fe_mul(r.Y, h, g);
  fe_carry(g);
  fe_mul(r.Z, g, f);
  fe_mul(r.T, e, h);
}

// one lane per (generator, window, block of FB_BUILD_BLOCK entries): entries d * 2^(wbits*w) * P, normalised to affine
// (one inversion each; the table is built once per parameter set)
__global__ void __launch_bounds__(64) k_fb_build(const niels *__restrict__ gens, uint32_t n_gens, FbGeom geo,
                                                 fbent *__restrict__ tbl) {
  const uint32_t blocks = (geo.entries + FB_BUILD_BLOCK - 1) / FB_BUILD_BLOCK;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_gens * geo.windows * blocks) return;
  const uint32_t blk = i % blocks, gw = i / blocks, g = gw / geo.windows, w = gw % geo.windows;
  // the last slot of a geometry with an unsigned top window continues the slot before it: entries + 1 .. 2 * entries
  const bool ext = fb_top_unsigned(geo) && w + 1 == geo.windows;
  const uint32_t wexp = ext ? w - 1 : w;
  ge base;
  ge_identity(base);
  const niels q = gens[g];
  ge_madd(base, base, q);
  if (wexp) ge_dbl_n(base, base, (int)(geo.wbits * wexp));
  cached cb;
  ge_to_cached(cb, base);
  // acc = first * base by double-and-add from the top bit
  const uint32_t pos = blk * FB_BUILD_BLOCK, first = pos + 1 + (ext ? geo.entries : 0u);
  ge acc;
  ge_identity(acc);
  for (int bit = 31; bit >= 0; bit--) {
    if ((first >> bit) == 0) continue;
    ge_dbl(acc, acc);
    if ((first >> bit) & 1u) ge_add_cached(acc, acc, cb);
  }
  fbent *out = tbl + ((size_t)g * geo.windows + w) * geo.entries + pos;
  const uint32_t cnt = geo.entries - pos < FB_BUILD_BLOCK ? geo.entries - pos : FB_BUILD_BLOCK;
  for (uint32_t d = 0; d < cnt; d++) {
    fe zi, x, y;
    fe_invert(zi, acc.Z);
    fe_mul(x, acc.X, zi);
    fe_mul(y, acc.Y, zi);
    niels e;
    niels_from_affine(e, x, y);
    fe_carry(e.yminusx);
    out[d].q = e;
    if (d + 1 < cnt) ge_add_cached(acc, acc, cb);
  }
}

// out[o] = sum_i scal[o][i] * Gen[gidx[o][i]], i < count[o]; one workgroup of FB_THREADS lanes per output.
// scal: canonical scalars, row stride `stride`; gidx rows likewise.  Result left in extended coordinates (k_compress_ge
// turns a whole launch's outputs into bytes, one lane each, instead of one busy lane per workgroup here).
//   phase 1 (per chunk of FB_CHUNK terms): recode scalars to signed digits in LDS (fb_recode)
//   phase 2: flat (term, window) work items, stride FB_THREADS -> every lane gets the same number of additions; the table
//            line of item k+1 is requested before item k is added.
#define FB_THREADS 256
#define FB_CHUNK 512
struct FbStage {
  int16_t dig[FB_CHUNK * FB_MAX_WINDOWS];
  uint32_t gi[FB_CHUNK];
};
union FbShared {
  ge red[FB_THREADS];
  FbStage st;
};

BPP_D void fb_fetch(niels &q, int &d, const FbStage &st, const fbent *__restrict__ tbl, const FbGeom &geo, uint32_t it) {
  const uint32_t i = it / geo.items, w = it - i * geo.items;
  d = st.dig[it];
  const uint32_t mag = (uint32_t)(d < 0 ? -d : d);
  // y+x / y-x exchanged by address for a negative digit (ge_madd_swapped does the rest of the negation)
  niels_load_swapped(q, &tbl[((size_t)st.gi[i] * geo.windows + w) * geo.entries + (mag ? mag - 1u : 0u)].q, d < 0);
}

// mont != 0: the scalars arrive in Montgomery form (the prover's round kernels leave them so: the conversion is one product per
// term, done here by every lane for its own terms instead of by the round kernel's single wavefront on the call's serial path)
__global__ void __launch_bounds__(FB_THREADS) k_fb_msm(const sc *__restrict__ scal, const uint32_t *__restrict__ gidx,
                                                       const uint32_t *__restrict__ count, uint32_t stride,
                                                       const fbent *__restrict__ tbl, FbGeom geo, ge *__restrict__ out, uint32_t mont) {
  // Outputs are dealt to the workgroups even ones first, then the odd ones.  The prover's last launch pairs a long output
  // (A1: 2mn + t + 1 terms) with a short one (B: t + 1 terms) per proof; with o = blockIdx.x every long output sat on an even
  // workgroup index, i.e. (workgroups go round the 8 XCDs) on four of the eight XCDs, and that launch took 1.35 ms against
  // the 0.33 ms of a round with as many additions.  Equal outputs (every other launch) do not care.
  const uint32_t n_even = (gridDim.x + 1u) >> 1;
  const uint32_t o = blockIdx.x < n_even ? 2u * blockIdx.x : 2u * (blockIdx.x - n_even) + 1u;
  const uint32_t tid = threadIdx.x, nthr = blockDim.x;  // a whole number of wavefronts per output, at most FB_THREADS
  const uint32_t n = count[o];
  __shared__ FbShared sh;
  ge acc;
  ge_identity(acc);
  for (uint32_t base = 0; base < n; base += FB_CHUNK) {
    const uint32_t cn = n - base < FB_CHUNK ? n - base : FB_CHUNK;
    __syncthreads();  // the previous chunk's digits are no longer read
    for (uint32_t i = tid; i < cn; i += nthr) {
      sc s = scal[(size_t)o * stride + base + i];
      if (mont) sc_from_mont(s, s);
      fb_recode(sh.st.dig + (size_t)i * geo.items, s, geo);
      sh.st.gi[i] = gidx[(size_t)o * stride + base + i];
    }
    __syncthreads();
    const uint32_t items = cn * geo.items;
    uint32_t it = tid;
    niels nxt;
    int nd = 0;
    if (it < items) fb_fetch(nxt, nd, sh.st, tbl, geo, it);
    while (it < items) {
      niels cur = nxt;
      const int cd = nd;
      it += nthr;
      if (it < items) fb_fetch(nxt, nd, sh.st, tbl, geo, it);
      if (cd != 0) ge_madd_swapped(acc, acc, cur, cd < 0);
    }
  }
  __syncthreads();
  sh.red[tid] = acc;
  __syncthreads();
  // nthr is a multiple of 64, not necessarily a power of two (192 lanes = three wavefronts per output fill the chip's
  // 3-wavefronts-per-SIMD slots with exactly two rounds of 1024 proofs' L and R): the lanes above the largest power of
  // two fold into the low ones first
  uint32_t p2 = 64;
  while (p2 * 2 <= nthr) p2 *= 2;
  if (nthr > p2) {
    if (tid + p2 < nthr) {
      ge x = sh.red[tid], y2 = sh.red[tid + p2];
      ge_add(x, x, y2);
      sh.red[tid] = x;
    }
    __syncthreads();
  }
  for (uint32_t off = p2 / 2; off >= 1; off >>= 1) {
    if (tid < off) {
      ge x = sh.red[tid], y2 = sh.red[tid + off];
      ge_add(x, x, y2);
      sh.red[tid] = x;
    }
    __syncthreads();
  }
  if (tid == 0) out[o] = sh.red[0];
}

// ---- The same sum cut into independent WAVEFRONTS (round 5): output o is `parts` slices of <= FBP_MAX_PER terms, one 64-lane
// workgroup each, and a slice ends with its 64 per-lane partial sums in memory -- no reduction tree, no barrier beyond its own
// staging, 9 KB of LDS.  The workgroup form above spends 17 % of its time in its 8-level tree (one busy wavefront, three idle
// ones holding registers and 40 KB of LDS: tools/microbench/fb_probe.hip, B against A); here the tree is gone from the chip-filling
// kernel and the `parts` x 64 partial sums of an output are added up by the round kernel that consumes the point anyway
// (kp_round: fb_reduce_pair, both outputs of a proof side by side on the two halves of its wavefront), or by k_fb_sum where a
// plain point is wanted (the last launch, the unfused form, "ct" = 2).
#define FBP_MAX_PER 128
#define FBP_MAX_PARTS 8
struct FbPartStage {
  int16_t dig[FBP_MAX_PER * FB_MAX_WINDOWS];
  uint32_t gi[FBP_MAX_PER];
};
__global__ void __launch_bounds__(64) k_fb_part(const sc *__restrict__ scal, const uint32_t *__restrict__ gidx, const uint32_t *__restrict__ count,
                                                uint32_t stride, uint32_t parts, const fbent *__restrict__ tbl, FbGeom geo,
                                                ge *__restrict__ partial /* [outputs][parts][64] */, uint32_t mont) {
  const uint32_t o = blockIdx.x / parts, part = blockIdx.x - o * parts, lane = threadIdx.x;
  const uint32_t n = count[o], per = (n + parts - 1) / parts;
  const uint32_t lo = part * per < n ? part * per : n, hi = lo + per < n ? lo + per : n, cn = hi - lo;
  __shared__ FbPartStage st;
  for (uint32_t i = lane; i < cn; i += 64) {
    sc s = scal[(size_t)o * stride + lo + i];
    if (mont) sc_from_mont(s, s);
    fb_recode(st.dig + (size_t)i * geo.items, s, geo);
    st.gi[i] = gidx[(size_t)o * stride + lo + i];
  }
  __syncthreads();
  ge acc;
  ge_identity(acc);
  const uint32_t items = cn * geo.items;
  // item it = (term i, window w), it = lane + 64 k: (i, w) advance without a division
  const uint32_t di = 64u / geo.items, dw = 64u - di * geo.items;
  uint32_t it = lane, i = lane / geo.items, w = lane - i * geo.items;
  auto fetch = [&](niels &q, int &d) {
    d = st.dig[it];
    const uint32_t mag = (uint32_t)(d < 0 ? -d : d);
    niels_load_swapped(q, &tbl[((size_t)st.gi[i] * geo.windows + w) * geo.entries + (mag ? mag - 1u : 0u)].q, d < 0);
  };
  auto step = [&]() {
    it += 64u;
    i += di;
    w += dw;
    if (w >= geo.items) {
      w -= geo.items;
      i++;
    }
  };
  niels nxt;
  int nd = 0;
  if (it < items) fetch(nxt, nd);
  while (it < items) {
    niels cur = nxt;
    const int cd = nd;
    step();
    if (it < items) fetch(nxt, nd);
    if (cd != 0) ge_madd_swapped(acc, acc, cur, cd < 0);
  }
  partial[(size_t)blockIdx.x * 64u + lane] = acc;
  lds_wipe(st);  // (the digits of witness-derived scalars)
}
// out[o] = sum of output o's parts x 64 partial sums: one wavefront per output
__global__ void __launch_bounds__(64) k_fb_sum(const ge *__restrict__ partial, uint32_t parts, ge *__restrict__ out) {
  const uint32_t o = blockIdx.x, lane = threadIdx.x;
  __shared__ ge red[64];
  ge acc = partial[((size_t)o * parts) * 64u + lane];
  for (uint32_t p = 1; p < parts; p++) {
    const ge x = partial[((size_t)o * parts + p) * 64u + lane];
    ge_add(acc, acc, x);
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
  if (lane == 0) out[o] = red[0];
  lds_wipe(red);
}
// Outputs 2p and 2p + 1 (a proof's L and R) summed side by side by ONE wavefront: lanes 0..31 take the first, lanes 32..63 the
// second; a lane adds the 2 x parts partial sums of its two columns, then five levels through LDS inside each half.  The sums end
// up in red[0] and red[32].  All 64 lanes must call.
__device__ __forceinline__ void fb_reduce_pair(ge *red /* LDS, 64 */, const ge *__restrict__ partial, uint32_t parts, uint32_t p) {
  const uint32_t lane = threadIdx.x, half = lane >> 5, h = lane & 31u;
  const ge *src = partial + ((size_t)(2 * p + half) * parts) * 64u;
  ge acc = src[h];
  {
    const ge x = src[h + 32];
    ge_add(acc, acc, x);
  }
  for (uint32_t q = 1; q < parts; q++) {
    const ge x = src[(size_t)q * 64u + h], y2 = src[(size_t)q * 64u + h + 32];
    ge_add(acc, acc, x);
    ge_add(acc, acc, y2);
  }
  red[lane] = acc;
  __syncthreads();
  for (uint32_t off = 16; off >= 1; off >>= 1) {
    if (h < off) {
      ge x = red[lane], y2 = red[lane + off];
      ge_add(x, x, y2);
      red[lane] = x;
    }
    __syncthreads();
  }
}

// ONE output summed by ONE whole wavefront (kp_round with two or more wavefronts: L on the first, R on the second): lane l adds
// its column of the `parts` slices, then six levels through this wavefront's 64 entries of LDS.  Wavefront-local synchronisation
// (wstrobe.h: ws_sync): the other wavefront is doing the same for the other output at its own pace.  Result in red[0].
__device__ __forceinline__ void fb_reduce_one(ge *red /* LDS, this wavefront's 64 */, const ge *__restrict__ partial, uint32_t parts, uint32_t o) {
  const uint32_t lane = ws_lane();
  const ge *src = partial + ((size_t)o * parts) * 64u;
  ge acc = src[lane];
  for (uint32_t q = 1; q < parts; q++) {
    const ge x = src[(size_t)q * 64u + lane];
    ge_add(acc, acc, x);
  }
  red[lane] = acc;
  ws_sync();
  for (uint32_t off = 32; off >= 1; off >>= 1) {
    if (lane < off) {
      ge x = red[lane], y2 = red[lane + off];
      ge_add(x, x, y2);
      red[lane] = x;
    }
    ws_sync();
  }
}

#define CT_ROW 16u
// scalars per proof in the round kernels' vector buffer: a, b, cG, cH (mn each), y^0 .. y^(mn+1), and ("ct" = 2) the factors the
// final step's folded generators take over the public points of step rounds - back: fG[c], fH[c], c < 2^back <= KP_EX_CLASSES
#define KP_EX_CLASSES 8u
#define KP_VEC_LEN(mn) (5u * (mn) + 2u + 2u * KP_EX_CLASSES)
#define KP_MAX_THREADS 256u  // the round kernel's workgroup: 1, 2 or 4 wavefronts (option "prove_waves")  // terms per row of the final round's secret-only term lists (3 + t <= 9 used)

// ---------------------------------------------------------------- per-proof prover state
struct ProveDesc {
  uint32_t m;           // aggregation factor (uniform over a prove batch)
  uint32_t wit_off;     // byte offset of the witness bytes (v LE64 || r[0..t) per opening) in bytes[]
  uint32_t commit_off;  // m compressed commitments
  uint32_t ext_off;     // (rounds+3) x 32 bytes of external randomness
  uint32_t minval_idx;  // m minimum values / presence flags
  uint32_t state_idx;
  uint32_t flags;       // bit0: seed nonce present
  uint32_t seed_off;    // byte offset of the 32-byte seed nonce (if any)
};

struct ProveState {
  Strobe tr;   // the proof's merlin transcript
  Strobe rng;  // current TranscriptRng (src/transcripts.rs:185-194)
  sc y, z, e, einv, esq, einvsq, yinv_nhalf, yinv_prev, yinv1, r, s;
  sc alpha[6], dl[6], dr[6], dd[6], eta[6];
  uint32_t status;  // nonzero: proving failed (BPP_ERR_*)
};

#define PV_STATUS_COMMIT_MISMATCH 2u  // InvalidArgument: "Witness opening is invalid!" (:275-284)
#define PV_STATUS_TRANSCRIPT 1u       // VerificationFailed: identity point / zero challenge

// build_rng (src/transcripts.rs:185-194): clone, rekey with the witness bytes, finalize with 32 external bytes
__device__ __forceinline__ void pv_build_rng(Strobe &rng, const Strobe &tr, const uint8_t *wit, uint32_t wit_len,
                                             const uint8_t *ext32) {
  rng = tr;
  merlin_rng_rekey(rng, (const uint8_t *)"witness", 7, wit, wit_len);
  uint8_t r32[32];
  for (int i = 0; i < 32; i++) r32[i] = ext32[i];
  merlin_rng_finalize(rng, r32);
}
// Scalar::random_not_zero(transcript_rng)
__device__ __forceinline__ void pv_random(sc &out, Strobe &rng) {
  do {
    uint8_t w[64];
    merlin_rng_fill(rng, w, 64);
    sc_mont_from_wide(out, w);
  } while (sc_iszero(out));
}
__device__ __forceinline__ void pv_nonce_or_random(sc &out, Strobe &rng, const uint8_t *seed, bool has_seed, const char *label,
                                                   uint32_t llen, int j, int k) {
  if (has_seed)
    dev_nonce(out, seed, label, llen, j, k);
  else
    pv_random(out, rng);
}
__device__ __forceinline__ bool pv_validate_append(Strobe &tr, const char *label, uint32_t llen, const uint8_t *p32) {
  uint8_t b[32];
  uint32_t nz = 0;
  for (int i = 0; i < 32; i++) {
    b[i] = p32[i];
    nz |= b[i];
  }
  merlin_append_message(tr, (const uint8_t *)label, llen, b, 32);
  return nz != 0;
}

// ---- wavefront-cooperative helpers (wstrobe.h): one transcript per WAVEFRONT (lanes and synchronisation are the wavefront's) ----
#define PW_MAX_DRAWS 14  // r, s, d[6], eta[6]
struct ProveLds {
  uint64_t tr[25], rng[25];
  uint8_t buf[64];   // scratch of the wavefront that owns the transcript (challenge bytes)
  uint8_t buf1[64];  // scratch of the wavefront that owns the round's TranscriptRng when that is another one (kp_lane_body2)
  uint32_t trmeta[3];  // pos, pos_begin, cur_flags of the transcript, for the wavefront that clones it
  sc xch[2];
  uint64_t rng_bak[25];              // pw_randoms: the generator's state before a batch of draws
  uint8_t wide[PW_MAX_DRAWS][64];    //             the batch's 64-byte outputs
  uint32_t wk[2][WK_LDS_DWORDS_RC];  // the permutations' exchange images (wkeccak.h): one per wavefront that runs a sponge (kp_lane_body2: two)
};
// build_rng (src/transcripts.rs:185-194): clone, rekey with the witness bytes, finalize with 32 external bytes
__device__ __forceinline__ void pw_build_rng(WStrobe &rng, ProveLds &L, const KeccakLanes &K, const WStrobe &tr, const uint8_t *wit,
                                             uint32_t wit_len, const uint8_t *ext32) {
  ws_clone(rng, L.rng, tr);
  wm_rng_rekey(rng, K, "witness", 7, BytesAt{wit}, wit_len);
  wm_rng_finalize(rng, K, BytesAt{ext32});
}
// Scalar::random_not_zero(transcript_rng); every lane ends up with the same scalar
__device__ __forceinline__ void pw_random(sc &out, WStrobe &rng, uint8_t *buf64, const KeccakLanes &K) {
  do {
    wm_rng_fill(rng, K, buf64, 64);
    sc_mont_from_wide(out, buf64);
  } while (sc_iszero(out));
}
__device__ __forceinline__ void pw_random(sc &out, WStrobe &rng, ProveLds &L, const KeccakLanes &K) { pw_random(out, rng, L.buf, K); }
__device__ __forceinline__ bool pw_challenge(WStrobe &tr, ProveLds &L, const KeccakLanes &K, const char *label, uint32_t llen, sc &out) {
  wm_challenge_bytes(tr, K, label, llen, L.buf, 64);
  sc_mont_from_wide(out, L.buf);
  return !sc_iszero(out);
}
__device__ __forceinline__ bool pw_validate_append(WStrobe &tr, const KeccakLanes &K, const char *label, uint32_t llen,
                                                   const uint8_t *p32) {
  const bool nz = __ballot(ws_lane() < 32 && p32[ws_lane() & 31u] != 0) != 0;
  wm_append_message(tr, K, label, llen, BytesAt{p32}, 32);
  return nz;
}
// n = `counts` scalars in a row from the transcript RNG (Scalar::random_not_zero each, src/protocols/scalar_protocol.rs:23-30),
// written to dst[0][0..c0), dst[1][0..c1), ...  The generator is sequential -- one forced Keccak-f per draw -- but the wide
// reduction of a draw's 64 bytes is not: the bytes of all n draws are squeezed first, then lane k reduces draw k (as n
// reductions in a row, each on every lane, they were ~8 k cycles apiece on the round's serial path).  A zero scalar makes the
// reference draw again, which shifts every later draw: should any of the n be zero (probability n 2^-252) the generator is put
// back to where it was and the draws are made one by one as before.
__device__ __forceinline__ void pw_randoms(WStrobe &rng, ProveLds &L, const KeccakLanes &K, sc *const *dst, const uint32_t *counts,
                                           uint32_t groups, uint8_t *buf64 = nullptr) {
  if (!buf64) buf64 = L.buf;
  uint32_t n = 0;
  for (uint32_t g = 0; g < groups; g++) n += counts[g];
  ws_sync();
  if (ws_lane() < 25) L.rng_bak[ws_lane()] = rng.st[ws_lane()];
  const uint32_t pos0 = rng.pos, begin0 = rng.pos_begin, flags0 = rng.cur_flags;
  for (uint32_t k = 0; k < n; k++) wm_rng_fill(rng, K, L.wide[k], 64);
  sc v;
  sc_0(v);
  if (ws_lane() < n) sc_mont_from_wide(v, L.wide[ws_lane()]);
  const bool zero = ws_lane() < n && sc_iszero(v);
  if (__ballot(zero) == 0) {
    uint32_t k0 = 0;
    for (uint32_t g = 0; g < groups; g++) {
      if (ws_lane() >= k0 && ws_lane() < k0 + counts[g]) dst[g][ws_lane() - k0] = v;
      k0 += counts[g];
    }
    return;
  }
  ws_sync();
  if (ws_lane() < 25) rng.st[ws_lane()] = L.rng_bak[ws_lane()];
  rng.pos = pos0;
  rng.pos_begin = begin0;
  rng.cur_flags = flags0;
  ws_sync();
  for (uint32_t g = 0; g < groups; g++)
    for (uint32_t k = 0; k < counts[g]; k++) {
      sc x;
      pw_random(x, rng, buf64, K);
      if (ws_lane() == 0) dst[g][k] = x;
    }
}
// t scalars "label"[k]: nonces (lane k computes its own BLAKE2b) or sequential draws from the transcript RNG; dst in HBM
__device__ __forceinline__ void pw_nonces_or_randoms(sc *dst, uint32_t t, WStrobe &rng, ProveLds &L, const KeccakLanes &K,
                                                     const uint8_t *seed, bool has_seed, const char *label, uint32_t llen, int j) {
  if (has_seed) {
    if (ws_lane() < t) {
      sc v;
      dev_nonce(v, seed, label, llen, j, (int)ws_lane());
      dst[ws_lane()] = v;
    }
  } else {
    for (uint32_t k = 0; k < t; k++) {
      sc v;
      pw_random(v, rng, L, K);
      if (ws_lane() == 0) dst[k] = v;
    }
  }
}

// ---- stage 0, one wavefront per proof: RangeProofTranscript::new (:287-297), alpha (:325-333) ----
__global__ void __launch_bounds__(64) kp_init(const uint8_t *__restrict__ bytes, const ProveDesc *__restrict__ desc,
                                              const uint64_t *__restrict__ minvals, const uint8_t *__restrict__ states,
                                              const uint8_t *__restrict__ hg32, uint32_t n_bits, uint32_t t, uint32_t B,
                                              ProveState *__restrict__ ps) {
  const uint32_t p = blockIdx.x;
  if (p >= B) return;
  __shared__ ProveLds L;
  const KeccakLanes K = keccak_lanes(L.wk[0]);
  const ProveDesc d = desc[p];
  ProveState &st = ps[p];
  if (threadIdx.x == 0) st.status = 0;
  const uint8_t *sb = states + 203u * d.state_idx;
  for (uint32_t k = threadIdx.x; k < 200; k += 64) ((uint8_t *)L.tr)[k] = sb[k];
  WStrobe tr, rng;
  tr.st = L.tr;
  tr.pos = sb[200];
  tr.pos_begin = sb[201];
  tr.cur_flags = sb[202];
  __syncthreads();
  // (the call-wide head of RangeProofTranscript::new -- domain separator, H, G bases, N, T, M -- is already in `states`: the host
  // applies it once per distinct caller transcript, engine_prove.h)
  for (uint32_t j = 0; j < d.m; j++) wm_append_message(tr, K, "Ci", 2, BytesAt{bytes + d.commit_off + 32 * j}, 32);
  for (uint32_t j = 0; j < d.m; j++) wm_append_u64(tr, K, "vi - minimum_value", 18, minvals[d.minval_idx + j]);
  const uint32_t wit_len = d.m * (8 + 32 * t);
  const bool has_seed = d.flags & 1u;
  if (!has_seed) {
    pw_build_rng(rng, L, K, tr, bytes + d.wit_off, wit_len, bytes + d.ext_off);
    sc *const dst[1] = {st.alpha};
    const uint32_t cnt[1] = {t};
    pw_randoms(rng, L, K, dst, cnt, 1);
  } else {
    pw_nonces_or_randoms(st.alpha, t, rng, L, K, bytes + d.seed_off, true, "alpha", 5, -1);
  }
  ws_store(st.tr, tr);
  lds_wipe(L);
}

// ---- A = sum_{bit=1} G_i - sum_{bit=0} H_i + sum_k alpha_k G_k  (:300-345), one wavefront per proof ----
// a_L/a_R are never materialised as scalars here: they are 0/1 and 0/-1.
__global__ void __launch_bounds__(64) kp_A(const uint8_t *__restrict__ bytes, const ProveDesc *__restrict__ desc,
                                           const uint64_t *__restrict__ minvals, const uint8_t *__restrict__ min_present,
                                           const niels *__restrict__ gens, const fbent *__restrict__ tbl, FbGeom geo,
                                           uint32_t n_gen, uint32_t n_bits, uint32_t t, const ProveState *__restrict__ ps,
                                           uint8_t *__restrict__ a_out32) {
  const uint32_t p = blockIdx.x, lane = threadIdx.x;
  const ProveDesc d = desc[p];
  const uint32_t mn = d.m * n_bits;
  __shared__ ge red[64];
  ge acc;
  ge_identity(acc);
  for (uint32_t i = lane; i < mn; i += 64) {
    const uint32_t party = i / n_bits, bit_idx = i % n_bits;
    const uint8_t *w = bytes + d.wit_off + party * (8 + 32 * t);
    uint64_t v = 0;
    for (int k = 0; k < 8; k++) v |= (uint64_t)w[k] << (8 * k);
    if (min_present[d.minval_idx + party]) v -= minvals[d.minval_idx + party];
    const bool bit = (v >> bit_idx) & 1ULL;
    niels q = gens[2 * i + (bit ? 0 : 1)];  // G_i or H_i
    niels_cneg(q, !bit);                     // a_R = a_L - 1 = -1 where the bit is 0
    ge_madd(acc, acc, q);
  }
  // alpha_k * G_k through the fixed-base table: the t scalars' digits go to LDS and the t x windows lookups are spread over the
  // wavefront (one or two additions per lane; on t lanes they were a chain of 23 additions each with 61 lanes waiting)
  __shared__ int16_t s_dig[6 * FB_MAX_WINDOWS];
  if (lane < t) {
    sc s;
    sc_from_mont(s, ps[p].alpha[lane]);
    fb_recode(s_dig + lane * FB_MAX_WINDOWS, s, geo);
  }
  __syncthreads();
  for (uint32_t it = lane; it < t * geo.items; it += 64) {
    const uint32_t k = it / geo.items, w = it - k * geo.items;
    const int dgt = s_dig[k * FB_MAX_WINDOWS + w];
    if (dgt != 0) {
      const fbent *row = tbl + (size_t)(n_gen + k) * fb_stride(geo);
      const uint32_t mag = (uint32_t)(dgt < 0 ? -dgt : dgt);
      niels c = row[(size_t)w * geo.entries + (mag - 1)].q;
      niels_cneg(c, dgt < 0);
      ge_madd(acc, acc, c);
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
  if (lane == 0) {
    uint8_t c32[32];
    ristretto_compress(c32, red[0]);
    for (int k = 0; k < 32; k++) a_out32[(size_t)p * 32 + k] = c32[k];
  }
}

// ---- Fiat-Shamir kernel, step j = 0..r (one wavefront per proof, wstrobe.h): challenges, RNG draws, inversions ----
//   j == 0      : challenges_y_z(A) (:348), then the draws of round 0
//   1 <= j <= r : challenge_round_e(L_{j-1}, R_{j-1}) (:498-508), alpha update (:535-537), then the draws of round j
//                 (j < r: d_L, d_R :437-464, y^-n :426-432;  j == r: r, s, d, eta :542-571)
// The per-round TranscriptRng is a throw-away clone (src/transcripts.rs:185-194): it is only built when something is
// drawn from it (no seed nonce, or the final round's r and s).
__device__ __forceinline__ void kp_lane_body(const uint8_t *__restrict__ bytes, const ProveDesc *__restrict__ desc,
                                             uint32_t n_bits, uint32_t t, uint32_t B, uint32_t j, uint32_t rounds,
                                             const uint8_t *__restrict__ a32, const uint8_t *lr32 /* [B][2][32] of round j-1 */,
                                             ProveState *ps, ProveLds &L) {
  const uint32_t p = blockIdx.x, lane = threadIdx.x;
  if (p >= B) return;
  const KeccakLanes K = keccak_lanes(L.wk[0]);
  const ProveDesc d = desc[p];
  ProveState &st = ps[p];
  const uint32_t mn = d.m * n_bits, wit_len = d.m * (8 + 32 * t);
  const bool has_seed = d.flags & 1u;
  const uint8_t *seed = bytes + d.seed_off;
  bool ok = true;
  WStrobe tr, rng;
  KP_T0();
  ws_load(tr, L.tr, st.tr);
  sc e, y;
  // the round's TranscriptRng is cloned after the points are appended and BEFORE the challenge is drawn
  // (src/transcripts.rs:126-131,142-147)
  const bool need_rng = !has_seed || j == rounds;
  if (j == 0) {
    ok = pw_validate_append(tr, K, "A", 1, a32 + (size_t)p * 32) && ok;
    if (need_rng) pw_build_rng(rng, L, K, tr, bytes + d.wit_off, wit_len, bytes + d.ext_off + 32 * (1 + j));
    sc z;
    ok = pw_challenge(tr, L, K, "y", 1, y) && ok;
    ok = pw_challenge(tr, L, K, "z", 1, z) && ok;
    if (lane == 0) {
      st.y = y;
      st.z = z;
    }
    sc_copy(e, y);  // placeholder operand of the paired inversion below
  } else {
    ok = pw_validate_append(tr, K, "L", 1, lr32 + (size_t)p * 64) && ok;
    ok = pw_validate_append(tr, K, "R", 1, lr32 + (size_t)p * 64 + 32) && ok;
    KP_MARK(1);
    if (need_rng) pw_build_rng(rng, L, K, tr, bytes + d.wit_off, wit_len, bytes + d.ext_off + 32 * (1 + j));
    KP_MARK(2);
    ok = pw_challenge(tr, L, K, "e", 1, e) && ok;
    y = st.y;
    KP_MARK(3);
  }
  // ONE inversion per round, the same input in every lane: e (rounds >= 1), y in step 0.  y^-(n / 2^(j+1)) is a power of y^-1
  // (kept from step 0) -- until round 4 it was the inverse of y^(n / 2^(j+1)), taken on the odd lanes beside e's on the even
  // ones: two different inputs in one wavefront make the variable-time inversion walk both lanes' branches one after the other.
  const uint32_t n_half = mn >> (j + 1);
  sc inv, einv, yinv, yinv1;
  {
    sc x;
#pragma unroll
    for (int q = 0; q < 8; q++) x.v[q] = j == 0 ? y.v[q] : e.v[q];
    sc_mont_invert_vartime(inv, x);
  }
  KP_MARK(4);
  if (j == 0) {
    yinv1 = inv;
    if (lane == 0) st.yinv1 = inv;
    sc_copy(einv, inv);  // (unused in step 0)
  } else {
    yinv1 = st.yinv1;
    einv = inv;
  }
  if (j < rounds) sc_mont_pow_u32(yinv, yinv1, n_half);
  else sc_copy(yinv, yinv1);  // (unused in the last step)
  if (j > 0) {
    sc esq, einvsq;
    sc_montsq(esq, e);
    sc_montsq(einvsq, einv);
    if (lane == 0) {
      st.e = e;
      st.einv = einv;
      st.esq = esq;
      st.einvsq = einvsq;
    }
    if (lane < t) {  // alpha_k += d_L,k e^2 + d_R,k e^-2
      sc a = st.alpha[lane], u, v;
      sc_montmul(u, st.dl[lane], esq);
      sc_montmul(v, st.dr[lane], einvsq);
      sc_add(a, a, u);
      sc_add(a, a, v);
      st.alpha[lane] = a;
    }
    ws_sync();  // dl / dr are overwritten below
  }
  KP_MARK(5);
  if (j < rounds) {
    if (has_seed) {
      pw_nonces_or_randoms(st.dl, t, rng, L, K, seed, true, "dL", 2, (int)j);
      pw_nonces_or_randoms(st.dr, t, rng, L, K, seed, true, "dR", 2, (int)j);
    } else {  // d_L[0..t), d_R[0..t): 2 t draws in a row (:437-464)
      sc *const dst[2] = {st.dl, st.dr};
      const uint32_t cnt[2] = {t, t};
      pw_randoms(rng, L, K, dst, cnt, 2);
    }
    if (lane == 0) {
      st.yinv_prev = st.yinv_nhalf;  // the fold of step j still needs round j-1's y^-n
      st.yinv_nhalf = yinv;
    }
  } else {
    if (lane == 0) st.yinv_prev = st.yinv_nhalf;
    if (has_seed) {  // r and s are always drawn (:542-546); d and eta are nonces
      sc *const dst[2] = {&st.r, &st.s};
      const uint32_t cnt[2] = {1, 1};
      pw_randoms(rng, L, K, dst, cnt, 2);
      pw_nonces_or_randoms(st.dd, t, rng, L, K, seed, true, "d", 1, -1);
      pw_nonces_or_randoms(st.eta, t, rng, L, K, seed, true, "eta", 3, -1);
    } else {  // r, s, d[0..t), eta[0..t) (:542-571)
      sc *const dst[4] = {&st.r, &st.s, st.dd, st.eta};
      const uint32_t cnt[4] = {1, 1, t, t};
      pw_randoms(rng, L, K, dst, cnt, 4);
    }
  }
  KP_MARK(6);
  ws_store(st.tr, tr);
  if (!ok && lane == 0) st.status |= PV_STATUS_TRANSCRIPT;
  KP_MARK(7);
}

// The same step on TWO wavefronts of a larger workgroup (kp_round with prove_waves >= 2).  What a round draws from its
// TranscriptRng does not depend on the round's challenge -- the generator is a clone of the transcript taken BEFORE the challenge
// (src/transcripts.rs:142-147) -- so the two halves of the step run side by side:
//   wavefront 0 (owns the transcript): append the points | challenge, inversion, powers, squares, alpha update, store
//   wavefront 1 (owns the generator) :                   | clone, rekey with the witness, finalize, the round's draws
// with workgroup barriers at the three points where one needs what the other made (the appended state to clone; the clone taken
// before the challenge disturbs the state; the end).  On one wavefront the two halves were 113 k + 151 k cycles in a row
// (profiles/r05_kp_phases_slices.json).  Every thread of the workgroup must call (wavefronts >= 2 only meet the barriers).
__device__ __forceinline__ void kp_lane_body2(const uint8_t *__restrict__ bytes, const ProveDesc *__restrict__ desc,
                                              uint32_t n_bits, uint32_t t, uint32_t B, uint32_t j, uint32_t rounds,
                                              const uint8_t *__restrict__ a32, const uint8_t *lr32, ProveState *ps, ProveLds &L) {
  const uint32_t p = blockIdx.x, lane = ws_lane(), wave = threadIdx.x >> 6;
  const KeccakLanes K = keccak_lanes(L.wk[wave & 1u]);  // (wavefronts 0 and 1 run the two sponges; 2 and 3 only meet the barriers)
  const ProveDesc d = desc[p];
  ProveState &st = ps[p];
  const uint32_t mn = d.m * n_bits, wit_len = d.m * (8 + 32 * t);
  const bool has_seed = d.flags & 1u;
  const uint8_t *seed = bytes + d.seed_off;
  const bool need_rng = !has_seed || j == rounds;
  bool ok = true;
  WStrobe tr;
  sc old_dl, old_dr;  // d_L, d_R of the round before: the alpha update reads them while the other wavefront writes the new ones
  sc_0(old_dl);
  sc_0(old_dr);
  KP_T0();
  if (wave == 0) {
    ws_load(tr, L.tr, st.tr);
    if (j == 0) {
      ok = pw_validate_append(tr, K, "A", 1, a32 + (size_t)p * 32) && ok;
    } else {
      ok = pw_validate_append(tr, K, "L", 1, lr32 + (size_t)p * 64) && ok;
      ok = pw_validate_append(tr, K, "R", 1, lr32 + (size_t)p * 64 + 32) && ok;
      if (lane < t) {
        old_dl = st.dl[lane];
        old_dr = st.dr[lane];
      }
    }
    if (lane == 0) {
      L.trmeta[0] = tr.pos;
      L.trmeta[1] = tr.pos_begin;
      L.trmeta[2] = tr.cur_flags;
    }
  }
  KP_MARK(1);
  __syncthreads();  // the transcript with the round's points in it: what the generator is a clone of
  WStrobe rng;
  if (wave == 1 && need_rng) {
    WStrobe view;
    view.st = L.tr;
    view.pos = L.trmeta[0];
    view.pos_begin = L.trmeta[1];
    view.cur_flags = L.trmeta[2];
    ws_clone(rng, L.rng, view);
  }
  __syncthreads();  // cloned: the challenge may now disturb the transcript
  if (wave == 0) {
    sc e, y;
    if (j == 0) {
      sc z;
      ok = pw_challenge(tr, L, K, "y", 1, y) && ok;
      ok = pw_challenge(tr, L, K, "z", 1, z) && ok;
      if (lane == 0) {
        st.y = y;
        st.z = z;
      }
      sc_copy(e, y);
    } else {
      ok = pw_challenge(tr, L, K, "e", 1, e) && ok;
      y = st.y;
    }
    KP_MARK(3);
    const uint32_t n_half = mn >> (j + 1);
    sc inv, einv, yinv, yinv1;
    {
      sc x;
#pragma unroll
      for (int q = 0; q < 8; q++) x.v[q] = j == 0 ? y.v[q] : e.v[q];
      sc_mont_invert_vartime(inv, x);
    }
    KP_MARK(4);
    if (j == 0) {
      yinv1 = inv;
      if (lane == 0) st.yinv1 = inv;
      sc_copy(einv, inv);
    } else {
      yinv1 = st.yinv1;
      einv = inv;
    }
    if (j < rounds) sc_mont_pow_u32(yinv, yinv1, n_half);
    else sc_copy(yinv, yinv1);
    if (j > 0) {
      sc esq, einvsq;
      sc_montsq(esq, e);
      sc_montsq(einvsq, einv);
      if (lane == 0) {
        st.e = e;
        st.einv = einv;
        st.esq = esq;
        st.einvsq = einvsq;
      }
      if (lane < t) {  // alpha_k += d_L,k e^2 + d_R,k e^-2 (:535-537), with the previous round's d_L, d_R
        sc a = st.alpha[lane], u, v;
        sc_montmul(u, old_dl, esq);
        sc_montmul(v, old_dr, einvsq);
        sc_add(a, a, u);
        sc_add(a, a, v);
        st.alpha[lane] = a;
      }
    }
    if (lane == 0) {
      st.yinv_prev = st.yinv_nhalf;  // the fold of step j still needs round j-1's y^-n
      if (j < rounds) st.yinv_nhalf = yinv;
    }
    KP_MARK(5);
    ws_store(st.tr, tr);
    if (!ok && lane == 0) st.status |= PV_STATUS_TRANSCRIPT;
  } else if (wave == 1) {
    if (need_rng) {
      wm_rng_rekey(rng, K, "witness", 7, BytesAt{bytes + d.wit_off}, wit_len);
      wm_rng_finalize(rng, K, BytesAt{bytes + d.ext_off + 32 * (1 + j)});
    }
    if (j < rounds) {
      if (has_seed) {
        pw_nonces_or_randoms(st.dl, t, rng, L, K, seed, true, "dL", 2, (int)j);
        pw_nonces_or_randoms(st.dr, t, rng, L, K, seed, true, "dR", 2, (int)j);
      } else {
        sc *const dst[2] = {st.dl, st.dr};
        const uint32_t cnt[2] = {t, t};
        pw_randoms(rng, L, K, dst, cnt, 2, L.buf1);
      }
    } else if (has_seed) {
      sc *const dst[2] = {&st.r, &st.s};
      const uint32_t cnt[2] = {1, 1};
      pw_randoms(rng, L, K, dst, cnt, 2, L.buf1);
      pw_nonces_or_randoms(st.dd, t, rng, L, K, seed, true, "d", 1, -1);
      pw_nonces_or_randoms(st.eta, t, rng, L, K, seed, true, "eta", 3, -1);
    } else {
      sc *const dst[4] = {&st.r, &st.s, st.dd, st.eta};
      const uint32_t cnt[4] = {1, 1, t, t};
      pw_randoms(rng, L, K, dst, cnt, 4, L.buf1);
    }
  }
  KP_MARK(6);
  __syncthreads();
}

// ---- wave kernel, step j = 0..r (one wavefront per proof): vector prep / fold / inner products / MSM term lists ----
// vec layout per proof (Montgomery): a[mn] | b[mn] | cG[mn] | cH[mn] | ypow[mn+2]
// term rows per proof: 2 outputs x stride; gidx uses the table order (2i = G_i, 2i+1 = H_i, n_gen + k = G_k, n_gen + t = H)
__device__ __forceinline__ void kp_wave_body(const uint8_t *__restrict__ bytes, const ProveDesc *__restrict__ desc,
                                             const uint64_t *__restrict__ minvals, const uint8_t *__restrict__ min_present,
                                             uint32_t n_bits, uint32_t t, uint32_t n_gen, uint32_t j, uint32_t rounds,
                                             uint32_t stride, ProveState *ps, sc *__restrict__ vec,
                                             sc *__restrict__ term_scal, uint32_t *__restrict__ term_gidx,
                                             uint32_t *__restrict__ term_count, sc *__restrict__ ct_scal,
                                             uint32_t *__restrict__ ct_idx, uint32_t *__restrict__ ct_count, sc *__restrict__ ex_scal,
                                             uint32_t *__restrict__ ex_gidx, uint32_t *__restrict__ ex_count, uint32_t ex_back,
                                             sc *red /* LDS, blockDim.x */) {
  // (`lane` = the thread's index in the workgroup, nthr = 64, 128 or 256 threads: every loop below strides by nthr)
  const uint32_t p = blockIdx.x, lane = threadIdx.x, nthr = blockDim.x;
  const ProveDesc d = desc[p];
  ProveState &st = ps[p];
  const uint32_t mn = d.m * n_bits;
  sc *a = vec + (size_t)p * KP_VEC_LEN(mn), *b = a + mn, *cG = b + mn, *cH = cG + mn, *ypow = cH + mn;
  sc *fG = ypow + mn + 2, *fH = fG + KP_EX_CLASSES;  // ("ct" = 2, see below)
  sc one;
  sc_mont_one(one);
  KP_T0();

  if (j == 0) {
    // y powers (:353-359) by lane-strided exponentiation, d (:362-373), a_L - z, a_R + d*y^(mn-i) + z (:376-381)
    const sc y = st.y, z = st.z;
    sc z_square;
    sc_montsq(z_square, z);
    for (uint32_t i = lane; i < mn + 2; i += nthr) {
      sc v;
      sc_mont_pow_u32(v, y, i);
      ypow[i] = v;
    }
    __syncthreads();
    for (uint32_t i = lane; i < mn; i += nthr) {
      const uint32_t party = i / n_bits, bit_idx = i % n_bits;
      const uint8_t *w = bytes + d.wit_off + party * (8 + 32 * t);
      uint64_t v = 0;
      for (int k = 0; k < 8; k++) v |= (uint64_t)w[k] << (8 * k);
      if (min_present[d.minval_idx + party]) v -= minvals[d.minval_idx + party];
      const bool bit = (v >> bit_idx) & 1ULL;
      sc al, ar, di, two_k, u;
      if (bit) {
        al = one;
        sc_0(ar);
      } else {
        sc_0(al);
        sc_neg(ar, one);
      }
      di = z_square;
      for (uint32_t q = 0; q < party; q++) sc_montmul(di, di, z_square);
      sc_mont_from_u64(two_k, 1ULL << bit_idx);
      sc_montmul(di, di, two_k);
      sc_sub(al, al, z);
      sc_montmul(u, di, ypow[mn - i]);
      sc_add(u, u, z);
      sc_add(ar, ar, u);
      a[i] = al;
      b[i] = ar;
      cG[i] = one;
      cH[i] = one;
    }
    // alpha[k] += sum_j z^(2(j+1)) * r_{j,k} * y^(mn+1)  (:382-392)
    __syncthreads();
    if (lane < t) {
      sc acc = st.alpha[lane], zp = one;
      const sc ymn1 = ypow[mn + 1];
      for (uint32_t party = 0; party < d.m; party++) {
        sc rjk, u;
        sc_montmul(zp, zp, z_square);
        sc_load_mont(rjk, bytes + d.wit_off + party * (8 + 32 * t) + 8 + 32 * lane);
        sc_montmul(u, zp, rjk);
        sc_montmul(u, u, ymn1);
        sc_add(acc, acc, u);
      }
      st.alpha[lane] = acc;
    }
    KP_MARK(13);
  } else {
    // fold with e_{j-1} (:511-533): len = mn >> (j-1)
    const uint32_t len = mn >> (j - 1), nh = len >> 1;
    const sc e = st.e, einv = st.einv, yinv = st.yinv_prev;
    sc yn, e_yinv;
    yn = ypow[nh];
    sc_montmul(e_yinv, e, yinv);
    // a' = a_lo*e + (a_hi*y^n)*e^-1 ; b' = b_lo*e^-1 + b_hi*e   (each lane reads both halves before anyone writes)
    for (uint32_t base = 0; base < nh; base += nthr) {
      const uint32_t i = base + lane;
      sc na, nb;
      if (i < nh) {
        sc u, v;
        sc_montmul(u, a[i], e);
        sc_montmul(v, a[nh + i], yn);
        sc_montmul(v, v, einv);
        sc_add(na, u, v);
        sc_montmul(u, b[i], einv);
        sc_montmul(v, b[nh + i], e);
        sc_add(nb, u, v);
      }
      __syncthreads();
      if (i < nh) {
        a[i] = na;
        b[i] = nb;
      }
    }
    KP_MARK(8);
    for (uint32_t u = lane; u < mn; u += nthr) {
      const bool lo = (u & (len - 1)) < nh;
      sc g = cG[u], h = cH[u], fg, fh;
      // (word-wise selects: `lo ? einv : e_yinv` as an operand selects between the ADDRESSES of two locals, which then live in
      // scratch memory)
#pragma unroll
      for (int q = 0; q < 8; q++) {
        fg.v[q] = lo ? einv.v[q] : e_yinv.v[q];
        fh.v[q] = lo ? e.v[q] : einv.v[q];
      }
      sc_montmul(g, g, fg);
      sc_montmul(h, h, fh);
      cG[u] = g;
      cH[u] = h;
      // "ct" = 2: the same factors, from step rounds - ex_back on, for the classes u < 2^ex_back alone: what Gf[0] / Hf[0] take
      // over the public points made at that step (below)
      if (ex_scal && j + ex_back > rounds && u < (1u << ex_back)) {
        sc x = fG[u], y2 = fH[u];
        sc_montmul(x, x, fg);
        sc_montmul(y2, y2, fh);
        fG[u] = x;
        fH[u] = y2;
      }
    }
    KP_MARK(9);
  }
  __syncthreads();

  sc *ts = term_scal + (size_t)p * 2 * stride;
  uint32_t *tg = term_gidx + (size_t)p * 2 * stride;
  if (j < rounds) {
    // round j: len = mn >> j
    const uint32_t len = mn >> j, nh = len >> 1;
    const sc yinv = st.yinv_nhalf, yn = ypow[nh];
    // c_L = sum a_lo[i] y^(i+1) b_hi[i];  c_R = sum a_hi[i] y^(nh+1+i) b_lo[i]   (:468-479)
    sc cl, cr;
    sc_0(cl);
    sc_0(cr);
    for (uint32_t i = lane; i < nh; i += nthr) {
      sc u;
      sc_montmul(u, a[i], ypow[i + 1]);
      sc_montmul(u, u, b[nh + i]);
      sc_add(cl, cl, u);
      sc_montmul(u, a[nh + i], ypow[nh + 1 + i]);
      sc_montmul(u, u, b[i]);
      sc_add(cr, cr, u);
    }
    for (int pass = 0; pass < 2; pass++) {
      {
        sc pick;
#pragma unroll
        for (int q = 0; q < 8; q++) pick.v[q] = pass ? cr.v[q] : cl.v[q];
        red[lane] = pick;
      }
      __syncthreads();
      for (uint32_t off = nthr / 2; off >= 1; off >>= 1) {
        if (lane < off) {
          sc x = red[lane], y2 = red[lane + off];
          sc_add(x, x, y2);
          red[lane] = x;
        }
        __syncthreads();
      }
      if (pass) cr = red[0];
      else cl = red[0];
      __syncthreads();
    }
    KP_MARK(10);
    // term lists.  L (:482-488): c_L H, d_L G_k, (a_lo y^-n) on Gf_hi, b_hi on Hf_lo.  R (:489-495) symmetric.
    // output 0 = L, output 1 = R; every original G_u / H_u goes to exactly one of them.
    for (uint32_t u = lane; u < mn; u += nthr) {
      const uint32_t fi = u & (len - 1);
      const bool lo = fi < nh;
      const uint32_t i = lo ? fi : fi - nh;
      sc sg, sh;
      if (lo) {  // G_u in R with a_hi*y^n ; H_u in L with b_hi
        sc_montmul(sg, a[nh + i], yn);
        sc_montmul(sg, sg, cG[u]);
        sc_montmul(sh, b[nh + i], cH[u]);
      } else {  // G_u in L with a_lo*y^-n ; H_u in R with b_lo
        sc_montmul(sg, a[i], yinv);
        sc_montmul(sg, sg, cG[u]);
        sc_montmul(sh, b[i], cH[u]);
      }
      // (left in Montgomery form: the fixed-base MSM converts while it recodes)
      // position within the output's row: the u-th G term and u-th H term of each half, packed densely
      const uint32_t rank = (u / len) * nh + i;  // index among the mn/2 generators of this kind in this output
      const uint32_t og = lo ? 1u : 0u, oh = lo ? 0u : 1u;
      ts[og * stride + rank] = sg;
      tg[og * stride + rank] = 2 * u;
      ts[oh * stride + (mn >> 1) + rank] = sh;
      tg[oh * stride + (mn >> 1) + rank] = 2 * u + 1;
    }
    if (lane <= t) {
      // lanes 0..t-1: d_L[k] G_k / d_R[k] G_k ; lane t: c_L H / c_R H
      sc sl, sr;
      if (lane < t) {
        sl = st.dl[lane];
        sr = st.dr[lane];
      } else {
        sl = cl;
        sr = cr;
      }
      ts[0 * stride + mn + lane] = sl;
      tg[0 * stride + mn + lane] = n_gen + lane;
      ts[1 * stride + mn + lane] = sr;
      tg[1 * stride + mn + lane] = n_gen + lane;
    }
    if (lane == 0) {
      term_count[2 * p] = mn + t + 1;
      term_count[2 * p + 1] = mn + t + 1;
    }
    if (ex_scal && j + ex_back == rounds) {
      // "ct" = 2, ex_back rounds before the end (the vectors have 2^ex_back elements left): the folded generators of the final step
      // are Gf[0] = sum_c fG[c] GE_c and Hf[0] = sum_c fH[c] HE_c over the classes c = u mod 2^ex_back, with GE_c (HE_c) the sum of
      // cG[u] G_u (cH[u] H_u) over the class as the coefficients stand NOW -- PUBLIC points, functions of the challenges so far --
      // and fG[c], fH[c] the products of the remaining rounds' fold factors (updated beside cG / cH above).  The 2 x 2^ex_back
      // points are more outputs of this round's fixed-base MSM (mn / 2^ex_back terms each: as much work as a round's L and R);
      // their multiples by 16^w are made while the remaining rounds run (ct.h: k_ct_pow16), so that the secret scalars r fG[c],
      // s fH[c] find everything they need when they exist (k_ct_var).
      const uint32_t nc = 1u << ex_back, hn = mn >> ex_back;
      sc *es = ex_scal + (size_t)p * 2 * mn;
      uint32_t *eg = ex_gidx + (size_t)p * 2 * mn;
      for (uint32_t u = lane; u < mn; u += nthr) {
        const uint32_t c = u & (nc - 1), r2 = u >> ex_back;
        es[c * hn + r2] = cG[u];  // (Montgomery form, as every fixed-base term list)
        eg[c * hn + r2] = 2 * u;
        es[(nc + c) * hn + r2] = cH[u];
        eg[(nc + c) * hn + r2] = 2 * u + 1;
      }
      if (lane < 2 * nc) ex_count[2 * nc * p + lane] = hn;
      if (lane < nc) {
        fG[lane] = one;
        fH[lane] = one;
      }
    }
    KP_MARK(11);
  } else {
    // final step (:574-584): A1 = r Gf[0] + s Hf[0] + (r y b + s y a) H + sum d_k G_k ;  B = (r y s) H + sum eta_k G_k
    const sc r = st.r, s = st.s, y = st.y;
    if (ct_scal) {
      // The reference multiplies by r, s, d_k, eta_k and the H scalars in constant time (`&P * Scalar`, :574-584): every scalar of
      // A1 and B is a secret.  None of them reaches a fixed-base table here (whose addresses would be their digits):
      //   row 2p   (k_ct_fixed):  (r y b + s y a) H, d_k G_k      -> A1's part over the Pedersen bases
      //   row 2p+1 (k_ct_fixed):  (r y s) H, eta_k G_k            -> B
      //   words 8..15 of row 2p, then of row 2p+1 (k_ct_var): r fG[c], then s fH[c], over the public points GE_c, HE_c of step
      //   rounds - ex_back (see there)
      // and this step has no fixed-base MSM at all.
      sc *fs = ct_scal + (size_t)p * 2 * CT_ROW;
      uint32_t *fi = ct_idx + (size_t)p * 2 * CT_ROW;
      if (lane <= t) {
        sc s1v, s2v;
        if (lane < t) {
          s1v = st.dd[lane];
          s2v = st.eta[lane];
        } else {
          sc u, v;
          sc_montmul(u, r, y);
          sc_montmul(u, u, b[0]);
          sc_montmul(v, s, y);
          sc_montmul(v, v, a[0]);
          sc_add(s1v, u, v);
          sc_montmul(s2v, r, y);
          sc_montmul(s2v, s2v, s);
        }
        sc_from_mont(s1v, s1v);
        sc_from_mont(s2v, s2v);
        // lane t carries the H terms (position 0 of either row), lanes k < t the G_k terms
        const uint32_t pos = lane < t ? 1 + lane : 0;
        fs[pos] = s1v;
        fi[pos] = n_gen + lane;
        fs[CT_ROW + pos] = s2v;
        fi[CT_ROW + pos] = n_gen + lane;
      }
      {  // the secret factors of the public points: r fG[c] and s fH[c], 2^ex_back each, eight slots from word 8 of either row
        const uint32_t nc = 1u << ex_back;
        if (lane < 2 * nc) {
          const bool hside = lane >= nc;
          sc v, f = hside ? fH[lane - nc] : fG[lane];
#pragma unroll
          for (int q = 0; q < 8; q++) v.v[q] = hside ? s.v[q] : r.v[q];  // (word-wise: no addresses of locals)
          sc_montmul(v, v, f);
          sc_from_mont(v, v);
          fs[(lane >> 3) * CT_ROW + 8 + (lane & 7u)] = v;
        }
      }
      if (lane == 0) {
        ct_count[2 * p] = 1 + t;
        ct_count[2 * p + 1] = 1 + t;
      }
      KP_MARK(12);
      return;
    }
    // THREE outputs per proof in the last launch: A1 = A1g + A1h with A1g = r Gf[0] (mn terms), A1h = s Hf[0] + the Pedersen-base
    // terms (mn + t + 1), and B (t + 1).  As ONE 2 mn + t + 1-term output A1 was twice as long as any output of a round and sat
    // beside a B of four terms: half as many busy workgroups, each twice as long -- the launch took 460-470 us against the
    // ~300 us of a round with as many additions (profiles/r04_v6_prover_launches.txt), on the call's last stretch.  The rows
    // (FINAL_ROW(mn, t) = mn + t + 1 terms each) are packed from the start of the term arrays -- 3 (mn + t + 1) <= 2 (2 mn + t + 1)
    // per proof -- which is safe because no workgroup reads a term row in this step; kp_final_points adds the halves.
    const uint32_t fstride = mn + t + 1;
    sc *f_s = term_scal + (size_t)3 * p * fstride;
    uint32_t *f_g = term_gidx + (size_t)3 * p * fstride;
    for (uint32_t u = lane; u < mn; u += nthr) {
      sc sg, sh;
      sc_montmul(sg, r, cG[u]);
      sc_montmul(sh, s, cH[u]);
      f_s[u] = sg;
      f_g[u] = 2 * u;
      f_s[fstride + u] = sh;
      f_g[fstride + u] = 2 * u + 1;
    }
    if (lane <= t) {
      sc s1v, s2v;
      if (lane < t) {
        s1v = st.dd[lane];
        s2v = st.eta[lane];
      } else {
        sc u, v;
        sc_montmul(u, r, y);
        sc_montmul(u, u, b[0]);
        sc_montmul(v, s, y);
        sc_montmul(v, v, a[0]);
        sc_add(s1v, u, v);
        sc_montmul(s2v, r, y);
        sc_montmul(s2v, s2v, s);
      }
      f_s[fstride + mn + lane] = s1v;
      f_g[fstride + mn + lane] = n_gen + lane;
      f_s[2 * fstride + lane] = s2v;
      f_g[2 * fstride + lane] = n_gen + lane;
    }
    if (lane == 0) {
      term_count[3 * p] = mn;
      term_count[3 * p + 1] = mn + t + 1;
      term_count[3 * p + 2] = t + 1;
    }
    KP_MARK(12);
  }
}

// The two steps as kernels of their own (prove_fused = 0; tests run both forms) ...
__global__ void __launch_bounds__(64) kp_lane(const uint8_t *__restrict__ bytes, const ProveDesc *__restrict__ desc, uint32_t n_bits,
                                              uint32_t t, uint32_t B, uint32_t j, uint32_t rounds, const uint8_t *__restrict__ a32,
                                              const uint8_t *lr32, ProveState *ps) {
  __shared__ ProveLds L;
  kp_lane_body(bytes, desc, n_bits, t, B, j, rounds, a32, lr32, ps, L);
  lds_wipe(L);
}
__global__ void __launch_bounds__(64) kp_wave(const uint8_t *__restrict__ bytes, const ProveDesc *__restrict__ desc,
                                              const uint64_t *__restrict__ minvals, const uint8_t *__restrict__ min_present, uint32_t n_bits,
                                              uint32_t t, uint32_t n_gen, uint32_t j, uint32_t rounds, uint32_t stride, ProveState *ps,
                                              sc *__restrict__ vec, sc *__restrict__ term_scal, uint32_t *__restrict__ term_gidx,
                                              uint32_t *__restrict__ term_count, sc *__restrict__ ct_scal, uint32_t *__restrict__ ct_idx,
                                              uint32_t *__restrict__ ct_count, sc *__restrict__ ex_scal, uint32_t *__restrict__ ex_gidx,
                                              uint32_t *__restrict__ ex_count, uint32_t ex_back) {
  __shared__ sc red[64];
  kp_wave_body(bytes, desc, minvals, min_present, n_bits, t, n_gen, j, rounds, stride, ps, vec, term_scal, term_gidx, term_count, ct_scal,
               ct_idx, ct_count, ex_scal, ex_gidx, ex_count, ex_back, red);
  lds_wipe(red);
}
// ... and as ONE launch per round (round 4): the encoding of the previous round's L and R (two lanes, ristretto_compress), the
// Fiat-Shamir step and the vector step of a proof are consecutive phases of the same 64-lane workgroup.  As three launches per
// round they were three latency-bound kernels of a few wavefronts each, every one of them queueing for wave slots behind the
// other sub-batch's chip-filling fixed-base MSM (70 / 135 / 45 us alone, 70-200 / 130-180 / 45-175 us in a call:
// profiles/r04_prover_launches.txt); whatever a phase writes to memory for the next one is read by the same workgroup behind a
// barrier.  ge_prev == null: nothing to encode (round 0).
template <int W>
__global__ void __launch_bounds__(64 * W) kp_round(const uint8_t *__restrict__ bytes, const ProveDesc *__restrict__ desc,
                                               const uint64_t *__restrict__ minvals, const uint8_t *__restrict__ min_present, uint32_t n_bits,
                                               uint32_t t, uint32_t n_gen, uint32_t B, uint32_t j, uint32_t rounds, uint32_t stride,
                                               const uint8_t *__restrict__ a32, const ge *__restrict__ ge_prev, uint32_t prev_parts,
                                               uint8_t *lr_prev, ProveState *ps, sc *__restrict__ vec, sc *__restrict__ term_scal,
                                               uint32_t *__restrict__ term_gidx, uint32_t *__restrict__ term_count,
                                               sc *__restrict__ ct_scal, uint32_t *__restrict__ ct_idx, uint32_t *__restrict__ ct_count,
                                               sc *__restrict__ ex_scal, uint32_t *__restrict__ ex_gidx, uint32_t *__restrict__ ex_count,
                                               uint32_t ex_back) {
  // blockDim.x = 64 x (1, 2 or 4) wavefronts per proof (option "prove_waves").  With two or more: L and R are summed and encoded
  // on a wavefront each, the Fiat-Shamir step runs its two halves side by side (kp_lane_body2), the vector step strides by the
  // whole workgroup.  One wavefront: the same three phases in a row (the form of round 4; tests run every form).
  static_assert(W == 1 || W == 2 || W == 4, "wavefronts per proof");
  const uint32_t p = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  constexpr bool multi = W >= 2;  // (an instantiation per form: each carries only its own Fiat-Shamir step)
  if (p >= B) return;
  // The three phases use LDS one after the other -- the points being summed (10 or 20 KB), the sponges of the Fiat-Shamir step,
  // the vector step's reduction buffer -- and SHARE it: what this workgroup holds decides how many of them fit on a compute unit
  // beside the other sub-batch's fixed-base MSM (a dozen 9 KB workgroups per CU): at 32 KB it was one, and the 512 workgroups of a
  // launch ran in two turns.
  union KpShared {
    ge red[multi ? 128 : 64];
    ProveLds L;
    sc red_sc[64 * W];
  };
  __shared__ KpShared sh;
  ge *red = sh.red;
  KP_T0();
  if (ge_prev) {
    // prev_parts == 0: ge_prev holds the previous round's L and R as points; otherwise their slices' partial sums (k_fb_part),
    // summed here first.
    if constexpr (multi) {
      if (wave < 2) {
        if (prev_parts) fb_reduce_one(red + 64 * wave, ge_prev, prev_parts, 2 * p + wave);
        else if (lane == 0) red[64 * wave] = ge_prev[2 * (size_t)p + wave];
      }
    } else if (prev_parts) {
      fb_reduce_pair(red, ge_prev, prev_parts, p);
    } else if ((lane & 31u) == 0) {
      red[lane] = ge_prev[2 * (size_t)p + (lane >> 5)];
    }
    KP_MARK(14);
    // the encodings: lane 0 of the first two wavefronts, or lanes 0 and 32 of the only one
    const bool enc = multi ? (wave < 2 && lane == 0) : ((lane & 31u) == 0);
    if (enc) {
      const uint32_t which = multi ? wave : (lane >> 5);
      uint8_t c32[32];
      ristretto_compress(c32, red[multi ? 64 * wave : lane]);
      uint32_t *o = (uint32_t *)(lr_prev + (size_t)p * 64 + 32 * which);
#pragma unroll
      for (int k = 0; k < 8; k++)
        o[k] = (uint32_t)c32[4 * k] | ((uint32_t)c32[4 * k + 1] << 8) | ((uint32_t)c32[4 * k + 2] << 16) | ((uint32_t)c32[4 * k + 3] << 24);
    }
  }
  __syncthreads();
  KP_MARK(0);
  if constexpr (multi) {
    kp_lane_body2(bytes, desc, n_bits, t, B, j, rounds, a32, lr_prev, ps, sh.L);
  } else {
    kp_lane_body(bytes, desc, n_bits, t, B, j, rounds, a32, lr_prev, ps, sh.L);
    __syncthreads();
  }
  kp_wave_body(bytes, desc, minvals, min_present, n_bits, t, n_gen, j, rounds, stride, ps, vec, term_scal, term_gidx, term_count, ct_scal,
               ct_idx, ct_count, ex_scal, ex_gidx, ex_count, ex_back, sh.red_sc);
  lds_wipe(sh);
}

// ---- final lane kernel: challenge_final_e, responses, wire bytes (:587-607, to_bytes :1120-1150) ----
// proof layout: [t] d1[t] A A1 B r1 s1 (L_j R_j)...
__global__ void __launch_bounds__(64) kp_finish(const ProveDesc *__restrict__ desc, uint32_t n_bits, uint32_t t, uint32_t B,
                                                uint32_t rounds, const uint8_t *__restrict__ a32,
                                                const uint8_t *__restrict__ lr_all /* [rounds][B][2][32] */,
                                                const uint8_t *__restrict__ a1b32 /* [B][2][32] */, const sc *__restrict__ vec,
                                                ProveState *__restrict__ ps, uint8_t *__restrict__ proofs, uint32_t proof_stride) {
  // one wavefront per proof (round 4; before: one lane per proof on the one-lane sponge, three Keccak-f of ~26 us each on the
  // call's last stretch with nothing beside them): the transcript's last three operations on the cooperative sponge of the
  // Fiat-Shamir steps, then lane k < t makes d1_k, lanes t and t + 1 make r1 and s1, and the wire bytes are written by all lanes
  const uint32_t p = blockIdx.x, lane = threadIdx.x;
  if (p >= B) return;
  __shared__ ProveLds L;
  const KeccakLanes K = keccak_lanes(L.wk[0]);
  ProveState &st = ps[p];
  const uint32_t mn = desc[p].m * n_bits;
  const sc *a = vec + (size_t)p * KP_VEC_LEN(mn), *b = a + mn;
  WStrobe tr;
  ws_load(tr, L.tr, st.tr);
  const uint8_t *pa1 = a1b32 + (size_t)p * 64;
  bool ok = pw_validate_append(tr, K, "A1", 2, pa1);
  ok = pw_validate_append(tr, K, "B", 1, pa1 + 32) && ok;
  sc e;
  ok = pw_challenge(tr, L, K, "e", 1, e) && ok;
  uint8_t *o = proofs + (size_t)p * proof_stride;
  // wire format (src/range_proof.rs:1120-1150): [t] d1[t] A A1 B r1 s1 (L_j R_j)_j
  if (lane < t + 2) {
    sc x, u;
    if (lane < t) {  // d1_k = eta_k + d_k e + alpha_k e^2
      sc esq, v;
      sc_montsq(esq, e);
      sc_montmul(v, st.dd[lane], e);
      sc_add(x, st.eta[lane], v);
      sc_montmul(v, st.alpha[lane], esq);
      sc_add(x, x, v);
    } else if (lane == t) {  // r1 = r + a e
      sc_montmul(u, a[0], e);
      sc_add(x, st.r, u);
    } else {  // s1 = s + b e
      sc_montmul(u, b[0], e);
      sc_add(x, st.s, u);
    }
    sc_from_mont(x, x);
    uint8_t tmp[32];
    sc_store_words(tmp, x);
    uint8_t *dst = o + 1 + (lane < t ? 32 * lane : 32 * t + 96 + 32 * (lane - t));
    for (int i = 0; i < 32; i++) dst[i] = tmp[i];
  }
  if (lane == 0) o[0] = (uint8_t)t;
  for (uint32_t i = lane; i < 96; i += 64)  // A | A1 | B
    o[1 + 32 * t + i] = i < 32 ? a32[(size_t)p * 32 + i] : a1b32[(size_t)p * 64 + (i - 32)];
  uint8_t *olr = o + 1 + 32 * t + 96 + 64;
  for (uint32_t i = lane; i < 64 * rounds; i += 64) olr[i] = lr_all[((size_t)(i >> 6) * B + p) * 64 + (i & 63u)];
  if (!ok && lane == 0) st.status |= PV_STATUS_TRANSCRIPT;
  lds_wipe(L);
}

// the last launch's three outputs per proof (kp_wave_body, final step) -> the encodings of A1 = A1g + A1h and of B, a1b32[p][2][32]
__global__ void __launch_bounds__(64) kp_final_points(const ge *__restrict__ g3, uint32_t B, uint8_t *__restrict__ a1b32) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * B) return;
  const uint32_t p = i >> 1, which = i & 1u;
  ge x = g3[3 * (size_t)p + (which ? 2 : 0)];
  if (!which) {
    const ge y2 = g3[3 * (size_t)p + 1];
    ge_add(x, x, y2);
  }
  uint8_t c32[32];
  ristretto_compress(c32, x);
  uint32_t *o = (uint32_t *)(a1b32 + (size_t)i * 32);
#pragma unroll
  for (int k = 0; k < 8; k++)
    o[k] = (uint32_t)c32[4 * k] | ((uint32_t)c32[4 * k + 1] << 8) | ((uint32_t)c32[4 * k + 2] << 16) | ((uint32_t)c32[4 * k + 3] << 24);
}

// commitment check (:275-284): compare the engine's commit(v_j, r_j) with the statement's commitments
__global__ void kp_check_commitments(const uint8_t *__restrict__ bytes, const ProveDesc *__restrict__ desc,
                                     const uint8_t *__restrict__ computed32 /* [B][m][32] */, uint32_t B, ProveState *__restrict__ ps) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= B) return;
  const ProveDesc d = desc[p];
  uint32_t diff = 0;
  for (uint32_t i = 0; i < 32 * d.m; i++) diff |= (uint32_t)(bytes[d.commit_off + i] ^ computed32[(size_t)p * 32 * d.m + i]);
  if (diff) ps[p].status |= PV_STATUS_COMMIT_MISMATCH;
}

// term lists for the commitment check: output (p, j) = v_j H + sum_k r_{j,k} G_k
__global__ void kp_commit_terms(const uint8_t *__restrict__ bytes, const ProveDesc *__restrict__ desc, uint32_t t, uint32_t n_gen,
                                uint32_t B, uint32_t m, uint32_t stride, sc *__restrict__ ts, uint32_t *__restrict__ tg,
                                uint32_t *__restrict__ tc) {
  const uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= B * m) return;
  const uint32_t p = o / m, j = o % m;
  const uint8_t *w = bytes + desc[p].wit_off + j * (8 + 32 * t);
  sc v;
  sc_0(v);
  v.v[0] = (uint32_t)w[0] | ((uint32_t)w[1] << 8) | ((uint32_t)w[2] << 16) | ((uint32_t)w[3] << 24);
  v.v[1] = (uint32_t)w[4] | ((uint32_t)w[5] << 8) | ((uint32_t)w[6] << 16) | ((uint32_t)w[7] << 24);
  ts[(size_t)o * stride] = v;
  tg[(size_t)o * stride] = n_gen + t;
  for (uint32_t k = 0; k < t; k++) {
    sc r;
    sc_load_words(r, w + 8 + 32 * k);
    ts[(size_t)o * stride + 1 + k] = r;
    tg[(size_t)o * stride + 1 + k] = n_gen + k;
  }
  tc[o] = 1 + t;
}

}  // namespace bpp
