// Uniform-access scalar multiplication for the terms whose scalars are SECRET AND NOTHING ELSE, where the reference itself is
// constant-time:
//   PedersenGens::commit            src/generators/pedersen_gens.rs:112-122   P::multiscalar_mul (dalek's constant-time Straus)
//   the prover's witness check      src/range_proof.rs:275-284                (calls commit)
//   A1 and B of the final round     src/range_proof.rs:572-584                `&P * Scalar` (constant-time variable-base mult)
// Everything else on the prover's path is variable-time in the reference as well (A: vartime_mixed_multiscalar_mul :339-345;
// every L and R: vartime_multiscalar_mul :482-495) and runs through the engine's fixed-base tables, whose ADDRESSES are digits of
// witness-derived scalars (kernels_prove.h: k_fb_msm, kp_A) -- DESIGN.md, "What is variable-time in secrets".
//
// The form (the one dalek uses for those calls, restated): signed radix-16 digits of the scalar, a table of the multiples
// 0 P .. 8 P per term, and for every digit position four doublings and ONE addition of the entry the digit names -- found by
// reading ALL nine entries and keeping one under an arithmetic mask, the sign applied by a select.  No branch and no address
// depends on the scalar: the sequence of table reads is the same for every scalar (tests/test_host_arith.py records it through
// BPP_CT_TOUCH in a host build and compares two scalars' traces), and so is the instruction stream.
//
// On the device one term runs on one QUAD (msm.h: lane q of the quad holds coordinate q of the accumulator, a doubling or an
// addition is two rounds of four field products), a workgroup of 64 lanes sums up to 16 terms of one output.
#pragma once
#include "point.h"
#include "scalar.h"
#if defined(__HIPCC__)
#include "msm.h"  // the quad forms of the point operations
#endif

namespace bpp {

#define BPP_CT_DIGITS 64
#define BPP_CT_ENTRIES 9  // 0 P .. 8 P

#ifndef BPP_CT_TOUCH
#define BPP_CT_TOUCH(entry) \
  do {                      \
  } while (0)
#endif

// signed radix-16 digits of a canonical scalar (< 2^253): 64 digits, d_0 .. d_62 in [-8, 8), d_63 in [0, 2].  Branch-free.
BPP_HD void ct_recode16(int8_t d[BPP_CT_DIGITS], const sc &s) {
  int32_t carry = 0;
#pragma unroll
  for (int i = 0; i < BPP_CT_DIGITS; i++) {
    int32_t v = (int32_t)((s.v[i >> 3] >> (4 * (i & 7))) & 15u) + carry;
    carry = (i + 1 < BPP_CT_DIGITS) ? ((v + 8) >> 4) : 0;
    v -= carry << 4;
    d[i] = (int8_t)v;
  }
}

// out[0..N) = the row that holds multiple `mag`, rows = multiples first, first + 1, .., first + E - 1 (zero when mag is none of
// them): EVERY row is read, one is kept.  `table` = E rows of `row_stride` words.
template <int N, int E = BPP_CT_ENTRIES>
BPP_HD void ct_select_words(uint32_t (&out)[N], const uint32_t *table, uint32_t row_stride, uint32_t mag, uint32_t first = 0) {
#pragma unroll
  for (int k = 0; k < N; k++) out[k] = 0;
#pragma unroll
  for (uint32_t j = 0; j < (uint32_t)E; j++) {
    BPP_CT_TOUCH(j);
    const uint32_t keep = 0u - ((((j + first) ^ mag) - 1u) >> 31);  // all ones iff j + first == mag (the xor is < 2^31)
#pragma unroll
    for (int k = 0; k < N; k++) out[k] |= table[(size_t)j * row_stride + k] & keep;
  }
}

// |d| and the sign of a digit without a branch
BPP_HD void ct_digit_parts(uint32_t &mag, uint32_t &neg, int32_t d) {
  const int32_t m = d >> 31;  // 0 or -1
  mag = (uint32_t)((d ^ m) - m);
  neg = (uint32_t)m & 1u;
}

// h = neg ? -f : f by word-wise selects (f reduced)
BPP_HD void fe_cneg_select(fe &h, const fe &f, uint32_t neg) {
  fe n;
  fe_neg(n, f);
  const uint32_t keep = 0u - neg;
#pragma unroll
  for (int i = 0; i < 10; i++) h.v[i] = (n.v[i] & keep) | (f.v[i] & ~keep);
}

// ---- the whole multiplication on ONE lane: the host probe's form (and the definition the quad form is held to) ----
BPP_HD void ct_scalarmul(ge &r, const ge &p, const sc &s) {
  ge tab[BPP_CT_ENTRIES];
  ge_identity(tab[0]);
  tab[1] = p;
  for (int j = 2; j < BPP_CT_ENTRIES; j++) ge_add(tab[j], tab[j - 1], p);
  // entries with carried limbs: what ct_select_words ORs together must be valid limbs of ONE entry
  for (int j = 0; j < BPP_CT_ENTRIES; j++) {
    fe_carry(tab[j].X);
    fe_carry(tab[j].Y);
    fe_carry(tab[j].Z);
    fe_carry(tab[j].T);
  }
  int8_t d[BPP_CT_DIGITS];
  ct_recode16(d, s);
  ge acc;
  ge_identity(acc);
  for (int i = BPP_CT_DIGITS - 1; i >= 0; i--) {
    if (i != BPP_CT_DIGITS - 1)
      for (int k = 0; k < 4; k++) ge_dbl(acc, acc);
    uint32_t mag, neg;
    ct_digit_parts(mag, neg, d[i]);
    uint32_t w[40];
    ct_select_words<40>(w, (const uint32_t *)tab, 40, mag);
    ge q;
    for (int k = 0; k < 10; k++) {
      q.X.v[k] = w[k];
      q.Y.v[k] = w[10 + k];
      q.Z.v[k] = w[20 + k];
      q.T.v[k] = w[30 + k];
    }
    fe_cneg_select(q.X, q.X, neg);
    fe_cneg_select(q.T, q.T, neg);
    ge_add(acc, acc, q);
  }
  r = acc;
}

// ---- FIXED bases (the Pedersen bases H, G_k of a parameter set): no doublings at all.  Table line [base][position w][j - 1] =
// j * 16^w * Base as an affine Niels entry, j = 1..8 (64 positions x 8 lines x 128 B = 64 KB per base, built once per parameter
// set); digit position w of a term contributes the line its digit names -- all eight lines of the position are read, one is kept
// under a mask, a zero digit keeps none and becomes the neutral entry (1, 1, 0), the sign exchanges y+x / y-x and negates 2dxy by
// selects -- and the 64 positions are 64 independent mixed additions.
#define BPP_CTF_ENTRIES 8
#define BPP_CTF_LINE_WORDS 32  // one 128-byte line
BPP_HD void ct_fixed_select(niels &q, const niels *position_lines, int32_t digit) {
  uint32_t mag, neg;
  ct_digit_parts(mag, neg, digit);
  uint32_t w[30];
  ct_select_words<30, BPP_CTF_ENTRIES>(w, (const uint32_t *)position_lines, BPP_CTF_LINE_WORDS, mag, 1);
  const uint32_t isz = (mag - 1u) >> 31;  // 1 iff the digit is zero: the neutral entry
  w[0] |= isz;
  w[10] |= isz;
  const uint32_t sw = 0u - neg;
  fe t2d;
#pragma unroll
  for (int k = 0; k < 10; k++) {
    const uint32_t a = w[k], b = w[10 + k];
    q.yplusx.v[k] = (b & sw) | (a & ~sw);
    q.yminusx.v[k] = (a & sw) | (b & ~sw);
    t2d.v[k] = w[20 + k];
  }
  fe_cneg_select(q.xy2d, t2d, neg);
}
// one lane's model of the whole sum (host probe): acc = sum_w line(w, digit_w)
BPP_HD void ct_fixed_scalarmul(ge &r, const niels *base_lines /* [64][8] */, const sc &s) {
  int8_t d[BPP_CT_DIGITS];
  ct_recode16(d, s);
  ge acc;
  ge_identity(acc);
  for (int w = 0; w < BPP_CT_DIGITS; w++) {
    niels q;
    ct_fixed_select(q, base_lines + (size_t)w * BPP_CTF_ENTRIES, d[w]);
    ge_madd(acc, acc, q);
  }
  r = acc;
}

// ---- VARIABLE bases known before their secret scalars (the prover's A1: r Gf[0] + s Hf[0], src/range_proof.rs:574-576): the
// 63 x 4 doublings of the ladder above are doublings of a PUBLIC point -- they need not wait for the scalar, and they need not be
// uniform.  With pow[w] = 16^w P made ahead (k_ct_pow16), digit position w contributes d_w * pow[w], d_w in [-8, 8]: the lane of
// position w walks 1 pow[w], 2 pow[w], .. 8 pow[w] by seven additions and keeps the one its digit names under an arithmetic mask
// (all eight are computed whatever the digit; zero keeps the neutral element), the sign is a select, and the 64 positions are
// summed by a tree.  No table, no address and no branch that depends on the scalar; what follows the scalar on the call's
// critical path is 7 + 1 additions and a 6-level tree instead of 252 doublings and 64 additions.
BPP_HD void ct_pos_multiple(ge &out, const ge &pw, int32_t digit) {
  uint32_t mag, neg;
  ct_digit_parts(mag, neg, digit);
  ge m = pw, sel;
  ge_identity(sel);
#pragma unroll 1
  for (uint32_t k = 1; k <= 8; k++) {
    if (k > 1) ge_add(m, m, pw);
    ge c = m;  // (carried limbs: what the mask keeps must be valid limbs of ONE multiple)
    fe_carry(c.X);
    fe_carry(c.Y);
    fe_carry(c.Z);
    fe_carry(c.T);
    const uint32_t keep = 0u - (((k ^ mag) - 1u) >> 31);  // all ones iff k == mag
#pragma unroll
    for (int q = 0; q < 10; q++) {
      sel.X.v[q] = (c.X.v[q] & keep) | (sel.X.v[q] & ~keep);
      sel.Y.v[q] = (c.Y.v[q] & keep) | (sel.Y.v[q] & ~keep);
      sel.Z.v[q] = (c.Z.v[q] & keep) | (sel.Z.v[q] & ~keep);
      sel.T.v[q] = (c.T.v[q] & keep) | (sel.T.v[q] & ~keep);
    }
  }
  fe_cneg_select(sel.X, sel.X, neg);
  fe_cneg_select(sel.T, sel.T, neg);
  out = sel;
}
// one lane's model of the whole product (host probe): pow[w] by doublings, then the sum over the positions
BPP_HD void ct_var_scalarmul(ge &r, const ge &p, const sc &s) {
  int8_t d[BPP_CT_DIGITS];
  ct_recode16(d, s);
  ge pw = p, acc;
  ge_identity(acc);
  for (int w = 0; w < BPP_CT_DIGITS; w++) {
    if (w)
      for (int k = 0; k < 4; k++) ge_dbl(pw, pw);
    ge q;
    ct_pos_multiple(q, pw, d[w]);
    ge_add(acc, acc, q);
  }
  r = acc;
}

#if defined(__HIPCC__)
#define CTF_MAX_TERMS 8
struct CtFixedShared {
  int8_t dig[CTF_MAX_TERMS][BPP_CT_DIGITS];
  ge red[64];
};
// out[o] = sum_{i < count[o]} scal[o][i] * Base[bidx[o][i] - idx_off], count[o] <= CTF_MAX_TERMS: one wavefront per output, lane w
// owns digit position w of every term.  `tbl` = [base][64][8] lines (k_fb_build with 4-bit windows).
__global__ void __launch_bounds__(64) k_ct_fixed(const sc *__restrict__ scal, const uint32_t *__restrict__ bidx,
                                                 const uint32_t *__restrict__ count, uint32_t stride, uint32_t idx_off,
                                                 const niels *__restrict__ tbl, ge *__restrict__ out) {
  const uint32_t o = blockIdx.x, lane = threadIdx.x;
  const uint32_t n = count[o] < CTF_MAX_TERMS ? count[o] : CTF_MAX_TERMS;
  __shared__ CtFixedShared sh;
  if (lane < CTF_MAX_TERMS) {  // (lanes beyond the output's terms recode its first term again: the same instructions, unused digits)
    const sc s = scal[(size_t)o * stride + (lane < n ? lane : 0u)];
    ct_recode16(sh.dig[lane], s);
  }
  __syncthreads();
  ge acc;
  ge_identity(acc);
#pragma unroll 1
  for (uint32_t i = 0; i < n; i++) {  // the number of terms is public
    const uint32_t b = bidx[(size_t)o * stride + i] - idx_off;
    niels q;
    ct_fixed_select(q, tbl + ((size_t)b * BPP_CT_DIGITS + lane) * BPP_CTF_ENTRIES, (int32_t)sh.dig[i][lane]);
    ge_madd(acc, acc, q);
  }
  sh.red[lane] = acc;
  __syncthreads();
  for (uint32_t off = 32; off >= 1; off >>= 1) {
    if (lane < off) {
      ge x = sh.red[lane], y2 = sh.red[lane + off];
      ge_add(x, x, y2);
      sh.red[lane] = x;
    }
    __syncthreads();
  }
  if (lane == 0) out[o] = sh.red[0];
  __syncthreads();
  for (uint32_t k = lane; k < sizeof(CtFixedShared) / 4; k += 64) ((uint32_t *)&sh)[k] = 0;  // digits and partial sums are secret-derived
}

// pow[pt][w] = 16^w * pts[pt], w = 0..63: one quad per point (msm.h's quad doubling), 252 dependent doublings of PUBLIC points
__global__ void __launch_bounds__(64) k_ct_pow16(const ge *__restrict__ pts, uint32_t n_pts, ge *__restrict__ pow) {
  const uint32_t lane = threadIdx.x, qi = lane & 3u;
  uint32_t pt = blockIdx.x * 16u + (lane >> 2);
  const bool live = pt < n_pts;
  if (!live) pt = n_pts - 1;  // (idle quads walk the last point again and store nothing)
  const QuadMask q = quad_mask(qi);
  const ge p = pts[pt];
  fe m;
  quad_load(m, q, p);
#pragma unroll 1
  for (uint32_t w = 0; w < BPP_CT_DIGITS; w++) {
    if (w) {
#pragma unroll 1
      for (int k = 0; k < 4; k++) quad_ge_dbl(m, q);
    }
    fe c = m;
    fe_carry(c);
    if (live) {
      uint32_t *dst = reinterpret_cast<uint32_t *>(pow + (size_t)pt * BPP_CT_DIGITS + w) + 10u * qi;  // coordinate qi of the entry
#pragma unroll
      for (int k = 0; k < 10; k++) dst[k] = c.v[k];
    }
  }
}

// prod[p * nt + v] = scal(p, v) * pts[p * nt + v], the points given as their multiples pow[p * nt + v][w] = 16^w pts[..]
// (k_ct_pow16): one wavefront per term, lane w owns digit position w (ct_pos_multiple), a six-level tree at the end.
// scal(p, v): rows of `row_stride` scalars per proof, term v at word 8 + (v & 7) of row v >> 3 (the prover's final-step rows).
struct CtVarShared {
  int8_t dig[BPP_CT_DIGITS];
  ge red[64];
};
__global__ void __launch_bounds__(64) k_ct_var(const ge *__restrict__ pow, const sc *__restrict__ scal, uint32_t row_stride, uint32_t nt,
                                               ge *__restrict__ prod) {
  const uint32_t p = blockIdx.x / nt, v = blockIdx.x - p * nt, w = threadIdx.x;
  __shared__ CtVarShared sh;
  if (w == 0) {
    const sc s = scal[((size_t)2 * p + (v >> 3)) * row_stride + 8u + (v & 7u)];
    ct_recode16(sh.dig, s);
  }
  __syncthreads();
  const ge pw = pow[((size_t)p * nt + v) * BPP_CT_DIGITS + w];
  ge mine;
  ct_pos_multiple(mine, pw, (int32_t)sh.dig[w]);
  sh.red[w] = mine;
  __syncthreads();
  for (uint32_t off = 32; off >= 1; off >>= 1) {
    if (w < off) {
      ge x = sh.red[w], y2 = sh.red[w + off];
      ge_add(x, x, y2);
      sh.red[w] = x;
    }
    __syncthreads();
  }
  if (w == 0) prod[(size_t)p * nt + v] = sh.red[0];
  __syncthreads();
  for (uint32_t k = w; k < sizeof(CtVarShared) / 4; k += 64) ((uint32_t *)&sh)[k] = 0;  // digits and partial sums are secret-derived
}
// acc[acc_stride * p] += sum_{v < nt} prod[p * nt + v]: one lane per proof (nt <= 16 additions in a row)
__global__ void __launch_bounds__(64) k_ct_sum(const ge *__restrict__ prod, uint32_t nt, uint32_t n_proofs, ge *__restrict__ acc,
                                               uint32_t acc_stride) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_proofs) return;
  ge x = acc[(size_t)acc_stride * p];
  for (uint32_t v = 0; v < nt; v++) {
    const ge y2 = prod[(size_t)p * nt + v];
    ge_add(x, x, y2);
  }
  acc[(size_t)acc_stride * p] = x;
}

#define CT_MAX_TERMS 16
struct CtShared {
  uint32_t tab[CT_MAX_TERMS][BPP_CT_ENTRIES][4][10];  // [term][entry][coordinate][limb]
  int8_t dig[CT_MAX_TERMS][BPP_CT_DIGITS];
  ge part[CT_MAX_TERMS];
};

// out[o] = sum_{i < count[o]} scal[o][i] * point(tidx[o][i]),  count[o] <= CT_MAX_TERMS; one workgroup of 64 lanes per output, one
// quad per term.  point(idx) = dyn[idx & 0x7fffffff] when bit 31 is set (a per-call point in extended coordinates), else the
// affine-Niels table line bases[idx] (the parameter set's generator table: Pedersen bases at n_gen + k, H at n_gen + t).
#define BPP_CT_DYN 0x80000000u
__global__ void __launch_bounds__(64) k_ct_msm(const sc *__restrict__ scal, const uint32_t *__restrict__ tidx,
                                               const uint32_t *__restrict__ count, uint32_t stride, const niels *__restrict__ bases,
                                               const ge *__restrict__ dyn, ge *__restrict__ out) {
  const uint32_t o = blockIdx.x, lane = threadIdx.x, qi = lane & 3u, term = lane >> 2;
  const QuadMask q = quad_mask(qi);
  const uint32_t n = count[o] < CT_MAX_TERMS ? count[o] : CT_MAX_TERMS;
  __shared__ CtShared sh;
  const bool active = term < n;  // (idle quads run the same instruction stream on the output's first term and write nothing)
  const uint32_t it = active ? term : 0u;
  const uint32_t idx = tidx[(size_t)o * stride + it];
  ge p;
  if (idx & BPP_CT_DYN) {
    p = dyn[idx & ~BPP_CT_DYN];
  } else {
    const niels b = bases[idx];
    ge_from_niels(p, b);
  }
  fe d2, one;
  fe_const(d2, FE_D2);
  fe_1(one);
  // table 0 P .. 8 P, every entry's coordinate qi written by lane qi with carried limbs
  auto store_entry = [&](uint32_t j, const fe &m) {
    fe c = m;
    fe_carry(c);
#pragma unroll
    for (int k = 0; k < 10; k++) sh.tab[term][j][qi][k] = c.v[k];
  };
  fe m;
  {
    ge id;
    ge_identity(id);
    quad_load(m, q, id);
    store_entry(0, m);
  }
  quad_load(m, q, p);
  store_entry(1, m);
#pragma unroll 1
  for (uint32_t j = 2; j < BPP_CT_ENTRIES; j++) {
    quad_ge_add(m, q, p, d2, one);
    store_entry(j, m);
  }
  if (qi == 0) {
    const sc s = scal[(size_t)o * stride + it];
    ct_recode16(sh.dig[term], s);
  }
  __syncthreads();
  {
    ge id;
    ge_identity(id);
    quad_load(m, q, id);
  }
#pragma unroll 1
  for (int i = BPP_CT_DIGITS - 1; i >= 0; i--) {
    if (i != BPP_CT_DIGITS - 1) {
#pragma unroll 1
      for (int k = 0; k < 4; k++) quad_ge_dbl(m, q);
    }
    uint32_t mag, neg;
    ct_digit_parts(mag, neg, (int32_t)sh.dig[term][i]);
    uint32_t w[10];
    ct_select_words<10>(w, &sh.tab[term][0][qi][0], 40, mag);  // this lane's coordinate of ALL nine entries
    fe mine, flipped;
#pragma unroll
    for (int k = 0; k < 10; k++) mine.v[k] = w[k];
    fe_cneg_select(flipped, mine, neg);
    // -P = (-X, Y, Z, -T): lanes 0 and 3 take the (possibly) negated coordinate; a select on the lane's position, not a branch
    {
      const uint32_t xt = 0u - (uint32_t)((qi == 0u) | (qi == 3u));
#pragma unroll
      for (int k = 0; k < 10; k++) mine.v[k] = (flipped.v[k] & xt) | (mine.v[k] & ~xt);
    }
    ge other;
    quad_gather(other, mine);
    quad_ge_add(m, q, other, d2, one);
  }
  {
    ge acc;
    quad_gather(acc, m);
    if (qi == 0) sh.part[term] = acc;
  }
  __syncthreads();
  if (term == 0) {  // the output's terms, summed by the first quad (count is public)
    ge first = sh.part[0];
    quad_load(m, q, first);
#pragma unroll 1
    for (uint32_t j = 1; j < n; j++) {
      const ge pj = sh.part[j];
      quad_ge_add(m, q, pj, d2, one);
    }
    ge acc;
    quad_gather(acc, m);
    if (qi == 0) out[o] = acc;
  }
  // the digits and the table of multiples are secret-derived: leave nothing in LDS for the next workgroup on this CU
  __syncthreads();
  for (uint32_t k = lane; k < sizeof(CtShared) / 4; k += 64) ((uint32_t *)&sh)[k] = 0;
}
#endif

}  // namespace bpp
