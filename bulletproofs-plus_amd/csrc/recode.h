// Scalar recodings shared by host and device (host-compiled by tests/: every digit string must add up to its scalar).
//   msm_recode : signed digits of the Pippenger MSM, uneven windows (MsmPlan)
//   fb_recode  : signed digits of the prover's fixed-base tables (FbGeom)
#pragma once
#include <stdlib.h>

#include "scalar.h"

namespace bpp {

// Window layout: 253 = K_wide * c + (K - K_wide) * (c - 1).  Canonical scalars are < l < 2^252 + 2^125, so 253 bits and
// no carry out of the top window; with equal widths the top window would hold only 253 mod c bits and concentrate every
// term of a group in a handful of buckets (one lane adding thousands of points in sequence) unless c divides 253 (c =
// 11).  Windows of c-1 bits simply leave the upper half of their 2^(c-1) buckets empty.
struct MsmPlan {
  uint32_t c;        // window bits (wide windows)
  uint32_t K;        // windows
  uint32_t K_wide;   // windows [0, K_wide) have c bits, windows [K_wide, K) have c - 1
  uint32_t nb;       // buckets per window = 2^(c-1)
  uint32_t G;        // groups
  uint32_t n_terms;  // total terms
  uint32_t split;    // 1: half-scalar plan -- every term appears twice, (s mod 2^126, P) and (s >> 126, 2^126 P), the windows
                     // cover 127 bits: half the doublings in the final Horner step (small calls, msm.h)
};
// Canonical scalars are < l = 2^252 + d, d < 2^125.  Split at bit 126: the low half has 126 bits; the high half has 126 bits
// too, except for s >= 2^252, where it is EXACTLY 2^126 (s - 2^252 < 2^125 leaves nothing below bit 126 of the shifted
// value).  With windows over 127 bits (BPP_MSM_SPLIT_BITS) both halves fill every window evenly -- the top window keeps its
// spare top bit, so its raw value stays under half its range and nothing carries out; the one exception, the lone bit 126,
// makes the top window's raw value exactly half with no carry coming in (everything below is zero): digit +half, a bucket
// that exists.  (Splitting at bit 127 over 128-bit windows left the top window with 7-8 significant bits: a quarter of its
// buckets got all of its terms, lists of 40-60 additions: 0.10 ms of accumulation for 256 proofs instead of 0.04.)
#define BPP_MSM_SPLIT_BIT 126u
#define BPP_MSM_SPLIT_BITS 127u
#define BPP_TERM_HI 0x80000000u   // term_sidx: take the high half of the scalar
#define BPP_POINT_HI 0x40000000u  // term_pidx / sorted[]: the point's 2^126 multiple (bit 31 of sorted[] is the sign)

// digits of one canonical scalar, written with stride `stride` (window-major layout of k_msm_digits: stride = terms of the group)
BPP_HD void msm_recode(int16_t *out, size_t stride, const sc &s, const MsmPlan &plan) {
  uint32_t carry = 0, bit = 0;
  for (uint32_t k = 0; k < plan.K; k++) {
    const uint32_t wd = k < plan.K_wide ? plan.c : plan.c - 1;  // this window's width
    const uint32_t wi = bit >> 5, sh = bit & 31;
    uint32_t raw = 0;
    if (wi < 8) {
      uint64_t two = (uint64_t)s.v[wi] | ((wi + 1 < 8) ? ((uint64_t)s.v[wi + 1] << 32) : 0ULL);
      raw = (uint32_t)(two >> sh) & ((1u << wd) - 1u);
    }
    uint32_t v = raw + carry;
    int32_t dgt;
    if (v > (1u << (wd - 1))) {  // digits in (-2^(wd-1), 2^(wd-1)]
      dgt = (int32_t)v - (int32_t)(1u << wd);
      carry = 1;
    } else {
      dgt = (int32_t)v;
      carry = 0;
    }
    out[(size_t)k * stride] = (int16_t)dgt;
    bit += wd;
  }
}

// The same digit for ONE window, computed from the words around it (what the fused sort kernel does: it owns one window of
// every term and never sees the whole digit string).  The carry into window k is decided by window k-1 alone unless that
// window's raw value is exactly half its range, in which case it is the carry into k-1 (and so on down): carry(j+1) =
// raw_j > half_j, or raw_j == half_j and carry(j).  `w` = the scalar's eight 32-bit words, read on demand.
BPP_HD uint32_t msm_window_bit(const MsmPlan &plan, uint32_t k) {
  return k <= plan.K_wide ? k * plan.c : plan.K_wide * plan.c + (k - plan.K_wide) * (plan.c - 1);
}
BPP_HD uint32_t msm_window_raw(const uint32_t *w, uint32_t bit, uint32_t wd) {
  const uint32_t wi = bit >> 5, sh = bit & 31;
  if (wi >= 8) return 0;
  uint32_t v = w[wi] >> sh;
  if (sh + wd > 32 && wi + 1 < 8) v |= w[wi + 1] << (32 - sh);
  return v & ((1u << wd) - 1u);
}
BPP_HD int32_t msm_digit_at(const uint32_t *w, const MsmPlan &plan, uint32_t k) {
  uint32_t carry = 0;
  for (uint32_t j = k; j-- > 0;) {
    const uint32_t wdj = j < plan.K_wide ? plan.c : plan.c - 1;
    const uint32_t rawj = msm_window_raw(w, msm_window_bit(plan, j), wdj), half = 1u << (wdj - 1);
    if (rawj != half) {
      carry = rawj > half ? 1u : 0u;
      break;
    }
  }
  const uint32_t wd = k < plan.K_wide ? plan.c : plan.c - 1;
  const uint32_t v = msm_window_raw(w, msm_window_bit(plan, k), wd) + carry;
  return v > (1u << (wd - 1)) ? (int32_t)v - (int32_t)(1u << wd) : (int32_t)v;
}

// plan for a group of `terms` terms with window width c
// bits = 253 (canonical scalars), or BPP_MSM_SPLIT_BITS for the half-scalar plan
inline MsmPlan msm_make_plan(uint32_t c, uint32_t G, uint32_t n_terms, uint32_t bits = 253) {
  MsmPlan plan;
  plan.c = c;
  plan.K = (bits + c - 1) / c;
  plan.K_wide = plan.K - (plan.K * c - bits);  // bits = K_wide * c + (K - K_wide) * (c - 1)
  plan.nb = 1u << (c - 1);
  plan.G = G;
  plan.n_terms = n_terms;
  plan.split = bits == 253 ? 0u : 1u;
  return plan;
}
// the half of a canonical scalar a split plan's term stands for, as eight words (upper ones zero)
BPP_HD void msm_half_words(uint32_t h[8], const uint32_t *w, bool hi) {
  constexpr uint32_t W = BPP_MSM_SPLIT_BIT >> 5, SH = BPP_MSM_SPLIT_BIT & 31u;  // 126 = 3 * 32 + 30
  static_assert(SH != 0, "the split bit is not word-aligned");
  for (int i = 0; i < 8; i++) h[i] = 0;
  if (!hi) {
    for (uint32_t i = 0; i < W; i++) h[i] = w[i];
    h[W] = w[W] & ((1u << SH) - 1u);
  } else {
    for (uint32_t i = 0; W + i < 8; i++) h[i] = (w[W + i] >> SH) | (W + i + 1 < 8 ? w[W + i + 1] << (32u - SH) : 0u);
  }
}

// Window width is chosen per parameter set (fb_geometry): the widest window whose table stays under ~1.8 GB, because
// random 128-byte lines come at 21 G lines/s out of <= 2 GB but only ~10 G lines/s out of larger tables (TLB reach,
// tools/microbench/rand_lines.hip).  11 bits (<= 600 generators) = 23 additions per term out of 24 slots of 1024 entries.
//
// A canonical scalar has 253 bits.  Signed digits need one position more than ceil(253 / wbits) only when the top window is
// full (253 = 23 x 11): its carry would be a 24th digit that is 0 or 1 -- a whole addition per term for one bit.  In that
// case the top window is kept UNSIGNED instead (digit 0 .. 2^wbits, no carry out) and its entries 2^(wbits-1)+1 .. 2^wbits
// live in the table slot the carry digit would have used: same table size, same addressing (slot w, entry digit - 1 simply
// runs on into slot w + 1), one addition per term fewer.  `items` = digit positions, `windows` = table slots per generator.
struct FbGeom {
  uint32_t wbits;    // window width
  uint32_t windows;  // table slots per generator: ceil(254 / wbits)
  uint32_t entries;  // entries per slot: 2^(wbits-1) signed multiples 1..2^(wbits-1)
  uint32_t items;    // digit positions per scalar: windows, or windows - 1 with the unsigned top window
};
#define FB_MAX_WINDOWS 32
#define FB_BUILD_BLOCK 128  // entries per lane of k_fb_build
inline FbGeom fb_geometry(uint32_t n_gens) {  // host side
  uint32_t w = 8;
  const char *forced = getenv("BPP_FB_WBITS");  // tests: exercise every geometry on small parameter sets
  const uint32_t top = forced ? (uint32_t)atoi(forced) : 11u;
  for (uint32_t cand = (top >= 8 && top <= 11) ? top : 11u; cand > 8; cand--) {
    const uint64_t bytes = (uint64_t)n_gens * ((254 + cand - 1) / cand) * (1ull << (cand - 1)) * 128ull;
    if (bytes <= 1800ull << 20) {
      w = cand;
      break;
    }
  }
  FbGeom g;
  g.wbits = w;
  g.windows = (254 + w - 1) / w;
  g.entries = 1u << (w - 1);
  g.items = (253u % w == 0u) ? 253u / w : g.windows;
  return g;
}
BPP_HD size_t fb_stride(const FbGeom &g) { return (size_t)g.windows * g.entries; }  // entries per generator
BPP_HD bool fb_top_unsigned(const FbGeom &g) { return g.items < g.windows; }

// word i of a scalar (0 for i >= 8) without indexing its register array at run time
BPP_HD uint32_t sc_word(const sc &s, uint32_t i) {
  uint32_t r = 0;
#pragma unroll
  for (uint32_t k = 0; k < 8; k++) r = (i == k) ? s.v[k] : r;
  return r;
}

// digits of a canonical scalar: signed, in [-(2^(w-1) - 1), 2^(w-1)]; the top one in [0, 2^w] when fb_top_unsigned
BPP_HD void fb_recode(int16_t *dig, const sc &s, const FbGeom &g) {
  uint32_t carry = 0;
  for (uint32_t w = 0; w < g.items; w++) {
    const uint32_t bit = w * g.wbits, wi = bit >> 5, sh = bit & 31u;
    uint32_t raw = 0;
    if (wi < 8) {
      // (sc_word: a select chain over the eight words; s.v[wi] with a run-time wi parks the scalar in scratch memory)
      const uint64_t two = (uint64_t)sc_word(s, wi) | ((uint64_t)sc_word(s, wi + 1) << 32);
      raw = (uint32_t)(two >> sh) & ((1u << g.wbits) - 1u);
    }
    const uint32_t v = raw + carry;
    carry = (v > g.entries && !(fb_top_unsigned(g) && w + 1 == g.items)) ? 1u : 0u;
    dig[w] = (int16_t)((int32_t)v - (int32_t)(carry << g.wbits));
  }
}

}  // namespace bpp
