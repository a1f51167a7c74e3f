// GF(2^255 - 19) for gfx950.
//
// Representation: 10 unsigned limbs, radix 2^25.5 (26,25,26,25,... bits).  Products are accumulated in
// 64-bit columns, which hipcc lowers to v_mad_u64_u32 (one instruction per limb product, no carry
// chains inside the product).  Limb bounds, E = bound of the even limbs (odd limbs: half of it):
//   reduced  E <= 2^26           what fe_mul / fe_sq / fe_carry return (limb 1 may carry 2^19 more)
//   loose    E <= 1.5 * 2^27     a sum of two reduced values, or fe_sub_lazy of two reduced values
//   wide     E <= 2^28           2 * reduced + 2p - reduced (the "d - c" of a point addition)
// fe_mul(h, f, g) takes f up to wide and g up to loose (g is the operand that is pre-multiplied by 19 for the wrapped
// columns: 19 * 1.5 * 2^27 < 2^32); fe_sq takes loose.  The heaviest column of such a product is
// 124.5 * E_f * E_g <= 2^62.5.  Point formulas in point.h are written against these classes, so that no carry pass runs
// outside the products; the host build checks every bound when BPP_FE_BOUNDS_CHECK is defined (sanitizer harness).
//
// Replaces (reference boundary): curve25519-dalek FieldElement reached through
// src/range_proof.rs:1067-1109 (decompress) and every point operation of the final MSM (:1050-1057).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BPP_HD __host__ __device__ __forceinline__
#define BPP_D __device__ __forceinline__
// helpers that are called many times or sit off the critical chain: real calls keep register pressure of the callers low
#define BPP_HD_NOINLINE __host__ __device__ __attribute__((noinline))
#else
#define BPP_HD inline
#define BPP_D inline
#define BPP_HD_NOINLINE inline
#endif

#define BPP_CONST static constexpr

namespace bpp {

struct fe {
  uint32_t v[10];
};

BPP_HD constexpr uint32_t fe_bits(int i) { return (i & 1) ? 25u : 26u; }
BPP_HD constexpr uint32_t fe_mask(int i) { return (i & 1) ? 0x1ffffffu : 0x3ffffffu; }

BPP_HD void fe_0(fe &h) {
#pragma unroll
  for (int i = 0; i < 10; i++) h.v[i] = 0;
}
BPP_HD void fe_1(fe &h) {
  fe_0(h);
  h.v[0] = 1;
}
BPP_HD void fe_copy(fe &h, const fe &f) {
#pragma unroll
  for (int i = 0; i < 10; i++) h.v[i] = f.v[i];
}

// one carry sweep; input limbs < 2^32 - 2^7, output reduced
BPP_HD void fe_carry(fe &h) {
  uint32_t c;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    c = h.v[i] >> fe_bits(i);
    h.v[i] &= fe_mask(i);
    h.v[i + 1] += c;
  }
  c = h.v[9] >> 25;
  h.v[9] &= 0x1ffffffu;
  h.v[0] += 19u * c;
  c = h.v[0] >> 26;
  h.v[0] &= 0x3ffffffu;
  h.v[1] += c;
}

// h = f + g, no carry: reduced + reduced -> limbs < 2^27
BPP_HD void fe_add(fe &h, const fe &f, const fe &g) {
#pragma unroll
  for (int i = 0; i < 10; i++) h.v[i] = f.v[i] + g.v[i];
}

// h = f - g (f limbs < 2^27, g limbs < 2^27); adds 4p limb-wise, then carries -> reduced
BPP_HD void fe_sub(fe &h, const fe &f, const fe &g) {
  h.v[0] = f.v[0] + 0xfffffb4u - g.v[0];  // 4*(2^26-19)
#pragma unroll
  for (int i = 1; i < 10; i++) h.v[i] = f.v[i] + ((i & 1) ? 0x7fffffcu : 0xffffffcu) - g.v[i];
  fe_carry(h);
}

// h = f - g + 2p without a carry pass: g must be reduced (its limbs do not exceed those of 2p); f reduced -> h loose
BPP_HD void fe_sub_lazy(fe &h, const fe &f, const fe &g) {
  h.v[0] = f.v[0] + 0x7ffffdau - g.v[0];  // 2 * (2^26 - 19)
#pragma unroll
  for (int i = 1; i < 10; i++) h.v[i] = f.v[i] + ((i & 1) ? 0x3fffffeu : 0x7fffffeu) - g.v[i];
}

// h = 2 f + g (no carry): f, g reduced -> loose
BPP_HD void fe_dbl_add(fe &h, const fe &f, const fe &g) {
#pragma unroll
  for (int i = 0; i < 10; i++) h.v[i] = 2u * f.v[i] + g.v[i];
}

// h = 2 f - g + 2p (no carry): f, g reduced -> wide
BPP_HD void fe_dbl_sub_lazy(fe &h, const fe &f, const fe &g) {
  h.v[0] = 2u * f.v[0] + 0x7ffffdau - g.v[0];
#pragma unroll
  for (int i = 1; i < 10; i++) h.v[i] = 2u * f.v[i] + ((i & 1) ? 0x3fffffeu : 0x7fffffeu) - g.v[i];
}

BPP_HD void fe_neg(fe &h, const fe &f) {
  fe z;
  fe_0(z);
  fe_sub(h, z, f);
}

#if defined(BPP_FE_BOUNDS_CHECK) && !defined(__HIP_DEVICE_COMPILE__)
}  // namespace bpp
#include <stdio.h>
#include <stdlib.h>
namespace bpp {
inline void fe_bound_fail(const char *what) {
  fprintf(stderr, "field bound violated: %s\n", what);
  abort();
}
#define BPP_FE_CHECK(cond, what) \
  do {                           \
    if (!(cond)) fe_bound_fail(what); \
  } while (0)
#else
#define BPP_FE_CHECK(cond, what) \
  do {                           \
  } while (0)
#endif

// a * b + c with c as the instruction's own 64-bit addend.  Written as `(uint64_t)a * b + c` the compiler re-associates
// the sums of a column (products first, carry-in last, to shorten the dependency chain) and pays a separate 64-bit
// addition per limb; pinned, the carry-in rides in the first v_mad_u64_u32 of the column for free.
#ifndef BPP_FE_PLAIN_MAD
#define BPP_FE_PLAIN_MAD 0
#endif
BPP_HD uint64_t fe_mad(uint32_t a, uint32_t b, uint64_t c) {
#if defined(__HIP_DEVICE_COMPILE__) && !BPP_FE_PLAIN_MAD
  uint64_t d, carry_out;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(carry_out) : "v"(a), "v"(b), "v"(c));
  return d;
#else
#if defined(BPP_FE_BOUNDS_CHECK)
  BPP_FE_CHECK((((unsigned __int128)a * b + c) >> 64) == 0, "64-bit column overflow");
#endif
  return (uint64_t)a * b + c;
#endif
}

// N multiply-adds into one 64-bit accumulator as ONE asm statement.  Between two separate inline-asm statements of which
// the second reads what the first wrote, the compiler inserts `s_nop 0` (it must assume a forwarding hazard for
// instructions it cannot see), and that costs the wavefront a whole issue turn: a dependent chain of v_mad_u64_u32 runs at
// 9.5 cycles per multiply with the s_nop and 5.2 without, and three wavefronts per SIMD reach 73 % instead of 95 % of the
// multiplier's rate (tools/microbench/mad_nops.hip).  Inside one statement the hardware interlocks by itself, as it does
// for the chains the compiler generates from C.  One s_nop is left per column (before the mask / shift that follows).
template <int N>
BPP_HD uint64_t fe_mad_chain(uint64_t acc, const uint32_t (&a)[N], const uint32_t (&b)[N]) {
#if defined(__HIP_DEVICE_COMPILE__) && !BPP_FE_PLAIN_MAD
  static_assert(N >= 1 && N <= 10, "chain length");
#define BPP_M(A, B) "v_mad_u64_u32 %0, vcc, %" #A ", %" #B ", %0\n\t"
  if constexpr (N == 1) asm(BPP_M(1, 2) : "+v"(acc) : "v"(a[0]), "v"(b[0]) : "vcc");
  if constexpr (N == 2) asm(BPP_M(1, 3) BPP_M(2, 4) : "+v"(acc) : "v"(a[0]), "v"(a[1]), "v"(b[0]), "v"(b[1]) : "vcc");
  if constexpr (N == 3) asm(BPP_M(1, 4) BPP_M(2, 5) BPP_M(3, 6) : "+v"(acc) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(b[0]), "v"(b[1]), "v"(b[2]) : "vcc");
  if constexpr (N == 4) asm(BPP_M(1, 5) BPP_M(2, 6) BPP_M(3, 7) BPP_M(4, 8) : "+v"(acc) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]) : "vcc");
  if constexpr (N == 5) asm(BPP_M(1, 6) BPP_M(2, 7) BPP_M(3, 8) BPP_M(4, 9) BPP_M(5, 10) : "+v"(acc) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]) : "vcc");
  if constexpr (N == 6) asm(BPP_M(1, 7) BPP_M(2, 8) BPP_M(3, 9) BPP_M(4, 10) BPP_M(5, 11) BPP_M(6, 12) : "+v"(acc) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]) : "vcc");
  if constexpr (N == 7) asm(BPP_M(1, 8) BPP_M(2, 9) BPP_M(3, 10) BPP_M(4, 11) BPP_M(5, 12) BPP_M(6, 13) BPP_M(7, 14) : "+v"(acc) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]) : "vcc");
  if constexpr (N == 8) asm(BPP_M(1, 9) BPP_M(2, 10) BPP_M(3, 11) BPP_M(4, 12) BPP_M(5, 13) BPP_M(6, 14) BPP_M(7, 15) BPP_M(8, 16) : "+v"(acc) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]) : "vcc");
  if constexpr (N == 9) asm(BPP_M(1, 10) BPP_M(2, 11) BPP_M(3, 12) BPP_M(4, 13) BPP_M(5, 14) BPP_M(6, 15) BPP_M(7, 16) BPP_M(8, 17) BPP_M(9, 18) : "+v"(acc) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]), "v"(b[8]) : "vcc");
  if constexpr (N == 10) asm(BPP_M(1, 11) BPP_M(2, 12) BPP_M(3, 13) BPP_M(4, 14) BPP_M(5, 15) BPP_M(6, 16) BPP_M(7, 17) BPP_M(8, 18) BPP_M(9, 19) BPP_M(10, 20) : "+v"(acc) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]), "v"(a[9]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]), "v"(b[8]), "v"(b[9]) : "vcc");
#undef BPP_M
  return acc;
#else
#pragma unroll
  for (int i = 0; i < N; i++) acc = fe_mad(a[i], b[i], acc);
  return acc;
#endif
}

// Products: column k of f*g is sum_{i+j = k (mod 10)} f_i g_j (x2 when i and j are both odd: radix 2^25.5; x19 when
// i + j >= 10: 2^255 = 19), accumulated in one 64-bit register pair by v_mad_u64_u32.  The columns are computed in order
// and column k's carry SEEDS column k+1's accumulator (the first v_mad_u64_u32 of a column takes it as its 64-bit addend),
// so the carry chain costs a shift and a mask per limb and no 64-bit addition.  Bounds: limbs up to 1.5 * 2^27 on both
// sides give column 0 (the heaviest: 267 L^2) 2^63.3, carries stay below 2^38.
BPP_HD void fe_mul(fe &h, const fe &f, const fe &g) {
  uint32_t g19[10], f2[10];
#pragma unroll
  for (int i = 0; i < 10; i++) {
    BPP_FE_CHECK((uint64_t)19u * g.v[i] < (1ull << 32) && (uint64_t)2u * f.v[i] < (1ull << 32), "fe_mul operand limb too large");
    g19[i] = 19u * g.v[i];
    f2[i] = 2u * f.v[i];
  }
  uint64_t c = 0;
  uint32_t r[10];
#pragma unroll
  for (int k = 0; k < 10; k++) {
    uint32_t a[10], b[10];
#pragma unroll
    for (int i = 0; i < 10; i++) {
      const int j = (k - i + 10) % 10;
      const bool wrap = i > k;
      const bool dbl = (i & 1) && (j & 1);
      a[i] = dbl ? f2[i] : f.v[i];
      b[i] = wrap ? g19[j] : g.v[j];
    }
    const uint64_t acc = fe_mad_chain<10>(c, a, b);
    r[k] = (uint32_t)acc & fe_mask(k);
    c = acc >> fe_bits(k);
  }
  // c < 2^39 wraps into limb 0 with weight 19; limb 1 absorbs what limb 0 spills (r[1] < 2^25 + 2^19)
  const uint64_t t0 = (uint64_t)r[0] + 19u * c;
  h.v[0] = (uint32_t)t0 & 0x3ffffffu;
  h.v[1] = r[1] + (uint32_t)(t0 >> 26);
#pragma unroll
  for (int i = 2; i < 10; i++) h.v[i] = r[i];
}

// Squaring: 55 products.  Both operands of every product are pre-scaled limbs, so a column is a plain chain of
// v_mad_u64_u32 (no doubling of 64-bit products); carries seed the next column as in fe_mul.  The factor of a pair
// (i <= j) is 2 (off-diagonal) x 2 (both odd) x 19 (wrapped, i + j >= 10).  A wrapped pair takes its 19 -- and for an odd
// j, whose bound is half that of the even limbs, one factor 2 as well -- on f_j; whatever factor 2 is left goes on f_i.
// That needs 13 pre-scaled limbs in all: 2 f_0..2 f_7, 19 f_6, 19 f_8, 38 f_5, 38 f_7, 38 f_9; loose input (even limbs
// <= 1.5 * 2^27, odd <= 1.5 * 2^26) keeps every one of them below 2^32.
template <int K>
BPP_HD void fe_sq_column(uint64_t &c, uint32_t (&r)[10], const fe &f, const uint32_t (&f2)[10], const uint32_t (&fw)[10]) {
  constexpr int N = (K & 1) ? 5 : 6;  // pairs (i <= j) with i + j = K (mod 10)
  uint32_t a[N], b[N];
  int n = 0;
#pragma unroll
  for (int i = 0; i < 10; i++) {
    const int j = (K - i + 10) % 10;
    if (i > j) continue;
    const bool wrap = i > K;  // i + j == K + 10
    const bool dbl = (i & 1) && (j & 1);
    const bool off = i != j;
    const int total = (off ? 2 : 1) * (dbl ? 2 : 1) * (wrap ? 19 : 1);
    const int on_j = wrap ? ((j & 1) ? 38 : 19) : ((dbl && off) ? 2 : 1);
    b[n] = wrap ? fw[j] : ((dbl && off) ? f2[j] : f.v[j]);
    a[n] = (total / on_j == 2) ? f2[i] : f.v[i];  // the quotient is 1 or 2 for every pair
    n++;
  }
  const uint64_t acc = fe_mad_chain<N>(c, a, b);
  r[K] = (uint32_t)acc & fe_mask(K);
  c = acc >> fe_bits(K);
}
BPP_HD void fe_sq(fe &h, const fe &f) {
  uint32_t f2[10], fw[10];  // fw[j] = 19 f_j (j even) or 38 f_j (j odd)
#pragma unroll
  for (int i = 0; i < 10; i++) {
    BPP_FE_CHECK((uint64_t)((i & 1) ? 38u : 19u) * f.v[i] < (1ull << 32), "fe_sq operand limb too large");
    f2[i] = 2u * f.v[i];
    fw[i] = ((i & 1) ? 38u : 19u) * f.v[i];
  }
  uint64_t c = 0;
  uint32_t r[10];
  fe_sq_column<0>(c, r, f, f2, fw);
  fe_sq_column<1>(c, r, f, f2, fw);
  fe_sq_column<2>(c, r, f, f2, fw);
  fe_sq_column<3>(c, r, f, f2, fw);
  fe_sq_column<4>(c, r, f, f2, fw);
  fe_sq_column<5>(c, r, f, f2, fw);
  fe_sq_column<6>(c, r, f, f2, fw);
  fe_sq_column<7>(c, r, f, f2, fw);
  fe_sq_column<8>(c, r, f, f2, fw);
  fe_sq_column<9>(c, r, f, f2, fw);
  const uint64_t t0 = (uint64_t)r[0] + 19u * c;
  h.v[0] = (uint32_t)t0 & 0x3ffffffu;
  h.v[1] = r[1] + (uint32_t)(t0 >> 26);
#pragma unroll
  for (int i = 2; i < 10; i++) h.v[i] = r[i];
}

// n squarings, two per loop iteration through a second set of registers: squaring in place makes the compiler copy every
// limb that is overwritten while still needed (8 moves per squaring)
BPP_HD void fe_sqn(fe &h, const fe &f, int n) {
  fe_sq(h, f);
  int i = 1;
  for (; i + 1 < n; i += 2) {
    fe t;
    fe_sq(t, h);
    fe_sq(h, t);
  }
  if (i < n) fe_sq(h, h);
}

// canonical value as 8 little-endian 32-bit words (the primitive: byte arrays cost one register per byte on the GPU)
BPP_HD void fe_towords(uint32_t w[8], const fe &f) {
  fe h;
  fe_copy(h, f);
  fe_carry(h);
  fe_carry(h);
  // h < 2^255 + small; q = 1 iff h >= p
  uint32_t q = (h.v[0] + 19u) >> 26;
#pragma unroll
  for (int i = 1; i < 10; i++) q = (h.v[i] + q) >> fe_bits(i);
  h.v[0] += 19u * q;
  uint32_t c;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    c = h.v[i] >> fe_bits(i);
    h.v[i] &= fe_mask(i);
    h.v[i + 1] += c;
  }
  h.v[9] &= 0x1ffffffu;
  // pack 26/25-bit limbs at bit offsets 0,26,51,77,102,128,153,179,204,230
  w[0] = h.v[0] | (h.v[1] << 26);
  w[1] = (h.v[1] >> 6) | (h.v[2] << 19);
  w[2] = (h.v[2] >> 13) | (h.v[3] << 13);
  w[3] = (h.v[3] >> 19) | (h.v[4] << 6);
  w[4] = h.v[5] | (h.v[6] << 25);
  w[5] = (h.v[6] >> 7) | (h.v[7] << 19);
  w[6] = (h.v[7] >> 13) | (h.v[8] << 12);
  w[7] = (h.v[8] >> 20) | (h.v[9] << 6);
}

// canonical little-endian bytes
BPP_HD void fe_tobytes(uint8_t s[32], const fe &f) {
  uint32_t w[8];
  fe_towords(w, f);
#pragma unroll
  for (int i = 0; i < 8; i++) {
    s[4 * i + 0] = (uint8_t)(w[i]);
    s[4 * i + 1] = (uint8_t)(w[i] >> 8);
    s[4 * i + 2] = (uint8_t)(w[i] >> 16);
    s[4 * i + 3] = (uint8_t)(w[i] >> 24);
  }
}

// bit 255 ignored (dalek FieldElement::from_bytes)
BPP_HD void fe_fromwords(fe &h, const uint32_t w[8]) {
  h.v[0] = w[0] & 0x3ffffffu;
  h.v[1] = ((w[0] >> 26) | (w[1] << 6)) & 0x1ffffffu;
  h.v[2] = ((w[1] >> 19) | (w[2] << 13)) & 0x3ffffffu;
  h.v[3] = ((w[2] >> 13) | (w[3] << 19)) & 0x1ffffffu;
  h.v[4] = (w[3] >> 6) & 0x3ffffffu;
  h.v[5] = w[4] & 0x1ffffffu;
  h.v[6] = ((w[4] >> 25) | (w[5] << 7)) & 0x3ffffffu;
  h.v[7] = ((w[5] >> 19) | (w[6] << 13)) & 0x1ffffffu;
  h.v[8] = ((w[6] >> 12) | (w[7] << 20)) & 0x3ffffffu;
  h.v[9] = (w[7] >> 6) & 0x1ffffffu;
}
BPP_HD void fe_frombytes(fe &h, const uint8_t s[32]) {
  uint32_t w[8];
#pragma unroll
  for (int i = 0; i < 8; i++)
    w[i] = (uint32_t)s[4 * i] | ((uint32_t)s[4 * i + 1] << 8) | ((uint32_t)s[4 * i + 2] << 16) |
           ((uint32_t)s[4 * i + 3] << 24);
  fe_fromwords(h, w);
}

BPP_HD bool fe_isnegative(const fe &f) {
  uint32_t w[8];
  fe_towords(w, f);
  return w[0] & 1u;
}

BPP_HD bool fe_iszero(const fe &f) {
  uint32_t w[8];
  fe_towords(w, f);
  uint32_t r = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) r |= w[i];
  return r == 0;
}

BPP_HD bool fe_eq(const fe &f, const fe &g) {
  fe d;
  fe_sub(d, f, g);
  return fe_iszero(d);
}

BPP_HD void fe_cmov(fe &h, const fe &g, bool b) {
#pragma unroll
  for (int i = 0; i < 10; i++) h.v[i] = b ? g.v[i] : h.v[i];
}

// |f|: negate if canonical f is odd
BPP_HD void fe_abs(fe &h, const fe &f) {
  fe n;
  fe_neg(n, f);
  bool neg = fe_isnegative(f);
  fe_copy(h, f);
  fe_cmov(h, n, neg);
}

// z^(2^252 - 3)
// Scheduling fence: the ten limbs pass through an empty asm, so the instruction scheduler cannot interleave the field
// operation before it with the one after it.  The addition chains below are strictly sequential, yet without fences
// the scheduler overlaps neighbouring operations until fe_pow22523 alone is allocated 364 VGPRs (k_decompress: 256 +
// 218 spilled); with them it is ~150 and the kernels around it fit 3-4 wavefronts per SIMD without scratch.
BPP_HD void fe_fence(fe &x) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(x.v[0]), "+v"(x.v[1]), "+v"(x.v[2]), "+v"(x.v[3]), "+v"(x.v[4]), "+v"(x.v[5]), "+v"(x.v[6]), "+v"(x.v[7]),
               "+v"(x.v[8]), "+v"(x.v[9]));
#else
  (void)x;
#endif
}

BPP_HD void fe_pow22523(fe &out, const fe &z) {
  fe t0, t1, t2;
  fe_sq(t0, z);          // 2
  fe_fence(t0);
  fe_sqn(t1, t0, 2);     // 8
  fe_fence(t1);
  fe_mul(t1, z, t1);     // 9
  fe_fence(t1);
  fe_mul(t0, t0, t1);    // 11
  fe_fence(t0);
  fe_sq(t0, t0);         // 22
  fe_fence(t0);
  fe_mul(t0, t1, t0);    // 31 = 2^5 - 1
  fe_fence(t0);
  fe_sqn(t1, t0, 5);
  fe_fence(t1);
  fe_mul(t0, t1, t0);    // 2^10 - 1
  fe_fence(t0);
  fe_sqn(t1, t0, 10);
  fe_fence(t1);
  fe_mul(t1, t1, t0);    // 2^20 - 1
  fe_fence(t1);
  fe_sqn(t2, t1, 20);
  fe_fence(t2);
  fe_mul(t1, t2, t1);    // 2^40 - 1
  fe_fence(t1);
  fe_sqn(t1, t1, 10);
  fe_fence(t1);
  fe_mul(t0, t1, t0);    // 2^50 - 1
  fe_fence(t0);
  fe_sqn(t1, t0, 50);
  fe_fence(t1);
  fe_mul(t1, t1, t0);    // 2^100 - 1
  fe_fence(t1);
  fe_sqn(t2, t1, 100);
  fe_fence(t2);
  fe_mul(t1, t2, t1);    // 2^200 - 1
  fe_fence(t1);
  fe_sqn(t1, t1, 50);
  fe_fence(t1);
  fe_mul(t0, t1, t0);    // 2^250 - 1
  fe_fence(t0);
  fe_sqn(t0, t0, 2);     // 2^252 - 4
  fe_fence(t0);
  fe_mul(out, t0, z);    // 2^252 - 3
}

// z^(p-2)
BPP_HD void fe_invert(fe &out, const fe &z) {
  fe t0, t1, t2, t3;
  fe_sq(t0, z);          // 2
  fe_fence(t0);
  fe_sqn(t1, t0, 2);     // 8
  fe_fence(t1);
  fe_mul(t1, z, t1);     // 9
  fe_fence(t1);
  fe_mul(t0, t0, t1);    // 11
  fe_fence(t0);
  fe_sq(t2, t0);         // 22
  fe_fence(t2);
  fe_mul(t1, t1, t2);    // 31
  fe_fence(t1);
  fe_sqn(t2, t1, 5);
  fe_fence(t2);
  fe_mul(t1, t2, t1);    // 2^10 - 1
  fe_fence(t1);
  fe_sqn(t2, t1, 10);
  fe_fence(t2);
  fe_mul(t2, t2, t1);    // 2^20 - 1
  fe_fence(t2);
  fe_sqn(t3, t2, 20);
  fe_fence(t3);
  fe_mul(t2, t3, t2);    // 2^40 - 1
  fe_fence(t2);
  fe_sqn(t2, t2, 10);
  fe_fence(t2);
  fe_mul(t1, t2, t1);    // 2^50 - 1
  fe_fence(t1);
  fe_sqn(t2, t1, 50);
  fe_fence(t2);
  fe_mul(t2, t2, t1);    // 2^100 - 1
  fe_fence(t2);
  fe_sqn(t3, t2, 100);
  fe_fence(t3);
  fe_mul(t2, t3, t2);    // 2^200 - 1
  fe_fence(t2);
  fe_sqn(t2, t2, 50);
  fe_fence(t2);
  fe_mul(t1, t2, t1);    // 2^250 - 1
  fe_fence(t1);
  fe_sqn(t1, t1, 5);     // 2^255 - 32
  fe_fence(t1);
  fe_mul(out, t1, t0);   // 2^255 - 21
}

}  // namespace bpp
