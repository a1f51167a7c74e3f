// ristretto255 group operations for gfx950 (twisted Edwards a = -1, extended coordinates).
//
//   ge      (X:Y:Z:T)            accumulator form, 160 B
//   niels   (y+x, y-x, 2dxy)     affine operand form, 120 B padded to ONE aligned 128-byte line -- every MSM input point
//                                (generator tables, a batch's decoded points, fixed-base windows) is stored like this, so a
//                                gather touches exactly one line (an unpadded entry straddles two of them 94 % of the time)
//
// Replaces (reference boundary): RistrettoPoint / CompressedRistretto of curve25519-dalek as used at
// src/range_proof.rs:1050-1057 (final MSM), :1067-1109 (decompress), :348,:499-504 (compress),
// src/generators/generators_chain.rs:43-49 (from_uniform_bytes).  Algorithms: RFC 9496 4.3.1-4.3.4.
#pragma once
#include "field.h"

namespace bpp {

#include "field_consts.inc"

BPP_HD void fe_const(fe &h, const uint32_t c[10]) {
#pragma unroll
  for (int i = 0; i < 10; i++) h.v[i] = c[i];
}

struct ge {
  fe X, Y, Z, T;
};
struct alignas(128) niels {
  fe yplusx, yminusx, xy2d;
};
static_assert(sizeof(niels) == 128, "one table entry = one 128-byte line");

// table entry -> registers for a term with sign `neg`: -(x, y) = (-x, y) exchanges y+x and y-x, which costs nothing when
// done by address while loading; the sign of 2dxy is handled inside ge_madd_swapped
BPP_HD void niels_load_swapped(niels &q, const niels *src, bool neg) {
  const fe *pair = &src->yplusx;  // pair[0] = y+x, pair[1] = y-x
  q.yplusx = pair[neg ? 1 : 0];
  q.yminusx = pair[neg ? 0 : 1];
  q.xy2d = src->xy2d;
}

BPP_HD void ge_identity(ge &r) {
  fe_0(r.X);
  fe_1(r.Y);
  fe_1(r.Z);
  fe_0(r.T);
}

BPP_HD void niels_identity(niels &r) {
  fe_1(r.yplusx);
  fe_1(r.yminusx);
  fe_0(r.xy2d);
}

// affine (x, y) -> niels
BPP_HD void niels_from_affine(niels &r, const fe &x, const fe &y) {
  fe t, d2;
  fe_add(r.yplusx, y, x);
  fe_carry(r.yplusx);
  fe_sub(r.yminusx, y, x);
  fe_mul(t, x, y);
  fe_const(d2, FE_D2);
  fe_mul(r.xy2d, t, d2);
}

// q = neg ? -q : q   (-(x,y) = (-x,y): swap y+x / y-x, negate 2dxy)
BPP_HD void niels_cneg(niels &q, bool neg) {
  fe n;
  fe_neg(n, q.xy2d);
#pragma unroll
  for (int i = 0; i < 10; i++) {
    const uint32_t a = q.yplusx.v[i], b = q.yminusx.v[i];
    q.yplusx.v[i] = neg ? b : a;
    q.yminusx.v[i] = neg ? a : b;
    q.xy2d.v[i] = neg ? n.v[i] : q.xy2d.v[i];
  }
}

// r = p + q (q affine niels): 7 mul.  Lazily reduced (field.h limb classes): with reduced coordinates in, Y+X and 2Z+c
// are loose, Y-X and a-b are lazy differences (loose), 2Z-c is wide; every product pairs a wide-or-smaller first operand
// with a loose-or-smaller second one, so no carry pass runs outside the seven products.
BPP_HD void ge_madd(ge &r, const ge &p, const niels &q) {
  fe a, b, c, e, f, g, h;
  fe_add(a, p.Y, p.X);
  fe_sub_lazy(b, p.Y, p.X);
  fe_mul(a, a, q.yplusx);
  fe_mul(b, b, q.yminusx);
  fe_mul(c, q.xy2d, p.T);
  fe_sub_lazy(e, a, b);
  fe_add(h, a, b);
  fe_dbl_add(g, p.Z, c);       // d + c, d = 2Z
  fe_dbl_sub_lazy(f, p.Z, c);  // d - c
  fe_mul(r.X, f, e);
  fe_mul(r.Y, g, h);
  fe_mul(r.Z, f, g);
  fe_mul(r.T, e, h);
}

// r = p + (neg ? -q : q) where the CALLER has already exchanged q.yplusx / q.yminusx when neg (-(x, y) = (-x, y) swaps
// y+x with y-x: free if done while loading the table entry) and q.xy2d is the entry's own: negating 2dxy negates c, which
// only makes d - c and d + c trade places.  Branch-free: lanes of one wavefront mix additions and subtractions.
BPP_HD void ge_madd_swapped(ge &r, const ge &p, const niels &q, bool neg) {
  fe a, b, c, e, f, g, h, u, v;
  fe_add(a, p.Y, p.X);
  fe_mul(a, a, q.yplusx);
  fe_sub_lazy(b, p.Y, p.X);
  fe_mul(b, b, q.yminusx);
  fe_mul(c, q.xy2d, p.T);
  fe_sub_lazy(e, a, b);
  fe_add(h, a, b);
  fe_dbl_add(u, p.Z, c);       // loose
  fe_dbl_sub_lazy(v, p.Z, c);  // wide
  fe_mul(r.Z, v, u);  // g * f either way
#pragma unroll
  for (int i = 0; i < 10; i++) {  // (and / xor selects instead of these 20 v_cndmask_b32: measured, no difference)
    f.v[i] = neg ? u.v[i] : v.v[i];
    g.v[i] = neg ? v.v[i] : u.v[i];
  }
  fe_mul(r.X, f, e);
  fe_mul(r.Y, g, h);
  fe_mul(r.T, e, h);
}

// r = p - q (q affine niels)
BPP_HD void ge_msub(ge &r, const ge &p, const niels &q) {
  fe a, b, c, e, f, g, h;
  fe_add(a, p.Y, p.X);
  fe_sub_lazy(b, p.Y, p.X);
  fe_mul(a, a, q.yminusx);
  fe_mul(b, b, q.yplusx);
  fe_mul(c, q.xy2d, p.T);
  fe_sub_lazy(e, a, b);
  fe_add(h, a, b);
  fe_dbl_sub_lazy(g, p.Z, c);  // d - c
  fe_dbl_add(f, p.Z, c);       // d + c
  fe_mul(r.X, f, e);
  fe_mul(r.Y, g, h);
  fe_mul(r.Z, g, f);
  fe_mul(r.T, e, h);
}

// r = p + q, both extended: 9 mul, no carry pass outside the products (same limb classes as ge_madd)
BPP_HD void ge_add(ge &r, const ge &p, const ge &q) {
  fe a, b, c, zz, e, f, g, h, t, d2;
  fe_sub_lazy(a, p.Y, p.X);
  fe_sub_lazy(t, q.Y, q.X);
  fe_mul(a, a, t);
  fe_add(b, p.Y, p.X);
  fe_add(t, q.Y, q.X);
  fe_mul(b, b, t);
  fe_const(d2, FE_D2);
  fe_mul(c, p.T, q.T);
  fe_mul(c, c, d2);
  fe_mul(zz, p.Z, q.Z);
  fe_sub_lazy(e, b, a);
  fe_add(h, b, a);
  fe_dbl_sub_lazy(f, zz, c);  // d - c, d = 2 Z1 Z2: wide
  fe_dbl_add(g, zz, c);       // d + c: loose
  fe_mul(r.X, f, e);
  fe_mul(r.Y, g, h);
  fe_mul(r.Z, f, g);
  fe_mul(r.T, e, h);
}

// one doubling step on (X, Y, Z): e (wide), f (reduced), g, h (loose) with 2p = (e f : g h : f g : e h)
BPP_HD void ge_dbl_efgh(fe &e, fe &f, fe &g, fe &h, const fe &X, const fe &Y, const fe &Z) {
  fe a, b, zz, t;
  fe_sq(a, X);
  fe_sq(b, Y);
  fe_sq(zz, Z);
  fe_add(t, X, Y);
  fe_sq(t, t);
  fe_add(h, a, b);
  fe_sub_lazy(e, h, t);  // a + b - (X + Y)^2 + 2p: wide
  fe_sub_lazy(g, a, b);
  fe_dbl_add(f, zz, g);  // 2 Z^2 + a - b + 2p <= 2.5 * 2^27: the one value that needs a carry pass
  fe_carry(f);
}

// r = 2p: 4 sq + 4 mul
BPP_HD void ge_dbl(ge &r, const ge &p) {
  fe e, f, g, h;
  ge_dbl_efgh(e, f, g, h, p.X, p.Y, p.Z);
  fe_mul(r.X, e, f);
  fe_mul(r.Y, g, h);
  fe_mul(r.Z, g, f);
  fe_mul(r.T, e, h);
}

// r = 2^n p (n >= 1): only the last doubling needs T (3M + 4S for the others)
BPP_HD void ge_dbl_n(ge &r, const ge &p, int n) {
  fe X, Y, Z;
  fe_copy(X, p.X);
  fe_copy(Y, p.Y);
  fe_copy(Z, p.Z);
  for (int i = 0; i < n; i++) {
    fe e, f, g, h;
    ge_dbl_efgh(e, f, g, h, X, Y, Z);
    fe_mul(X, e, f);
    fe_mul(Y, g, h);
    fe_mul(Z, g, f);
    if (i == n - 1) fe_mul(r.T, e, h);
  }
  fe_copy(r.X, X);
  fe_copy(r.Y, Y);
  fe_copy(r.Z, Z);
}

BPP_HD void ge_neg(ge &r, const ge &p) {
  fe_neg(r.X, p.X);
  fe_copy(r.Y, p.Y);
  fe_copy(r.Z, p.Z);
  fe_neg(r.T, p.T);
}

// RFC 9496 4.2 SQRT_RATIO_M1
BPP_HD bool fe_sqrt_ratio_m1(fe &r_out, const fe &u, const fe &v) {
  fe v3, v7, r, check, t, neg_u, neg_u_i, sqrt_m1, rp;
  fe_sq(v3, v);
  fe_mul(v3, v3, v);
  fe_sq(v7, v3);
  fe_mul(v7, v7, v);
  fe_mul(t, u, v7);
  fe_pow22523(r, t);
  fe_mul(t, u, v3);
  fe_mul(r, r, t);
  fe_sq(check, r);
  fe_mul(check, check, v);
  fe_const(sqrt_m1, FE_SQRT_M1);
  fe_neg(neg_u, u);
  fe_mul(neg_u_i, neg_u, sqrt_m1);
  bool correct_sign = fe_eq(check, u);
  bool flipped_sign = fe_eq(check, neg_u);
  bool flipped_sign_i = fe_eq(check, neg_u_i);
  fe_mul(rp, r, sqrt_m1);
  fe_cmov(r, rp, flipped_sign || flipped_sign_i);
  fe_abs(r_out, r);
  return correct_sign || flipped_sign;
}

// RFC 9496 4.3.1 Decode -> affine niels.  false = not a canonical encoding of a point.
BPP_HD bool ristretto_decompress(niels &out, const uint8_t s_bytes[32]) {
  fe s, ss, u1, u2, u2_sqr, v, t, one, d, invsqrt, den_x, den_y, x, y;
  fe_frombytes(s, s_bytes);
  // canonical: re-encoding must reproduce the input (rejects >= p and bit 255); non-negative: low bit 0
  uint8_t chk[32];
  fe_tobytes(chk, s);
  uint32_t diff = 0;
#pragma unroll
  for (int i = 0; i < 32; i++) diff |= (uint32_t)(chk[i] ^ s_bytes[i]);
  bool ok = (diff == 0) && ((s_bytes[0] & 1) == 0);
  fe_1(one);
  fe_sq(ss, s);
  fe_sub(u1, one, ss);
  fe_add(u2, one, ss);
  fe_carry(u2);
  fe_sq(u2_sqr, u2);
  fe_const(d, FE_D);
  fe_sq(t, u1);
  fe_mul(t, t, d);
  fe_neg(t, t);
  fe_sub(v, t, u2_sqr);
  fe_mul(t, v, u2_sqr);
  bool was_square = fe_sqrt_ratio_m1(invsqrt, one, t);
  fe_mul(den_x, invsqrt, u2);
  fe_mul(den_y, invsqrt, den_x);
  fe_mul(den_y, den_y, v);
  fe_mul(x, s, den_x);
  fe_add(x, x, x);
  fe_abs(x, x);
  fe_mul(y, u1, den_y);
  fe_mul(t, x, y);
  ok = ok && was_square && !fe_isnegative(t) && !fe_iszero(y);
  niels_from_affine(out, x, y);
  return ok;
}

// Same function, scheduled for registers: the 254-squaring chain runs with only (w, accumulator) live; everything the
// epilogue needs (u1, u2, v) is RECOMPUTED from the input bytes afterwards (6 multiplications out of ~270) instead of
// being kept alive across the chain.  The compiler barrier keeps it from merging the two computations again.
// `spill` (optional): 30 words of this lane in a limb-major scratch array (word k at spill[k * stride]).  With it the three
// values the epilogue shares with the prologue (s^2, v, w) are parked there across the chain instead of being recomputed:
// 60 memory instructions instead of 3 squarings + 2 multiplications (~660 VALU instructions, 2.4 % of the function).
BPP_HD bool ristretto_decompress_lean(niels &out, const uint8_t *s_bytes, uint32_t *spill = nullptr, size_t stride = 0) {
  fe r;
  {
    fe s, ss, u1, u2, u2_sqr, v, t, one, d, w, v3, v7;
    uint32_t sw[8];
#pragma unroll
    for (int i = 0; i < 8; i++)
      sw[i] = (uint32_t)s_bytes[4 * i] | ((uint32_t)s_bytes[4 * i + 1] << 8) | ((uint32_t)s_bytes[4 * i + 2] << 16) |
              ((uint32_t)s_bytes[4 * i + 3] << 24);
    fe_fromwords(s, sw);
    fe_1(one);
    fe_sq(ss, s);
  fe_fence(ss);
    fe_sub(u1, one, ss);
    fe_add(u2, one, ss);
    fe_carry(u2);
    fe_sq(u2_sqr, u2);
  fe_fence(u2_sqr);
    fe_const(d, FE_D);
    fe_sq(t, u1);
  fe_fence(t);
    fe_mul(t, t, d);
  fe_fence(t);
    fe_neg(t, t);
    fe_sub(v, t, u2_sqr);
    fe_mul(w, v, u2_sqr);
  fe_fence(w);
    // SQRT_RATIO_M1(1, w).  dalek computes r0 = w^3 (w^7)^((p-5)/8); r = w^((p-5)/8) differs from it by the factor
    // c^3, c = w^((p-1)/4) a fourth root of unity, so w r^2 = c where dalek sees c^7 = c^-1: "1" and "-1" (w is a square:
    // the root is r resp. r*sqrt(-1)) are recognised identically and give the same non-negative root; for a non-square
    // the decoding fails either way and r is not used.  Saves 2 squarings + 3 multiplications per point.
    if (spill) {
#pragma unroll
      for (int i = 0; i < 10; i++) {
        spill[(size_t)i * stride] = ss.v[i];
        spill[(size_t)(10 + i) * stride] = v.v[i];
        spill[(size_t)(20 + i) * stride] = w.v[i];
      }
    }
    fe_pow22523(r, w);
    (void)v3;
    (void)v7;
  }
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" ::: "memory");
#endif
  fe s, ss, u1, u2, u2_sqr, v, t, one, d, w, check, sqrt_m1, neg_one, neg_i, rp, den_x, den_y, x, y;
  uint32_t sw[8], chk[8];
#pragma unroll
  for (int i = 0; i < 8; i++)
    sw[i] = (uint32_t)s_bytes[4 * i] | ((uint32_t)s_bytes[4 * i + 1] << 8) | ((uint32_t)s_bytes[4 * i + 2] << 16) |
            ((uint32_t)s_bytes[4 * i + 3] << 24);
  fe_fromwords(s, sw);
  fe_towords(chk, s);
  uint32_t diff = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) diff |= chk[i] ^ sw[i];  // canonical: re-encoding gives the same 256 bits (bit 255 clear)
  bool ok = (diff == 0) && ((sw[0] & 1u) == 0);
  fe_1(one);
  if (spill) {
#pragma unroll
    for (int i = 0; i < 10; i++) {
      ss.v[i] = spill[(size_t)i * stride];
      v.v[i] = spill[(size_t)(10 + i) * stride];
      w.v[i] = spill[(size_t)(20 + i) * stride];
    }
    fe_sub(u1, one, ss);
    fe_add(u2, one, ss);
    fe_carry(u2);
  } else {
    fe_sq(ss, s);
    fe_fence(ss);
    fe_sub(u1, one, ss);
    fe_add(u2, one, ss);
    fe_carry(u2);
    fe_sq(u2_sqr, u2);
    fe_fence(u2_sqr);
    fe_const(d, FE_D);
    fe_sq(t, u1);
    fe_fence(t);
    fe_mul(t, t, d);
    fe_fence(t);
    fe_neg(t, t);
    fe_sub(v, t, u2_sqr);
    fe_mul(w, v, u2_sqr);
    fe_fence(w);
  }
  fe_sq(check, r);
  fe_fence(check);
  fe_mul(check, check, w);
  fe_fence(check);
  fe_const(sqrt_m1, FE_SQRT_M1);
  fe_neg(neg_one, one);
  fe_neg(neg_i, sqrt_m1);
  const bool correct_sign = fe_eq(check, one);
  const bool flipped_sign = fe_eq(check, neg_one);
  const bool flipped_sign_i = fe_eq(check, neg_i);
  fe_mul(rp, r, sqrt_m1);
  fe_fence(rp);
  fe_cmov(r, rp, flipped_sign || flipped_sign_i);
  fe_abs(r, r);
  const bool was_square = correct_sign || flipped_sign;
  fe_mul(den_x, r, u2);
  fe_fence(den_x);
  fe_mul(den_y, r, den_x);
  fe_fence(den_y);
  fe_mul(den_y, den_y, v);
  fe_fence(den_y);
  fe_mul(x, s, den_x);
  fe_fence(x);
  fe_add(x, x, x);
  fe_abs(x, x);
  fe_mul(y, u1, den_y);
  fe_fence(y);
  fe_mul(t, x, y);
  fe_fence(t);
  ok = ok && was_square && !fe_isnegative(t) && !fe_iszero(y);
  niels_from_affine(out, x, y);
  return ok;
}

// RFC 9496 4.3.2 Encode
BPP_HD void ristretto_compress(uint8_t out[32], const ge &p) {
  fe u1, u2, t, one, invsqrt, den1, den2, z_inv, ix0, iy0, ench, x, y, den_inv, sqrt_m1, c, s;
  fe_add(u1, p.Z, p.Y);
  fe_sub(t, p.Z, p.Y);
  fe_mul(u1, u1, t);
  fe_mul(u2, p.X, p.Y);
  fe_sq(t, u2);
  fe_mul(t, t, u1);
  fe_1(one);
  fe_sqrt_ratio_m1(invsqrt, one, t);
  fe_mul(den1, invsqrt, u1);
  fe_mul(den2, invsqrt, u2);
  fe_mul(z_inv, den1, den2);
  fe_mul(z_inv, z_inv, p.T);
  fe_const(sqrt_m1, FE_SQRT_M1);
  fe_mul(ix0, p.X, sqrt_m1);
  fe_mul(iy0, p.Y, sqrt_m1);
  fe_const(c, FE_INVSQRT_A_MINUS_D);
  fe_mul(ench, den1, c);
  fe_mul(t, p.T, z_inv);
  bool rotate = fe_isnegative(t);
  fe_copy(x, p.X);
  fe_copy(y, p.Y);
  fe_copy(den_inv, den2);
  fe_cmov(x, iy0, rotate);
  fe_cmov(y, ix0, rotate);
  fe_cmov(den_inv, ench, rotate);
  fe_mul(t, x, z_inv);
  fe ny;
  fe_neg(ny, y);
  fe_cmov(y, ny, fe_isnegative(t));
  fe_sub(t, p.Z, y);
  fe_mul(s, den_inv, t);
  fe_abs(s, s);
  fe_tobytes(out, s);
}

// RFC 9496 4.3.4 MAP (Elligator 2 to the Jacobi quartic, then to Edwards)
BPP_HD void ristretto_elligator(ge &out, const fe &t_in) {
  fe r, u, v, c, s, s_prime, n, w0, w1, w2, w3, one, d, t, k;
  fe_1(one);
  fe_const(k, FE_SQRT_M1);
  fe_sq(r, t_in);
  fe_mul(r, r, k);  // r = i t^2
  fe_add(u, r, one);
  fe_const(k, FE_ONE_MINUS_D_SQ);
  fe_mul(u, u, k);
  fe_const(d, FE_D);
  fe_mul(t, r, d);
  fe_add(t, t, one);
  fe_neg(t, t);  // -1 - r d
  fe_add(v, r, d);
  fe_mul(v, v, t);
  bool was_square = fe_sqrt_ratio_m1(s, u, v);
  fe_mul(s_prime, s, t_in);
  fe_abs(s_prime, s_prime);
  fe_neg(s_prime, s_prime);
  fe_cmov(s, s_prime, !was_square);
  fe_neg(c, one);
  fe_cmov(c, r, !was_square);
  fe_sub(t, r, one);
  fe_mul(n, c, t);
  fe_const(k, FE_D_MINUS_ONE_SQ);
  fe_mul(n, n, k);
  fe_sub(n, n, v);
  fe_mul(w0, s, v);
  fe_add(w0, w0, w0);
  fe_const(k, FE_SQRT_AD_MINUS_ONE);
  fe_mul(w1, n, k);
  fe_sq(t, s);
  fe_sub(w2, one, t);
  fe_add(w3, one, t);
  fe_carry(w3);
  fe_mul(out.X, w0, w3);
  fe_mul(out.Y, w2, w1);
  fe_mul(out.Z, w1, w3);
  fe_mul(out.T, w0, w2);
}

// RistrettoPoint::from_uniform_bytes
BPP_HD void ristretto_from_uniform(ge &out, const uint8_t b[64]) {
  fe r0, r1;
  ge p0, p1;
  fe_frombytes(r0, b);
  fe_frombytes(r1, b + 32);
  ristretto_elligator(p0, r0);
  ristretto_elligator(p1, r1);
  ge_add(out, p0, p1);
}

// affine niels -> extended with ONE multiplication: y+x and y-x give e = 2x and h = 2y, and (2e : 2h : 4 : e h) is the
// point (T = X Y / Z).  This is what "identity + q" evaluates to (every other product of the mixed addition is then a
// multiplication by 2 or 4), so a bucket's first term costs one product instead of seven.  A sign applied by
// niels_load_swapped carries over by itself: e changes sign, and T = e h with it.  Output coordinates are reduced.
BPP_HD void ge_from_niels_first(ge &r, const niels &q) {
  fe e, h;
  fe_sub_lazy(e, q.yplusx, q.yminusx);
  fe_add(h, q.yplusx, q.yminusx);
  fe_mul(r.T, e, h);
#pragma unroll
  for (int i = 0; i < 10; i++) {
    r.X.v[i] = 2u * e.v[i];
    r.Y.v[i] = 2u * h.v[i];
  }
  fe_carry(r.X);
  fe_carry(r.Y);
  fe_0(r.Z);
  r.Z.v[0] = 4;
}

// affine niels -> extended, no inversion: (2x : 2y : 2 : 2xy) with 2xy = (2dxy) / d
BPP_HD void ge_from_niels(ge &r, const niels &q) {
  fe dinv;
  fe_const(dinv, FE_D_INV);
  fe_sub(r.X, q.yplusx, q.yminusx);
  fe_carry(r.X);
  fe_add(r.Y, q.yplusx, q.yminusx);
  fe_carry(r.Y);
  fe_0(r.Z);
  r.Z.v[0] = 2;
  fe_mul(r.T, q.xy2d, dinv);
}

// extended -> affine niels (one inversion)
BPP_HD void ge_to_niels(niels &out, const ge &p) {
  fe zi, x, y;
  fe_invert(zi, p.Z);
  fe_mul(x, p.X, zi);
  fe_mul(y, p.Y, zi);
  niels_from_affine(out, x, y);
}

// ristretto identity test: X == 0 or Y == 0 (dalek ct_eq against (0,1,1,0))
BPP_HD bool ge_is_ristretto_identity(const ge &p) { return fe_iszero(p.X) || fe_iszero(p.Y); }

}  // namespace bpp
