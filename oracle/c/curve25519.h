/* ORACLE (test infrastructure / CPU baseline only -- never linked into the product).
 *
 * Plain-C restatement of the curve25519-dalek 4.1.3 arithmetic the reference reaches (serial u64 backend):
 *   field   5 x 51-bit limbs, 128-bit products                       [dalek backend/serial/u64/field.rs]
 *   scalar  4 x 64-bit limbs, Montgomery (dalek: 5 x 52 Montgomery)   [dalek backend/serial/u64/scalar.rs]
 *   points  extended / projective-niels / affine-niels, P2 doubling   [dalek backend/serial/curve_models]
 *   MSM     Straus with width-5 NAF (dynamic) + width-8 NAF precomputed affine-niels tables (static), shared
 *           doublings; Pippenger with signed radix-2^w digits above 190 terms  [dalek scalar_mul/{straus,
 *           precomputed_straus,pippenger}.rs]  -- "what dalek does" per SURVEY 2.1 K1/K2, written from the published
 *           algorithms, not from dalek source (not available here).
 * dalek is not under /root/reference (pinned supply-chain/config.toml:76-77).  Pinned by RFC 9496 KATs and by
 * agreement with oracle/pyref (tests/test_oracle_c.py).
 */
#ifndef ORACLE_CURVE25519_H
#define ORACLE_CURVE25519_H
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------ field */
typedef struct { uint64_t v[5]; } fe;
#define M51 0x7ffffffffffffULL

static inline void fe_0(fe *h) { memset(h, 0, sizeof(*h)); }
static inline void fe_1(fe *h) { fe_0(h); h->v[0] = 1; }
static inline void fe_add(fe *h, const fe *f, const fe *g) { for (int i = 0; i < 5; i++) h->v[i] = f->v[i] + g->v[i]; }
static inline void fe_carry(fe *h) {
  uint64_t c;
  for (int i = 0; i < 4; i++) { c = h->v[i] >> 51; h->v[i] &= M51; h->v[i + 1] += c; }
  c = h->v[4] >> 51; h->v[4] &= M51; h->v[0] += 19 * c;
  c = h->v[0] >> 51; h->v[0] &= M51; h->v[1] += c;
}
/* f - g with a 16p bias, carried */
static inline void fe_sub(fe *h, const fe *f, const fe *g) {
  h->v[0] = f->v[0] + 0x7ffffffffffed0ULL - g->v[0];
  for (int i = 1; i < 5; i++) h->v[i] = f->v[i] + 0x7ffffffffffff0ULL - g->v[i];
  fe_carry(h);
}
static inline void fe_neg(fe *h, const fe *f) { fe z; fe_0(&z); fe_sub(h, &z, f); }
static inline void fe_mul(fe *h, const fe *f, const fe *g) {
  const uint64_t *a = f->v, *b = g->v;
  uint64_t b1 = 19 * b[1], b2 = 19 * b[2], b3 = 19 * b[3], b4 = 19 * b[4];
  u128 r0 = (u128)a[0] * b[0] + (u128)a[1] * b4 + (u128)a[2] * b3 + (u128)a[3] * b2 + (u128)a[4] * b1;
  u128 r1 = (u128)a[0] * b[1] + (u128)a[1] * b[0] + (u128)a[2] * b4 + (u128)a[3] * b3 + (u128)a[4] * b2;
  u128 r2 = (u128)a[0] * b[2] + (u128)a[1] * b[1] + (u128)a[2] * b[0] + (u128)a[3] * b4 + (u128)a[4] * b3;
  u128 r3 = (u128)a[0] * b[3] + (u128)a[1] * b[2] + (u128)a[2] * b[1] + (u128)a[3] * b[0] + (u128)a[4] * b4;
  u128 r4 = (u128)a[0] * b[4] + (u128)a[1] * b[3] + (u128)a[2] * b[2] + (u128)a[3] * b[1] + (u128)a[4] * b[0];
  uint64_t c;
  r1 += (uint64_t)(r0 >> 51); h->v[0] = (uint64_t)r0 & M51;
  r2 += (uint64_t)(r1 >> 51); h->v[1] = (uint64_t)r1 & M51;
  r3 += (uint64_t)(r2 >> 51); h->v[2] = (uint64_t)r2 & M51;
  r4 += (uint64_t)(r3 >> 51); h->v[3] = (uint64_t)r3 & M51;
  c = (uint64_t)(r4 >> 51); h->v[4] = (uint64_t)r4 & M51;
  h->v[0] += 19 * c; c = h->v[0] >> 51; h->v[0] &= M51; h->v[1] += c;
}
static inline void fe_sq(fe *h, const fe *f) { fe_mul(h, f, f); }
static inline void fe_sqn(fe *h, const fe *f, int n) { fe_sq(h, f); for (int i = 1; i < n; i++) fe_sq(h, h); }
static inline void fe_tobytes(uint8_t s[32], const fe *f) {
  fe h = *f; fe_carry(&h); fe_carry(&h);
  uint64_t q = (h.v[0] + 19) >> 51;
  for (int i = 1; i < 5; i++) q = (h.v[i] + q) >> 51;
  h.v[0] += 19 * q;
  uint64_t c;
  for (int i = 0; i < 4; i++) { c = h.v[i] >> 51; h.v[i] &= M51; h.v[i + 1] += c; }
  h.v[4] &= M51;
  uint64_t w[4] = { h.v[0] | (h.v[1] << 51), (h.v[1] >> 13) | (h.v[2] << 38), (h.v[2] >> 26) | (h.v[3] << 25),
                    (h.v[3] >> 39) | (h.v[4] << 12) };
  for (int i = 0; i < 4; i++) for (int k = 0; k < 8; k++) s[8 * i + k] = (uint8_t)(w[i] >> (8 * k));
}
static inline uint64_t load64(const uint8_t *p) { uint64_t w = 0; for (int k = 0; k < 8; k++) w |= (uint64_t)p[k] << (8 * k); return w; }
static inline void fe_frombytes(fe *h, const uint8_t s[32]) { /* bit 255 ignored */
  uint64_t w0 = load64(s), w1 = load64(s + 8), w2 = load64(s + 16), w3 = load64(s + 24);
  h->v[0] = w0 & M51; h->v[1] = ((w0 >> 51) | (w1 << 13)) & M51; h->v[2] = ((w1 >> 38) | (w2 << 26)) & M51;
  h->v[3] = ((w2 >> 25) | (w3 << 39)) & M51; h->v[4] = (w3 >> 12) & M51;
}
static inline int fe_isnegative(const fe *f) { uint8_t s[32]; fe_tobytes(s, f); return s[0] & 1; }
static inline int fe_iszero(const fe *f) { uint8_t s[32]; fe_tobytes(s, f); uint8_t r = 0; for (int i = 0; i < 32; i++) r |= s[i]; return r == 0; }
static inline int fe_eq(const fe *f, const fe *g) { fe d; fe_sub(&d, f, g); return fe_iszero(&d); }
static inline void fe_abs(fe *h, const fe *f) { fe n; fe_neg(&n, f); *h = fe_isnegative(f) ? n : *f; }
static void fe_pow22523(fe *out, const fe *z) {
  fe t0, t1, t2;
  fe_sq(&t0, z); fe_sqn(&t1, &t0, 2); fe_mul(&t1, z, &t1); fe_mul(&t0, &t0, &t1); fe_sq(&t0, &t0); fe_mul(&t0, &t1, &t0);
  fe_sqn(&t1, &t0, 5); fe_mul(&t0, &t1, &t0); fe_sqn(&t1, &t0, 10); fe_mul(&t1, &t1, &t0); fe_sqn(&t2, &t1, 20);
  fe_mul(&t1, &t2, &t1); fe_sqn(&t1, &t1, 10); fe_mul(&t0, &t1, &t0); fe_sqn(&t1, &t0, 50); fe_mul(&t1, &t1, &t0);
  fe_sqn(&t2, &t1, 100); fe_mul(&t1, &t2, &t1); fe_sqn(&t1, &t1, 50); fe_mul(&t0, &t1, &t0); fe_sqn(&t0, &t0, 2);
  fe_mul(out, &t0, z);
}
static void fe_invert(fe *out, const fe *z) {
  fe t0, t1, t2, t3;
  fe_sq(&t0, z); fe_sqn(&t1, &t0, 2); fe_mul(&t1, z, &t1); fe_mul(&t0, &t0, &t1); fe_sq(&t2, &t0); fe_mul(&t1, &t1, &t2);
  fe_sqn(&t2, &t1, 5); fe_mul(&t1, &t2, &t1); fe_sqn(&t2, &t1, 10); fe_mul(&t2, &t2, &t1); fe_sqn(&t3, &t2, 20);
  fe_mul(&t2, &t3, &t2); fe_sqn(&t2, &t2, 10); fe_mul(&t1, &t2, &t1); fe_sqn(&t2, &t1, 50); fe_mul(&t2, &t2, &t1);
  fe_sqn(&t3, &t2, 100); fe_mul(&t2, &t3, &t2); fe_sqn(&t2, &t2, 50); fe_mul(&t1, &t2, &t1); fe_sqn(&t1, &t1, 5);
  fe_mul(out, &t1, &t0);
}

/* constants, derived at start-up from p and d = -121665/121666 (no magic tables) */
static fe FE_D, FE_D2, FE_SQRT_M1, FE_ONE_MINUS_D_SQ, FE_D_MINUS_ONE_SQ, FE_SQRT_AD_MINUS_ONE, FE_INVSQRT_A_MINUS_D;
static void fe_from_u64(fe *h, uint64_t x) { fe_0(h); h->v[0] = x & M51; h->v[1] = x >> 51; }
static int fe_sqrt_ratio_m1(fe *r_out, const fe *u, const fe *v);

/* ------------------------------------------------------------------ scalars */
typedef struct { uint64_t v[4]; } sc;
static const uint64_t SC_L[4] = { 0x5812631a5cf5d3edULL, 0x14def9dea2f79cd6ULL, 0, 0x1000000000000000ULL };
static uint64_t SC_LFACTOR;          /* -l^-1 mod 2^64 */
static sc SC_R1, SC_R2, SC_R3;       /* R, R^2, R^3 mod l */

static inline int sc_geq_l(const uint64_t a[4]) {
  for (int i = 3; i >= 0; i--) { if (a[i] > SC_L[i]) return 1; if (a[i] < SC_L[i]) return 0; }
  return 1;
}
static inline void sc_cond_sub(uint64_t a[4], uint64_t top) {
  if (top || sc_geq_l(a)) { u128 b = 0; for (int i = 0; i < 4; i++) { u128 d = (u128)a[i] - SC_L[i] - (uint64_t)b; a[i] = (uint64_t)d; b = (d >> 127) & 1; } }
}
static inline void sc_add(sc *r, const sc *a, const sc *b) {
  u128 c = 0; uint64_t t[4];
  for (int i = 0; i < 4; i++) { c += (u128)a->v[i] + b->v[i]; t[i] = (uint64_t)c; c >>= 64; }
  sc_cond_sub(t, (uint64_t)c); memcpy(r->v, t, 32);
}
static inline void sc_sub(sc *r, const sc *a, const sc *b) {
  uint64_t t[4]; u128 br = 0;
  for (int i = 0; i < 4; i++) { u128 d = (u128)a->v[i] - b->v[i] - (uint64_t)br; t[i] = (uint64_t)d; br = (d >> 127) & 1; }
  if (br) { u128 c = 0; for (int i = 0; i < 4; i++) { c += (u128)t[i] + SC_L[i]; t[i] = (uint64_t)c; c >>= 64; } }
  memcpy(r->v, t, 32);
}
static inline void sc_neg(sc *r, const sc *a) { sc z = { {0, 0, 0, 0} }; sc_sub(r, &z, a); }
static inline int sc_iszero(const sc *a) { return (a->v[0] | a->v[1] | a->v[2] | a->v[3]) == 0; }
static inline int sc_eq(const sc *a, const sc *b) { return memcmp(a->v, b->v, 32) == 0; }
static void sc_montmul(sc *r, const sc *a, const sc *b) {
  uint64_t t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    u128 c = 0;
    for (int j = 0; j < 4; j++) { c += (u128)a->v[j] * b->v[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
    c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
    uint64_t m = t[0] * SC_LFACTOR;
    c = (u128)m * SC_L[0] + t[0]; c >>= 64;
    for (int j = 1; j < 4; j++) { c += (u128)m * SC_L[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
    c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
  }
  sc_cond_sub(t, t[4]); memcpy(r->v, t, 32);
}
static inline void sc_to_mont(sc *r, const sc *a) { sc_montmul(r, a, &SC_R2); }
static inline void sc_from_mont(sc *r, const sc *a) { sc one = { {1, 0, 0, 0} }; sc_montmul(r, a, &one); }
/* the oracle keeps scalars in NORMAL form; multiplication = two Montgomery products */
static inline void sc_mul(sc *r, const sc *a, const sc *b) { sc t; sc_montmul(&t, a, b); sc_montmul(r, &t, &SC_R2); }
static inline void sc_from_bytes(sc *r, const uint8_t s[32]) { for (int i = 0; i < 4; i++) r->v[i] = load64(s + 8 * i); }
static inline void sc_to_bytes(uint8_t s[32], const sc *a) { for (int i = 0; i < 4; i++) for (int k = 0; k < 8; k++) s[8 * i + k] = (uint8_t)(a->v[i] >> (8 * k)); }
static inline int sc_is_canonical(const uint8_t s[32]) { sc a; sc_from_bytes(&a, s); return !sc_geq_l(a.v); }
static inline void sc_from_u64(sc *r, uint64_t x) { r->v[0] = x; r->v[1] = r->v[2] = r->v[3] = 0; }
/* Scalar::from_bytes_mod_order_wide */
static void sc_from_wide(sc *r, const uint8_t s[64]) {
  sc lo, hi, a, b; sc_from_bytes(&lo, s); sc_from_bytes(&hi, s + 32);
  sc_montmul(&a, &lo, &SC_R2); sc_montmul(&b, &hi, &SC_R3); sc_add(&a, &a, &b); sc_from_mont(r, &a);
}
/* Scalar::invert = a^(l-2) */
static void sc_invert(sc *r, const sc *a) {
  sc base, acc; sc_to_mont(&base, a); acc = SC_R1;
  uint64_t e[4] = { SC_L[0] - 2, SC_L[1], SC_L[2], SC_L[3] };
  for (int i = 252; i >= 0; i--) { sc_montmul(&acc, &acc, &acc); if ((e[i >> 6] >> (i & 63)) & 1) sc_montmul(&acc, &acc, &base); }
  sc_from_mont(r, &acc);
}
/* Scalar::batch_invert: Montgomery's trick; returns the product of all inverses (dalek semantics) */
static void sc_batch_invert(sc *x, size_t n, sc *prod_of_inverses) {
  sc *pre = (sc *)malloc(sizeof(sc) * (n + 1)); sc acc; sc_from_u64(&acc, 1);
  for (size_t i = 0; i < n; i++) { pre[i] = acc; sc_mul(&acc, &acc, &x[i]); }
  sc inv; sc_invert(&inv, &acc); *prod_of_inverses = inv;
  for (size_t i = n; i-- > 0;) { sc t; sc_mul(&t, &inv, &pre[i]); sc_mul(&inv, &inv, &x[i]); x[i] = t; }
  free(pre);
}
static void sc_pow_vartime(sc *r, const sc *a, uint64_t e) {
  sc acc, base = *a; sc_from_u64(&acc, 1);
  while (e) { if (e & 1) sc_mul(&acc, &acc, &base); e >>= 1; if (e) sc_mul(&base, &base, &base); }
  *r = acc;
}

/* ------------------------------------------------------------------ points */
typedef struct { fe X, Y, Z, T; } ge_p3;          /* extended */
typedef struct { fe X, Y, Z; } ge_p2;             /* projective */
typedef struct { fe X, Y, Z, T; } ge_p1p1;        /* completed */
typedef struct { fe YplusX, YminusX, Z, T2d; } ge_cached;  /* projective niels */
typedef struct { fe yplusx, yminusx, xy2d; } ge_precomp;   /* affine niels */

static void ge_p3_0(ge_p3 *h) { fe_0(&h->X); fe_1(&h->Y); fe_1(&h->Z); fe_0(&h->T); }
static void ge_p1p1_to_p3(ge_p3 *r, const ge_p1p1 *p) { fe_mul(&r->X, &p->X, &p->T); fe_mul(&r->Y, &p->Y, &p->Z); fe_mul(&r->Z, &p->Z, &p->T); fe_mul(&r->T, &p->X, &p->Y); }
static void ge_p1p1_to_p2(ge_p2 *r, const ge_p1p1 *p) { fe_mul(&r->X, &p->X, &p->T); fe_mul(&r->Y, &p->Y, &p->Z); fe_mul(&r->Z, &p->Z, &p->T); }
static void ge_p3_to_cached(ge_cached *r, const ge_p3 *p) { fe_add(&r->YplusX, &p->Y, &p->X); fe_carry(&r->YplusX); fe_sub(&r->YminusX, &p->Y, &p->X); r->Z = p->Z; fe_mul(&r->T2d, &p->T, &FE_D2); }
static void ge_p2_dbl(ge_p1p1 *r, const fe *X, const fe *Y, const fe *Z) {
  fe xx, yy, zz2, xpy, xpy2;
  fe_sq(&xx, X); fe_sq(&yy, Y); fe_sq(&zz2, Z); fe_add(&zz2, &zz2, &zz2); fe_add(&xpy, X, Y); fe_sq(&xpy2, &xpy);
  fe_add(&r->Y, &yy, &xx); fe_carry(&r->Y); fe_sub(&r->Z, &yy, &xx); fe_sub(&r->X, &xpy2, &r->Y); fe_sub(&r->T, &zz2, &r->Z);
}
static void ge_add_cached(ge_p1p1 *r, const ge_p3 *p, const ge_cached *q, int sub) {
  fe a, b, pp, mm, tt, zz;
  fe_add(&a, &p->Y, &p->X); fe_sub(&b, &p->Y, &p->X);
  fe_mul(&pp, &a, sub ? &q->YminusX : &q->YplusX); fe_mul(&mm, &b, sub ? &q->YplusX : &q->YminusX);
  fe_mul(&tt, &p->T, &q->T2d); fe_mul(&zz, &p->Z, &q->Z); fe_add(&zz, &zz, &zz);
  fe_sub(&r->X, &pp, &mm); fe_add(&r->Y, &pp, &mm); fe_carry(&r->Y);
  if (!sub) { fe_add(&r->Z, &zz, &tt); fe_carry(&r->Z); fe_sub(&r->T, &zz, &tt); }
  else { fe_sub(&r->Z, &zz, &tt); fe_add(&r->T, &zz, &tt); fe_carry(&r->T); }
}
static void ge_add_precomp(ge_p1p1 *r, const ge_p3 *p, const ge_precomp *q, int sub) {
  fe a, b, pp, mm, tt, zz;
  fe_add(&a, &p->Y, &p->X); fe_sub(&b, &p->Y, &p->X);
  fe_mul(&pp, &a, sub ? &q->yminusx : &q->yplusx); fe_mul(&mm, &b, sub ? &q->yplusx : &q->yminusx);
  fe_mul(&tt, &p->T, &q->xy2d); fe_add(&zz, &p->Z, &p->Z);
  fe_sub(&r->X, &pp, &mm); fe_add(&r->Y, &pp, &mm); fe_carry(&r->Y);
  if (!sub) { fe_add(&r->Z, &zz, &tt); fe_carry(&r->Z); fe_sub(&r->T, &zz, &tt); }
  else { fe_sub(&r->Z, &zz, &tt); fe_add(&r->T, &zz, &tt); fe_carry(&r->T); }
}
static void ge_add(ge_p3 *r, const ge_p3 *p, const ge_p3 *q) { ge_cached c; ge_p1p1 t; ge_p3_to_cached(&c, q); ge_add_cached(&t, p, &c, 0); ge_p1p1_to_p3(r, &t); }
static void ge_sub(ge_p3 *r, const ge_p3 *p, const ge_p3 *q) { ge_cached c; ge_p1p1 t; ge_p3_to_cached(&c, q); ge_add_cached(&t, p, &c, 1); ge_p1p1_to_p3(r, &t); }
static void ge_dbl(ge_p3 *r, const ge_p3 *p) { ge_p1p1 t; ge_p2_dbl(&t, &p->X, &p->Y, &p->Z); ge_p1p1_to_p3(r, &t); }
static void ge_p3_to_precomp(ge_precomp *r, const ge_p3 *p) {
  fe zi, x, y, t; fe_invert(&zi, &p->Z); fe_mul(&x, &p->X, &zi); fe_mul(&y, &p->Y, &zi);
  fe_add(&r->yplusx, &y, &x); fe_carry(&r->yplusx); fe_sub(&r->yminusx, &y, &x); fe_mul(&t, &x, &y); fe_mul(&r->xy2d, &t, &FE_D2);
}
/* RistrettoPoint equality: X1*Y2 == Y1*X2 || X1*X2 == Y1*Y2 */
static int ristretto_eq(const ge_p3 *a, const ge_p3 *b) {
  fe l, r; fe_mul(&l, &a->X, &b->Y); fe_mul(&r, &a->Y, &b->X); if (fe_eq(&l, &r)) return 1;
  fe_mul(&l, &a->X, &b->X); fe_mul(&r, &a->Y, &b->Y); return fe_eq(&l, &r);
}
static int ristretto_is_identity(const ge_p3 *a) { ge_p3 id; ge_p3_0(&id); return ristretto_eq(a, &id); }

static int fe_sqrt_ratio_m1(fe *r_out, const fe *u, const fe *v) {
  fe v3, v7, r, check, t, nu, nui, rp;
  fe_sq(&v3, v); fe_mul(&v3, &v3, v); fe_sq(&v7, &v3); fe_mul(&v7, &v7, v); fe_mul(&t, u, &v7); fe_pow22523(&r, &t);
  fe_mul(&t, u, &v3); fe_mul(&r, &r, &t); fe_sq(&check, &r); fe_mul(&check, &check, v);
  fe_neg(&nu, u); fe_mul(&nui, &nu, &FE_SQRT_M1);
  int correct = fe_eq(&check, u), flipped = fe_eq(&check, &nu), flipped_i = fe_eq(&check, &nui);
  fe_mul(&rp, &r, &FE_SQRT_M1); if (flipped || flipped_i) r = rp;
  fe_abs(r_out, &r);
  return correct || flipped;
}
static int ristretto_decompress(ge_p3 *out, const uint8_t sb[32]) {
  fe s, ss, u1, u2, u2s, v, t, one, inv, dx, dy, x, y; uint8_t chk[32];
  fe_frombytes(&s, sb); fe_tobytes(chk, &s);
  if (memcmp(chk, sb, 32) != 0 || (sb[0] & 1)) return 0;
  fe_1(&one); fe_sq(&ss, &s); fe_sub(&u1, &one, &ss); fe_add(&u2, &one, &ss); fe_sq(&u2s, &u2);
  fe_sq(&t, &u1); fe_mul(&t, &t, &FE_D); fe_neg(&t, &t); fe_sub(&v, &t, &u2s); fe_mul(&t, &v, &u2s);
  int was_square = fe_sqrt_ratio_m1(&inv, &one, &t);
  fe_mul(&dx, &inv, &u2); fe_mul(&dy, &inv, &dx); fe_mul(&dy, &dy, &v);
  fe_mul(&x, &s, &dx); fe_add(&x, &x, &x); fe_abs(&x, &x); fe_mul(&y, &u1, &dy); fe_mul(&t, &x, &y);
  if (!was_square || fe_isnegative(&t) || fe_iszero(&y)) return 0;
  fe_carry(&x); out->X = x; out->Y = y; fe_1(&out->Z); out->T = t;
  return 1;
}
static void ristretto_compress(uint8_t out[32], const ge_p3 *p) {
  fe u1, u2, t, one, inv, den1, den2, zinv, ix, iy, ench, x, y, deninv, s, ny;
  fe_add(&u1, &p->Z, &p->Y); fe_sub(&t, &p->Z, &p->Y); fe_mul(&u1, &u1, &t); fe_mul(&u2, &p->X, &p->Y);
  fe_sq(&t, &u2); fe_mul(&t, &t, &u1); fe_1(&one); fe_sqrt_ratio_m1(&inv, &one, &t);
  fe_mul(&den1, &inv, &u1); fe_mul(&den2, &inv, &u2); fe_mul(&zinv, &den1, &den2); fe_mul(&zinv, &zinv, &p->T);
  fe_mul(&ix, &p->X, &FE_SQRT_M1); fe_mul(&iy, &p->Y, &FE_SQRT_M1); fe_mul(&ench, &den1, &FE_INVSQRT_A_MINUS_D);
  fe_mul(&t, &p->T, &zinv);
  if (fe_isnegative(&t)) { x = iy; y = ix; deninv = ench; } else { x = p->X; y = p->Y; deninv = den2; }
  fe_mul(&t, &x, &zinv); if (fe_isnegative(&t)) { fe_neg(&ny, &y); y = ny; }
  fe_sub(&t, &p->Z, &y); fe_mul(&s, &deninv, &t); fe_abs(&s, &s); fe_tobytes(out, &s);
}
static void ristretto_elligator(ge_p3 *out, const fe *t_in) {
  fe r, u, v, c, s, sp, n, w0, w1, w2, w3, one, t;
  fe_1(&one); fe_sq(&r, t_in); fe_mul(&r, &r, &FE_SQRT_M1); fe_add(&u, &r, &one); fe_mul(&u, &u, &FE_ONE_MINUS_D_SQ);
  fe_mul(&t, &r, &FE_D); fe_add(&t, &t, &one); fe_neg(&t, &t); fe_add(&v, &r, &FE_D); fe_mul(&v, &v, &t);
  int was_square = fe_sqrt_ratio_m1(&s, &u, &v);
  fe_mul(&sp, &s, t_in); fe_abs(&sp, &sp); fe_neg(&sp, &sp);
  fe_neg(&c, &one); if (!was_square) { s = sp; c = r; }
  fe_sub(&t, &r, &one); fe_mul(&n, &c, &t); fe_mul(&n, &n, &FE_D_MINUS_ONE_SQ); fe_sub(&n, &n, &v);
  fe_mul(&w0, &s, &v); fe_add(&w0, &w0, &w0); fe_mul(&w1, &n, &FE_SQRT_AD_MINUS_ONE); fe_sq(&t, &s);
  fe_sub(&w2, &one, &t); fe_add(&w3, &one, &t);
  fe_mul(&out->X, &w0, &w3); fe_mul(&out->Y, &w2, &w1); fe_mul(&out->Z, &w1, &w3); fe_mul(&out->T, &w0, &w2);
}
static void ristretto_from_uniform(ge_p3 *out, const uint8_t b[64]) {
  fe r0, r1; ge_p3 p0, p1; fe_frombytes(&r0, b); fe_frombytes(&r1, b + 32);
  ristretto_elligator(&p0, &r0); ristretto_elligator(&p1, &r1); ge_add(out, &p0, &p1);
}

/* ------------------------------------------------------------------ start-up constants */
static ge_p3 GE_BASEPOINT;
static void curve_init(void) {
  static int done = 0; if (done) return; done = 1;
  /* scalar constants */
  uint64_t inv = 1; for (int i = 0; i < 6; i++) inv *= 2 - SC_L[0] * inv;   /* Newton: l^-1 mod 2^64 */
  SC_LFACTOR = (uint64_t)0 - inv;
  /* R mod l by doubling 1, 256 times */
  sc r; sc_from_u64(&r, 1); for (int i = 0; i < 256; i++) sc_add(&r, &r, &r); SC_R1 = r;
  /* R^2 mod l: double R another 256 times (R * 2^256) */
  sc r2 = r; for (int i = 0; i < 256; i++) sc_add(&r2, &r2, &r2); SC_R2 = r2;
  sc_montmul(&SC_R3, &SC_R2, &SC_R2);  /* R^2*R^2/R = R^3 */
  /* field constants */
  fe a, b, one, t; fe_from_u64(&a, 121665); fe_from_u64(&b, 121666); fe_invert(&b, &b); fe_mul(&t, &a, &b); fe_neg(&FE_D, &t);
  fe_add(&FE_D2, &FE_D, &FE_D); fe_carry(&FE_D2);
  /* sqrt(-1) = 2^((p-1)/4): (p-1)/4 = 2^253 - 5 -> 2^(2^253-5) = (2^(2^252-3))^2 * 2  */
  fe two; fe_from_u64(&two, 2); fe_pow22523(&t, &two); fe_sq(&t, &t); fe_mul(&FE_SQRT_M1, &t, &two);
  fe_1(&one); fe_sq(&t, &FE_D); fe_sub(&FE_ONE_MINUS_D_SQ, &one, &t);
  fe_sub(&t, &FE_D, &one); fe_sq(&FE_D_MINUS_ONE_SQ, &t);
  /* sqrt(a*d - 1) = sqrt(-d-1) and 1/sqrt(a-d) = 1/sqrt(-1-d); signs fixed by RFC 9496 4.1 (first: odd value, second: even) */
  fe md1; fe_add(&t, &FE_D, &one); fe_neg(&md1, &t);
  fe_sqrt_ratio_m1(&FE_SQRT_AD_MINUS_ONE, &md1, &one);  /* returns the even root */
  /* RFC value 2506...0235 is odd -> negate */
  fe_neg(&FE_SQRT_AD_MINUS_ONE, &FE_SQRT_AD_MINUS_ONE);
  fe_sqrt_ratio_m1(&FE_INVSQRT_A_MINUS_D, &one, &md1);  /* RFC value 5446...7578 is even -> keep */
  /* basepoint: y = 4/5, x even */
  fe four, five, y, x, u, v; fe_from_u64(&four, 4); fe_from_u64(&five, 5); fe_invert(&five, &five); fe_mul(&y, &four, &five);
  fe_sq(&t, &y); fe_sub(&u, &t, &one); fe_mul(&v, &t, &FE_D); fe_add(&v, &v, &one); fe_carry(&v);
  fe_sqrt_ratio_m1(&x, &u, &v);
  GE_BASEPOINT.X = x; GE_BASEPOINT.Y = y; fe_1(&GE_BASEPOINT.Z); fe_mul(&GE_BASEPOINT.T, &x, &y);
}

/* ------------------------------------------------------------------ NAF + Straus + Pippenger */
/* width-w non-adjacent form of a canonical scalar, 256 digits (dalek Scalar::non_adjacent_form) */
static void sc_naf(int8_t naf[256], const sc *s, int w) {
  uint64_t x[5] = { s->v[0], s->v[1], s->v[2], s->v[3], 0 };
  memset(naf, 0, 256);
  const uint64_t width = 1ULL << w, window_mask = width - 1;
  int pos = 0; uint64_t carry = 0;
  while (pos < 256) {
    int idx = pos / 64, bit = pos % 64; uint64_t bit_buf;
    if (bit < 64 - w) bit_buf = x[idx] >> bit; else bit_buf = (x[idx] >> bit) | (x[idx + 1] << (64 - bit));
    uint64_t window = carry + (bit_buf & window_mask);
    if ((window & 1) == 0) { pos += 1; continue; }
    if (window < width / 2) { carry = 0; naf[pos] = (int8_t)window; }
    else { carry = 1; naf[pos] = (int8_t)((int64_t)window - (int64_t)width); }
    pos += w;
  }
}
/* odd multiples table [P, 3P, 5P, ...] as projective niels (count entries) */
static void odd_multiples_cached(ge_cached *tbl, int count, const ge_p3 *p) {
  ge_p3 p2, cur = *p; ge_dbl(&p2, p); ge_cached c2; ge_p3_to_cached(&c2, &p2);
  ge_p3_to_cached(&tbl[0], &cur);
  for (int i = 1; i < count; i++) { ge_p1p1 t; ge_add_cached(&t, &cur, &c2, 0); ge_p1p1_to_p3(&cur, &t); ge_p3_to_cached(&tbl[i], &cur); }
}
/* VartimePrecomputedStraus: 64 odd multiples per static point in affine niels form */
typedef struct { ge_precomp (*tbl)[64]; size_t n; } static_tables;
static void static_tables_build(static_tables *st, const ge_p3 *pts, size_t n) {
  st->n = n; st->tbl = (ge_precomp (*)[64])malloc(sizeof(ge_precomp) * 64 * (n ? n : 1));
  for (size_t i = 0; i < n; i++) {
    ge_p3 p2, cur = pts[i]; ge_dbl(&p2, &pts[i]); ge_cached c2; ge_p3_to_cached(&c2, &p2);
    for (int k = 0; k < 64; k++) { ge_p3_to_precomp(&st->tbl[i][k], &cur); ge_p1p1 t; ge_add_cached(&t, &cur, &c2, 0); ge_p1p1_to_p3(&cur, &t); }
  }
}
/* vartime_mixed_multiscalar_mul: static scalars (width-8 NAF over st) + dynamic (width-5 NAF), shared doublings */
static void straus_mixed(ge_p3 *out, const static_tables *st, const sc *ss, size_t ns, const sc *ds, const ge_p3 *dp, size_t nd) {
  int8_t (*snaf)[256] = (int8_t (*)[256])malloc(256 * (ns ? ns : 1));
  int8_t (*dnaf)[256] = (int8_t (*)[256])malloc(256 * (nd ? nd : 1));
  ge_cached (*dtbl)[8] = (ge_cached (*)[8])malloc(sizeof(ge_cached) * 8 * (nd ? nd : 1));
  for (size_t i = 0; i < ns; i++) sc_naf(snaf[i], &ss[i], 8);
  for (size_t i = 0; i < nd; i++) { sc_naf(dnaf[i], &ds[i], 5); odd_multiples_cached(dtbl[i], 8, &dp[i]); }
  int top = 255;
  for (; top >= 0; top--) { int any = 0; for (size_t i = 0; i < ns && !any; i++) any |= snaf[i][top] != 0; for (size_t i = 0; i < nd && !any; i++) any |= dnaf[i][top] != 0; if (any) break; }
  ge_p3 acc; ge_p3_0(&acc);
  ge_p2 s; s.X = acc.X; s.Y = acc.Y; s.Z = acc.Z;
  for (int j = top; j >= 0; j--) {
    ge_p1p1 t; ge_p2_dbl(&t, &s.X, &s.Y, &s.Z);
    for (size_t i = 0; i < nd; i++) { int8_t d = dnaf[i][j]; if (d) { ge_p1p1_to_p3(&acc, &t); ge_add_cached(&t, &acc, &dtbl[i][(d > 0 ? d : -d) >> 1], d < 0); } }
    for (size_t i = 0; i < ns; i++) { int8_t d = snaf[i][j]; if (d) { ge_p1p1_to_p3(&acc, &t); ge_add_precomp(&t, &acc, &st->tbl[i][(d > 0 ? d : -d) >> 1], d < 0); } }
    ge_p1p1_to_p2(&s, &t);
    if (j == 0) ge_p1p1_to_p3(&acc, &t);
  }
  if (top < 0) ge_p3_0(&acc);
  *out = acc; free(snaf); free(dnaf); free(dtbl);
}
/* Pippenger (dalek: size >= 190; w = 6 (<500), 7 (<800), else 8; signed radix 2^w) */
static void pippenger(ge_p3 *out, const sc *s, const ge_p3 *p, size_t n) {
  int w = n < 500 ? 6 : (n < 800 ? 7 : 8);
  int digits_count = (256 + w - 1) / w + 1; size_t buckets_count = (size_t)1 << (w - 1);
  int16_t *dig = (int16_t *)malloc(sizeof(int16_t) * n * digits_count);
  for (size_t i = 0; i < n; i++) {
    uint64_t x[5] = { s[i].v[0], s[i].v[1], s[i].v[2], s[i].v[3], 0 }; int64_t carry = 0;
    for (int k = 0; k < digits_count; k++) {
      int bit = k * w, idx = bit / 64, off = bit % 64; uint64_t raw;
      if (idx >= 4) raw = 0; else if (off + w <= 64) raw = x[idx] >> off; else raw = (x[idx] >> off) | (x[idx + 1] << (64 - off));
      int64_t coef = carry + (int64_t)(raw & ((1ULL << w) - 1));
      carry = (coef + (1LL << (w - 1))) >> w; dig[i * digits_count + k] = (int16_t)(coef - (carry << w));
    }
  }
  ge_cached *pc = (ge_cached *)malloc(sizeof(ge_cached) * (n ? n : 1));
  for (size_t i = 0; i < n; i++) ge_p3_to_cached(&pc[i], &p[i]);
  ge_p3 *buckets = (ge_p3 *)malloc(sizeof(ge_p3) * buckets_count);
  ge_p3 total; ge_p3_0(&total);
  for (int k = digits_count - 1; k >= 0; k--) {
    for (size_t b = 0; b < buckets_count; b++) ge_p3_0(&buckets[b]);
    for (size_t i = 0; i < n; i++) {
      int d = dig[i * digits_count + k]; ge_p1p1 t;
      if (d > 0) { ge_add_cached(&t, &buckets[d - 1], &pc[i], 0); ge_p1p1_to_p3(&buckets[d - 1], &t); }
      else if (d < 0) { ge_add_cached(&t, &buckets[-d - 1], &pc[i], 1); ge_p1p1_to_p3(&buckets[-d - 1], &t); }
    }
    ge_p3 isum = buckets[buckets_count - 1], sum = buckets[buckets_count - 1];
    for (size_t b = buckets_count - 1; b-- > 0;) { ge_add(&isum, &isum, &buckets[b]); ge_add(&sum, &sum, &isum); }
    for (int i = 0; i < w; i++) ge_dbl(&total, &total);
    ge_add(&total, &total, &sum);
  }
  *out = total; free(dig); free(pc); free(buckets);
}
/* VartimeMultiscalarMul::vartime_multiscalar_mul: Straus below 190 terms, Pippenger above */
static void vartime_multiscalar_mul(ge_p3 *out, const sc *s, const ge_p3 *p, size_t n) {
  if (n < 190) { static_tables none = { NULL, 0 }; straus_mixed(out, &none, NULL, 0, s, p, n); }
  else pippenger(out, s, p, n);
}
/* &P * Scalar (constant-time in dalek; plain here) */
static void ge_scalarmult(ge_p3 *out, const ge_p3 *p, const sc *s) { vartime_multiscalar_mul(out, s, p, 1); }

#endif
