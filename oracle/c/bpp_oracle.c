/* ORACLE (test infrastructure / CPU baseline only -- never linked into or called by the product path).
 *
 * Plain-C port of the reference's hot path using the reference's ALGORITHMS, so that it can stand in as the
 * "reference-equivalent CPU restatement" timed by bench.py's cpu_baseline leg (kind = "port"):
 *   RangeProof::verify          src/range_proof.rs:756-1065   (two passes, batch inversion, precomputed Straus MSM)
 *   RangeProof::prove_with_rng  src/range_proof.rs:232-608    (generator folding with 2-term MSMs, like the reference)
 *   RangeProofTranscript        src/transcripts.rs:59-200
 *   nonce / padding             src/utils/generic.rs:30-82
 *   BulletproofGens / Pedersen  src/generators/*.rs, src/ristretto.rs:67-112
 * Single-threaded, like the reference.  Checked against oracle/pyref on shared inputs (tests/test_oracle_c.py).
 * Parity with the Rust crate itself: unpinned (see oracle/README.md). */
#include <stdio.h>
#include <time.h>

#include "curve25519.h"
#include "hashes.h"

#define ERR_VERIFICATION_FAILED 1
#define ERR_INVALID_ARGUMENT 2
#define ERR_INVALID_LENGTH 3
#define ERR_INVALID_BLAKE2B 4
#define ERR_SIZE_OVERFLOW 5

typedef struct {
  uint32_t n, m_max, t;
  ge_p3 *gi, *hi;            /* party-major, n*m_max each */
  ge_p3 g_base[6], h_base;
  uint8_t g_comp[6][32], h_comp[32];
  static_tables precomp;     /* interleaved G0,H0,G1,H1,...  (src/generators/bulletproof_gens.rs:99-103) */
} oracle_params;

typedef struct {
  const uint8_t *proof; size_t proof_len;
  const uint8_t *commitments32; uint32_t m;
  const uint64_t *min_values; const uint8_t *min_present;
  const uint8_t *seed_nonce32;
  const uint8_t *transcript_label; size_t label_len;
} oracle_item;

/* ------------------------------------------------------------------ parameters */
oracle_params *oracle_params_new(uint32_t n, uint32_t m_max, uint32_t t) {
  curve_init();
  if (!n || (n & (n - 1)) || n > 64 || !m_max || (m_max & (m_max - 1)) || t < 1 || t > 6 || (size_t)n * m_max > 4096) return NULL;
  oracle_params *P = (oracle_params *)calloc(1, sizeof(*P));
  P->n = n; P->m_max = m_max; P->t = t;
  size_t nm = (size_t)n * m_max;
  P->gi = (ge_p3 *)malloc(sizeof(ge_p3) * nm); P->hi = (ge_p3 *)malloc(sizeof(ge_p3) * nm);
  uint8_t *stream = (uint8_t *)malloc((size_t)64 * n);
  for (uint32_t party = 0; party < m_max; party++) {           /* bulletproof_gens.rs:88-97 */
    for (int which = 0; which < 2; which++) {
      uint8_t seed[20]; memcpy(seed, "GeneratorsChain", 15); seed[15] = which ? 'H' : 'G'; le32(seed + 16, party);
      shake256(seed, 20, stream, (size_t)64 * n);               /* generators_chain.rs:23-49 */
      for (uint32_t i = 0; i < n; i++) ristretto_from_uniform(which ? &P->hi[party * n + i] : &P->gi[party * n + i], stream + 64 * i);
    }
  }
  free(stream);
  for (uint32_t k = 0; k < t; k++) {                            /* ristretto.rs:88-95 */
    char label[64]; int ll = snprintf(label, sizeof(label), "RISTRETTO_MASKING_BASEPOINT_%u", k + 1);
    uint8_t h[64]; sha3_512((const uint8_t *)label, (size_t)ll, h); ristretto_from_uniform(&P->g_base[k], h);
    ristretto_compress(P->g_comp[k], &P->g_base[k]);
  }
  P->h_base = GE_BASEPOINT; ristretto_compress(P->h_comp, &P->h_base);
  ge_p3 *inter = (ge_p3 *)malloc(sizeof(ge_p3) * 2 * nm);
  for (size_t i = 0; i < nm; i++) { inter[2 * i] = P->gi[i]; inter[2 * i + 1] = P->hi[i]; }
  static_tables_build(&P->precomp, inter, 2 * nm);
  free(inter);
  return P;
}
void oracle_params_free(oracle_params *P) { if (!P) return; free(P->gi); free(P->hi); free(P->precomp.tbl); free(P); }
void oracle_params_export(const oracle_params *P, uint8_t *gi32, uint8_t *hi32, uint8_t *h32, uint8_t *g32) {
  size_t nm = (size_t)P->n * P->m_max;
  for (size_t i = 0; i < nm; i++) { if (gi32) ristretto_compress(gi32 + 32 * i, &P->gi[i]); if (hi32) ristretto_compress(hi32 + 32 * i, &P->hi[i]); }
  if (h32) memcpy(h32, P->h_comp, 32);
  if (g32) for (uint32_t k = 0; k < P->t; k++) memcpy(g32 + 32 * k, P->g_comp[k], 32);
}

/* PedersenGens::commit (pedersen_gens.rs:112-122) */
static int commit_point(const oracle_params *P, ge_p3 *out, uint64_t v, const sc *blind, uint32_t nb) {
  if (nb == 0 || nb > P->t) return ERR_INVALID_LENGTH;
  sc s[7]; ge_p3 p[7]; sc_from_u64(&s[0], v); p[0] = P->h_base;
  for (uint32_t k = 0; k < nb; k++) { s[k + 1] = blind[k]; p[k + 1] = P->g_base[k]; }
  vartime_multiscalar_mul(out, s, p, nb + 1);
  return 0;
}
int oracle_commit(const oracle_params *P, uint64_t v, const uint8_t *blind32, uint32_t nb, uint8_t out32[32]) {
  sc b[6]; if (nb > 6) return ERR_INVALID_LENGTH;
  for (uint32_t k = 0; k < nb; k++) sc_from_bytes(&b[k], blind32 + 32 * k);
  ge_p3 c; int rc = commit_point(P, &c, v, b, nb); if (rc) return rc;
  ristretto_compress(out32, &c); return 0;
}

/* ------------------------------------------------------------------ utilities */
static int nonce(sc *out, const sc *seed, const char *label, int j, int k) {   /* utils/generic.rs:30-60 */
  size_t ll = strlen(label); if (ll > 16) return ERR_INVALID_LENGTH;
  uint8_t key[43]; size_t n = 0; key[n++] = 0; sc_to_bytes(key + n, seed); n += 32;
  if (j >= 0) { key[n++] = 'j'; le32(key + n, (uint32_t)j); n += 4; }
  if (k >= 0) { key[n++] = 'k'; le32(key + n, (uint32_t)k); n += 4; }
  uint8_t h[64]; blake2b_mac512_empty(h, key, n, (const uint8_t *)label, ll); sc_from_wide(out, h);
  return 0;
}
static void random_not_zero(sc *out, strobe_t *rng) {                          /* scalar_protocol.rs:23-30 */
  do { uint8_t w[64]; rng_fill(rng, w, 64); sc_from_wide(out, w); } while (sc_iszero(out));
}
static int is_zero32(const uint8_t *p) { uint8_t r = 0; for (int i = 0; i < 32; i++) r |= p[i]; return r == 0; }
static int validate_and_append(transcript_t *t, const char *label, const uint8_t p[32]) { /* transcript_protocol.rs:48-61 */
  if (is_zero32(p)) return ERR_VERIFICATION_FAILED; transcript_append(t, label, p, 32); return 0;
}
static int challenge_scalar(transcript_t *t, const char *label, sc *out) {      /* transcript_protocol.rs:67-78 */
  uint8_t b[64]; transcript_challenge(t, label, b, 64); sc_from_wide(out, b); return sc_iszero(out) ? ERR_VERIFICATION_FAILED : 0;
}

/* RangeProofTranscript (transcripts.rs) */
typedef struct { transcript_t *tr; const uint8_t *wit; size_t wit_len; const uint8_t *ext; size_t ext_len, ext_off; strobe_t rng; } rp_transcript;
static int rpt_build_rng(rp_transcript *r) {                                    /* :185-194 */
  r->rng = *r->tr;
  if (r->wit) rng_rekey(&r->rng, "witness", r->wit, r->wit_len);
  uint8_t rnd[32]; memset(rnd, 0, 32);
  if (r->ext) { if (r->ext_off + 32 > r->ext_len) return ERR_INVALID_LENGTH; memcpy(rnd, r->ext + r->ext_off, 32); r->ext_off += 32; }
  rng_finalize(&r->rng, rnd); return 0;
}
static int rpt_new(rp_transcript *r, transcript_t *tr, const oracle_params *P, uint32_t m, const uint8_t *commitments32,
                   const uint64_t *min_values, const uint8_t *min_present, const uint8_t *wit, size_t wit_len,
                   const uint8_t *ext, size_t ext_len) {                        /* :59-121 */
  int rc; memset(r, 0, sizeof(*r)); r->tr = tr; r->wit = wit; r->wit_len = wit_len; r->ext = ext; r->ext_len = ext_len;
  transcript_append(tr, "dom-sep", "Bulletproofs+ Range Proof", 25);
  if ((rc = validate_and_append(tr, "H", P->h_comp))) return rc;
  for (uint32_t k = 0; k < P->t; k++) if ((rc = validate_and_append(tr, "G", P->g_comp[k]))) return rc;
  transcript_append_u64(tr, "N", P->n); transcript_append_u64(tr, "T", P->t); transcript_append_u64(tr, "M", m);
  for (uint32_t j = 0; j < m; j++) transcript_append(tr, "Ci", commitments32 + 32 * j, 32);
  for (uint32_t j = 0; j < m; j++) transcript_append_u64(tr, "vi - minimum_value", (min_present && min_present[j]) ? min_values[j] : 0);
  return rpt_build_rng(r);
}
static int rpt_challenges_y_z(rp_transcript *r, const uint8_t a[32], sc *y, sc *z) { int rc;
  if ((rc = validate_and_append(r->tr, "A", a))) return rc; if ((rc = rpt_build_rng(r))) return rc;
  if ((rc = challenge_scalar(r->tr, "y", y))) return rc; return challenge_scalar(r->tr, "z", z); }
static int rpt_challenge_round_e(rp_transcript *r, const uint8_t l[32], const uint8_t rr[32], sc *e) { int rc;
  if ((rc = validate_and_append(r->tr, "L", l))) return rc; if ((rc = validate_and_append(r->tr, "R", rr))) return rc;
  if ((rc = rpt_build_rng(r))) return rc; return challenge_scalar(r->tr, "e", e); }
static int rpt_challenge_final_e(rp_transcript *r, const uint8_t a1[32], const uint8_t b[32], sc *e) { int rc;
  if ((rc = validate_and_append(r->tr, "A1", a1))) return rc; if ((rc = validate_and_append(r->tr, "B", b))) return rc;
  if ((rc = rpt_build_rng(r))) return rc; return challenge_scalar(r->tr, "e", e); }

/* ------------------------------------------------------------------ prover (src/range_proof.rs:232-608) */
int oracle_prove(const oracle_params *P, const uint8_t *label, size_t label_len, uint32_t m, const uint64_t *values,
                 const uint8_t *blindings32 /* m x t x 32 */, const uint64_t *min_values, const uint8_t *min_present,
                 const uint8_t *seed_nonce32, const uint8_t *ext_rng, size_t ext_rng_len, uint8_t *proof_out,
                 size_t *proof_len, uint8_t *commitments_out /* m x 32, may be NULL */) {
  const uint32_t n = P->n, t = P->t; const size_t mn = (size_t)n * m; int rc;
  if (!m || (m & (m - 1)) || m > P->m_max) return ERR_INVALID_ARGUMENT;
  if (seed_nonce32 && m > 1) return ERR_INVALID_ARGUMENT;
  for (uint32_t j = 0; j < m; j++) if (n < 64 && (values[j] >> n)) return ERR_INVALID_LENGTH;      /* :264-271 */
  sc *blind = (sc *)malloc(sizeof(sc) * m * t); uint8_t *comm = (uint8_t *)malloc(32 * (size_t)m);
  for (uint32_t i = 0; i < m * t; i++) sc_from_bytes(&blind[i], blindings32 + 32 * i);
  for (uint32_t j = 0; j < m; j++) { ge_p3 c; commit_point(P, &c, values[j], &blind[j * t], t); ristretto_compress(comm + 32 * j, &c); }  /* :275-284 */
  if (commitments_out) memcpy(commitments_out, comm, 32 * (size_t)m);
  /* witness bytes (transcripts.rs:91-109) */
  size_t wl = (size_t)m * (8 + 32 * t); uint8_t *wit = (uint8_t *)malloc(wl); size_t wo = 0;
  for (uint32_t j = 0; j < m; j++) { for (int k = 0; k < 8; k++) wit[wo++] = (uint8_t)(values[j] >> (8 * k)); memcpy(wit + wo, blindings32 + 32 * (size_t)j * t, 32 * (size_t)t); wo += 32 * t; }
  transcript_t tr; transcript_new(&tr, label, label_len);
  rp_transcript rpt;
  if ((rc = rpt_new(&rpt, &tr, P, m, comm, min_values, min_present, wit, wl, ext_rng, ext_rng_len))) goto fail0;
  sc seed; if (seed_nonce32) sc_from_bytes(&seed, seed_nonce32);
  sc one, zero; sc_from_u64(&one, 1); sc_from_u64(&zero, 0);
  sc *a_li = (sc *)malloc(sizeof(sc) * mn), *a_ri = (sc *)malloc(sizeof(sc) * mn);
  for (uint32_t j = 0; j < m; j++) {                                                                 /* :300-322 */
    uint64_t off = values[j];
    if (min_present && min_present[j]) { if (values[j] < min_values[j]) { rc = ERR_INVALID_ARGUMENT; goto fail1; } off = values[j] - min_values[j]; }
    for (uint32_t i = 0; i < n; i++) { uint64_t bit = (off >> i) & 1; sc_from_u64(&a_li[j * n + i], bit); sc_sub(&a_ri[j * n + i], &a_li[j * n + i], &one); }
  }
  sc alpha[6];                                                                                       /* :325-333 */
  for (uint32_t k = 0; k < t; k++) { if (seed_nonce32) nonce(&alpha[k], &seed, "alpha", -1, (int)k); else random_not_zero(&alpha[k], &rpt.rng); }
  ge_p3 A;                                                                                           /* :339-345 */
  { sc *ss = (sc *)malloc(sizeof(sc) * 2 * mn); for (size_t i = 0; i < mn; i++) { ss[2 * i] = a_li[i]; ss[2 * i + 1] = a_ri[i]; }
    straus_mixed(&A, &P->precomp, ss, 2 * mn, alpha, P->g_base, t); free(ss); }
  uint8_t a_comp[32]; ristretto_compress(a_comp, &A);
  sc y, z; if ((rc = rpt_challenges_y_z(&rpt, a_comp, &y, &z))) goto fail1;
  sc z_square; sc_mul(&z_square, &z, &z);
  sc *y_powers = (sc *)malloc(sizeof(sc) * (mn + 2)); y_powers[0] = one; for (size_t i = 1; i < mn + 2; i++) sc_mul(&y_powers[i], &y_powers[i - 1], &y);
  sc *d = (sc *)malloc(sizeof(sc) * mn); d[0] = z_square;                                            /* :362-373 */
  for (uint32_t i = 1; i < n; i++) sc_add(&d[i], &d[i - 1], &d[i - 1]);
  for (uint32_t j = 1; j < m; j++) for (uint32_t i = 0; i < n; i++) sc_mul(&d[j * n + i], &d[(j - 1) * n + i], &z_square);
  for (size_t i = 0; i < mn; i++) { sc tt; sc_sub(&a_li[i], &a_li[i], &z); sc_mul(&tt, &d[i], &y_powers[mn - i]); sc_add(&tt, &tt, &z); sc_add(&a_ri[i], &a_ri[i], &tt); }
  { sc zp = one; for (uint32_t j = 0; j < m; j++) { sc_mul(&zp, &zp, &z_square); for (uint32_t k = 0; k < t; k++) { sc tt; sc_mul(&tt, &zp, &blind[j * t + k]); sc_mul(&tt, &tt, &y_powers[mn + 1]); sc_add(&alpha[k], &alpha[k], &tt); } } }
  ge_p3 *gi = (ge_p3 *)malloc(sizeof(ge_p3) * mn), *hi = (ge_p3 *)malloc(sizeof(ge_p3) * mn);      /* :395-396 */
  memcpy(gi, P->gi, sizeof(ge_p3) * mn); memcpy(hi, P->hi, sizeof(ge_p3) * mn);
  uint32_t rounds = 0; while (((size_t)1 << rounds) < mn) rounds++;
  uint8_t (*li)[32] = (uint8_t (*)[32])malloc(32 * (rounds + 1)), (*ri)[32] = (uint8_t (*)[32])malloc(32 * (rounds + 1));
  size_t nn = mn; uint32_t round = 0;
  sc *ms = (sc *)malloc(sizeof(sc) * (mn + 8)); ge_p3 *mp = (ge_p3 *)malloc(sizeof(ge_p3) * (mn + 8));
  while (nn > 1) {                                                                                   /* :409-538 */
    nn /= 2;
    sc y_n_inverse; sc_invert(&y_n_inverse, &y_powers[nn]);
    sc *a_lo = a_li, *a_hi = a_li + nn, *b_lo = a_ri, *b_hi = a_ri + nn;
    sc *a_lo_off = (sc *)malloc(sizeof(sc) * nn), *a_hi_off = (sc *)malloc(sizeof(sc) * nn);
    for (size_t i = 0; i < nn; i++) { sc_mul(&a_lo_off[i], &a_lo[i], &y_n_inverse); sc_mul(&a_hi_off[i], &a_hi[i], &y_powers[nn]); }
    sc d_l[6], d_r[6];
    for (uint32_t k = 0; k < t; k++) { if (seed_nonce32) nonce(&d_l[k], &seed, "dL", (int)round, (int)k); else random_not_zero(&d_l[k], &rpt.rng); }
    for (uint32_t k = 0; k < t; k++) { if (seed_nonce32) nonce(&d_r[k], &seed, "dR", (int)round, (int)k); else random_not_zero(&d_r[k], &rpt.rng); }
    round++;
    sc c_l = zero, c_r = zero;
    for (size_t i = 0; i < nn; i++) { sc tt; sc_mul(&tt, &a_lo[i], &y_powers[i + 1]); sc_mul(&tt, &tt, &b_hi[i]); sc_add(&c_l, &c_l, &tt);
                                      sc_mul(&tt, &a_hi[i], &y_powers[nn + 1 + i]); sc_mul(&tt, &tt, &b_lo[i]); sc_add(&c_r, &c_r, &tt); }
    size_t q = 0; ge_p3 Lp, Rp;
    ms[q] = c_l; mp[q++] = P->h_base; for (uint32_t k = 0; k < t; k++) { ms[q] = d_l[k]; mp[q++] = P->g_base[k]; }
    for (size_t i = 0; i < nn; i++) { ms[q] = a_lo_off[i]; mp[q++] = gi[nn + i]; } for (size_t i = 0; i < nn; i++) { ms[q] = b_hi[i]; mp[q++] = hi[i]; }
    vartime_multiscalar_mul(&Lp, ms, mp, q);
    q = 0; ms[q] = c_r; mp[q++] = P->h_base; for (uint32_t k = 0; k < t; k++) { ms[q] = d_r[k]; mp[q++] = P->g_base[k]; }
    for (size_t i = 0; i < nn; i++) { ms[q] = a_hi_off[i]; mp[q++] = gi[i]; } for (size_t i = 0; i < nn; i++) { ms[q] = b_lo[i]; mp[q++] = hi[nn + i]; }
    vartime_multiscalar_mul(&Rp, ms, mp, q);
    ristretto_compress(li[round - 1], &Lp); ristretto_compress(ri[round - 1], &Rp);
    sc e; if ((rc = rpt_challenge_round_e(&rpt, li[round - 1], ri[round - 1], &e))) { free(a_lo_off); free(a_hi_off); goto fail2; }
    sc e_square, e_inverse, e_inverse_square, e_y_n_inverse;
    sc_mul(&e_square, &e, &e); sc_invert(&e_inverse, &e); sc_mul(&e_inverse_square, &e_inverse, &e_inverse); sc_mul(&e_y_n_inverse, &e, &y_n_inverse);
    for (size_t i = 0; i < nn; i++) {                                                                /* :512-521: 2-term MSMs */
      sc s2[2]; ge_p3 p2[2], o;
      s2[0] = e_inverse; s2[1] = e_y_n_inverse; p2[0] = gi[i]; p2[1] = gi[nn + i]; vartime_multiscalar_mul(&o, s2, p2, 2); gi[i] = o;
      s2[0] = e; s2[1] = e_inverse; p2[0] = hi[i]; p2[1] = hi[nn + i]; vartime_multiscalar_mul(&o, s2, p2, 2); hi[i] = o;
    }
    for (size_t i = 0; i < nn; i++) { sc t1, t2; sc_mul(&t1, &a_lo[i], &e); sc_mul(&t2, &a_hi_off[i], &e_inverse); sc_add(&a_li[i], &t1, &t2);
                                      sc_mul(&t1, &b_lo[i], &e_inverse); sc_mul(&t2, &b_hi[i], &e); sc_add(&a_ri[i], &t1, &t2); }
    for (uint32_t k = 0; k < t; k++) { sc t1, t2; sc_mul(&t1, &d_l[k], &e_square); sc_mul(&t2, &d_r[k], &e_inverse_square); sc_add(&t1, &t1, &t2); sc_add(&alpha[k], &alpha[k], &t1); }
    free(a_lo_off); free(a_hi_off);
  }
  {                                                                                                  /* :542-607 */
    sc r, s, dd[6], eta[6]; random_not_zero(&r, &rpt.rng); random_not_zero(&s, &rpt.rng);
    for (uint32_t k = 0; k < t; k++) { if (seed_nonce32) nonce(&dd[k], &seed, "d", -1, (int)k); else random_not_zero(&dd[k], &rpt.rng); }
    for (uint32_t k = 0; k < t; k++) { if (seed_nonce32) nonce(&eta[k], &seed, "eta", -1, (int)k); else random_not_zero(&eta[k], &rpt.rng); }
    sc hs, t1, t2; sc_mul(&t1, &r, &y_powers[1]); sc_mul(&t1, &t1, &a_ri[0]); sc_mul(&t2, &s, &y_powers[1]); sc_mul(&t2, &t2, &a_li[0]); sc_add(&hs, &t1, &t2);
    size_t q = 0; ms[q] = r; mp[q++] = gi[0]; ms[q] = s; mp[q++] = hi[0]; ms[q] = hs; mp[q++] = P->h_base;
    for (uint32_t k = 0; k < t; k++) { ms[q] = dd[k]; mp[q++] = P->g_base[k]; }
    ge_p3 A1, Bp; vartime_multiscalar_mul(&A1, ms, mp, q);
    q = 0; sc_mul(&t1, &r, &y_powers[1]); sc_mul(&t1, &t1, &s); ms[q] = t1; mp[q++] = P->h_base;
    for (uint32_t k = 0; k < t; k++) { ms[q] = eta[k]; mp[q++] = P->g_base[k]; }
    vartime_multiscalar_mul(&Bp, ms, mp, q);
    uint8_t a1c[32], bc[32]; ristretto_compress(a1c, &A1); ristretto_compress(bc, &Bp);
    sc e; if ((rc = rpt_challenge_final_e(&rpt, a1c, bc, &e))) goto fail2;
    sc e_square, r1, s1; sc_mul(&e_square, &e, &e);
    sc_mul(&t1, &a_li[0], &e); sc_add(&r1, &r, &t1); sc_mul(&t1, &a_ri[0], &e); sc_add(&s1, &s, &t1);
    uint8_t *o = proof_out; *o++ = (uint8_t)t;                                                       /* to_bytes :1120-1150 */
    for (uint32_t k = 0; k < t; k++) { sc d1; sc_mul(&t1, &dd[k], &e); sc_mul(&t2, &alpha[k], &e_square); sc_add(&d1, &eta[k], &t1); sc_add(&d1, &d1, &t2); sc_to_bytes(o, &d1); o += 32; }
    memcpy(o, a_comp, 32); o += 32; memcpy(o, a1c, 32); o += 32; memcpy(o, bc, 32); o += 32; sc_to_bytes(o, &r1); o += 32; sc_to_bytes(o, &s1); o += 32;
    for (uint32_t j = 0; j < rounds; j++) { memcpy(o, li[j], 32); o += 32; memcpy(o, ri[j], 32); o += 32; }
    *proof_len = (size_t)(o - proof_out); rc = 0;
  }
fail2:
  free(ms); free(mp); free(li); free(ri); free(gi); free(hi); free(y_powers); free(d);
fail1:
  free(a_li); free(a_ri);
fail0:
  free(blind); free(comm); free(wit);
  return rc;
}

/* ------------------------------------------------------------------ verifier (src/range_proof.rs:756-1065) */
typedef struct {            /* optional outputs for differential tests; any pointer may be NULL */
  uint8_t *challenges;      /* n x (rmax+3) x 32 : y, z, e_j.., e  (zero padded) */
  uint32_t rmax;
  uint8_t *rng_out;         /* n x 32 */
  uint8_t *weights;         /* n x 32 */
  uint8_t *static_scalars;  /* (2*max_mn + t + 1) x 32 */
  uint8_t *dynamic_scalars; /* total_dyn x 32 */
  uint8_t *msm_result;      /* 32 */
} oracle_trace;

int oracle_verify(const oracle_params *P, const oracle_item *items, size_t n_items, int action, uint8_t *masks_out,
                  uint8_t *mask_present, oracle_trace *trace) {
  const uint32_t n = P->n, t = P->t; int rc = 0;
  if (!items || n_items == 0) return ERR_INVALID_ARGUMENT;
  /* from_bytes + consistency (:610-709) */
  uint32_t *rounds = (uint32_t *)malloc(4 * n_items); size_t max_mn = 0, total_dyn = (size_t)t + 1;
  for (size_t p = 0; p < n_items; p++) {
    const oracle_item *it = &items[p];
    if (it->proof_len < 1) { rc = ERR_INVALID_LENGTH; goto out0; }
    uint32_t pt = it->proof[0]; if (pt < 1 || pt > 6) { rc = ERR_INVALID_ARGUMENT; goto out0; }
    size_t body = it->proof_len - 1; if (body % 32) { rc = ERR_INVALID_LENGTH; goto out0; }
    size_t nch = body / 32; if (nch < pt + 5 + 2 || ((nch - pt - 5) & 1)) { rc = ERR_INVALID_LENGTH; goto out0; }
    for (uint32_t k = 0; k < pt; k++) if (!sc_is_canonical(it->proof + 1 + 32 * k)) { rc = ERR_INVALID_ARGUMENT; goto out0; }
    if (!sc_is_canonical(it->proof + 1 + 32 * (pt + 3)) || !sc_is_canonical(it->proof + 1 + 32 * (pt + 4))) { rc = ERR_INVALID_ARGUMENT; goto out0; }
    if (pt != t) { rc = ERR_INVALID_ARGUMENT; goto out0; }
    rounds[p] = (uint32_t)((nch - pt - 5) / 2);
    if (!it->m || (it->m & (it->m - 1)) || it->m > P->m_max) { rc = ERR_INVALID_ARGUMENT; goto out0; }
    for (uint32_t j = 0; j < it->m; j++) if (it->min_present && it->min_present[j] && n < 64 && (it->min_values[j] >> n)) { rc = ERR_INVALID_LENGTH; goto out0; }
    if ((size_t)it->m * n > max_mn) max_mn = (size_t)it->m * n;
    total_dyn += it->m + 3 + 2 * rounds[p];
  }
  {
  sc two_n_minus_one, one, two; sc_from_u64(&one, 1); sc_from_u64(&two, 2);
  sc_pow_vartime(&two_n_minus_one, &two, n); sc_sub(&two_n_minus_one, &two_n_minus_one, &one);       /* :781-782 */
  sc g_base_scalars[6], h_base_scalar; memset(g_base_scalars, 0, sizeof(g_base_scalars)); memset(&h_base_scalar, 0, sizeof(sc));
  sc *gi_s = (sc *)calloc(max_mn, sizeof(sc)), *hi_s = (sc *)calloc(max_mn, sizeof(sc));
  sc *dyn_s = (sc *)malloc(sizeof(sc) * total_dyn); ge_p3 *dyn_p = (ge_p3 *)malloc(sizeof(ge_p3) * total_dyn); size_t nd = 0;
  /* PASS 1 (:811-853) */
  sc *chal = (sc *)malloc(sizeof(sc) * n_items * 16);   /* y z e_0..e_11 e : slot 15 = final e */
  transcript_t wt; transcript_new(&wt, "Bulletproofs+ verifier weights", 30);
  for (size_t p = 0; p < n_items; p++) {
    const oracle_item *it = &items[p]; const uint8_t *pr = it->proof, *pA = pr + 1 + 32 * t, *pLR = pA + 160;
    transcript_t tr; transcript_new(&tr, it->transcript_label, it->label_len);
    rp_transcript rpt; sc *c = chal + p * 16;
    if ((rc = rpt_new(&rpt, &tr, P, it->m, it->commitments32, it->min_values, it->min_present, NULL, 0, NULL, 0))) goto out1;
    if ((rc = rpt_challenges_y_z(&rpt, pA, &c[0], &c[1]))) goto out1;
    if (rounds[p] > 12) { rc = ERR_INVALID_LENGTH; goto out1; }
    for (uint32_t j = 0; j < rounds[p]; j++) if ((rc = rpt_challenge_round_e(&rpt, pLR + 64 * j, pLR + 64 * j + 32, &c[2 + j]))) goto out1;
    if ((rc = rpt_challenge_final_e(&rpt, pA + 32, pA + 64, &c[15]))) goto out1;
    transcript_append(&tr, "r1", pA + 96, 32); transcript_append(&tr, "s1", pA + 128, 32);          /* to_verifier_rng :166-179 */
    for (uint32_t k = 0; k < t; k++) transcript_append(&tr, "d1", pr + 1 + 32 * k, 32);
    rpt_build_rng(&rpt);
    uint8_t bytes[32]; rng_fill(&rpt.rng, bytes, 32); transcript_append(&wt, "proof", bytes, 32);   /* :845-849 */
    if (trace && trace->rng_out) memcpy(trace->rng_out + 32 * p, bytes, 32);
    if (trace && trace->challenges) {
      uint8_t *o = trace->challenges + (size_t)p * (trace->rmax + 3) * 32; memset(o, 0, (size_t)(trace->rmax + 3) * 32);
      sc_to_bytes(o, &c[0]); sc_to_bytes(o + 32, &c[1]); for (uint32_t j = 0; j < rounds[p]; j++) sc_to_bytes(o + 64 + 32 * j, &c[2 + j]);
      sc_to_bytes(o + 64 + 32 * rounds[p], &c[15]);
    }
  }
  strobe_t wrng = wt; { uint8_t z32[32]; memset(z32, 0, 32); rng_finalize(&wrng, z32); }             /* :853 */
  /* PASS 2 (:856-1033) */
  for (size_t p = 0; p < n_items; p++) {
    const oracle_item *it = &items[p]; const uint8_t *pr = it->proof, *pA = pr + 1 + 32 * t, *pLR = pA + 160;
    const uint32_t m = it->m, r = rounds[p]; const size_t mn = (size_t)m * n;
    ge_p3 a, a1, b, lpt[12], rpt_[12], cm[64];
    if (m > 64) { rc = ERR_SIZE_OVERFLOW; goto out1; }
    if (!ristretto_decompress(&a, pA) || !ristretto_decompress(&a1, pA + 32) || !ristretto_decompress(&b, pA + 64)) { rc = ERR_INVALID_ARGUMENT; goto out1; }
    for (uint32_t j = 0; j < r; j++) if (!ristretto_decompress(&lpt[j], pLR + 64 * j)) { rc = ERR_INVALID_ARGUMENT; goto out1; }
    for (uint32_t j = 0; j < r; j++) if (!ristretto_decompress(&rpt_[j], pLR + 64 * j + 32)) { rc = ERR_INVALID_ARGUMENT; goto out1; }
    if (r >= 32) { rc = ERR_SIZE_OVERFLOW; goto out1; }
    if (((size_t)1 << r) != mn) { rc = ERR_INVALID_LENGTH; goto out1; }
    for (uint32_t j = 0; j < m; j++) if (!ristretto_decompress(&cm[j], it->commitments32 + 32 * j)) { rc = ERR_INVALID_ARGUMENT; goto out1; }
    sc r1, s1, d1[6]; sc_from_bytes(&r1, pA + 96); sc_from_bytes(&s1, pA + 128); for (uint32_t k = 0; k < t; k++) sc_from_bytes(&d1[k], pr + 1 + 32 * k);
    sc *c = chal + p * 16; sc y = c[0], z = c[1], e = c[15];
    sc weight; random_not_zero(&weight, &wrng);                                                      /* :894 */
    if (trace && trace->weights) sc_to_bytes(trace->weights + 32 * p, &weight);
    sc inv[14], prod_inv; for (uint32_t j = 0; j < r; j++) inv[j] = c[2 + j]; inv[r] = y; sc_sub(&inv[r + 1], &y, &one);
    sc ym1 = inv[r + 1]; sc_batch_invert(inv, r + 2, &prod_inv);                                     /* :897-905 */
    sc challenges_inv_prod; sc_mul(&challenges_inv_prod, &prod_inv, &y); sc_mul(&challenges_inv_prod, &challenges_inv_prod, &ym1);
    sc y_1_inverse = inv[r + 1], y_inverse = inv[r];
    sc z_square, e_square, csq[12], csqi[12], y_nm, y_nm_1, y_sum, tt;
    sc_mul(&z_square, &z, &z); sc_mul(&e_square, &e, &e);
    for (uint32_t j = 0; j < r; j++) { sc_mul(&csq[j], &c[2 + j], &c[2 + j]); sc_mul(&csqi[j], &inv[j], &inv[j]); }
    sc_pow_vartime(&y_nm, &y, mn); sc_mul(&y_nm_1, &y_nm, &y);
    sc_sub(&tt, &y_nm, &one); sc_mul(&tt, &tt, &y); sc_mul(&y_sum, &tt, &y_1_inverse);              /* :916 */
    sc *d = (sc *)malloc(sizeof(sc) * mn); d[0] = z_square;                                          /* :919-929 */
    for (uint32_t i = 1; i < n; i++) sc_mul(&d[i], &two, &d[i - 1]);
    for (uint32_t j = 1; j < m; j++) for (uint32_t i = 0; i < n; i++) sc_mul(&d[j * n + i], &d[(j - 1) * n + i], &z_square);
    sc d_sum = z_square, d_tmp = z_square;                                                           /* :932-938 */
    for (uint32_t mm = m; mm > 1; mm >>= 1) { sc_mul(&tt, &d_sum, &d_tmp); sc_add(&d_sum, &d_sum, &tt); sc_mul(&d_tmp, &d_tmp, &d_tmp); }
    sc_mul(&d_sum, &d_sum, &two_n_minus_one);
    if (mask_present) mask_present[p] = 0;
    if (action != 0) {                                                                               /* :941-969 */
      if (it->seed_nonce32) {
        sc seed; sc_from_bytes(&seed, it->seed_nonce32); sc e2i, zyi; sc_invert(&e2i, &e_square); sc_mul(&tt, &z_square, &y_nm_1); sc_invert(&zyi, &tt);
        for (uint32_t k = 0; k < t; k++) {
          sc mk, nn1, nn2; nonce(&nn1, &seed, "eta", -1, (int)k); sc_sub(&mk, &d1[k], &nn1); nonce(&nn2, &seed, "d", -1, (int)k); sc_mul(&nn2, &nn2, &e); sc_sub(&mk, &mk, &nn2);
          sc_mul(&mk, &mk, &e2i); nonce(&nn1, &seed, "alpha", -1, (int)k); sc_sub(&mk, &mk, &nn1);
          for (uint32_t j = 0; j < r; j++) { nonce(&nn1, &seed, "dL", (int)j, (int)k); sc_mul(&nn1, &nn1, &csq[j]); sc_sub(&mk, &mk, &nn1);
                                             nonce(&nn2, &seed, "dR", (int)j, (int)k); sc_mul(&nn2, &nn2, &csqi[j]); sc_sub(&mk, &mk, &nn2); }
          sc_mul(&mk, &mk, &zyi); if (masks_out) sc_to_bytes(masks_out + ((size_t)p * t + k) * 32, &mk);
        }
        if (mask_present) mask_present[p] = 1;
      }
      if (action == 2) { free(d); continue; }
    }
    sc *s = (sc *)malloc(sizeof(sc) * mn); s[0] = challenges_inv_prod;                               /* :975-986 */
    for (size_t i = 1; i < mn; i++) { uint32_t log_i = 63 - (uint32_t)__builtin_clzll(i); sc_mul(&s[i], &s[i - ((size_t)1 << log_i)], &csq[r - log_i - 1]); }
    sc r1_e, s1_e, e_square_z, y_inv_i = one, y_nm_i = y_nm; sc_mul(&r1_e, &r1, &e); sc_mul(&s1_e, &s1, &e); sc_mul(&e_square_z, &e_square, &z);
    for (size_t i = 0; i < mn; i++) {                                                                /* :987-1003 */
      sc g, h, u; sc_mul(&g, &r1_e, &y_inv_i); sc_mul(&g, &g, &s[i]); sc_mul(&h, &s1_e, &s[mn - 1 - i]);
      sc_add(&g, &g, &e_square_z); sc_mul(&g, &weight, &g); sc_add(&gi_s[i], &gi_s[i], &g);
      sc_mul(&u, &d[i], &y_nm_i); sc_add(&u, &u, &z); sc_mul(&u, &e_square, &u); sc_sub(&h, &h, &u); sc_mul(&h, &weight, &h); sc_add(&hi_s[i], &hi_s[i], &h);
      sc_mul(&y_inv_i, &y_inv_i, &y_inverse); sc_mul(&y_nm_i, &y_nm_i, &y_inverse);
    }
    sc neg_e_square, zp = one; sc_neg(&neg_e_square, &e_square);                                     /* :1006-1015 */
    for (uint32_t j = 0; j < m; j++) {
      sc weighted; sc_mul(&zp, &zp, &z_square); sc_mul(&weighted, &neg_e_square, &zp); sc_mul(&weighted, &weighted, &y_nm_1); sc_mul(&weighted, &weight, &weighted);
      dyn_s[nd] = weighted; dyn_p[nd++] = cm[j];
      if (it->min_present && it->min_present[j]) { sc mv; sc_from_u64(&mv, it->min_values[j]); sc_mul(&mv, &weighted, &mv); sc_sub(&h_base_scalar, &h_base_scalar, &mv); }
    }
    { sc a0, b0, u0; sc_mul(&a0, &r1, &y); sc_mul(&a0, &a0, &s1); sc_mul(&b0, &y_nm_1, &z); sc_mul(&b0, &b0, &d_sum); sc_sub(&u0, &z_square, &z); sc_mul(&u0, &u0, &y_sum);
      sc_add(&b0, &b0, &u0); sc_mul(&b0, &e_square, &b0); sc_add(&a0, &a0, &b0); sc_mul(&a0, &weight, &a0); sc_add(&h_base_scalar, &h_base_scalar, &a0); }   /* :1017 */
    for (uint32_t k = 0; k < t; k++) { sc_mul(&tt, &weight, &d1[k]); sc_add(&g_base_scalars[k], &g_base_scalars[k], &tt); }
    sc ne, w_ne2; sc_neg(&ne, &e); sc_mul(&dyn_s[nd], &weight, &ne); dyn_p[nd++] = a1;              /* :1022-1032 */
    sc_neg(&dyn_s[nd], &weight); dyn_p[nd++] = b;
    sc_mul(&w_ne2, &weight, &neg_e_square); dyn_s[nd] = w_ne2; dyn_p[nd++] = a;
    for (uint32_t j = 0; j < r; j++) { sc_mul(&dyn_s[nd], &w_ne2, &csq[j]); dyn_p[nd++] = lpt[j]; }
    for (uint32_t j = 0; j < r; j++) { sc_mul(&dyn_s[nd], &w_ne2, &csqi[j]); dyn_p[nd++] = rpt_[j]; }
    free(s); free(d);
  }
  if (trace && trace->dynamic_scalars) for (size_t i = 0; i < nd; i++) sc_to_bytes(trace->dynamic_scalars + 32 * i, &dyn_s[i]);
  if (trace && trace->static_scalars) {
    uint8_t *o = trace->static_scalars; for (size_t i = 0; i < max_mn; i++) { sc_to_bytes(o, &gi_s[i]); o += 32; sc_to_bytes(o, &hi_s[i]); o += 32; }
    for (uint32_t k = 0; k < t; k++) { sc_to_bytes(o, &g_base_scalars[k]); o += 32; } sc_to_bytes(o, &h_base_scalar);
  }
  if (action != 2) {                                                                                 /* :1039-1062 */
    for (uint32_t k = 0; k < t; k++) { dyn_s[nd] = g_base_scalars[k]; dyn_p[nd++] = P->g_base[k]; }
    dyn_s[nd] = h_base_scalar; dyn_p[nd++] = P->h_base;
    sc *ss = (sc *)malloc(sizeof(sc) * 2 * (max_mn ? max_mn : 1)); for (size_t i = 0; i < max_mn; i++) { ss[2 * i] = gi_s[i]; ss[2 * i + 1] = hi_s[i]; }
    ge_p3 res; straus_mixed(&res, &P->precomp, ss, 2 * max_mn, dyn_s, dyn_p, nd); free(ss);
    if (trace && trace->msm_result) ristretto_compress(trace->msm_result, &res);
    if (!ristretto_is_identity(&res)) rc = ERR_VERIFICATION_FAILED;
  }
out1:
  free(gi_s); free(hi_s); free(dyn_s); free(dyn_p); free(chal);
  }
out0:
  free(rounds);
  return rc;
}

/* time `iters` VerifyOnly runs of the same batch, chunked by `chunk` proofs per reference batch (0 = one batch) */
int oracle_verify_timed(const oracle_params *P, const oracle_item *items, size_t n_items, size_t chunk, int iters, double *seconds) {
  struct timespec t0, t1; int rc = 0; if (chunk == 0 || chunk > n_items) chunk = n_items;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int it = 0; it < iters && !rc; it++) for (size_t off = 0; off < n_items && !rc; off += chunk)
    rc = oracle_verify(P, items + off, (n_items - off < chunk) ? n_items - off : chunk, 0, NULL, NULL, NULL);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  *seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
  return rc;
}

/* the same on `threads` host threads: thread k verifies the chunk-sized slice k (mod the number of slices) `iters` times;
   wall time over all threads.  The reference is single-threaded; this is SURVEY 8(d)'s "same code across all host cores,
   one 256-proof chunk per thread" leg of the CPU baseline. */
#include <pthread.h>
typedef struct { const oracle_params *P; const oracle_item *items; size_t n; int iters; int rc; } mt_job;
static void *mt_worker(void *arg) { mt_job *j = (mt_job *)arg; j->rc = 0;
  for (int it = 0; it < j->iters && !j->rc; it++) j->rc = oracle_verify(j->P, j->items, j->n, 0, NULL, NULL, NULL);
  return NULL; }
int oracle_verify_timed_mt(const oracle_params *P, const oracle_item *items, size_t n_items, size_t chunk, int iters, int threads, double *seconds) {
  if (threads < 1 || threads > 1024) return -1;
  if (chunk == 0 || chunk > n_items) chunk = n_items;
  const size_t slices = n_items / chunk; if (slices == 0) return -1;
  curve_init();
  pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * threads); mt_job *jobs = (mt_job *)malloc(sizeof(mt_job) * threads);
  struct timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int k = 0; k < threads; k++) { jobs[k] = (mt_job){P, items + (k % slices) * chunk, chunk, iters, 0}; pthread_create(&th[k], NULL, mt_worker, &jobs[k]); }
  int rc = 0; for (int k = 0; k < threads; k++) { pthread_join(th[k], NULL); if (jobs[k].rc) rc = jobs[k].rc; }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  *seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
  free(th); free(jobs); return rc;
}

/* bench.py's per-config cpu_baseline legs: `threads` host threads, thread k runs the chunk-sized slice k (mod the number of
   slices) `iters` times under `action` (0 VerifyOnly, 1 RecoverAndVerify, 2 RecoverOnly: src/range_proof.rs:941-969); wall time.
   Masks go to a per-thread scratch buffer and are dropped. */
typedef struct { const oracle_params *P; const oracle_item *items; size_t n; int iters, action, rc; } act_job;
static void *act_worker(void *arg) { act_job *j = (act_job *)arg; j->rc = 0;
  uint8_t *masks = (uint8_t *)malloc(32 * (size_t)j->P->t * j->n + 1), *present = (uint8_t *)malloc(j->n + 1);
  for (int it = 0; it < j->iters && !j->rc; it++) j->rc = oracle_verify(j->P, j->items, j->n, j->action, masks, present, NULL);
  free(masks); free(present); return NULL; }
int oracle_verify_action_timed_mt(const oracle_params *P, const oracle_item *items, size_t n_items, size_t chunk, int action, int iters,
                                  int threads, double *seconds) {
  if (threads < 1 || threads > 1024 || action < 0 || action > 2) return -1;
  if (chunk == 0 || chunk > n_items) chunk = n_items;
  const size_t slices = n_items / chunk; if (slices == 0) return -1;
  curve_init();
  pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * threads); act_job *jobs = (act_job *)malloc(sizeof(act_job) * threads);
  struct timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int k = 0; k < threads; k++) { jobs[k] = (act_job){P, items + (k % slices) * chunk, chunk, iters, action, 0}; pthread_create(&th[k], NULL, act_worker, &jobs[k]); }
  int rc = 0; for (int k = 0; k < threads; k++) { pthread_join(th[k], NULL); if (jobs[k].rc) rc = jobs[k].rc; }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  *seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
  free(th); free(jobs); return rc;
}

/* the prover timed the same way (benches/range_proof.rs:43-107): `n_items` witnesses as contiguous arrays (values n x m,
   blindings n x m x t x 32, minimum values / presence n x m, seed nonces n x 32 or NULL, external randomness n x ext_len), thread k
   proves items k, k + threads, ... `iters` times each; wall time.  proofs_out (n x proof_stride, may be NULL) receives the last
   pass's bytes so that the caller can hold them against the engine's. */
typedef struct { const oracle_params *P; const uint8_t *label; size_t label_len; uint32_t m; const uint64_t *values; const uint8_t *blind;
                 const uint64_t *minv; const uint8_t *minp; const uint8_t *seeds; const uint8_t *ext; size_t ext_len, n, first, step;
                 int iters, rc; uint8_t *out; size_t stride; } prove_job;
static void *prove_worker(void *arg) { prove_job *j = (prove_job *)arg; j->rc = 0; const uint32_t m = j->m, t = j->P->t;
  uint8_t buf[4096]; size_t len = 0;
  for (int it = 0; it < j->iters && !j->rc; it++) for (size_t i = j->first; i < j->n && !j->rc; i += j->step) {
    j->rc = oracle_prove(j->P, j->label, j->label_len, m, j->values + i * m, j->blind + i * (size_t)m * t * 32, j->minv + i * m,
                         j->minp ? j->minp + i * m : NULL, j->seeds ? j->seeds + 32 * i : NULL, j->ext + i * j->ext_len, j->ext_len, buf, &len, NULL);
    if (!j->rc && j->out && len <= j->stride) memcpy(j->out + i * j->stride, buf, len);
  }
  return NULL; }
int oracle_prove_timed_mt(const oracle_params *P, const uint8_t *label, size_t label_len, uint32_t m, const uint64_t *values,
                          const uint8_t *blindings32, const uint64_t *min_values, const uint8_t *min_present, const uint8_t *seed_nonces32,
                          const uint8_t *ext_rng, size_t ext_len, size_t n_items, int iters, int threads, uint8_t *proofs_out,
                          size_t proof_stride, double *seconds) {
  if (threads < 1 || threads > 1024 || n_items == 0) return -1;
  curve_init();
  pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * threads); prove_job *jobs = (prove_job *)malloc(sizeof(prove_job) * threads);
  struct timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int k = 0; k < threads; k++) {
    jobs[k] = (prove_job){P, label, label_len, m, values, blindings32, min_values, min_present, seed_nonces32, ext_rng, ext_len, n_items,
                          (size_t)k, (size_t)threads, iters, 0, proofs_out, proof_stride};
    pthread_create(&th[k], NULL, prove_worker, &jobs[k]);
  }
  int rc = 0; for (int k = 0; k < threads; k++) { pthread_join(th[k], NULL); if (jobs[k].rc) rc = jobs[k].rc; }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  *seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
  free(th); free(jobs); return rc;
}

/* ------------------------------------------------------------------ primitive probes for KAT tests */
void oracle_from_uniform(const uint8_t in[64], uint8_t out[32]) { curve_init(); ge_p3 p; ristretto_from_uniform(&p, in); ristretto_compress(out, &p); }
int oracle_decompress_compress(const uint8_t in[32], uint8_t out[32]) { curve_init(); ge_p3 p; if (!ristretto_decompress(&p, in)) return 0; ristretto_compress(out, &p); return 1; }
int oracle_msm(const uint8_t *scalars32, const uint8_t *points32, size_t n, uint8_t out[32]) {
  curve_init(); sc *s = (sc *)malloc(sizeof(sc) * (n ? n : 1)); ge_p3 *p = (ge_p3 *)malloc(sizeof(ge_p3) * (n ? n : 1));
  for (size_t i = 0; i < n; i++) { sc_from_bytes(&s[i], scalars32 + 32 * i); if (!ristretto_decompress(&p[i], points32 + 32 * i)) { free(s); free(p); return 0; } }
  ge_p3 r; vartime_multiscalar_mul(&r, s, p, n); ristretto_compress(out, &r); free(s); free(p); return 1;
}
void oracle_sc_wide(const uint8_t in[64], uint8_t out[32]) { curve_init(); sc r; sc_from_wide(&r, in); sc_to_bytes(out, &r); }
void oracle_sc_mul_inv(const uint8_t a[32], const uint8_t b[32], uint8_t mul[32], uint8_t inva[32]) { curve_init(); sc x, y, r; sc_from_bytes(&x, a); sc_from_bytes(&y, b); sc_mul(&r, &x, &y); sc_to_bytes(mul, &r); sc_invert(&r, &x); sc_to_bytes(inva, &r); }
void oracle_nonce(const uint8_t seed[32], const char *label, int j, int k, uint8_t out[32]) { curve_init(); sc s, r; sc_from_bytes(&s, seed); nonce(&r, &s, label, j, k); sc_to_bytes(out, &r); }
void oracle_merlin_kat(const uint8_t *label, size_t ll, const char *ml, const uint8_t *msg, size_t mlen, const char *cl, uint8_t *out, size_t n) {
  transcript_t t; transcript_new(&t, label, ll); transcript_append(&t, ml, msg, mlen); transcript_challenge(&t, cl, out, n); }
void oracle_shake256(const uint8_t *in, size_t n, uint8_t *out, size_t outlen) { shake256(in, n, out, outlen); }
void oracle_sha3_512(const uint8_t *in, size_t n, uint8_t out[64]) { sha3_512(in, n, out); }
uint64_t oracle_keccak_count(void) { return g_keccak_count; }
