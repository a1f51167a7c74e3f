// Host-compiled probes of the shared host/device arithmetic headers (used only by tests/, CPU suite).
#include <string.h>
#include "point.h"
#include "scalar.h"
#include "merlin.h"
#include "blake2b.h"
#include "chain_host.h"
#include "wkeccak.h"
#include "recode.h"
#include "upload_host.h"
// the uniform-access scalar multiplication (ct.h) with its table reads RECORDED: BPP_CT_TOUCH(entry) appends the entry index
#include <vector>
static thread_local std::vector<uint8_t> *g_ct_trace = nullptr;
#define BPP_CT_TOUCH(entry)                                      \
  do {                                                           \
    if (g_ct_trace) g_ct_trace->push_back((uint8_t)(entry));     \
  } while (0)
#include "ct.h"
using namespace bpp;
extern "C" {
// scalar * P through ct_scalarmul -> compressed result; trace_out (may be null) receives the sequence of table entries read
// (trace_cap bytes at most), *trace_len their number.  0 if P does not decode.
int ht_ct_scalarmul(const uint8_t point32[32], const uint8_t scalar32[32], uint8_t out32[32], uint8_t *trace_out, size_t trace_cap,
                    size_t *trace_len) {
  niels n; if (!ristretto_decompress(n, point32)) return 0;
  ge p; ge_from_niels(p, n);
  sc s; sc_load_words(s, scalar32);
  std::vector<uint8_t> tr;
  g_ct_trace = &tr;
  ge r; ct_scalarmul(r, p, s);
  g_ct_trace = nullptr;
  ristretto_compress(out32, r);
  if (trace_len) *trace_len = tr.size();
  if (trace_out) memcpy(trace_out, tr.data(), tr.size() < trace_cap ? tr.size() : trace_cap);
  return 1; }
// the fixed-base form (k_ct_fixed's per-lane steps, all 64 positions on one lane): lines built here as k_fb_build does (4-bit
// windows: j * 16^w * P, j = 1..8), then ct_fixed_scalarmul; the trace holds the line index of every read
int ht_ct_fixed_scalarmul(const uint8_t point32[32], const uint8_t scalar32[32], uint8_t out32[32], uint8_t *trace_out, size_t trace_cap,
                          size_t *trace_len) {
  niels n; if (!ristretto_decompress(n, point32)) return 0;
  std::vector<niels> lines((size_t)BPP_CT_DIGITS * BPP_CTF_ENTRIES);
  ge base; ge_from_niels(base, n);
  for (int w = 0; w < BPP_CT_DIGITS; w++) {
    ge acc = base;
    for (int j = 0; j < BPP_CTF_ENTRIES; j++) {
      fe zi, x, y; fe_invert(zi, acc.Z); fe_mul(x, acc.X, zi); fe_mul(y, acc.Y, zi);
      niels e; niels_from_affine(e, x, y); fe_carry(e.yminusx); fe_carry(e.yplusx);
      lines[(size_t)w * BPP_CTF_ENTRIES + j] = e;
      ge_add(acc, acc, base);
    }
    for (int k = 0; k < 4; k++) ge_dbl(base, base);
  }
  sc s; sc_load_words(s, scalar32);
  std::vector<uint8_t> tr;
  g_ct_trace = &tr;
  ge r; ct_fixed_scalarmul(r, lines.data(), s);
  g_ct_trace = nullptr;
  ristretto_compress(out32, r);
  if (trace_len) *trace_len = tr.size();
  if (trace_out) memcpy(trace_out, tr.data(), tr.size() < trace_cap ? tr.size() : trace_cap);
  return 1; }
// the digit-parallel form over the multiples 16^w P of a point known ahead of its scalar (k_ct_pow16 + k_ct_var's per-lane steps,
// all 64 positions on one lane): no table at all -- every position walks 1 .. 8 times its multiple and keeps one under a mask
int ht_ct_var_scalarmul(const uint8_t point32[32], const uint8_t scalar32[32], uint8_t out32[32]) {
  niels n; if (!ristretto_decompress(n, point32)) return 0;
  ge p; ge_from_niels(p, n);
  sc s; sc_load_words(s, scalar32);
  ge r; ct_var_scalarmul(r, p, s);
  ristretto_compress(out32, r);
  return 1; }
// the recoding alone: 64 signed radix-16 digits whose weighted sum is the scalar
void ht_ct_recode16(const uint8_t scalar32[32], int8_t digits[64]) { sc s; sc_load_words(s, scalar32); ct_recode16(digits, s); }
void ht_fe_mul(const uint8_t a[32], const uint8_t b[32], uint8_t out[32]) { fe x, y, z; fe_frombytes(x, a); fe_frombytes(y, b); fe_mul(z, x, y); fe_tobytes(out, z); }
void ht_fe_sq(const uint8_t a[32], uint8_t out[32]) { fe x, z; fe_frombytes(x, a); fe_sq(z, x); fe_tobytes(out, z); }
void ht_fe_addsubmul(const uint8_t a[32], const uint8_t b[32], uint8_t out[32]) {  // (a+b)*(a-b) with lazy limbs
  fe x, y, s, d, z; fe_frombytes(x, a); fe_frombytes(y, b); fe_add(s, x, y); fe_sub(d, x, y); fe_mul(z, s, d); fe_tobytes(out, z); }
void ht_fe_invert(const uint8_t a[32], uint8_t out[32]) { fe x, z; fe_frombytes(x, a); fe_invert(z, x); fe_tobytes(out, z); }
void ht_sc_mul(const uint8_t a[32], const uint8_t b[32], uint8_t out[32]) {
  sc x, y, z; sc_load_words(x, a); sc_load_words(y, b); sc_to_mont(x, x); sc_to_mont(y, y); sc_montmul(z, x, y); sc_from_mont(z, z); sc_store_words(out, z); }
// (a b c) through the unpacked forms: lazy product a*b kept as limbs, times c, packed; and a * 2^e through the constant table
void ht_sc9_mul3(const uint8_t a[32], const uint8_t b[32], const uint8_t c[32], uint32_t e, uint8_t out[32], uint8_t out2[32]) {
  sc x, y, z, r; sc_load_words(x, a); sc_load_words(y, b); sc_load_words(z, c); sc_to_mont(x, x); sc_to_mont(y, y); sc_to_mont(z, z);
  sc9 x9, y9, z9, t9; sc9_from(x9, x); sc9_from(y9, y); sc9_from(z9, z);
  sc9_montmul_lazy(t9, x9, y9); sc9_montmul(r, t9, z9); sc_from_mont(r, r); sc_store_words(out, r);
  sc9 p2; for (int i = 0; i < 9; i++) p2.l[i] = SC_POW2_R29[e & 63][i];
  sc9_montmul(r, x9, p2); sc_from_mont(r, r); sc_store_words(out2, r); }
// a*b + c*d under one reduction (the h-row of k_scalars_lanes)
void ht_sc9_mul2(const uint8_t a[32], const uint8_t b[32], const uint8_t c[32], const uint8_t d[32], uint8_t out[32]) {
  sc x, y, z, w, r; sc_load_words(x, a); sc_load_words(y, b); sc_load_words(z, c); sc_load_words(w, d);
  sc_to_mont(x, x); sc_to_mont(y, y); sc_to_mont(z, z); sc_to_mont(w, w);
  sc9 x9, y9, z9, w9; sc9_from(x9, x); sc9_from(y9, y); sc9_from(z9, z); sc9_from(w9, w);
  sc9_montmul2(r, x9, y9, z9, w9); sc_from_mont(r, r); sc_store_words(out, r); }
void ht_sc_addsub(const uint8_t a[32], const uint8_t b[32], uint8_t sum[32], uint8_t dif[32]) {
  sc x, y, z; sc_load_words(x, a); sc_load_words(y, b); sc_add(z, x, y); sc_store_words(sum, z); sc_sub(z, x, y); sc_store_words(dif, z); }
void ht_sc_wide(const uint8_t a[64], uint8_t out[32]) { sc z; sc_mont_from_wide(z, a); sc_from_mont(z, z); sc_store_words(out, z); }
void ht_host_wide(const uint8_t a[64], uint8_t out[32]) { host_wide_reduce(out, a); }  // the host weight chains' 64-bit form
void ht_sc_invert(const uint8_t a[32], uint8_t out[32]) { sc x; sc_load_words(x, a); sc_to_mont(x, x); sc_mont_invert(x, x); sc_from_mont(x, x); sc_store_words(out, x); }
void ht_sc_invert_vartime(const uint8_t a[32], uint8_t out[32]) { sc x; sc_load_words(x, a); sc_to_mont(x, x); sc_mont_invert_vartime(x, x); sc_from_mont(x, x); sc_store_words(out, x); }
void ht_sc_invert_plain(const uint8_t a[32], int which, uint8_t out[32]) { sc x, y; sc_load_words(x, a); if (which) sc_invert_vartime_plain(y, x); else sc_invert_xgcd_plain(y, x); sc_store_words(out, y); }
int ht_sc_canonical(const uint8_t a[32]) { return sc_is_canonical(a) ? 1 : 0; }
int ht_decompress_compress(const uint8_t in[32], uint8_t out[32]) {
  niels n; bool ok = ristretto_decompress(n, in); if (!ok) return 0;
  ge p; ge_identity(p); ge_madd(p, p, n); ristretto_compress(out, p); return 1; }
// decode, rebuild the extended point straight from the Niels form (ge_from_niels), negate optionally, add P: 2P or identity
int ht_from_niels(const uint8_t in[32], int neg, uint8_t out[32]) {
  niels n; if (!ristretto_decompress(n, in)) return 0;
  niels m = n; niels_cneg(m, neg != 0); ge p; ge_from_niels(p, m); ge_madd(p, p, n); ristretto_compress(out, p); return 1; }
// 2P +- P with the sign applied the way the MSM kernels do it (niels_load_swapped + ge_madd_swapped): 3P or P
int ht_madd_swapped(const uint8_t in[32], int neg, uint8_t out[32]) {
  niels n; if (!ristretto_decompress(n, in)) return 0;
  ge p; ge_identity(p); ge_madd(p, p, n); ge_dbl(p, p);
  niels q; niels_load_swapped(q, &n, neg != 0); ge_madd_swapped(p, p, q, neg != 0); ristretto_compress(out, p); return 1; }
// P (or -P) as an accumulator built by ge_from_niels_first from a sign-swapped load, then + P: 2P or the identity
int ht_from_niels_first(const uint8_t in[32], int neg, uint8_t out[32]) {
  niels n; if (!ristretto_decompress(n, in)) return 0;
  niels q; niels_load_swapped(q, &n, neg != 0); ge p; ge_from_niels_first(p, q); ge_madd(p, p, n); ristretto_compress(out, p); return 1; }
// the engine's RangeProof::from_bytes (upload_host.h: what bpp_batch_upload applies to raw proof bytes): 0 + (t, rounds), or the ProofError code
int ht_parse_proof(const uint8_t *p, size_t len, uint32_t *t, uint32_t *rounds) {
  try { ParsedItem pi; parse_proof(p, len, pi); *t = pi.t; *rounds = pi.rounds; return 0; } catch (const ProofErr &e) { return e.code; } }
// the device's decoding schedule (k_decompress) on the host: same verdict and point as the plain schedule
int ht_decompress_lean(const uint8_t in[32], uint8_t out[32]) {
  niels n; if (!ristretto_decompress_lean(n, in)) return 0;
  ge p; ge_identity(p); ge_madd(p, p, n); ristretto_compress(out, p); return 1; }
void ht_from_uniform(const uint8_t in[64], uint8_t out[32]) { ge p; ristretto_from_uniform(p, in); ristretto_compress(out, p); }
// out = compress(k*P) by double-and-add using dbl/madd/add; also exercises msub and ge_to_niels
int ht_scalarmult(const uint8_t k[32], const uint8_t pin[32], uint8_t out[32]) {
  niels n; if (!ristretto_decompress(n, pin)) return 0;
  ge acc; ge_identity(acc);
  for (int i = 255; i >= 0; i--) { ge_dbl(acc, acc); if ((k[i >> 3] >> (i & 7)) & 1) ge_madd(acc, acc, n); }
  // acc = acc + P - P through ge_add / msub paths
  ge q; ge_identity(q); ge_madd(q, q, n); ge t; ge_add(t, acc, q); ge_msub(t, t, n);
  niels tn; ge_to_niels(tn, t); ge r; ge_identity(r); ge_madd(r, r, tn);
  ristretto_compress(out, r); return 1; }
int ht_is_identity(const uint8_t pin[32]) { niels n; if (!ristretto_decompress(n, pin)) return -1; ge p; ge_identity(p); ge_madd(p, p, n); return ge_is_ristretto_identity(p) ? 1 : 0; }
void ht_merlin_kat(const uint8_t *label, uint32_t llen, const uint8_t *ml, uint32_t mll, const uint8_t *msg, uint32_t mlen, const uint8_t *cl, uint32_t cll, uint8_t *out, uint32_t n, uint8_t state_out[203]) {
  Strobe s; merlin_new(s, label, llen); merlin_append_message(s, ml, mll, msg, mlen); merlin_challenge_bytes(s, cl, cll, out, n); strobe_to_bytes(state_out, s); }
void ht_merlin_rng(const uint8_t state[203], const uint8_t *wit, uint32_t wlen, const uint8_t rnd[32], uint8_t *out, uint32_t n) {
  Strobe s; strobe_from_bytes(s, state); if (wlen) merlin_rng_rekey(s, (const uint8_t *)"witness", 7, wit, wlen); merlin_rng_finalize(s, rnd); merlin_rng_fill(s, out, n); }
// nonce() through the word-level BLAKE2b the kernels use (blake2b.h: nonce_hash_words); j, k < 0: absent
void ht_nonce_words(const uint8_t seed32[32], const char *label, uint32_t llen, int j, int k, uint8_t out[64]) {
  uint64_t h[8];
  nonce_hash_words(h, seed32, label, llen, j, k);
  for (int i = 0; i < 8; i++) for (int b = 0; b < 8; b++) out[8 * i + b] = (uint8_t)(h[i] >> (8 * b));
}
void ht_blake2b(const uint8_t *key, uint32_t klen, const uint8_t *persona, uint32_t plen, uint8_t out[64]) { blake2b512_keyed_personal_empty(out, key, klen, persona, plen); }
// batch-weight chains (chain_host.h): `width` chains of n proofs each in lock-step (1 = the template's scalar instance,
// 4 = AVX2, 8 = AVX-512); returns 0 when the CPU lacks the instruction set
int ht_weight_chains(const uint8_t *rng /* [width][n][32] */, uint32_t n, uint32_t width, uint8_t *out /* [width][n][32] */) {
  const uint8_t *in[8]; uint8_t *o[8];
  for (uint32_t k = 0; k < width && k < 8; k++) { in[k] = rng + (size_t)k * n * 32; o[k] = out + (size_t)k * n * 32; }
  if (width == 1) { weights_chain_multi_impl<1>(in, n, o); return 1; }
  if (width == 4) { if (!__builtin_cpu_supports("avx2")) return 0; weights_chain_x4(in, n, o); return 1; }
  if (width == 8) { if (!(__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl"))) return 0; weights_chain_x8(in, n, o); return 1; }
  return -1; }
// the WIDE forms (round 6: the host squeezes 64 bytes per proof, the device reduces them): `width` chains of n proofs each, the 64
// PRF bytes per proof as the sponge leaves them.  width 1 = wide_chain_single, 4 / 8 = the lock-step bundles (0 when the CPU lacks them)
int ht_wide_chains(const uint8_t *rng /* [width][n][32] */, uint32_t n, uint32_t width, uint8_t *out /* [width][n][64] */) {
  const uint8_t *in[8];
  uint8_t *o[8];
  for (uint32_t k = 0; k < width && k < 8; k++) { in[k] = rng + (size_t)k * n * 32; o[k] = out + (size_t)k * n * 64; }
  if (width == 1) { wide_chain_single(in[0], n, o[0]); return 1; }
  if (width == 4) { if (!__builtin_cpu_supports("avx2")) return 0; wide_chain_x4(in, n, o); return 1; }
  if (width == 8) { if (!(__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl"))) return 0; wide_chain_x8(in, n, o); return 1; }
  return -1; }
// bit interleaving of wkeccak.h (host-evaluable): halves of a word and back
void ht_wk_halves(uint64_t w, uint32_t out2[2]) { out2[0] = wk_half(w, 0); out2[1] = wk_half(w, 1); }
uint64_t ht_wk_word(uint32_t even, uint32_t odd) { return wk_word(even, odd); }
// one chain: form 0 = merlin.h's generic sponge, 1 = fast form without BMI, 2 = fast form compiled for BMI (0 when the CPU lacks it),
// 3 = whatever the engine picks at run time
int ht_weight_chain_single(const uint8_t *rng /* [n][32] */, uint32_t n, uint32_t form, uint8_t *out /* [n][32] */) {
  if (form == 0) { weights_chain_generic(rng, n, out); return 1; }
  if (form == 1) { weights_chain_single_plain(rng, n, out); return 1; }
  if (form == 2) { if (!(__builtin_cpu_supports("bmi") && __builtin_cpu_supports("bmi2"))) return 0; weights_chain_single_bmi(rng, n, out); return 1; }
  if (form == 3) { weights_chain_single(rng, n, out); return 1; }
  return -1; }
// recodings: digits of a canonical scalar for MSM window width c (uneven windows) / fixed-base window width w
int ht_msm_recode(const uint8_t a[32], uint32_t c, int16_t *digits /* K */, uint32_t *widths /* K */) {
  sc x; sc_load_words(x, a);
  const MsmPlan plan = msm_make_plan(c, 1, 1);
  msm_recode(digits, 1, x, plan);
  for (uint32_t k = 0; k < plan.K; k++)  // the one-window form must give the same digit (msm.h: k_msm_prelude uses it)
    if (msm_digit_at(x.v, plan, k) != (int32_t)digits[k]) return -1;
  for (uint32_t k = 0; k < plan.K; k++) widths[k] = k < plan.K_wide ? plan.c : plan.c - 1;
  return (int)plan.K; }
// half-scalar plan: digits of the two halves (128-bit windows) -> out_lo[K], out_hi[K]; returns K
int ht_msm_recode_split(const uint8_t a[32], uint32_t c, int16_t *out_lo, int16_t *out_hi, uint32_t *widths) {
  sc x; sc_load_words(x, a);
  const MsmPlan plan = msm_make_plan(c, 1, 2, BPP_MSM_SPLIT_BITS);
  uint32_t lo[8], hi[8];
  msm_half_words(lo, x.v, false);
  msm_half_words(hi, x.v, true);
  for (uint32_t k = 0; k < plan.K; k++) {
    out_lo[k] = (int16_t)msm_digit_at(lo, plan, k);
    out_hi[k] = (int16_t)msm_digit_at(hi, plan, k);
    widths[k] = k < plan.K_wide ? plan.c : plan.c - 1;
  }
  return (int)plan.K; }
int ht_fb_recode(const uint8_t a[32], uint32_t n_gens, int16_t *digits /* 32 */, uint32_t *wbits) {
  sc x; sc_load_words(x, a);
  const FbGeom g = fb_geometry(n_gens);
  fb_recode(digits, x, g);
  *wbits = g.wbits;
  return (int)g.items; }
}
