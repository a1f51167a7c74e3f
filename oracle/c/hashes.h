/* ORACLE (test infrastructure / CPU baseline only).
 * Keccak-f[1600] -> SHAKE256 / SHA3-512 (sha3 0.10.8), STROBE-128 + Merlin 3.0.0 transcript & RNG, BLAKE2b-512 MAC
 * (blake2 0.10.6).  None of these crates is under /root/reference; published algorithms restated (FIPS 202,
 * STROBE v1.0.2, merlin.cool, RFC 7693) and pinned by hashlib / merlin KATs in tests/test_oracle_c.py.
 * Reference call sites: src/transcripts.rs:59-200, src/protocols/transcript_protocol.rs:39-79,
 * src/generators/generators_chain.rs:23-33, src/protocols/curve_point_protocol.rs:31-35, src/utils/generic.rs:30-60. */
#ifndef ORACLE_HASHES_H
#define ORACLE_HASHES_H
#include <stdint.h>
#include <string.h>

static __thread uint64_t g_keccak_count = 0; /* instrumentation (per thread: the all-cores baseline must not share a counter line) */

static void keccakf(uint64_t st[25]) {
  static const uint64_t rc[24] = {
      0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
      0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
      0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
      0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
      0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
  static const int rotc[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
  static const int piln[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
  g_keccak_count++;
  for (int r = 0; r < 24; r++) {
    uint64_t bc[5], t;
    for (int i = 0; i < 5; i++) bc[i] = st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20];
    for (int i = 0; i < 5; i++) {
      t = bc[(i + 4) % 5] ^ ((bc[(i + 1) % 5] << 1) | (bc[(i + 1) % 5] >> 63));
      for (int j = 0; j < 25; j += 5) st[j + i] ^= t;
    }
    t = st[1];
    for (int i = 0; i < 24; i++) {
      int j = piln[i]; uint64_t b = st[j];
      st[j] = (t << rotc[i]) | (t >> (64 - rotc[i])); t = b;
    }
    for (int j = 0; j < 25; j += 5) {
      for (int i = 0; i < 5; i++) bc[i] = st[j + i];
      for (int i = 0; i < 5; i++) st[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
    }
    st[0] ^= rc[r];
  }
}

/* generic sponge on a little-endian host */
static void keccak_sponge(const uint8_t *in, size_t inlen, uint8_t *out, size_t outlen, unsigned rate, uint8_t pad) {
  uint64_t st[25]; memset(st, 0, sizeof(st)); uint8_t *sb = (uint8_t *)st;
  while (inlen >= rate) { for (unsigned i = 0; i < rate; i++) sb[i] ^= in[i]; keccakf(st); in += rate; inlen -= rate; }
  for (size_t i = 0; i < inlen; i++) sb[i] ^= in[i];
  sb[inlen] ^= pad; sb[rate - 1] ^= 0x80; keccakf(st);
  while (outlen) { size_t take = outlen < rate ? outlen : rate; memcpy(out, sb, take); out += take; outlen -= take; if (outlen) keccakf(st); }
}
static void shake256(const uint8_t *in, size_t inlen, uint8_t *out, size_t outlen) { keccak_sponge(in, inlen, out, outlen, 136, 0x1f); }
static void sha3_512(const uint8_t *in, size_t inlen, uint8_t out[64]) { keccak_sponge(in, inlen, out, 64, 72, 0x06); }

/* ---- STROBE-128 as merlin drives it ---- */
typedef struct { uint8_t st[200]; uint8_t pos, pos_begin, cur_flags; } strobe_t;
enum { ST_R = 166, FL_I = 1, FL_A = 2, FL_C = 4, FL_T = 8, FL_M = 16, FL_K = 32 };

static void strobe_runf(strobe_t *s) {
  s->st[s->pos] ^= s->pos_begin; s->st[s->pos + 1] ^= 0x04; s->st[ST_R + 1] ^= 0x80;
  uint64_t w[25]; memcpy(w, s->st, 200); keccakf(w); memcpy(s->st, w, 200);
  s->pos = 0; s->pos_begin = 0;
}
static void strobe_absorb(strobe_t *s, const uint8_t *d, size_t n) { for (size_t i = 0; i < n; i++) { s->st[s->pos++] ^= d[i]; if (s->pos == ST_R) strobe_runf(s); } }
static void strobe_overwrite(strobe_t *s, const uint8_t *d, size_t n) { for (size_t i = 0; i < n; i++) { s->st[s->pos++] = d[i]; if (s->pos == ST_R) strobe_runf(s); } }
static void strobe_squeeze(strobe_t *s, uint8_t *d, size_t n) { for (size_t i = 0; i < n; i++) { d[i] = s->st[s->pos]; s->st[s->pos++] = 0; if (s->pos == ST_R) strobe_runf(s); } }
static void strobe_begin(strobe_t *s, uint8_t flags, int more) {
  if (more) return;
  uint8_t hdr[2] = { s->pos_begin, flags };
  s->pos_begin = s->pos + 1; s->cur_flags = flags; strobe_absorb(s, hdr, 2);
  if ((flags & (FL_C | FL_K)) && s->pos != 0) strobe_runf(s);
}
static void strobe_meta_ad(strobe_t *s, const void *d, size_t n, int more) { strobe_begin(s, FL_M | FL_A, more); strobe_absorb(s, (const uint8_t *)d, n); }
static void strobe_ad(strobe_t *s, const void *d, size_t n, int more) { strobe_begin(s, FL_A, more); strobe_absorb(s, (const uint8_t *)d, n); }
static void strobe_prf(strobe_t *s, uint8_t *d, size_t n, int more) { strobe_begin(s, FL_I | FL_A | FL_C, more); strobe_squeeze(s, d, n); }
static void strobe_key(strobe_t *s, const void *d, size_t n, int more) { strobe_begin(s, FL_A | FL_C, more); strobe_overwrite(s, (const uint8_t *)d, n); }
static void strobe_new(strobe_t *s, const void *label, size_t n) {
  memset(s, 0, sizeof(*s));
  const uint8_t hdr[6] = {1, ST_R + 2, 1, 0, 1, 96}; memcpy(s->st, hdr, 6); memcpy(s->st + 6, "STROBEv1.0.2", 12);
  uint64_t w[25]; memcpy(w, s->st, 200); keccakf(w); memcpy(s->st, w, 200);
  strobe_meta_ad(s, label, n, 0);
}

/* ---- merlin::Transcript ---- */
typedef strobe_t transcript_t;
static void le32(uint8_t o[4], uint32_t x) { o[0] = (uint8_t)x; o[1] = (uint8_t)(x >> 8); o[2] = (uint8_t)(x >> 16); o[3] = (uint8_t)(x >> 24); }
static void transcript_append(transcript_t *t, const char *label, const void *msg, size_t n) {
  uint8_t l4[4]; le32(l4, (uint32_t)n);
  strobe_meta_ad(t, label, strlen(label), 0); strobe_meta_ad(t, l4, 4, 1); strobe_ad(t, msg, n, 0);
}
static void transcript_append_u64(transcript_t *t, const char *label, uint64_t x) { uint8_t b[8]; for (int k = 0; k < 8; k++) b[k] = (uint8_t)(x >> (8 * k)); transcript_append(t, label, b, 8); }
static void transcript_challenge(transcript_t *t, const char *label, uint8_t *out, size_t n) {
  uint8_t l4[4]; le32(l4, (uint32_t)n);
  strobe_meta_ad(t, label, strlen(label), 0); strobe_meta_ad(t, l4, 4, 1); strobe_prf(t, out, n, 0);
}
static void transcript_new(transcript_t *t, const void *label, size_t n) { strobe_new(t, "Merlin v1.0", 11); uint8_t l4[4]; le32(l4, (uint32_t)n); strobe_meta_ad(t, "dom-sep", 7, 0); strobe_meta_ad(t, l4, 4, 1); strobe_ad(t, label, n, 0); }
/* TranscriptRngBuilder / TranscriptRng operate on a copy of the state */
static void rng_rekey(strobe_t *r, const char *label, const void *w, size_t n) { uint8_t l4[4]; le32(l4, (uint32_t)n); strobe_meta_ad(r, label, strlen(label), 0); strobe_meta_ad(r, l4, 4, 1); strobe_key(r, w, n, 0); }
static void rng_finalize(strobe_t *r, const uint8_t rnd[32]) { strobe_meta_ad(r, "rng", 3, 0); strobe_key(r, rnd, 32, 0); }
static void rng_fill(strobe_t *r, uint8_t *out, size_t n) { uint8_t l4[4]; le32(l4, (uint32_t)n); strobe_meta_ad(r, l4, 4, 0); strobe_prf(r, out, n, 0); }

/* ---- BLAKE2b-512, keyed + personalised, empty message (src/utils/generic.rs:56-57) ---- */
static inline uint64_t ror64(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
static void blake2b_mac512_empty(uint8_t out[64], const uint8_t *key, size_t klen, const uint8_t *persona, size_t plen) {
  static const uint64_t iv[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                                 0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
  static const uint8_t sigma[12][16] = {
      {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
      {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
      {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
      {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
      {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
      {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
  uint8_t param[64]; memset(param, 0, 64);
  param[0] = 64; param[1] = (uint8_t)klen; param[2] = 1; param[3] = 1; memcpy(param + 48, persona, plen);
  uint64_t h[8], m[16], v[16]; uint8_t block[128];
  for (int i = 0; i < 8; i++) { uint64_t w = 0; for (int k = 0; k < 8; k++) w |= (uint64_t)param[8 * i + k] << (8 * k); h[i] = iv[i] ^ w; }
  memset(block, 0, 128); memcpy(block, key, klen);
  for (int i = 0; i < 16; i++) { uint64_t w = 0; for (int k = 0; k < 8; k++) w |= (uint64_t)block[8 * i + k] << (8 * k); m[i] = w; }
  for (int i = 0; i < 8; i++) { v[i] = h[i]; v[i + 8] = iv[i]; }
  v[12] ^= 128; v[14] = ~v[14];
#define B2G(a, b, c, d, x, y) v[a] += v[b] + (x); v[d] = ror64(v[d] ^ v[a], 32); v[c] += v[d]; v[b] = ror64(v[b] ^ v[c], 24); \
  v[a] += v[b] + (y); v[d] = ror64(v[d] ^ v[a], 16); v[c] += v[d]; v[b] = ror64(v[b] ^ v[c], 63);
  for (int r = 0; r < 12; r++) {
    const uint8_t *s = sigma[r];
    B2G(0, 4, 8, 12, m[s[0]], m[s[1]]) B2G(1, 5, 9, 13, m[s[2]], m[s[3]]) B2G(2, 6, 10, 14, m[s[4]], m[s[5]]) B2G(3, 7, 11, 15, m[s[6]], m[s[7]])
    B2G(0, 5, 10, 15, m[s[8]], m[s[9]]) B2G(1, 6, 11, 12, m[s[10]], m[s[11]]) B2G(2, 7, 8, 13, m[s[12]], m[s[13]]) B2G(3, 4, 9, 14, m[s[14]], m[s[15]])
  }
#undef B2G
  for (int i = 0; i < 8; i++) { h[i] ^= v[i] ^ v[i + 8]; for (int k = 0; k < 8; k++) out[8 * i + k] = (uint8_t)(h[i] >> (8 * k)); }
}
#endif
