// Keccak-f[1600], STROBE-128 and the Merlin transcript operations the range-proof path uses.
// Host + device: the device replays one transcript per proof (PASS 1 of the verifier); the host runs
// the inherently sequential batch-weight chain and Transcript::new(label).
//
// Replaces (reference boundary): merlin::Transcript / TranscriptRng as driven by
// src/transcripts.rs:59-200 and src/protocols/transcript_protocol.rs:39-79; weight chain
// src/range_proof.rs:811,849,853,894.
#pragma once
#include "field.h"

namespace bpp {

BPP_HD uint64_t rotl64(uint64_t x, int n) { return (x << n) | (x >> (64 - n)); }

BPP_HD void keccak_f1600(uint64_t a[25]) {
  const uint64_t RC[24] = {
      0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808AULL, 0x8000000080008000ULL,
      0x000000000000808BULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
      0x000000000000008AULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000AULL,
      0x000000008000808BULL, 0x800000000000008BULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
      0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800AULL, 0x800000008000000AULL,
      0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
  uint64_t a00 = a[0], a01 = a[1], a02 = a[2], a03 = a[3], a04 = a[4];
  uint64_t a05 = a[5], a06 = a[6], a07 = a[7], a08 = a[8], a09 = a[9];
  uint64_t a10 = a[10], a11 = a[11], a12 = a[12], a13 = a[13], a14 = a[14];
  uint64_t a15 = a[15], a16 = a[16], a17 = a[17], a18 = a[18], a19 = a[19];
  uint64_t a20 = a[20], a21 = a[21], a22 = a[22], a23 = a[23], a24 = a[24];
#pragma unroll 1
  for (int rnd = 0; rnd < 24; rnd++) {
    // theta
    uint64_t c0 = a00 ^ a05 ^ a10 ^ a15 ^ a20;
    uint64_t c1 = a01 ^ a06 ^ a11 ^ a16 ^ a21;
    uint64_t c2 = a02 ^ a07 ^ a12 ^ a17 ^ a22;
    uint64_t c3 = a03 ^ a08 ^ a13 ^ a18 ^ a23;
    uint64_t c4 = a04 ^ a09 ^ a14 ^ a19 ^ a24;
    uint64_t d0 = c4 ^ rotl64(c1, 1);
    uint64_t d1 = c0 ^ rotl64(c2, 1);
    uint64_t d2 = c1 ^ rotl64(c3, 1);
    uint64_t d3 = c2 ^ rotl64(c4, 1);
    uint64_t d4 = c3 ^ rotl64(c0, 1);
    a00 ^= d0; a05 ^= d0; a10 ^= d0; a15 ^= d0; a20 ^= d0;
    a01 ^= d1; a06 ^= d1; a11 ^= d1; a16 ^= d1; a21 ^= d1;
    a02 ^= d2; a07 ^= d2; a12 ^= d2; a17 ^= d2; a22 ^= d2;
    a03 ^= d3; a08 ^= d3; a13 ^= d3; a18 ^= d3; a23 ^= d3;
    a04 ^= d4; a09 ^= d4; a14 ^= d4; a19 ^= d4; a24 ^= d4;
    // rho + pi: b[y][2x+3y] = rot(a[x][y])
    uint64_t b00 = a00;
    uint64_t b10 = rotl64(a01, 1);
    uint64_t b20 = rotl64(a02, 62);
    uint64_t b05 = rotl64(a03, 28);
    uint64_t b15 = rotl64(a04, 27);
    uint64_t b16 = rotl64(a05, 36);
    uint64_t b01 = rotl64(a06, 44);
    uint64_t b11 = rotl64(a07, 6);
    uint64_t b21 = rotl64(a08, 55);
    uint64_t b06 = rotl64(a09, 20);
    uint64_t b07 = rotl64(a10, 3);
    uint64_t b17 = rotl64(a11, 10);
    uint64_t b02 = rotl64(a12, 43);
    uint64_t b12 = rotl64(a13, 25);
    uint64_t b22 = rotl64(a14, 39);
    uint64_t b23 = rotl64(a15, 41);
    uint64_t b08 = rotl64(a16, 45);
    uint64_t b18 = rotl64(a17, 15);
    uint64_t b03 = rotl64(a18, 21);
    uint64_t b13 = rotl64(a19, 8);
    uint64_t b14 = rotl64(a20, 18);
    uint64_t b24 = rotl64(a21, 2);
    uint64_t b09 = rotl64(a22, 61);
    uint64_t b19 = rotl64(a23, 56);
    uint64_t b04 = rotl64(a24, 14);
    // chi
    a00 = b00 ^ (~b01 & b02); a01 = b01 ^ (~b02 & b03); a02 = b02 ^ (~b03 & b04); a03 = b03 ^ (~b04 & b00); a04 = b04 ^ (~b00 & b01);
    a05 = b05 ^ (~b06 & b07); a06 = b06 ^ (~b07 & b08); a07 = b07 ^ (~b08 & b09); a08 = b08 ^ (~b09 & b05); a09 = b09 ^ (~b05 & b06);
    a10 = b10 ^ (~b11 & b12); a11 = b11 ^ (~b12 & b13); a12 = b12 ^ (~b13 & b14); a13 = b13 ^ (~b14 & b10); a14 = b14 ^ (~b10 & b11);
    a15 = b15 ^ (~b16 & b17); a16 = b16 ^ (~b17 & b18); a17 = b17 ^ (~b18 & b19); a18 = b18 ^ (~b19 & b15); a19 = b19 ^ (~b15 & b16);
    a20 = b20 ^ (~b21 & b22); a21 = b21 ^ (~b22 & b23); a22 = b22 ^ (~b23 & b24); a23 = b23 ^ (~b24 & b20); a24 = b24 ^ (~b20 & b21);
    a00 ^= RC[rnd];
  }
  a[0] = a00; a[1] = a01; a[2] = a02; a[3] = a03; a[4] = a04;
  a[5] = a05; a[6] = a06; a[7] = a07; a[8] = a08; a[9] = a09;
  a[10] = a10; a[11] = a11; a[12] = a12; a[13] = a13; a[14] = a14;
  a[15] = a15; a[16] = a16; a[17] = a17; a[18] = a18; a[19] = a19;
  a[20] = a20; a[21] = a21; a[22] = a22; a[23] = a23; a[24] = a24;
}

// ---- STROBE-128 (rate 166) exactly as merlin 3.0.0 drives it ----
struct Strobe {
  uint64_t st[25];
  uint32_t pos, pos_begin, cur_flags;
};

#define BPP_STROBE_R 166u
#define BPP_FLAG_I 1u
#define BPP_FLAG_A 2u
#define BPP_FLAG_C 4u
#define BPP_FLAG_M 16u
#define BPP_FLAG_K 32u

BPP_HD void strobe_xor_byte(Strobe &s, uint32_t i, uint8_t b) { s.st[i >> 3] ^= (uint64_t)b << (8 * (i & 7)); }
BPP_HD uint8_t strobe_get_byte(const Strobe &s, uint32_t i) { return (uint8_t)(s.st[i >> 3] >> (8 * (i & 7))); }
BPP_HD void strobe_set_byte(Strobe &s, uint32_t i, uint8_t b) {
  uint64_t m = 0xffULL << (8 * (i & 7));
  s.st[i >> 3] = (s.st[i >> 3] & ~m) | ((uint64_t)b << (8 * (i & 7)));
}

BPP_HD void strobe_run_f(Strobe &s) {
  strobe_xor_byte(s, s.pos, (uint8_t)s.pos_begin);
  strobe_xor_byte(s, s.pos + 1, 0x04);
  strobe_xor_byte(s, BPP_STROBE_R + 1, 0x80);
  keccak_f1600(s.st);
  s.pos = 0;
  s.pos_begin = 0;
}

BPP_HD void strobe_absorb(Strobe &s, const uint8_t *data, uint32_t n) {
  for (uint32_t i = 0; i < n; i++) {
    strobe_xor_byte(s, s.pos, data[i]);
    s.pos++;
    if (s.pos == BPP_STROBE_R) strobe_run_f(s);
  }
}
BPP_HD void strobe_overwrite(Strobe &s, const uint8_t *data, uint32_t n) {
  for (uint32_t i = 0; i < n; i++) {
    strobe_set_byte(s, s.pos, data[i]);
    s.pos++;
    if (s.pos == BPP_STROBE_R) strobe_run_f(s);
  }
}
BPP_HD void strobe_squeeze(Strobe &s, uint8_t *out, uint32_t n) {
  for (uint32_t i = 0; i < n; i++) {
    out[i] = strobe_get_byte(s, s.pos);
    strobe_set_byte(s, s.pos, 0);
    s.pos++;
    if (s.pos == BPP_STROBE_R) strobe_run_f(s);
  }
}
BPP_HD void strobe_begin_op(Strobe &s, uint32_t flags, bool more) {
  if (more) return;
  uint8_t hdr[2] = {(uint8_t)s.pos_begin, (uint8_t)flags};
  s.pos_begin = s.pos + 1;
  s.cur_flags = flags;
  strobe_absorb(s, hdr, 2);
  bool force_f = (flags & (BPP_FLAG_C | BPP_FLAG_K)) != 0;
  if (force_f && s.pos != 0) strobe_run_f(s);
}
BPP_HD void strobe_meta_ad(Strobe &s, const uint8_t *d, uint32_t n, bool more) {
  strobe_begin_op(s, BPP_FLAG_M | BPP_FLAG_A, more);
  strobe_absorb(s, d, n);
}
BPP_HD void strobe_ad(Strobe &s, const uint8_t *d, uint32_t n, bool more) {
  strobe_begin_op(s, BPP_FLAG_A, more);
  strobe_absorb(s, d, n);
}
BPP_HD void strobe_prf(Strobe &s, uint8_t *out, uint32_t n, bool more) {
  strobe_begin_op(s, BPP_FLAG_I | BPP_FLAG_A | BPP_FLAG_C, more);
  strobe_squeeze(s, out, n);
}
BPP_HD void strobe_key(Strobe &s, const uint8_t *d, uint32_t n, bool more) {
  strobe_begin_op(s, BPP_FLAG_A | BPP_FLAG_C, more);
  strobe_overwrite(s, d, n);
}

BPP_HD void strobe_init(Strobe &s, const uint8_t *protocol_label, uint32_t n) {
  for (int i = 0; i < 25; i++) s.st[i] = 0;
  const uint8_t hdr[18] = {1, BPP_STROBE_R + 2, 1, 0, 1, 96, 'S', 'T', 'R', 'O', 'B', 'E', 'v', '1', '.', '0', '.', '2'};
  for (uint32_t i = 0; i < 18; i++) strobe_xor_byte(s, i, hdr[i]);
  keccak_f1600(s.st);
  s.pos = 0;
  s.pos_begin = 0;
  s.cur_flags = 0;
  strobe_meta_ad(s, protocol_label, n, false);
}

// 203-byte wire form: 200 state bytes, pos, pos_begin, cur_flags
BPP_HD void strobe_from_bytes(Strobe &s, const uint8_t *b) {
  for (int i = 0; i < 25; i++) {
    uint64_t w = 0;
    for (int k = 0; k < 8; k++) w |= (uint64_t)b[8 * i + k] << (8 * k);
    s.st[i] = w;
  }
  s.pos = b[200];
  s.pos_begin = b[201];
  s.cur_flags = b[202];
}
BPP_HD void strobe_to_bytes(uint8_t *b, const Strobe &s) {
  for (int i = 0; i < 25; i++)
    for (int k = 0; k < 8; k++) b[8 * i + k] = (uint8_t)(s.st[i] >> (8 * k));
  b[200] = (uint8_t)s.pos;
  b[201] = (uint8_t)s.pos_begin;
  b[202] = (uint8_t)s.cur_flags;
}

// ---- Merlin ----
BPP_HD void u32le(uint8_t o[4], uint32_t x) {
  o[0] = (uint8_t)x;
  o[1] = (uint8_t)(x >> 8);
  o[2] = (uint8_t)(x >> 16);
  o[3] = (uint8_t)(x >> 24);
}
BPP_HD void merlin_append_message(Strobe &s, const uint8_t *label, uint32_t llen, const uint8_t *msg, uint32_t mlen) {
  uint8_t len4[4];
  u32le(len4, mlen);
  strobe_meta_ad(s, label, llen, false);
  strobe_meta_ad(s, len4, 4, true);
  strobe_ad(s, msg, mlen, false);
}
BPP_HD void merlin_append_u64(Strobe &s, const uint8_t *label, uint32_t llen, uint64_t x) {
  uint8_t b[8];
  for (int k = 0; k < 8; k++) b[k] = (uint8_t)(x >> (8 * k));
  merlin_append_message(s, label, llen, b, 8);
}
BPP_HD void merlin_challenge_bytes(Strobe &s, const uint8_t *label, uint32_t llen, uint8_t *out, uint32_t n) {
  uint8_t len4[4];
  u32le(len4, n);
  strobe_meta_ad(s, label, llen, false);
  strobe_meta_ad(s, len4, 4, true);
  strobe_prf(s, out, n, false);
}
BPP_HD void merlin_new(Strobe &s, const uint8_t *label, uint32_t llen) {
  const uint8_t proto[11] = {'M', 'e', 'r', 'l', 'i', 'n', ' ', 'v', '1', '.', '0'};
  strobe_init(s, proto, 11);
  const uint8_t ds[7] = {'d', 'o', 'm', '-', 's', 'e', 'p'};
  merlin_append_message(s, ds, 7, label, llen);
}
// TranscriptRngBuilder on a CLONE of the transcript state
BPP_HD void merlin_rng_rekey(Strobe &rng, const uint8_t *label, uint32_t llen, const uint8_t *w, uint32_t wlen) {
  uint8_t len4[4];
  u32le(len4, wlen);
  strobe_meta_ad(rng, label, llen, false);
  strobe_meta_ad(rng, len4, 4, true);
  strobe_key(rng, w, wlen, false);
}
BPP_HD void merlin_rng_finalize(Strobe &rng, const uint8_t random32[32]) {
  const uint8_t l[3] = {'r', 'n', 'g'};
  strobe_meta_ad(rng, l, 3, false);
  strobe_key(rng, random32, 32, false);
}
BPP_HD void merlin_rng_fill(Strobe &rng, uint8_t *out, uint32_t n) {
  uint8_t len4[4];
  u32le(len4, n);
  strobe_meta_ad(rng, len4, 4, false);
  strobe_prf(rng, out, n, false);
}

}  // namespace bpp
