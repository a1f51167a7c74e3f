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

// The three primitives of a round.  On the device a 64-bit word is two registers: gfx950's three-input boolean
// instruction v_bitop3_b32 does theta's a ^ b ^ c (table 0x96) and chi's a ^ (~b & c) (table 0xD2) in one instruction per
// half instead of two, and a rotation by a constant is two v_alignbit_b32 (the compiler's 64-bit shift pair + or takes
// three): 290 -> 196 VALU instructions per round.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ uint64_t kk_join(uint32_t lo, uint32_t hi) { return (uint64_t)lo | ((uint64_t)hi << 32); }
__device__ __forceinline__ uint64_t kk_xor3(uint64_t a, uint64_t b, uint64_t c) {
  return kk_join(__builtin_amdgcn_bitop3_b32((uint32_t)a, (uint32_t)b, (uint32_t)c, 0x96),
                 __builtin_amdgcn_bitop3_b32((uint32_t)(a >> 32), (uint32_t)(b >> 32), (uint32_t)(c >> 32), 0x96));
}
__device__ __forceinline__ uint64_t kk_chi(uint64_t a, uint64_t b, uint64_t c) {
  return kk_join(__builtin_amdgcn_bitop3_b32((uint32_t)a, (uint32_t)b, (uint32_t)c, 0xD2),
                 __builtin_amdgcn_bitop3_b32((uint32_t)(a >> 32), (uint32_t)(b >> 32), (uint32_t)(c >> 32), 0xD2));
}
template <int N>
__device__ __forceinline__ uint64_t kk_rol(uint64_t x) {
  static_assert(N > 0 && N < 64, "rotation amount");
  const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
  if (N == 32) return kk_join(hi, lo);
  if (N < 32) return kk_join(__builtin_amdgcn_alignbit(lo, hi, 32 - N), __builtin_amdgcn_alignbit(hi, lo, 32 - N));
  return kk_join(__builtin_amdgcn_alignbit(hi, lo, 64 - N), __builtin_amdgcn_alignbit(lo, hi, 64 - N));
}
#else
BPP_HD uint64_t kk_xor3(uint64_t a, uint64_t b, uint64_t c) { return a ^ b ^ c; }
BPP_HD uint64_t kk_chi(uint64_t a, uint64_t b, uint64_t c) { return a ^ (~b & c); }
template <int N>
BPP_HD uint64_t kk_rol(uint64_t x) {
  return (x << N) | (x >> (64 - N));
}
#endif

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
    // theta: column parities, then every word takes parity(x-1) ^ rol(parity(x+1), 1) in one three-input xor
    const uint64_t c0 = kk_xor3(kk_xor3(a00, a05, a10), a15, a20);
    const uint64_t c1 = kk_xor3(kk_xor3(a01, a06, a11), a16, a21);
    const uint64_t c2 = kk_xor3(kk_xor3(a02, a07, a12), a17, a22);
    const uint64_t c3 = kk_xor3(kk_xor3(a03, a08, a13), a18, a23);
    const uint64_t c4 = kk_xor3(kk_xor3(a04, a09, a14), a19, a24);
    const uint64_t r0 = kk_rol<1>(c0), r1 = kk_rol<1>(c1), r2 = kk_rol<1>(c2), r3 = kk_rol<1>(c3), r4 = kk_rol<1>(c4);
    a00 = kk_xor3(a00, c4, r1); a05 = kk_xor3(a05, c4, r1); a10 = kk_xor3(a10, c4, r1); a15 = kk_xor3(a15, c4, r1); a20 = kk_xor3(a20, c4, r1);
    a01 = kk_xor3(a01, c0, r2); a06 = kk_xor3(a06, c0, r2); a11 = kk_xor3(a11, c0, r2); a16 = kk_xor3(a16, c0, r2); a21 = kk_xor3(a21, c0, r2);
    a02 = kk_xor3(a02, c1, r3); a07 = kk_xor3(a07, c1, r3); a12 = kk_xor3(a12, c1, r3); a17 = kk_xor3(a17, c1, r3); a22 = kk_xor3(a22, c1, r3);
    a03 = kk_xor3(a03, c2, r4); a08 = kk_xor3(a08, c2, r4); a13 = kk_xor3(a13, c2, r4); a18 = kk_xor3(a18, c2, r4); a23 = kk_xor3(a23, c2, r4);
    a04 = kk_xor3(a04, c3, r0); a09 = kk_xor3(a09, c3, r0); a14 = kk_xor3(a14, c3, r0); a19 = kk_xor3(a19, c3, r0); a24 = kk_xor3(a24, c3, r0);
    // rho + pi: b[y][2x+3y] = rot(a[x][y])
    const uint64_t b00 = a00;
    const uint64_t b10 = kk_rol<1>(a01);
    const uint64_t b20 = kk_rol<62>(a02);
    const uint64_t b05 = kk_rol<28>(a03);
    const uint64_t b15 = kk_rol<27>(a04);
    const uint64_t b16 = kk_rol<36>(a05);
    const uint64_t b01 = kk_rol<44>(a06);
    const uint64_t b11 = kk_rol<6>(a07);
    const uint64_t b21 = kk_rol<55>(a08);
    const uint64_t b06 = kk_rol<20>(a09);
    const uint64_t b07 = kk_rol<3>(a10);
    const uint64_t b17 = kk_rol<10>(a11);
    const uint64_t b02 = kk_rol<43>(a12);
    const uint64_t b12 = kk_rol<25>(a13);
    const uint64_t b22 = kk_rol<39>(a14);
    const uint64_t b23 = kk_rol<41>(a15);
    const uint64_t b08 = kk_rol<45>(a16);
    const uint64_t b18 = kk_rol<15>(a17);
    const uint64_t b03 = kk_rol<21>(a18);
    const uint64_t b13 = kk_rol<8>(a19);
    const uint64_t b14 = kk_rol<18>(a20);
    const uint64_t b24 = kk_rol<2>(a21);
    const uint64_t b09 = kk_rol<61>(a22);
    const uint64_t b19 = kk_rol<56>(a23);
    const uint64_t b04 = kk_rol<14>(a24);
    // chi
    a00 = kk_chi(b00, b01, b02); a01 = kk_chi(b01, b02, b03); a02 = kk_chi(b02, b03, b04); a03 = kk_chi(b03, b04, b00); a04 = kk_chi(b04, b00, b01);
    a05 = kk_chi(b05, b06, b07); a06 = kk_chi(b06, b07, b08); a07 = kk_chi(b07, b08, b09); a08 = kk_chi(b08, b09, b05); a09 = kk_chi(b09, b05, b06);
    a10 = kk_chi(b10, b11, b12); a11 = kk_chi(b11, b12, b13); a12 = kk_chi(b12, b13, b14); a13 = kk_chi(b13, b14, b10); a14 = kk_chi(b14, b10, b11);
    a15 = kk_chi(b15, b16, b17); a16 = kk_chi(b16, b17, b18); a17 = kk_chi(b17, b18, b19); a18 = kk_chi(b18, b19, b15); a19 = kk_chi(b19, b15, b16);
    a20 = kk_chi(b20, b21, b22); a21 = kk_chi(b21, b22, b23); a22 = kk_chi(b22, b23, b24); a23 = kk_chi(b23, b24, b20); a24 = kk_chi(b24, b20, b21);
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
