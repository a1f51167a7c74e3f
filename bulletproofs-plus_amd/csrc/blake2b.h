// BLAKE2b-512, keyed + personalised, EMPTY message: exactly the shape of the reference's nonce()
// (src/utils/generic.rs:30-60: Blake2bMac512::new_with_salt_and_personal(key, &[], label), no update).
// Host + device.  Output = 64 bytes, fed to the wide scalar reduction (src/protocols/scalar_protocol.rs:32-36).
#pragma once
#include "field.h"

namespace bpp {

BPP_HD uint64_t rotr64(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }

#define BPP_B2B_G(a, b, c, d, x, y) \
  do {                              \
    a = a + b + (x);                \
    d = rotr64(d ^ a, 32);          \
    c = c + d;                      \
    b = rotr64(b ^ c, 24);          \
    a = a + b + (y);                \
    d = rotr64(d ^ a, 16);          \
    c = c + d;                      \
    b = rotr64(b ^ c, 63);          \
  } while (0)

// The same hash with the key block and the personalisation handed over as WORDS, fully unrolled (every message-schedule
// index is a constant): nothing is indexed at run time, so on the device neither the block nor the state ever leaves the
// registers -- the byte-wise form below, called with a key assembled in a local byte array, kept that array and the block in
// scratch memory (k_masks 48 bytes, kp_lane 80).  m[0..7]: the key, zero padded (klen <= 64); p0 / p1: the personalisation.
BPP_HD void blake2b512_keyed_words(uint64_t h[8], const uint64_t mk[8], uint32_t klen, uint64_t p0, uint64_t p1) {
  const uint64_t IV0 = 0x6a09e667f3bcc908ULL, IV1 = 0xbb67ae8584caa73bULL, IV2 = 0x3c6ef372fe94f82bULL, IV3 = 0xa54ff53a5f1d36f1ULL,
                 IV4 = 0x510e527fade682d1ULL, IV5 = 0x9b05688c2b3e6c1fULL, IV6 = 0x1f83d9abfb41bd6bULL, IV7 = 0x5be0cd19137e2179ULL;
  h[0] = IV0 ^ 0x01010000ULL ^ ((uint64_t)klen << 8) ^ 64ULL;
  h[1] = IV1;
  h[2] = IV2;
  h[3] = IV3;
  h[4] = IV4;
  h[5] = IV5;
  h[6] = IV6 ^ p0;
  h[7] = IV7 ^ p1;
  const uint64_t m0 = mk[0], m1 = mk[1], m2 = mk[2], m3 = mk[3], m4 = mk[4], m5 = mk[5], m6 = mk[6], m7 = mk[7];
  const uint64_t m8 = 0, m9 = 0, m10 = 0, m11 = 0, m12 = 0, m13 = 0, m14 = 0, m15 = 0;  // the key fills at most half the block
  uint64_t v0 = h[0], v1 = h[1], v2 = h[2], v3 = h[3], v4 = h[4], v5 = h[5], v6 = h[6], v7 = h[7];
  uint64_t v8 = IV0, v9 = IV1, v10 = IV2, v11 = IV3, v12 = IV4 ^ 128ULL, v13 = IV5, v14 = ~IV6, v15 = IV7;
#define BPP_B2B_ROUND(a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11, a12, a13, a14, a15)   BPP_B2B_G(v0, v4, v8, v12, m##a0, m##a1);                                                     BPP_B2B_G(v1, v5, v9, v13, m##a2, m##a3);                                                     BPP_B2B_G(v2, v6, v10, v14, m##a4, m##a5);                                                    BPP_B2B_G(v3, v7, v11, v15, m##a6, m##a7);                                                    BPP_B2B_G(v0, v5, v10, v15, m##a8, m##a9);                                                    BPP_B2B_G(v1, v6, v11, v12, m##a10, m##a11);                                                  BPP_B2B_G(v2, v7, v8, v13, m##a12, m##a13);                                                   BPP_B2B_G(v3, v4, v9, v14, m##a14, m##a15)
  BPP_B2B_ROUND(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
  BPP_B2B_ROUND(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3);
  BPP_B2B_ROUND(11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4);
  BPP_B2B_ROUND(7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8);
  BPP_B2B_ROUND(9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13);
  BPP_B2B_ROUND(2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9);
  BPP_B2B_ROUND(12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11);
  BPP_B2B_ROUND(13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10);
  BPP_B2B_ROUND(6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5);
  BPP_B2B_ROUND(10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0);
  BPP_B2B_ROUND(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
  BPP_B2B_ROUND(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3);
#undef BPP_B2B_ROUND
  h[0] ^= v0 ^ v8;
  h[1] ^= v1 ^ v9;
  h[2] ^= v2 ^ v10;
  h[3] ^= v3 ^ v11;
  h[4] ^= v4 ^ v12;
  h[5] ^= v5 ^ v13;
  h[6] ^= v6 ^ v14;
  h[7] ^= v7 ^ v15;
}

// nonce() of the reference (src/utils/generic.rs:30-60) as 64 output bytes in eight words: key = 0x00 || seed (32 bytes) ||
// ['j' || u32le(j)] || ['k' || u32le(k)] (j, k < 0: absent), personalisation = label (at most 16 bytes), empty message.
// The key is put together by shifts at FIXED word positions (byte 33 and byte 38 both start in word 4), so no byte array exists.
BPP_HD uint64_t b2b_le64(const uint8_t *p) {
  uint64_t v = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) v |= (uint64_t)p[i] << (8 * i);
  return v;
}
BPP_HD void nonce_hash_words(uint64_t h[8], const uint8_t seed32[32], const char *label, uint32_t llen, int j, int k) {
  const uint64_t s0 = b2b_le64(seed32), s1 = b2b_le64(seed32 + 8), s2 = b2b_le64(seed32 + 16), s3 = b2b_le64(seed32 + 24);
  uint64_t m[8] = {s0 << 8, (s0 >> 56) | (s1 << 8), (s1 >> 56) | (s2 << 8), (s2 >> 56) | (s3 << 8), s3 >> 56, 0, 0, 0};
  uint32_t n = 33;
  if (j >= 0) {  // bytes 33..37: inside word 4
    m[4] |= ((uint64_t)'j' | ((uint64_t)(uint32_t)j << 8)) << 8;
    n = 38;
  }
  if (k >= 0) {  // bytes 33..37 (word 4) or 38..42 (two bytes in word 4, three in word 5)
    const uint64_t v = (uint64_t)'k' | ((uint64_t)(uint32_t)k << 8);
    if (n == 33) {
      m[4] |= v << 8;
    } else {
      m[4] |= v << 48;
      m[5] |= v >> 16;
    }
    n += 5;
  }
  uint64_t p0 = 0, p1 = 0;
  for (uint32_t i = 0; i < llen && i < 8; i++) p0 |= (uint64_t)(uint8_t)label[i] << (8 * i);
  for (uint32_t i = 8; i < llen && i < 16; i++) p1 |= (uint64_t)(uint8_t)label[i] << (8 * (i - 8));
  blake2b512_keyed_words(h, m, n, p0, p1);
}

// key: klen <= 64 bytes; persona: plen <= 16 bytes
BPP_HD void blake2b512_keyed_personal_empty(uint8_t out[64], const uint8_t *key, uint32_t klen, const uint8_t *persona,
                                            uint32_t plen) {
  const uint64_t IV[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                          0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
  const uint8_t SIGMA[12][16] = {
      {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
      {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
      {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
      {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
      {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
      {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
  uint64_t h[8];
  for (int i = 0; i < 8; i++) h[i] = IV[i];
  // parameter block word 0: digest_length=64 | key_length<<8 | fanout=1<<16 | depth=1<<24
  h[0] ^= 0x01010000ULL ^ ((uint64_t)klen << 8) ^ 64ULL;
  // words 4,5 = salt (zero); words 6,7 = personal
  uint64_t p0 = 0, p1 = 0;
  for (uint32_t i = 0; i < plen && i < 8; i++) p0 |= (uint64_t)persona[i] << (8 * i);
  for (uint32_t i = 8; i < plen && i < 16; i++) p1 |= (uint64_t)persona[i] << (8 * (i - 8));
  h[6] ^= p0;
  h[7] ^= p1;
  // single block = key padded to 128 bytes, t = 128, last block
  uint64_t m[16];
  for (int i = 0; i < 16; i++) m[i] = 0;
  for (uint32_t i = 0; i < klen; i++) m[i >> 3] |= (uint64_t)key[i] << (8 * (i & 7));
  uint64_t v[16];
  for (int i = 0; i < 8; i++) {
    v[i] = h[i];
    v[i + 8] = IV[i];
  }
  v[12] ^= 128ULL;  // t0
  v[14] = ~v[14];   // f0
  for (int r = 0; r < 12; r++) {
    const uint8_t *s = SIGMA[r];
    BPP_B2B_G(v[0], v[4], v[8], v[12], m[s[0]], m[s[1]]);
    BPP_B2B_G(v[1], v[5], v[9], v[13], m[s[2]], m[s[3]]);
    BPP_B2B_G(v[2], v[6], v[10], v[14], m[s[4]], m[s[5]]);
    BPP_B2B_G(v[3], v[7], v[11], v[15], m[s[6]], m[s[7]]);
    BPP_B2B_G(v[0], v[5], v[10], v[15], m[s[8]], m[s[9]]);
    BPP_B2B_G(v[1], v[6], v[11], v[12], m[s[10]], m[s[11]]);
    BPP_B2B_G(v[2], v[7], v[8], v[13], m[s[12]], m[s[13]]);
    BPP_B2B_G(v[3], v[4], v[9], v[14], m[s[14]], m[s[15]]);
  }
  for (int i = 0; i < 8; i++) {
    h[i] ^= v[i] ^ v[i + 8];
    for (int k = 0; k < 8; k++) out[8 * i + k] = (uint8_t)(h[i] >> (8 * k));
  }
}

}  // namespace bpp
