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
