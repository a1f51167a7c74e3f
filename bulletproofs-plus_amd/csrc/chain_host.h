// Host side of the batch-weight transcript (src/range_proof.rs:811,849,853,894), W independent chains in lockstep.
//
// One chain is a strictly sequential sponge (1.27 Keccak-f per proof), so a single chain cannot be parallelised --
// but the chains of DIFFERENT reference batches (chunks / groups) have identical control flow (same message lengths),
// so W of them share one instruction stream: the STROBE state is kept lane-transposed (st[i][w]) and Keccak-f runs on
// W-lane vectors (AVX-512: 8 chains, AVX2: 4, scalar: 1; chosen at run time).  Host only.
#pragma once
#include <string.h>

#include "merlin.h"
#include "scalar.h"

namespace bpp {

template <int W>
struct VecOps {
  typedef uint64_t vec __attribute__((vector_size(8 * W)));
};

template <int W>
static inline __attribute__((always_inline)) void keccak_f1600_vec(typename VecOps<W>::vec a[25]) {
  typedef typename VecOps<W>::vec V;
  static const uint64_t RC[24] = {
      0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808AULL, 0x8000000080008000ULL,
      0x000000000000808BULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
      0x000000000000008AULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000AULL,
      0x000000008000808BULL, 0x800000000000008BULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
      0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800AULL, 0x800000008000000AULL,
      0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
  static const int ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
  static const int PI[25] = {0, 10, 20, 5, 15, 16, 1, 11, 21, 6, 7, 17, 2, 12, 22, 23, 8, 18, 3, 13, 14, 24, 9, 19, 4};
#define BPP_VROL(x, n) (((x) << (n)) | ((x) >> (64 - (n))))
  for (int rnd = 0; rnd < 24; rnd++) {
    V c[5], d[5], b[25];
    for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
    for (int x = 0; x < 5; x++) d[x] = c[(x + 4) % 5] ^ BPP_VROL(c[(x + 1) % 5], 1);
    for (int i = 0; i < 25; i++) {
      V t = a[i] ^ d[i % 5];
      b[PI[i]] = ROT[i] ? BPP_VROL(t, ROT[i]) : t;
    }
    for (int y = 0; y < 25; y += 5)
      for (int x = 0; x < 5; x++) a[y + x] = b[y + x] ^ (~b[y + (x + 1) % 5] & b[y + (x + 2) % 5]);
    V rc;
    for (int w = 0; w < W; w++) rc[w] = RC[rnd];
    a[0] ^= rc;
  }
#undef BPP_VROL
}

// W lane-transposed STROBE-128 states with shared position bookkeeping
template <int W>
struct MultiStrobe {
  typedef typename VecOps<W>::vec V;
  V st[25];
  uint32_t pos, pos_begin, cur_flags;
};

template <int W>
static inline __attribute__((always_inline)) void ms_xor_byte(MultiStrobe<W> &s, uint32_t i, int w, uint8_t b) {
  s.st[i >> 3][w] ^= (uint64_t)b << (8 * (i & 7));
}
template <int W>
static inline __attribute__((always_inline)) void ms_run_f(MultiStrobe<W> &s) {
  for (int w = 0; w < W; w++) {
    ms_xor_byte(s, s.pos, w, (uint8_t)s.pos_begin);
    ms_xor_byte(s, s.pos + 1, w, 0x04);
    ms_xor_byte(s, BPP_STROBE_R + 1, w, 0x80);
  }
  keccak_f1600_vec<W>(s.st);
  s.pos = 0;
  s.pos_begin = 0;
}
// absorb the same-length message of every chain (data[w] + off)
template <int W>
static inline __attribute__((always_inline)) void ms_absorb(MultiStrobe<W> &s, const uint8_t *const data[W], size_t off, uint32_t n) {
  for (uint32_t i = 0; i < n; i++) {
    for (int w = 0; w < W; w++) ms_xor_byte(s, s.pos, w, data[w][off + i]);
    s.pos++;
    if (s.pos == BPP_STROBE_R) ms_run_f(s);
  }
}
template <int W>
static inline __attribute__((always_inline)) void ms_absorb_same(MultiStrobe<W> &s, const uint8_t *d, uint32_t n) {
  const uint8_t *rep[W];
  for (int w = 0; w < W; w++) rep[w] = d;
  ms_absorb<W>(s, rep, 0, n);
}
template <int W>
static inline __attribute__((always_inline)) void ms_begin_op(MultiStrobe<W> &s, uint32_t flags, bool more) {
  if (more) return;
  uint8_t hdr[2] = {(uint8_t)s.pos_begin, (uint8_t)flags};
  s.pos_begin = s.pos + 1;
  s.cur_flags = flags;
  ms_absorb_same<W>(s, hdr, 2);
  if ((flags & (BPP_FLAG_C | BPP_FLAG_K)) && s.pos != 0) ms_run_f(s);
}
template <int W>
static inline __attribute__((always_inline)) void ms_squeeze(MultiStrobe<W> &s, uint8_t *const out[W], uint32_t n) {
  for (uint32_t i = 0; i < n; i++) {
    const uint32_t sh = 8 * (s.pos & 7);
    for (int w = 0; w < W; w++) {
      out[w][i] = (uint8_t)(s.st[s.pos >> 3][w] >> sh);
      s.st[s.pos >> 3][w] &= ~(0xffULL << sh);
    }
    s.pos++;
    if (s.pos == BPP_STROBE_R) ms_run_f(s);
  }
}

// W chains: rng[w] -> n x 32 transcript-RNG bytes, out[w] -> n x 32 canonical weights
template <int W>
static inline __attribute__((always_inline)) void weights_chain_multi_impl(const uint8_t *const rng[W], size_t n, uint8_t *const out[W]) {
  // Transcript::new(b"Bulletproofs+ verifier weights") is the same for every chain
  Strobe t0;
  const char *lbl = "Bulletproofs+ verifier weights";
  merlin_new(t0, (const uint8_t *)lbl, (uint32_t)strlen(lbl));
  MultiStrobe<W> s;
  for (int i = 0; i < 25; i++)
    for (int w = 0; w < W; w++) s.st[i][w] = t0.st[i];
  s.pos = t0.pos;
  s.pos_begin = t0.pos_begin;
  s.cur_flags = t0.cur_flags;
  uint8_t len4[4];
  for (size_t i = 0; i < n; i++) {  // append_message(b"proof", bytes)
    ms_begin_op<W>(s, BPP_FLAG_M | BPP_FLAG_A, false);
    ms_absorb_same<W>(s, (const uint8_t *)"proof", 5);
    u32le(len4, 32);
    ms_absorb_same<W>(s, len4, 4);  // meta_ad(len, more = true)
    ms_begin_op<W>(s, BPP_FLAG_A, false);
    ms_absorb<W>(s, rng, 32 * i, 32);
  }
  // build_rng().finalize(&mut NullRng): meta_ad(b"rng"), key(32 zero bytes)
  ms_begin_op<W>(s, BPP_FLAG_M | BPP_FLAG_A, false);
  ms_absorb_same<W>(s, (const uint8_t *)"rng", 3);
  ms_begin_op<W>(s, BPP_FLAG_A | BPP_FLAG_C, false);
  for (uint32_t i = 0; i < 32; i++) {  // overwrite with zeros
    const uint32_t sh = 8 * (s.pos & 7);
    for (int w = 0; w < W; w++) s.st[s.pos >> 3][w] &= ~(0xffULL << sh);
    s.pos++;
    if (s.pos == BPP_STROBE_R) ms_run_f(s);
  }
  uint8_t wide[W][64];
  uint8_t *wp[W];
  for (int w = 0; w < W; w++) wp[w] = wide[w];
  for (size_t i = 0; i < n; i++) {  // Scalar::random: fill_bytes(64) = meta_ad(u32le(64)) + prf(64)
    ms_begin_op<W>(s, BPP_FLAG_M | BPP_FLAG_A, false);
    u32le(len4, 64);
    ms_absorb_same<W>(s, len4, 4);
    ms_begin_op<W>(s, BPP_FLAG_I | BPP_FLAG_A | BPP_FLAG_C, false);
    ms_squeeze<W>(s, wp, 64);
    for (int w = 0; w < W; w++) {
      sc v;
      sc_mont_from_wide(v, wide[w]);
      // random_not_zero: a zero draw (probability 2^-252) would desynchronise the lockstep; flagged to the caller
      sc_from_mont(v, v);
      sc_store_words(out[w] + 32 * i, v);
    }
  }
}

__attribute__((target("avx512f,avx512vl,avx512bw,avx512dq"))) static void weights_chain_x8(const uint8_t *const rng[8], size_t n,
                                                                                          uint8_t *const out[8]) {
  weights_chain_multi_impl<8>(rng, n, out);
}
__attribute__((target("avx2"))) static void weights_chain_x4(const uint8_t *const rng[4], size_t n, uint8_t *const out[4]) {
  weights_chain_multi_impl<4>(rng, n, out);
}

static inline bool weight_is_zero(const uint8_t *w32) {
  uint8_t r = 0;
  for (int i = 0; i < 32; i++) r |= w32[i];
  return r == 0;
}

}  // namespace bpp
