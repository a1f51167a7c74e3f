// Host side of the batch-weight transcript (src/range_proof.rs:811,849,853,894), W independent chains in lockstep.
//
// One chain is a strictly sequential sponge (1.27 Keccak-f per proof), so a single chain cannot be parallelised --
// but the chains of DIFFERENT reference batches (chunks / groups) have identical control flow (same message lengths),
// so W of them share one instruction stream: the STROBE state is kept lane-transposed (st[i][w]) and Keccak-f runs on
// W-lane vectors (AVX-512: 8 chains, AVX2: 4, scalar: 1; chosen at run time).  Host only.
//
// A SINGLE chain (one reference batch per call, or fewer chains than workers) runs on `weights_chain_single`: the sponge
// state is addressed as bytes and absorbed eight at a time, squeeze / overwrite use the fixed words their forced
// permutation leaves them at, and Keccak-f is a fully unrolled 64-bit form compiled for BMI (andn, rorx) when the CPU
// has it: 0.27 instead of 0.39 us per proof on the GPU boxes' EPYC 9575F (merlin.h's byte-wise sponge, which stays the
// device / test form).  A plane-per-register AVX-512 Keccak-f for one state was measured and dropped: its five
// cross-lane permutes per round are a 1 130-cycle dependency chain on Zen 5 (0.227 us against 0.166 us scalar).
#pragma once
#include <string.h>

#include "merlin.h"
#include "scalar.h"

namespace bpp {

// the inner loops of the vector Keccak-f must be unrolled completely: rolled, a[] / b[] live in memory and are indexed
// through the PI / ROT tables (1 800 64-byte loads and stores per permutation: 2 us per 8-way permutation instead of 0.2)
#if defined(__clang__)
#define BPP_UNROLL_FULL _Pragma("clang loop unroll(full)")
#else
#define BPP_UNROLL_FULL _Pragma("GCC unroll 32")
#endif

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
  constexpr int ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
  constexpr int PI[25] = {0, 10, 20, 5, 15, 16, 1, 11, 21, 6, 7, 17, 2, 12, 22, 23, 8, 18, 3, 13, 14, 24, 9, 19, 4};
#define BPP_VROL(x, n) (((x) << (n)) | ((x) >> (64 - (n))))
  for (int rnd = 0; rnd < 24; rnd++) {
    V c[5], d[5], b[25];
    BPP_UNROLL_FULL
    for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
    BPP_UNROLL_FULL
    for (int x = 0; x < 5; x++) d[x] = c[(x + 4) % 5] ^ BPP_VROL(c[(x + 1) % 5], 1);
    BPP_UNROLL_FULL
    for (int i = 0; i < 25; i++) {
      V t = a[i] ^ d[i % 5];
      b[PI[i]] = ROT[i] ? BPP_VROL(t, ROT[i]) : t;
    }
    BPP_UNROLL_FULL
    for (int y = 0; y < 25; y += 5) {
      BPP_UNROLL_FULL
      for (int x = 0; x < 5; x++) a[y + x] = b[y + x] ^ (~b[y + (x + 1) % 5] & b[y + (x + 2) % 5]);
    }
    V rc;
    BPP_UNROLL_FULL
    for (int w = 0; w < W; w++) rc[w] = RC[rnd];
    a[0] ^= rc;
  }
#undef BPP_VROL
}

// ---- Scalar::from_bytes_mod_order_wide on the HOST: 64 little-endian bytes -> canonical 32 bytes of (value mod l).
// The device form (scalar.h: 29-bit limbs, 64-bit columns) is shaped for v_mad_u64_u32; compiled for x86-64 it costs ~400 ns
// per weight and was two thirds of a lock-step bundle's time.  Here: w = lo + hi 2^256 with two Montgomery products on
// four 64-bit limbs (R = 2^256, unsigned __int128): montmul(lo, R mod l) = lo mod l, montmul(hi, R^2 mod l) = hi 2^256 mod l.
struct HostScalar64 {
  static constexpr uint64_t L[4] = {0x5812631a5cf5d3edULL, 0x14def9dea2f79cd6ULL, 0x0ULL, 0x1000000000000000ULL};
  static constexpr uint64_t R1[4] = {0xd6ec31748d98951dULL, 0xc6ef5bf4737dcf70ULL, 0xfffffffffffffffeULL, 0x0fffffffffffffffULL};
  static constexpr uint64_t R2[4] = {0xa40611e3449c0f01ULL, 0xd00e1ba768859347ULL, 0xceec73d217f5be65ULL, 0x0399411b7c309a3dULL};
  static constexpr uint64_t N0 = 0xd2b51da312547e1bULL;  // -l^-1 mod 2^64
};
// r = a b 2^-256 mod l, fully reduced; a any 256-bit value, b < l  (CIOS, four limbs)
static inline void host_montmul64(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
  typedef unsigned __int128 u128;
  uint64_t t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    u128 c = 0;
    for (int j = 0; j < 4; j++) {
      c += (u128)a[j] * b[i] + t[j];
      t[j] = (uint64_t)c;
      c >>= 64;
    }
    c += t[4];
    t[4] = (uint64_t)c;
    t[5] = (uint64_t)(c >> 64);
    const uint64_t m = t[0] * HostScalar64::N0;
    c = (u128)m * HostScalar64::L[0] + t[0];
    c >>= 64;
    for (int j = 1; j < 4; j++) {
      c += (u128)m * HostScalar64::L[j] + t[j];
      t[j - 1] = (uint64_t)c;
      c >>= 64;
    }
    c += t[4];
    t[3] = (uint64_t)c;
    t[4] = t[5] + (uint64_t)(c >> 64);
  }
  // t < 2l: one conditional subtraction
  uint64_t d[4];
  unsigned __int128 br = 0;
  for (int j = 0; j < 4; j++) {
    const unsigned __int128 x = (unsigned __int128)t[j] - HostScalar64::L[j] - (uint64_t)br;
    d[j] = (uint64_t)x;
    br = (x >> 64) & 1;
  }
  const bool take = t[4] != 0 || br == 0;
  for (int j = 0; j < 4; j++) r[j] = take ? d[j] : t[j];
}
static inline void host_wide_reduce(uint8_t out32[32], const uint8_t wide64[64]) {
  uint64_t lo[4], hi[4], a[4], b[4];
  memcpy(lo, wide64, 32);  // little-endian host
  memcpy(hi, wide64 + 32, 32);
  host_montmul64(a, lo, HostScalar64::R1);
  host_montmul64(b, hi, HostScalar64::R2);
  // a + b mod l (both < l)
  typedef unsigned __int128 u128;
  uint64_t s4[4], d[4];
  u128 c = 0;
  for (int j = 0; j < 4; j++) {
    c += (u128)a[j] + b[j];
    s4[j] = (uint64_t)c;
    c >>= 64;
  }
  u128 br = 0;
  for (int j = 0; j < 4; j++) {
    const u128 x = (u128)s4[j] - HostScalar64::L[j] - (uint64_t)br;
    d[j] = (uint64_t)x;
    br = (x >> 64) & 1;
  }
  const bool take = (uint64_t)c != 0 || br == 0;
  for (int j = 0; j < 4; j++) s4[j] = take ? d[j] : s4[j];
  memcpy(out32, s4, 32);
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
// XOR up to 8 bytes per chain (little-endian words in c, `nb` of them significant, the rest zero) into the state at byte
// position pos; the caller guarantees pos + nb <= R.  One or two vector XORs instead of nb x W byte operations.
template <int W>
static inline __attribute__((always_inline)) void ms_xor_word(MultiStrobe<W> &s, typename VecOps<W>::vec c, uint32_t pos) {
  const uint32_t q = pos >> 3, sh = 8 * (pos & 7);
  s.st[q] ^= c << sh;
  if (sh) s.st[q + 1] ^= c >> (64 - sh);  // st has 25 words, R + 8 <= 200: q + 1 <= 21
}
static inline __attribute__((always_inline)) uint64_t ms_load_le(const uint8_t *p, uint32_t nb) {
  uint64_t v = 0;
  if (nb == 8) memcpy(&v, p, 8);  // little-endian host; the common case as ONE load (a run-time length is a library call)
  else memcpy(&v, p, nb);
  return v;
}
// absorb the same-length message of every chain (data[w] + off)
template <int W>
static inline __attribute__((always_inline)) void ms_absorb(MultiStrobe<W> &s, const uint8_t *const data[W], size_t off, uint32_t n) {
  typedef typename VecOps<W>::vec V;
  uint32_t i = 0;
  while (i < n) {
    const uint32_t room = BPP_STROBE_R - s.pos, nb = (n - i < 8u ? n - i : 8u) < room ? (n - i < 8u ? n - i : 8u) : room;
    V c;
    for (int w = 0; w < W; w++) c[w] = ms_load_le(data[w] + off + i, nb);
    ms_xor_word<W>(s, c, s.pos);
    s.pos += nb;
    i += nb;
    if (s.pos == BPP_STROBE_R) ms_run_f(s);
  }
}
template <int W>
static inline __attribute__((always_inline)) void ms_absorb_same(MultiStrobe<W> &s, const uint8_t *d, uint32_t n) {
  typedef typename VecOps<W>::vec V;
  uint32_t i = 0;
  while (i < n) {
    const uint32_t room = BPP_STROBE_R - s.pos, nb = (n - i < 8u ? n - i : 8u) < room ? (n - i < 8u ? n - i : 8u) : room;
    const uint64_t v = ms_load_le(d + i, nb);
    V c;
    for (int w = 0; w < W; w++) c[w] = v;
    ms_xor_word<W>(s, c, s.pos);
    s.pos += nb;
    i += nb;
    if (s.pos == BPP_STROBE_R) ms_run_f(s);
  }
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
// PRF output: copy n state bytes out per chain and zero them in the state (STROBE's squeeze), 8 bytes at a time
template <int W>
static inline __attribute__((always_inline)) void ms_squeeze(MultiStrobe<W> &s, uint8_t *const out[W], uint32_t n) {
  typedef typename VecOps<W>::vec V;
  uint32_t i = 0;
  while (i < n) {
    const uint32_t room = BPP_STROBE_R - s.pos, nb = (n - i < 8u ? n - i : 8u) < room ? (n - i < 8u ? n - i : 8u) : room;
    const uint32_t q = s.pos >> 3, sh = 8 * (s.pos & 7);
    const uint64_t mask = nb == 8 ? ~0ULL : ((1ULL << (8 * nb)) - 1ULL);
    V v = s.st[q] >> sh;
    if (sh) v |= s.st[q + 1] << (64 - sh);
    for (int w = 0; w < W; w++) {
      const uint64_t x = v[w] & mask;
      if (nb == 8) memcpy(out[w] + i, &x, 8);
      else memcpy(out[w] + i, &x, nb);
    }
    // zero the bytes just read
    V m0, m1;
    for (int w = 0; w < W; w++) {
      m0[w] = ~(mask << sh);
      m1[w] = sh ? ~(mask >> (64 - sh)) : ~0ULL;
    }
    s.st[q] &= m0;
    if (sh) s.st[q + 1] &= m1;
    s.pos += nb;
    i += nb;
    if (s.pos == BPP_STROBE_R) ms_run_f(s);
  }
}

// W chains: rng[w] -> n x 32 transcript-RNG bytes, out[w] -> n x 32 canonical weights.
// WIDE: out[w] -> n x 64 bytes as the sponge leaves them; Scalar::from_bytes_mod_order_wide -- half of a bundle's CPU time, and
// perfectly parallel -- is then the device's (chain_dev.h: k_chain_finish_bytes), and so is the look for a zero weight.
template <int W, bool WIDE = false>
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
  {  // key(32 zero bytes): overwrite with zeros = squeeze and discard
    uint8_t sink[W][32];
    uint8_t *sp[W];
    for (int w = 0; w < W; w++) sp[w] = sink[w];
    ms_squeeze<W>(s, sp, 32);
  }
  uint8_t wide[W][64];
  uint8_t *wp[W];
  for (int w = 0; w < W; w++) wp[w] = wide[w];
  for (size_t i = 0; i < n; i++) {  // Scalar::random: fill_bytes(64) = meta_ad(u32le(64)) + prf(64)
    ms_begin_op<W>(s, BPP_FLAG_M | BPP_FLAG_A, false);
    u32le(len4, 64);
    ms_absorb_same<W>(s, len4, 4);
    ms_begin_op<W>(s, BPP_FLAG_I | BPP_FLAG_A | BPP_FLAG_C, false);
    if (WIDE) {
      uint8_t *dst[W];
      for (int w = 0; w < W; w++) dst[w] = out[w] + 64 * i;
      ms_squeeze<W>(s, dst, 64);
      continue;
    }
    ms_squeeze<W>(s, wp, 64);
    // random_not_zero: a zero draw (probability 2^-252) would desynchronise the lockstep; flagged to the caller
    for (int w = 0; w < W; w++) host_wide_reduce(out[w] + 32 * i, wide[w]);
  }
}

__attribute__((target("avx512f,avx512vl,avx512bw,avx512dq"))) static void weights_chain_x8(const uint8_t *const rng[8], size_t n,
                                                                                          uint8_t *const out[8]) {
  weights_chain_multi_impl<8>(rng, n, out);
}
__attribute__((target("avx2"))) static void weights_chain_x4(const uint8_t *const rng[4], size_t n, uint8_t *const out[4]) {
  weights_chain_multi_impl<4>(rng, n, out);
}
__attribute__((target("avx512f,avx512vl,avx512bw,avx512dq"))) static void wide_chain_x8(const uint8_t *const rng[8], size_t n,
                                                                                       uint8_t *const out[8]) {
  weights_chain_multi_impl<8, true>(rng, n, out);
}
__attribute__((target("avx2"))) static void wide_chain_x4(const uint8_t *const rng[4], size_t n, uint8_t *const out[4]) {
  weights_chain_multi_impl<4, true>(rng, n, out);
}

static inline bool weight_is_zero(const uint8_t *w32) {
  uint8_t r = 0;
  for (int i = 0; i < 32; i++) r |= w32[i];
  return r == 0;
}

// ---- one chain, low latency ------------------------------------------------------------------------------------------
// Keccak-f[1600] on 64-bit registers, two rounds per step between two sets of lanes (no copies), all 24 rounds unrolled.
#define BPP_KROL(x, n) (((x) << (n)) | ((x) >> (64 - (n))))
#define BPP_KROUND(A, E, rc)                                                                                                  \
  {                                                                                                                           \
    const uint64_t c0 = A[0] ^ A[5] ^ A[10] ^ A[15] ^ A[20], c1 = A[1] ^ A[6] ^ A[11] ^ A[16] ^ A[21],                        \
                   c2 = A[2] ^ A[7] ^ A[12] ^ A[17] ^ A[22], c3 = A[3] ^ A[8] ^ A[13] ^ A[18] ^ A[23],                        \
                   c4 = A[4] ^ A[9] ^ A[14] ^ A[19] ^ A[24];                                                                  \
    const uint64_t d0 = c4 ^ BPP_KROL(c1, 1), d1 = c0 ^ BPP_KROL(c2, 1), d2 = c1 ^ BPP_KROL(c3, 1),                           \
                   d3 = c2 ^ BPP_KROL(c4, 1), d4 = c3 ^ BPP_KROL(c0, 1);                                                      \
    uint64_t b0, b1, b2, b3, b4;                                                                                              \
    b0 = A[0] ^ d0; b1 = BPP_KROL(A[6] ^ d1, 44); b2 = BPP_KROL(A[12] ^ d2, 43); b3 = BPP_KROL(A[18] ^ d3, 21);               \
    b4 = BPP_KROL(A[24] ^ d4, 14);                                                                                            \
    E[0] = b0 ^ (~b1 & b2) ^ rc; E[1] = b1 ^ (~b2 & b3); E[2] = b2 ^ (~b3 & b4); E[3] = b3 ^ (~b4 & b0);                      \
    E[4] = b4 ^ (~b0 & b1);                                                                                                   \
    b0 = BPP_KROL(A[3] ^ d3, 28); b1 = BPP_KROL(A[9] ^ d4, 20); b2 = BPP_KROL(A[10] ^ d0, 3); b3 = BPP_KROL(A[16] ^ d1, 45);  \
    b4 = BPP_KROL(A[22] ^ d2, 61);                                                                                            \
    E[5] = b0 ^ (~b1 & b2); E[6] = b1 ^ (~b2 & b3); E[7] = b2 ^ (~b3 & b4); E[8] = b3 ^ (~b4 & b0); E[9] = b4 ^ (~b0 & b1);   \
    b0 = BPP_KROL(A[1] ^ d1, 1); b1 = BPP_KROL(A[7] ^ d2, 6); b2 = BPP_KROL(A[13] ^ d3, 25); b3 = BPP_KROL(A[19] ^ d4, 8);    \
    b4 = BPP_KROL(A[20] ^ d0, 18);                                                                                            \
    E[10] = b0 ^ (~b1 & b2); E[11] = b1 ^ (~b2 & b3); E[12] = b2 ^ (~b3 & b4); E[13] = b3 ^ (~b4 & b0);                       \
    E[14] = b4 ^ (~b0 & b1);                                                                                                  \
    b0 = BPP_KROL(A[4] ^ d4, 27); b1 = BPP_KROL(A[5] ^ d0, 36); b2 = BPP_KROL(A[11] ^ d1, 10); b3 = BPP_KROL(A[17] ^ d2, 15); \
    b4 = BPP_KROL(A[23] ^ d3, 56);                                                                                            \
    E[15] = b0 ^ (~b1 & b2); E[16] = b1 ^ (~b2 & b3); E[17] = b2 ^ (~b3 & b4); E[18] = b3 ^ (~b4 & b0);                       \
    E[19] = b4 ^ (~b0 & b1);                                                                                                  \
    b0 = BPP_KROL(A[2] ^ d2, 62); b1 = BPP_KROL(A[8] ^ d3, 55); b2 = BPP_KROL(A[14] ^ d4, 39); b3 = BPP_KROL(A[15] ^ d0, 41); \
    b4 = BPP_KROL(A[21] ^ d1, 2);                                                                                             \
    E[20] = b0 ^ (~b1 & b2); E[21] = b1 ^ (~b2 & b3); E[22] = b2 ^ (~b3 & b4); E[23] = b3 ^ (~b4 & b0);                       \
    E[24] = b4 ^ (~b0 & b1);                                                                                                  \
  }
static inline __attribute__((always_inline)) void keccak_f1600_host_body(uint64_t st[25]) {
  static const uint64_t RC[24] = {
      0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808AULL, 0x8000000080008000ULL,
      0x000000000000808BULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
      0x000000000000008AULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000AULL,
      0x000000008000808BULL, 0x800000000000008BULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
      0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800AULL, 0x800000008000000AULL,
      0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
  uint64_t A[25], E[25];
  for (int i = 0; i < 25; i++) A[i] = st[i];
  BPP_UNROLL_FULL
  for (int r = 0; r < 24; r += 2) {
    BPP_KROUND(A, E, RC[r]);
    BPP_KROUND(E, A, RC[r + 1]);
  }
  for (int i = 0; i < 25; i++) st[i] = A[i];
}
#undef BPP_KROUND
#undef BPP_KROL
// real functions (not inlined into every absorb site: 20 KB of code each)
__attribute__((noinline)) static void keccak_f1600_host_plain(uint64_t st[25]) { keccak_f1600_host_body(st); }
__attribute__((noinline, target("bmi,bmi2"))) static void keccak_f1600_host_bmi(uint64_t st[25]) { keccak_f1600_host_body(st); }

// STROBE-128 with the state addressed as bytes (little-endian host)
struct FastStrobe {
  alignas(64) uint64_t st[25];
  uint32_t pos, pos_begin;
};
static inline __attribute__((always_inline)) void fs_xor(uint8_t *p, const uint8_t *d, uint32_t k) {
  while (k >= 8) {
    uint64_t a, b;
    memcpy(&a, p, 8);
    memcpy(&b, d, 8);
    a ^= b;
    memcpy(p, &a, 8);
    p += 8, d += 8, k -= 8;
  }
  if (k & 4) {
    uint32_t a, b;
    memcpy(&a, p, 4);
    memcpy(&b, d, 4);
    a ^= b;
    memcpy(p, &a, 4);
    p += 4, d += 4;
  }
  if (k & 2) {
    uint16_t a, b;
    memcpy(&a, p, 2);
    memcpy(&b, d, 2);
    a ^= b;
    memcpy(p, &a, 2);
    p += 2, d += 2;
  }
  if (k & 1) *p ^= *d;
}
template <void (*PERM)(uint64_t *)>
static inline __attribute__((always_inline)) void fs_run_f(FastStrobe &s) {
  uint8_t *b = (uint8_t *)s.st;
  b[s.pos] ^= (uint8_t)s.pos_begin;
  b[s.pos + 1] ^= 0x04;
  b[BPP_STROBE_R + 1] ^= 0x80;
  PERM(s.st);
  s.pos = 0;
  s.pos_begin = 0;
}
template <void (*PERM)(uint64_t *)>
static inline __attribute__((always_inline)) void fs_absorb(FastStrobe &s, const uint8_t *d, uint32_t n) {
  while (n) {
    const uint32_t room = BPP_STROBE_R - s.pos, k = n < room ? n : room;
    fs_xor((uint8_t *)s.st + s.pos, d, k);
    s.pos += k, d += k, n -= k;
    if (s.pos == BPP_STROBE_R) fs_run_f<PERM>(s);
  }
}
// a message of compile-time length that does not reach the end of the block: a handful of fixed-size loads and stores
template <void (*PERM)(uint64_t *), uint32_t N>
static inline __attribute__((always_inline)) void fs_absorb_fixed(FastStrobe &s, const uint8_t *d) {
  if (s.pos + N < BPP_STROBE_R) {
    fs_xor((uint8_t *)s.st + s.pos, d, N);
    s.pos += N;
  } else {
    fs_absorb<PERM>(s, d, N);
  }
}
template <void (*PERM)(uint64_t *)>
static inline __attribute__((always_inline)) void fs_begin_op(FastStrobe &s, uint32_t flags) {
  const uint8_t hdr[2] = {(uint8_t)s.pos_begin, (uint8_t)flags};
  s.pos_begin = s.pos + 1;
  fs_absorb_fixed<PERM, 2>(s, hdr);
  if ((flags & (BPP_FLAG_C | BPP_FLAG_K)) != 0 && s.pos != 0) fs_run_f<PERM>(s);
}
// rng: n x 32 transcript-RNG bytes -> out: n x 32 canonical non-zero weights (src/range_proof.rs:811,849,853,894)
template <void (*PERM)(uint64_t *), bool WIDE = false>
static inline __attribute__((always_inline)) void weights_chain_single_impl(const uint8_t *rng, size_t n, uint8_t *out) {
  Strobe t0;
  const char *lbl = "Bulletproofs+ verifier weights";
  merlin_new(t0, (const uint8_t *)lbl, (uint32_t)strlen(lbl));
  FastStrobe s;
  for (int i = 0; i < 25; i++) s.st[i] = t0.st[i];
  s.pos = t0.pos;
  s.pos_begin = t0.pos_begin;
  static const uint8_t proof_len[9] = {'p', 'r', 'o', 'o', 'f', 32, 0, 0, 0};  // label, then meta_ad(u32le(32), more = true)
  for (size_t i = 0; i < n; i++) {  // append_message(b"proof", bytes)
    fs_begin_op<PERM>(s, BPP_FLAG_M | BPP_FLAG_A);
    fs_absorb_fixed<PERM, 9>(s, proof_len);
    fs_begin_op<PERM>(s, BPP_FLAG_A);
    fs_absorb_fixed<PERM, 32>(s, rng + 32 * i);
  }
  // build_rng().finalize(&mut NullRng): meta_ad(b"rng"), key(32 zero bytes) = overwrite from position 0
  fs_begin_op<PERM>(s, BPP_FLAG_M | BPP_FLAG_A);
  fs_absorb<PERM>(s, (const uint8_t *)"rng", 3);
  fs_begin_op<PERM>(s, BPP_FLAG_A | BPP_FLAG_C);
  s.st[0] = s.st[1] = s.st[2] = s.st[3] = 0;
  s.pos = 32;
  static const uint8_t len64[4] = {64, 0, 0, 0};
  for (size_t i = 0; i < n; i++) {
    if (WIDE) {  // one draw per proof, 64 bytes out as they are (the device reduces them and looks for a zero)
      fs_begin_op<PERM>(s, BPP_FLAG_M | BPP_FLAG_A);
      fs_absorb_fixed<PERM, 4>(s, len64);
      fs_begin_op<PERM>(s, BPP_FLAG_I | BPP_FLAG_A | BPP_FLAG_C);
      memcpy(out + 64 * i, s.st, 64);
      for (int j = 0; j < 8; j++) s.st[j] = 0;
      s.pos = 64;
      continue;
    }
    uint8_t *w = out + 32 * i;
    do {  // Scalar::random_not_zero (src/protocols/scalar_protocol.rs:23-30): fill_bytes(64) = meta_ad(u32le(64)) + prf(64)
      fs_begin_op<PERM>(s, BPP_FLAG_M | BPP_FLAG_A);
      fs_absorb_fixed<PERM, 4>(s, len64);
      fs_begin_op<PERM>(s, BPP_FLAG_I | BPP_FLAG_A | BPP_FLAG_C);  // forces a permutation: the squeeze starts at position 0
      uint8_t wide[64];
      memcpy(wide, s.st, 64);
      for (int j = 0; j < 8; j++) s.st[j] = 0;
      s.pos = 64;
      host_wide_reduce(w, wide);
    } while (weight_is_zero(w));
  }
}
static void weights_chain_single_plain(const uint8_t *rng, size_t n, uint8_t *out) {
  weights_chain_single_impl<keccak_f1600_host_plain>(rng, n, out);
}
__attribute__((target("bmi,bmi2"))) static void weights_chain_single_bmi(const uint8_t *rng, size_t n, uint8_t *out) {
  weights_chain_single_impl<keccak_f1600_host_bmi>(rng, n, out);
}
static inline void weights_chain_single(const uint8_t *rng, size_t n, uint8_t *out) {
  static const bool bmi = __builtin_cpu_supports("bmi") && __builtin_cpu_supports("bmi2");
  if (bmi) weights_chain_single_bmi(rng, n, out);
  else weights_chain_single_plain(rng, n, out);
}
static void wide_chain_single_plain(const uint8_t *rng, size_t n, uint8_t *out) {
  weights_chain_single_impl<keccak_f1600_host_plain, true>(rng, n, out);
}
__attribute__((target("bmi,bmi2"))) static void wide_chain_single_bmi(const uint8_t *rng, size_t n, uint8_t *out) {
  weights_chain_single_impl<keccak_f1600_host_bmi, true>(rng, n, out);
}
// rng: n x 32 bytes -> out: n x 64 PRF bytes (one draw per proof, not reduced)
static inline void wide_chain_single(const uint8_t *rng, size_t n, uint8_t *out) {
  static const bool bmi = __builtin_cpu_supports("bmi") && __builtin_cpu_supports("bmi2");
  if (bmi) wide_chain_single_bmi(rng, n, out);
  else wide_chain_single_plain(rng, n, out);
}
// the same chain on merlin.h's generic byte-wise sponge (the form the device kernels and the oracle tests are written
// against): kept as the cross-check of the fast form (hosttest.cpp)
static inline void weights_chain_generic(const uint8_t *rng32, size_t n, uint8_t *weights32) {
  Strobe wt;
  const char *lbl = "Bulletproofs+ verifier weights";
  merlin_new(wt, (const uint8_t *)lbl, (uint32_t)strlen(lbl));
  for (size_t i = 0; i < n; i++) merlin_append_message(wt, (const uint8_t *)"proof", 5, rng32 + 32 * i, 32);
  uint8_t zero32[32] = {0};
  merlin_rng_finalize(wt, zero32);  // build_rng().finalize(&mut NullRng)
  for (size_t i = 0; i < n; i++) {
    uint8_t *w = weights32 + 32 * i;
    do {
      uint8_t wide[64];
      merlin_rng_fill(wt, wide, 64);
      host_wide_reduce(w, wide);
    } while (weight_is_zero(w));
  }
}

}  // namespace bpp
