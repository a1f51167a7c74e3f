// STROBE-128 / Merlin for ONE LANE PER TRANSCRIPT with the sponge state in LDS (device only).
//
// merlin.h's Strobe keeps the 200 state bytes in a struct and absorbs byte by byte at a run-time position: on the device the
// dynamically indexed state ends up in scratch memory, every absorbed byte becomes a dependent read-modify-write through
// memory and every inlined absorb loop carries its own copy of Keccak-f (k_transcripts: 463 KB of code, 0.75 ms for 117 M
// instructions at one wavefront per SIMD).  Here the state of lane l is 50 LDS words, word w at st[w * 64] (word-major: the 64
// lanes of a wavefront hit 64 different banks for the same w); absorbing is `ds_xor_b32` (no return value, nothing to wait
// for), four bytes at a time; Keccak-f is ONE non-inlined function that loads the 50 words, runs the 24 rounds in registers
// and stores them back.  Squeeze and overwrite only ever start at position 0 (their begin_op forces a permutation), so they
// touch fixed words.  Byte-for-byte the same duplex as merlin.h (tests compare challenges and transcript-RNG bytes with it).
//
// Replaces (reference boundary): merlin::Transcript / TranscriptRng as driven by src/transcripts.rs:59-200 and
// src/protocols/transcript_protocol.rs:39-79, for PASS 1 of the verifier (src/range_proof.rs:816-850).
#pragma once
#include "merlin.h"

namespace bpp {

#define BPP_LS_WORDS 50u
#define BPP_LS_STRIDE 64u  // lanes per workgroup of the kernels that use it

typedef __attribute__((address_space(3))) uint32_t lds_u32;

struct LStrobe {
  lds_u32 *st;  // this lane's word 0
  uint32_t pos, pos_begin;
};

__device__ __forceinline__ void ls_xor_word(const LStrobe &s, uint32_t w, uint32_t v) {
  (void)__hip_atomic_fetch_xor(s.st + w * BPP_LS_STRIDE, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

// Keccak-f[1600] on the LDS-resident state of the calling lane.  One copy in the code object.
__device__ __attribute__((noinline)) void ls_keccak(lds_u32 *st) {
  uint64_t a[25];
#pragma unroll
  for (int i = 0; i < 25; i++) a[i] = (uint64_t)st[(2 * i) * BPP_LS_STRIDE] | ((uint64_t)st[(2 * i + 1) * BPP_LS_STRIDE] << 32);
  keccak_f1600(a);
#pragma unroll
  for (int i = 0; i < 25; i++) {
    st[(2 * i) * BPP_LS_STRIDE] = (uint32_t)a[i];
    st[(2 * i + 1) * BPP_LS_STRIDE] = (uint32_t)(a[i] >> 32);
  }
}

__device__ __forceinline__ void ls_run_f(LStrobe &s) {
  // pos_begin at byte pos, 0x04 at pos + 1, 0x80 at byte R + 1 = 167 (word 41, byte 3)
  const uint32_t v = (s.pos_begin & 0xffu) | (0x04u << 8);  // two adjacent bytes, may straddle a word
  const uint32_t w = s.pos >> 2, sh = 8u * (s.pos & 3u);
  const uint64_t t = (uint64_t)v << sh;
  ls_xor_word(s, w, (uint32_t)t);
  if (sh == 24u) ls_xor_word(s, w + 1u, (uint32_t)(t >> 32));  // pos + 1 <= 167: inside the state
  ls_xor_word(s, (BPP_STROBE_R + 1u) >> 2, 0x80u << 24);
  ls_keccak(s.st);
  s.pos = 0;
  s.pos_begin = 0;
}

// up to 8 bytes held in a register (labels, lengths, u64 values), little-endian byte order
__device__ __forceinline__ void ls_absorb_reg(LStrobe &s, uint64_t v, uint32_t n) {
  for (uint32_t i = 0; i < n; i++) {
    ls_xor_word(s, s.pos >> 2, (uint32_t)((v >> (8u * i)) & 0xffu) << (8u * (s.pos & 3u)));
    s.pos++;
    if (s.pos == BPP_STROBE_R) ls_run_f(s);
  }
}

// n bytes from memory (any alignment), four at a time while they fit into the block
__device__ __forceinline__ void ls_absorb_mem(LStrobe &s, const uint8_t *p, uint32_t n) {
  uint32_t i = 0;
  while (i < n) {
    if (n - i >= 4u && s.pos + 4u <= BPP_STROBE_R) {
      uint32_t d;
      __builtin_memcpy(&d, p + i, 4);
      const uint64_t t = (uint64_t)d << (8u * (s.pos & 3u));
      ls_xor_word(s, s.pos >> 2, (uint32_t)t);
      ls_xor_word(s, (s.pos >> 2) + 1u, (uint32_t)(t >> 32));  // zero when pos is word-aligned; word <= 41
      s.pos += 4u;
      i += 4u;
    } else {
      ls_xor_word(s, s.pos >> 2, (uint32_t)p[i] << (8u * (s.pos & 3u)));
      s.pos++;
      i++;
    }
    if (s.pos == BPP_STROBE_R) ls_run_f(s);
  }
}

__device__ __forceinline__ void ls_begin_op(LStrobe &s, uint32_t flags) {
  const uint32_t old_begin = s.pos_begin;
  s.pos_begin = s.pos + 1u;
  ls_absorb_reg(s, (uint64_t)(old_begin & 0xffu) | ((uint64_t)(flags & 0xffu) << 8), 2);
  if ((flags & (BPP_FLAG_C | BPP_FLAG_K)) != 0 && s.pos != 0) ls_run_f(s);
}

// PRF / KEY start at position 0 (ls_begin_op just forced a permutation): fixed words
template <int WORDS>
__device__ __forceinline__ void ls_squeeze_words(LStrobe &s, uint32_t out[WORDS]) {
#pragma unroll
  for (int i = 0; i < WORDS; i++) {
    out[i] = s.st[i * BPP_LS_STRIDE];
    s.st[i * BPP_LS_STRIDE] = 0;
  }
  s.pos = 4u * WORDS;
}
__device__ __forceinline__ void ls_overwrite_zero32(LStrobe &s) {
#pragma unroll
  for (int i = 0; i < 8; i++) s.st[i * BPP_LS_STRIDE] = 0;
  s.pos = 32;
}

// 203-byte wire form of merlin.h's Strobe -> LDS
__device__ __forceinline__ void ls_from_bytes(LStrobe &s, lds_u32 *lane_base, const uint8_t *b) {
  s.st = lane_base;
  for (uint32_t i = 0; i < BPP_LS_WORDS; i++) {
    uint32_t d;
    __builtin_memcpy(&d, b + 4u * i, 4);
    s.st[i * BPP_LS_STRIDE] = d;
  }
  s.pos = b[200];
  s.pos_begin = b[201];
}

// ---- Merlin on top: labels arrive as little-endian register images (at most 8 bytes per piece)
__device__ __forceinline__ void lm_label_len(LStrobe &s, uint64_t label, uint32_t llen, uint32_t mlen) {
  ls_begin_op(s, BPP_FLAG_M | BPP_FLAG_A);
  ls_absorb_reg(s, label, llen);
  ls_absorb_reg(s, (uint64_t)mlen, 4);  // meta_ad(len, more = true): no new header
}
// a label of 17..24 characters in three pieces
__device__ __forceinline__ void lm_append_u64_long(LStrobe &s, uint64_t l0, uint64_t l1, uint64_t l2, uint32_t llen, uint64_t x) {
  ls_begin_op(s, BPP_FLAG_M | BPP_FLAG_A);
  ls_absorb_reg(s, l0, 8);
  ls_absorb_reg(s, l1, 8);
  ls_absorb_reg(s, l2, llen - 16u);
  ls_absorb_reg(s, 8ull, 4);
  ls_begin_op(s, BPP_FLAG_A);
  ls_absorb_reg(s, x, 8);
}
__device__ __forceinline__ void lm_append_mem(LStrobe &s, uint64_t label, uint32_t llen, const uint8_t *msg, uint32_t mlen) {
  lm_label_len(s, label, llen, mlen);
  ls_begin_op(s, BPP_FLAG_A);
  ls_absorb_mem(s, msg, mlen);
}
__device__ __forceinline__ void lm_append_u64(LStrobe &s, uint64_t label, uint32_t llen, uint64_t x) {
  lm_label_len(s, label, llen, 8);
  ls_begin_op(s, BPP_FLAG_A);
  ls_absorb_reg(s, x, 8);
}
// challenge_bytes(label, 64) -> 16 words
__device__ __forceinline__ void lm_challenge64(LStrobe &s, uint64_t label, uint32_t llen, uint32_t out[16]) {
  lm_label_len(s, label, llen, 64);
  ls_begin_op(s, BPP_FLAG_I | BPP_FLAG_A | BPP_FLAG_C);
  ls_squeeze_words<16>(s, out);
}
// TranscriptRngBuilder::finalize with an all-zero rng draw (NullRng), then 32 bytes of the TranscriptRng
__device__ __forceinline__ void lm_rng_finalize_zero(LStrobe &s) {
  ls_begin_op(s, BPP_FLAG_M | BPP_FLAG_A);
  ls_absorb_reg(s, 0x676e72ull /* "rng" */, 3);
  ls_begin_op(s, BPP_FLAG_A | BPP_FLAG_C);
  ls_overwrite_zero32(s);
}
__device__ __forceinline__ void lm_rng_fill32(LStrobe &s, uint32_t out[8]) {
  ls_begin_op(s, BPP_FLAG_M | BPP_FLAG_A);
  ls_absorb_reg(s, 32ull, 4);
  ls_begin_op(s, BPP_FLAG_I | BPP_FLAG_A | BPP_FLAG_C);
  ls_squeeze_words<8>(s, out);
}

// little-endian register image of a label of up to 8 characters
constexpr uint64_t lm_label(const char *t, int n) {
  uint64_t v = 0;
  for (int i = 0; i < n && i < 8; i++) v |= (uint64_t)(uint8_t)t[i] << (8 * i);
  return v;
}

}  // namespace bpp
