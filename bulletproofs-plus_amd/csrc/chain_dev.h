// The batch-weight chain ON THE DEVICE: one wavefront per reference batch (group).
//
// Reference: the weight transcript of RangeProof::verify -- Transcript::new(b"Bulletproofs+ verifier weights")
// (src/range_proof.rs:811), append_message(b"proof", 32 transcript-RNG bytes) per proof (:849), build_rng().finalize(&mut NullRng)
// (:853, src/utils/nullrng.rs:16-40: 32 zero bytes), then one Scalar::random_not_zero per proof (:894,
// src/protocols/scalar_protocol.rs:23-30: 64 bytes from the transcript RNG, reduced mod l, drawn again while zero).
// chain_host.h is the same chain on host cores (the default for calls that wait for it: a core runs one permutation in 0.2 us,
// a lone wavefront needs ~1.5); this form exists so that a rank's host footprint does not grow with its GPU's throughput: the
// chain becomes one more kernel on the stream, between PASS 1 and the weighted scalars, and the calling thread only enqueues.
//
// The sponge is strictly sequential (1.27 Keccak-f per proof), so the only thing to optimise is the latency of one
// permutation on one wavefront: wkeccak.h.  Around it:
//   * absorbing: the bytes since the last permutation are collected in an LDS image of the block (plain byte stores; a
//     45-byte "proof" record -- two operation headers, label, length, 32 data bytes -- is one store per lane when it does not
//     reach the end of the block) and XORed into the lanes' interleaved halves right before the permutation: the state never
//     leaves the registers and is interleaved once per block, never de-interleaved;
//   * after finalize every draw has the same shape (the sponge stands at byte 64 with no operation open): meta-AD header and
//     length, PRF header, padding are ten constant bytes in words 8, 9 and 20, so a draw is: XOR three per-lane constants,
//     permute, store words 0..7 (still interleaved: k_chain_finish undoes that for 65 536 proofs at once), clear them;
//   * k_chain_finish: 64 bytes -> Scalar::from_bytes_mod_order_wide -> canonical 32 bytes, one lane per proof; a zero weight
//     (probability 2^-252) cannot be redrawn without re-running the chain: it raises a flag and the call falls back to the
//     host chain, which redraws as the reference does.
#pragma once
#include "scalar.h"
#include "wkeccak.h"
#include "wstrobe.h"

namespace bpp {

#define BPP_CHAIN_STAGE 64u  // proofs whose transcript-RNG bytes are staged in LDS at a time (one 32-byte load per lane)

struct ChainLds {
  uint32_t img[WK_LDS_DWORDS];          // wkeccak.h's exchange image
  uint64_t blk[26];                     // bytes absorbed since the last permutation, at their place in the block (208 >= R + 2)
  uint8_t stage[BPP_CHAIN_STAGE * 32];  // transcript-RNG bytes of the next BPP_CHAIN_STAGE proofs
};

struct ChainSponge {
  uint32_t a;               // this lane's interleaved half of the state
  uint32_t pos, pos_begin;  // wave-uniform
};

// padding + permutation (STROBE's run_f): the block image is folded into the state first
__device__ __forceinline__ void cs_run_f(ChainSponge &s, ChainLds &L, const WkLanes &W, const WkRc &R) {
  uint8_t *bb = (uint8_t *)L.blk;
  ws_sync();
  if (ws_lane() == 0) {
    bb[s.pos] ^= (uint8_t)s.pos_begin;
    bb[s.pos + 1] ^= 0x04;
    bb[BPP_STROBE_R + 1] ^= 0x80;
  }
  ws_sync();
  const uint64_t w = L.blk[W.word < 25 ? W.word : 25];
  s.a ^= wk_half(w, W.half);
  ws_sync();
  if (ws_lane() < 26) L.blk[ws_lane()] = 0;
  ws_sync();
  s.a = wk_keccak_f1600(s.a, W, R);
  s.pos = 0;
  s.pos_begin = 0;
}
// absorb n bytes (byte_at(k), evaluated by lane k mod 64)
template <class F>
__device__ __forceinline__ void cs_absorb(ChainSponge &s, ChainLds &L, const WkLanes &W, const WkRc &R, F byte_at, uint32_t n) {
  uint8_t *bb = (uint8_t *)L.blk;
  uint32_t off = 0;
  while (off < n) {
    const uint32_t room = BPP_STROBE_R - s.pos, chunk = n - off < room ? n - off : room;
    for (uint32_t k = ws_lane(); k < chunk; k += 64) bb[s.pos + k] = byte_at(off + k);
    s.pos += chunk;
    off += chunk;
    if (s.pos == BPP_STROBE_R) cs_run_f(s, L, W, R);
  }
}
__device__ __forceinline__ void cs_begin_op(ChainSponge &s, ChainLds &L, const WkLanes &W, const WkRc &R, uint32_t flags) {
  const uint32_t old_begin = s.pos_begin;
  s.pos_begin = s.pos + 1;
  cs_absorb(s, L, W, R, [=](uint32_t k) { return (uint8_t)(k == 0 ? old_begin : flags); }, 2);
  if ((flags & (BPP_FLAG_C | BPP_FLAG_K)) && s.pos != 0) cs_run_f(s, L, W, R);
}

// bytes 8 w .. 8 w + 7 of a block in which `bytes` stand at byte position `at` (the rest zero), as this lane's half
BPP_HD constexpr uint64_t chain_word_of(uint32_t w, uint32_t at, const uint8_t *bytes, uint32_t n) {
  uint64_t v = 0;
  for (uint32_t k = 0; k < n; k++)
    if ((at + k) / 8 == w) v |= (uint64_t)bytes[k] << (8 * ((at + k) % 8));
  return v;
}

// t0: the transcript after Transcript::new(b"Bulletproofs+ verifier weights") (25 state words, pos, pos_begin; host-made).
// wide[p][16]: the 64 PRF bytes of proof p's draw as the interleaved halves of state words 0..7 ([word][half]).
__global__ void __launch_bounds__(64) k_weight_chain(const uint8_t *__restrict__ rng, const uint32_t *__restrict__ group_first, uint32_t G,
                                                     const Strobe *__restrict__ t0, uint32_t *__restrict__ wide) {
  const uint32_t g = blockIdx.x, lane = threadIdx.x;
  if (g >= G) return;
  __shared__ ChainLds L;
  const WkLanes W = wk_lanes(L.img);
  const WkRc R = wk_rc(W);
  const uint32_t p0 = group_first[g], n = group_first[g + 1] - p0;
  if (lane < 26) L.blk[lane] = 0;
  ChainSponge s;
  s.a = W.word < 25 ? wk_half(t0->st[W.word], W.half) : 0u;
  s.pos = t0->pos;
  s.pos_begin = t0->pos_begin;
  ws_sync();

  // ---- append_message(b"proof", rng bytes) per proof: [pos_begin, M|A] "proof" [32,0,0,0] [pos_begin', A] data[32]
  const uint32_t k = lane;  // this lane's byte of a 45-byte record
  const uint8_t rec_const[13] = {0, BPP_FLAG_M | BPP_FLAG_A, 'p', 'r', 'o', 'o', 'f', 32, 0, 0, 0, 0, BPP_FLAG_A};
  uint32_t my_const = 0;
#pragma unroll
  for (int q = 0; q < 13; q++) my_const = (q == (int)k) ? rec_const[q] : my_const;
  uint8_t *bb = (uint8_t *)L.blk;
  for (uint32_t i0 = 0; i0 < n; i0 += BPP_CHAIN_STAGE) {
    const uint32_t cnt = n - i0 < BPP_CHAIN_STAGE ? n - i0 : BPP_CHAIN_STAGE;
    ws_sync();
    if (lane < cnt) {
      const uint4 *src = reinterpret_cast<const uint4 *>(rng + (size_t)(p0 + i0 + lane) * 32);
      uint4 *dst = reinterpret_cast<uint4 *>(L.stage + lane * 32);
      dst[0] = src[0];
      dst[1] = src[1];
    }
    ws_sync();
    for (uint32_t j = 0; j < cnt; j++) {
      const uint8_t *data = L.stage + j * 32;
      if (s.pos + 45 < BPP_STROBE_R) {  // the whole record inside the block: one byte store per lane
        if (k < 45) bb[s.pos + k] = (uint8_t)(k == 0 ? s.pos_begin : k == 11 ? s.pos + 1 : k >= 13 ? data[k - 13] : my_const);
        s.pos_begin = s.pos + 12;
        s.pos += 45;
      } else {
        cs_begin_op(s, L, W, R, BPP_FLAG_M | BPP_FLAG_A);
        cs_absorb(s, L, W, R, [=](uint32_t q) { return (uint8_t)(q < 5 ? "proof"[q] : q == 5 ? 32 : 0); }, 9);  // label, then meta_ad(len, more = true)
        cs_begin_op(s, L, W, R, BPP_FLAG_A);
        cs_absorb(s, L, W, R, [=](uint32_t q) { return data[q]; }, 32);
      }
    }
  }
  // ---- build_rng().finalize(&mut NullRng): meta_ad(b"rng"), key(32 zero bytes)
  cs_begin_op(s, L, W, R, BPP_FLAG_M | BPP_FLAG_A);
  cs_absorb(s, L, W, R, [=](uint32_t q) { return (uint8_t) "rng"[q]; }, 3);
  cs_begin_op(s, L, W, R, BPP_FLAG_A | BPP_FLAG_C);  // its forced permutation leaves pos = 0, no operation open
  // overwrite(32 zero bytes): state bytes 0..31 become zero; pos = 32
  if (W.word < 4) s.a = 0;

  // ---- n draws of 64 bytes: meta_ad(u32le(64)) + prf(64).  First draw at byte 32, every later one at byte 64 (the previous
  // draw's 64 output bytes were cleared and pos_begin is 0 after the PRF's forced permutation):
  //   [0, M|A] [64,0,0,0] [at + 1, I|A|C], padding [at + 7] [0x04] at the next two bytes, 0x80 at byte R + 1
  auto draw_const = [&](uint32_t at) {
    const uint8_t fr[10] = {0, BPP_FLAG_M | BPP_FLAG_A, 64, 0, 0, 0, (uint8_t)(at + 1), BPP_FLAG_I | BPP_FLAG_A | BPP_FLAG_C, (uint8_t)(at + 7), 0x04};
    const uint8_t end[1] = {0x80};
    const uint64_t w = W.word < 25 ? (chain_word_of(W.word, at, fr, 10) | chain_word_of(W.word, BPP_STROBE_R + 1, end, 1)) : 0;
    return wk_half(w, W.half);
  };
  const uint32_t c_first = draw_const(32), c_next = draw_const(64);
  const bool out_lane = W.word < 8;
  uint32_t *dst = wide + (size_t)p0 * 16 + (out_lane ? W.word * 2 + W.half : 0);
  for (uint32_t i = 0; i < n; i++) {
    s.a ^= i == 0 ? c_first : c_next;
    s.a = wk_keccak_f1600(s.a, W, R);
    if (out_lane) {
      dst[(size_t)i * 16] = s.a;
      s.a = 0;
    }
  }
}

// wide -> weights: Scalar::from_bytes_mod_order_wide of every draw, canonical bytes out; *zero_flag = 1 if any weight is zero
// (test_zero: proof index + 1 whose weight is REPORTED as zero, for the tests of the fall-back; 0 = none)
__global__ void __launch_bounds__(64) k_chain_finish(const uint32_t *__restrict__ wide, uint32_t B, uint8_t *__restrict__ weights,
                                                     uint32_t *__restrict__ zero_flag, uint32_t test_zero) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= B) return;
  uint32_t w[16];
  const uint4 *src = reinterpret_cast<const uint4 *>(wide + (size_t)p * 16);
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const uint4 v = src[q];  // [even, odd] of words 2q, 2q + 1
    const uint64_t w0 = wk_word(v.x, v.y), w1 = wk_word(v.z, v.w);
    w[4 * q] = (uint32_t)w0;
    w[4 * q + 1] = (uint32_t)(w0 >> 32);
    w[4 * q + 2] = (uint32_t)w1;
    w[4 * q + 3] = (uint32_t)(w1 >> 32);
  }
  sc m, c;
  sc_mont_from_wide_words(m, w);
  sc_from_mont(c, m);
  if (sc_iszero(c) || p + 1 == test_zero) *zero_flag = 1u;  // (mapped host memory: every writer writes the same word)
  uint4 *dst = reinterpret_cast<uint4 *>(weights + (size_t)p * 32);
  dst[0] = make_uint4(c.v[0], c.v[1], c.v[2], c.v[3]);
  dst[1] = make_uint4(c.v[4], c.v[5], c.v[6], c.v[7]);
}

// The same for draws that come as plain bytes (the host's sponge with the reduction left to the device: chain_host.h, WIDE):
// wide64[p] = the 64 PRF bytes of proof p, in mapped host memory
__global__ void __launch_bounds__(64) k_chain_finish_bytes(const uint8_t *__restrict__ wide64, uint32_t B, uint8_t *__restrict__ weights,
                                                           uint32_t *__restrict__ zero_flag, uint32_t test_zero) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= B) return;
  uint32_t w[16];
  const uint4 *src = reinterpret_cast<const uint4 *>(wide64 + (size_t)p * 64);
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const uint4 v = src[q];
    w[4 * q] = v.x;
    w[4 * q + 1] = v.y;
    w[4 * q + 2] = v.z;
    w[4 * q + 3] = v.w;
  }
  sc m, c;
  sc_mont_from_wide_words(m, w);
  sc_from_mont(c, m);
  if (sc_iszero(c) || p + 1 == test_zero) *zero_flag = 1u;
  uint4 *dst = reinterpret_cast<uint4 *>(weights + (size_t)p * 32);
  dst[0] = make_uint4(c.v[0], c.v[1], c.v[2], c.v[3]);
  dst[1] = make_uint4(c.v[4], c.v[5], c.v[6], c.v[7]);
}

}  // namespace bpp
