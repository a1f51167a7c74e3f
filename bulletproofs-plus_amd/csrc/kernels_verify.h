// Device kernels of the batch verifier (everything except the MSM, which lives in msm.h).
//
// Data layout in HBM (one resident batch):
//   bytes[]        all proof wire bytes back to back, then all compressed commitments
//   desc[B]        ProofDesc: where each proof's pieces are
//   minvals[sumM]  u64 minimum-value promises (0 for None; indistinguishable in the protocol)
//   chal[B][CS]    Montgomery-form challenges per proof: y, z, e_0..e_{r-1}, e_final   (CS = rmax + 3)
//   rng_out[B][32] transcript-RNG bytes that feed the weight transcript
//   dynpts[total]  affine-niels points, proof order: C_0..C_{m-1}, A1, B, A, L_0.., R_0..
//   rows[B][cols]  per-proof contributions to the shared generator scalars (Montgomery form)
//   scal[...]      canonical scalars the MSM consumes: [G x cols static | total dynamic]
#pragma once
#include "blake2b.h"
#include "layout.h"
#include "merlin.h"
#include "wstrobe.h"
#include "lstrobe.h"
#include "point.h"
#include "recode.h"
#include "scalar.h"
#include "kernels_static_gemm.h"

namespace bpp {

__device__ __forceinline__ bool bytes32_all_zero(const uint8_t *p) {
  uint32_t r = 0;
  for (int i = 0; i < 32; i++) r |= p[i];
  return r == 0;
}

// challenge_scalar (src/protocols/transcript_protocol.rs:67-78): returns false on a zero challenge
__device__ __forceinline__ bool dev_challenge(Strobe &s, const uint8_t *label, uint32_t llen, sc &out) {
  uint8_t buf[64];
  merlin_challenge_bytes(s, label, llen, buf, 64);
  sc_mont_from_wide(out, buf);
  return !sc_iszero(out);
}

// PASS 1 of RangeProof::verify (src/range_proof.rs:816-850), one lane per proof.
// RangeProofTranscript::new / challenges_y_z / challenge_round_e / challenge_final_e / to_verifier_rng
// (src/transcripts.rs:59-179); the intermediate build_rng() clones are dead for a verifier and skipped.
#ifndef BPP_TRANSCRIPTS_WAVES
#define BPP_TRANSCRIPTS_WAVES 3  // the 12.8 KB LDS sponge of a wavefront (lstrobe.h) allows 12 wavefronts per CU anyway
#endif
__device__ __forceinline__ bool lm_challenge_scalar(LStrobe &s, uint64_t label, uint32_t llen, sc &out) {
  uint32_t w[16];
  lm_challenge64(s, label, llen, w);
  sc_mont_from_wide_words(out, w);
  return !sc_iszero(out);
}
__global__ void __launch_bounds__(64, BPP_TRANSCRIPTS_WAVES) k_transcripts(const uint8_t *__restrict__ bytes, const ProofDesc *__restrict__ desc,
                                                    const uint64_t *__restrict__ minvals,
                                                    const uint8_t *__restrict__ states, const uint8_t *__restrict__ hg32,
                                                    uint32_t n_bits, uint32_t t, uint32_t B, uint32_t cs,
                                                    sc *__restrict__ chal, uint8_t *__restrict__ rng_out,
                                                    uint32_t *__restrict__ status, uint8_t *__restrict__ rng_host) {
  __shared__ uint32_t sponge[BPP_LS_WORDS * BPP_LS_STRIDE];  // word w of lane l at [w * 64 + l] (lstrobe.h)
  uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  // The lanes past the end of the input replay the LAST proof once more (the same values to the same places) instead of
  // leaving: the host's copy at the end is stored by lane q for piece q of the wavefront's rows, so with B % 64 != 0 the
  // pieces of the last rows belong to lanes that have no proof of their own.  (Round 6: they used to return here; with
  // B % 64 = L the pieces L .. 2L-1 -- the bytes of the last L / 2 proofs -- never reached the host and the host chain of the
  // last reference batch ran on stale bytes: verdicts unchanged, weights not the reference's.  tests/test_gpu_round6.py holds
  // a ragged last batch of an input of more than BPP_TRANSCRIPTS_WAVE_MAX proofs to the oracle.)
  if (p >= B) p = B - 1;
  const ProofDesc d = desc[p];
  LStrobe s;
  ls_from_bytes(s, (lds_u32 *)sponge + threadIdx.x, states + 203u * d.state_idx);
  bool ok = true;
  const uint8_t *pr = bytes + d.proof_off;
  const uint8_t *pd1 = pr + 1;
  const uint8_t *pA = pr + 1 + 32 * t;
  const uint8_t *pA1 = pA + 32, *pB = pA + 64, *pr1 = pA + 96, *ps1 = pA + 128, *pLR = pA + 160;

  lm_append_mem(s, lm_label("dom-sep", 7), 7, (const uint8_t *)"Bulletproofs+ Range Proof", 25);
  lm_append_mem(s, lm_label("H", 1), 1, hg32, 32);  // validated at params creation
  for (uint32_t k = 0; k < t; k++) lm_append_mem(s, lm_label("G", 1), 1, hg32 + 32 * (k + 1), 32);
  lm_append_u64(s, lm_label("N", 1), 1, n_bits);
  lm_append_u64(s, lm_label("T", 1), 1, t);
  lm_append_u64(s, lm_label("M", 1), 1, d.m);
  for (uint32_t j = 0; j < d.m; j++) lm_append_mem(s, lm_label("Ci", 2), 2, bytes + d.commit_off + 32 * j, 32);  // identity allowed (q3)
  for (uint32_t j = 0; j < d.m; j++)
    lm_append_u64_long(s, lm_label("vi - min", 8), lm_label("imum_val", 8), lm_label("ue", 2), 18, minvals[d.minval_idx + j]);

  sc *c = chal + (size_t)p * cs;
  sc v;
  ok = ok && !bytes32_all_zero(pA);
  lm_append_mem(s, lm_label("A", 1), 1, pA, 32);
  ok = lm_challenge_scalar(s, lm_label("y", 1), 1, v) && ok;
  c[0] = v;
  ok = lm_challenge_scalar(s, lm_label("z", 1), 1, v) && ok;
  c[1] = v;
  for (uint32_t j = 0; j < d.rounds; j++) {
    ok = ok && !bytes32_all_zero(pLR + 64 * j) && !bytes32_all_zero(pLR + 64 * j + 32);
    lm_append_mem(s, lm_label("L", 1), 1, pLR + 64 * j, 32);
    lm_append_mem(s, lm_label("R", 1), 1, pLR + 64 * j + 32, 32);
    ok = lm_challenge_scalar(s, lm_label("e", 1), 1, v) && ok;
    if (j + 3 < cs) c[2 + j] = v;  // cs covers every well-formed proof; an oversized one is replayed for its status only
  }
  ok = ok && !bytes32_all_zero(pA1) && !bytes32_all_zero(pB);
  lm_append_mem(s, lm_label("A1", 2), 2, pA1, 32);
  lm_append_mem(s, lm_label("B", 1), 1, pB, 32);
  ok = lm_challenge_scalar(s, lm_label("e", 1), 1, v) && ok;
  if (d.rounds + 3 <= cs) c[2 + d.rounds] = v;
  // to_verifier_rng (src/transcripts.rs:166-179) + NullRng finalize + 32 bytes (src/range_proof.rs:845-848)
  lm_append_mem(s, lm_label("r1", 2), 2, pr1, 32);
  lm_append_mem(s, lm_label("s1", 2), 2, ps1, 32);
  for (uint32_t k = 0; k < t; k++) lm_append_mem(s, lm_label("d1", 2), 2, pd1 + 32 * k, 32);
  lm_rng_finalize_zero(s);
  uint32_t out[8];
  lm_rng_fill32(s, out);
#pragma unroll
  for (int i = 0; i < 8; i++) reinterpret_cast<uint32_t *>(rng_out)[(size_t)p * 8 + i] = out[i];
  if (!ok) atomicOr(&status[p], BPP_ST_TRANSCRIPT_FAIL);
  // The host's copy (mapped page-locked memory: what the batch-weight chain reads, src/range_proof.rs:845-853) leaves the
  // kernel as whole 64-byte lines: the wavefront's 2 KB are transposed through the (now dead) sponge and stored 16 bytes per
  // lane, 1 KB contiguous per store -- a lane's own eight dwords, 32 bytes apart from its neighbour's, would cross the bus as
  // partial lines.  The workgroup is one wavefront: its LDS operations execute in order.
  if (rng_host) {
#pragma unroll
    for (int i = 0; i < 8; i++) sponge[threadIdx.x * 8 + i] = out[i];
    __builtin_amdgcn_wave_barrier();
    const uint32_t p0 = blockIdx.x * blockDim.x, live = min(B - p0, 64u) * 2u;  // 16-byte pieces this wavefront owns
#pragma unroll
    for (uint32_t k = 0; k < 2; k++) {
      const uint32_t q = k * 64u + threadIdx.x;
      if (q < live) {
        const uint4 v = *reinterpret_cast<const uint4 *>(&sponge[q * 4u]);
        *reinterpret_cast<uint4 *>(rng_host + (size_t)p0 * 32 + (size_t)q * 16) = v;
      }
    }
  }
}

// The same PASS 1 with one proof per WAVEFRONT on the cooperative sponge of wstrobe.h (state in LDS, 25-lane Keccak-f).
// Per proof it uses ~7x the issue slots of the one-lane kernel but finishes in half the time, so the host picks it for
// small batches (B <= BPP_TRANSCRIPTS_WAVE_MAX), where PASS 1 is pure latency on a nearly idle chip: 0.115 ms up to 256
// proofs, 0.13 / 0.17 / 0.31 ms for 1024 / 2048 / 4096; the one-lane kernel on the LDS sponge takes 0.24 ms whatever the size.
#define BPP_TRANSCRIPTS_WAVE_MAX 2048u
#define BPP_TABLES_WAVE_MAX 2048u  // same idea for the scalar-stage tables (k_scalars_tables_wave)
#define BPP_SIDE_DECOMPRESS_MAX 2048u  // up to here decompression runs beside PASS 1 on a second stream (engine.hip)
struct TranscriptLds {
  uint64_t st[25];
  uint8_t buf[64];
  uint32_t wk[WK_LDS_DWORDS_RC];  // the permutation's exchange image (wkeccak.h)
};
__device__ __forceinline__ bool wave_challenge(WStrobe &s, TranscriptLds &L, const KeccakLanes &K, const char *label, uint32_t llen,
                                               sc &out) {
  wm_challenge_bytes(s, K, label, llen, L.buf, 64);
  sc_mont_from_wide(out, L.buf);
  return !sc_iszero(out);
}
__device__ __forceinline__ bool wave_nonzero32(const uint8_t *p32) {
  return __ballot(threadIdx.x < 32 && p32[threadIdx.x & 31u] != 0) != 0;
}
__global__ void __launch_bounds__(64) k_transcripts_wave(const uint8_t *__restrict__ bytes, const ProofDesc *__restrict__ desc,
                                                         const uint64_t *__restrict__ minvals,
                                                         const uint8_t *__restrict__ states, const uint8_t *__restrict__ hg32,
                                                         uint32_t n_bits, uint32_t t, uint32_t B, uint32_t cs,
                                                         sc *__restrict__ chal, uint8_t *__restrict__ rng_out,
                                                         uint32_t *__restrict__ status, uint8_t *__restrict__ rng_host) {
  const uint32_t p = blockIdx.x, lane = threadIdx.x;
  if (p >= B) return;
  __shared__ TranscriptLds L;
  const KeccakLanes K = keccak_lanes(L.wk);
  const ProofDesc d = desc[p];
  const uint8_t *sb = states + 203u * d.state_idx;
  for (uint32_t k = lane; k < 200; k += 64) ((uint8_t *)L.st)[k] = sb[k];
  WStrobe s;
  s.st = L.st;
  s.pos = sb[200];
  s.pos_begin = sb[201];
  s.cur_flags = sb[202];
  __syncthreads();
  bool ok = true;
  const uint8_t *pr = bytes + d.proof_off;
  const uint8_t *pd1 = pr + 1;
  const uint8_t *pA = pr + 1 + 32 * t;
  const uint8_t *pA1 = pA + 32, *pB = pA + 64, *pr1 = pA + 96, *ps1 = pA + 128, *pLR = pA + 160;

  wm_append_message(s, K, "dom-sep", 7, BytesAt{(const uint8_t *)"Bulletproofs+ Range Proof"}, 25);
  wm_append_message(s, K, "H", 1, BytesAt{hg32}, 32);
  for (uint32_t k = 0; k < t; k++) wm_append_message(s, K, "G", 1, BytesAt{hg32 + 32 * (k + 1)}, 32);
  wm_append_u64(s, K, "N", 1, n_bits);
  wm_append_u64(s, K, "T", 1, t);
  wm_append_u64(s, K, "M", 1, d.m);
  for (uint32_t j = 0; j < d.m; j++) wm_append_message(s, K, "Ci", 2, BytesAt{bytes + d.commit_off + 32 * j}, 32);
  for (uint32_t j = 0; j < d.m; j++) wm_append_u64(s, K, "vi - minimum_value", 18, minvals[d.minval_idx + j]);

  sc *c = chal + (size_t)p * cs;
  sc v;
  ok = ok && wave_nonzero32(pA);
  wm_append_message(s, K, "A", 1, BytesAt{pA}, 32);
  ok = wave_challenge(s, L, K, "y", 1, v) && ok;
  if (lane == 0) c[0] = v;
  ok = wave_challenge(s, L, K, "z", 1, v) && ok;
  if (lane == 0) c[1] = v;
  for (uint32_t j = 0; j < d.rounds; j++) {
    ok = ok && wave_nonzero32(pLR + 64 * j) && wave_nonzero32(pLR + 64 * j + 32);
    wm_append_message(s, K, "L", 1, BytesAt{pLR + 64 * j}, 32);
    wm_append_message(s, K, "R", 1, BytesAt{pLR + 64 * j + 32}, 32);
    ok = wave_challenge(s, L, K, "e", 1, v) && ok;
    if (lane == 0 && j + 3 < cs) c[2 + j] = v;
  }
  ok = ok && wave_nonzero32(pA1) && wave_nonzero32(pB);
  wm_append_message(s, K, "A1", 2, BytesAt{pA1}, 32);
  wm_append_message(s, K, "B", 1, BytesAt{pB}, 32);
  ok = wave_challenge(s, L, K, "e", 1, v) && ok;
  if (lane == 0 && d.rounds + 3 <= cs) c[2 + d.rounds] = v;
  // to_verifier_rng (src/transcripts.rs:166-179) + NullRng finalize + 32 bytes (src/range_proof.rs:845-848)
  wm_append_message(s, K, "r1", 2, BytesAt{pr1}, 32);
  wm_append_message(s, K, "s1", 2, BytesAt{ps1}, 32);
  for (uint32_t k = 0; k < t; k++) wm_append_message(s, K, "d1", 2, BytesAt{pd1 + 32 * k}, 32);
  wm_rng_finalize(s, K, ZeroAt{});
  wm_rng_fill(s, K, L.buf, 32);
  if (lane < 32) rng_out[(size_t)p * 32 + lane] = L.buf[lane];
  if (rng_host && lane < 8)  // the host's copy (mapped page-locked memory), one 32-byte write per proof
    reinterpret_cast<uint32_t *>(rng_host)[(size_t)p * 8 + lane] = reinterpret_cast<const uint32_t *>(L.buf)[lane];
  if (!ok && lane == 0) atomicOr(&status[p], BPP_ST_TRANSCRIPT_FAIL);
}

#ifndef BPP_DECOMPRESS_WAVES
#define BPP_DECOMPRESS_WAVES 3  // 168 VGPRs; the ~40 spilled registers are outside the squaring chain
#endif
// CompressedRistretto::decompress for every proof point and commitment (src/range_proof.rs:859-866,1067-1109),
// one lane per point.  src_off[i] = byte offset in bytes[]; owner[i] = proof index | (is_commitment << 31).
__global__ void __launch_bounds__(64, BPP_DECOMPRESS_WAVES) k_decompress(const uint8_t *__restrict__ bytes, const uint32_t *__restrict__ src_off,
                                                   const uint32_t *__restrict__ owner, const uint32_t *__restrict__ idx, uint32_t n,
                                                   niels *__restrict__ out, uint32_t *__restrict__ status,
                                                   uint32_t *__restrict__ spill /* [30][n] words or null */,
                                                   uint32_t *__restrict__ status_b = nullptr /* upload: the verifications' working copy */) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t e = idx[i];  // dynamic slot: the statement's commitments are decoded once at upload (the reference's
                              // RangeStatement holds points), the proof's own points on every verification
  niels q;
  bool ok = ristretto_decompress_lean(q, bytes + src_off[e], spill ? spill + i : nullptr, n);
  if (!ok) {
    niels_identity(q);
    uint32_t o = owner[e];
    atomicOr(&status[o & 0x7fffffffu], (o >> 31) ? BPP_ST_COMMIT_FAIL : BPP_ST_DECOMPRESS_FAIL);
    if (status_b) atomicOr(&status_b[o & 0x7fffffffu], (o >> 31) ? BPP_ST_COMMIT_FAIL : BPP_ST_DECOMPRESS_FAIL);
  }
  out[e] = q;
}

// Plain batch decompression for the B1 entry points (bpp_precomp_create / bpp_msm_*).
__global__ void __launch_bounds__(64) k_decompress_plain(const uint8_t *__restrict__ pts32, uint32_t n,
                                                         niels *__restrict__ out, uint32_t *__restrict__ bad) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint8_t s[32];
  for (int k = 0; k < 32; k++) s[k] = pts32[(size_t)i * 32 + k];
  niels q;
  if (!ristretto_decompress(q, s)) {
    niels_identity(q);
    atomicAdd(bad, 1u);
  }
  out[i] = q;
}

// RistrettoPoint::from_uniform_bytes for generator derivation (src/generators/generators_chain.rs:43-49,
// src/ristretto.rs:88-95): 64 uniform bytes -> affine niels + canonical encoding.
__global__ void __launch_bounds__(64) k_from_uniform(const uint8_t *__restrict__ uni64, uint32_t n, niels *__restrict__ out,
                                                     uint8_t *__restrict__ comp32) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint8_t b[64];
  for (int k = 0; k < 64; k++) b[k] = uni64[(size_t)i * 64 + k];
  ge p;
  ristretto_from_uniform(p, b);
  niels q;
  ge_to_niels(q, p);
  out[i] = q;
  uint8_t c[32];
  ristretto_compress(c, p);
  for (int k = 0; k < 32; k++) comp32[(size_t)i * 32 + k] = c[k];
}

// ---------------------------------------------------------------------------------------------
// PASS 2 scalar block (src/range_proof.rs:894-1033).  The host weight chain (1.3 ms per 64 x 1024 proofs) runs while
// PASS 1, decompression and k_scalars_shared execute; the weights enter in k_scalars_lanes' prologue:
//   k_scalars_shared    1 lane / proof : batch inversion (divsteps), powers, sums -> shr[p][*], and the weight-free
//                                        low / high tables of k_scalars_lanes -> tab[p][*]              (no weight)
//   k_scalars_lanes     1 wave / ppw proofs : w x (low tables, e^2 z) while the tables go to LDS, the dynamic scalars
//                                        (canonical) and the g / h base columns, one product per lane and job; then lanes
//                                        over (proof, generator index) -> WEIGHTED rows
//   k_reduce_static                    : per group, column sums of the rows (additions only)
// A product costs the same ~270 instructions whether one lane of the wavefront needs it or all 64: everything that exists
// once per proof or once per table entry is therefore computed with one lane per PROOF (64 proofs per wavefront), and
// only the 4 products per generator pair run with lanes over generator indices.  (Until r02_v5 the tables were built
// inside k_scalars_lanes, 54 busy lanes in four divergent branches: 5.5 k of that kernel's 9.6 k instructions per
// wavefront.)  Everything is Montgomery form until the final stores.
// ---------------------------------------------------------------------------------------------
#define BPP_MAX_ROUNDS 12  // mn <= 64 * 32 = 2048 -> 11 rounds
#define SH_Z 0
#define SH_Z2 1
#define SH_E 2
#define SH_E2 3
#define SH_Y 4
#define SH_YINV 5
#define SH_YNM 6
#define SH_YNM1 7
#define SH_R1E 10
#define SH_S1E 11
#define SH_E2Z 12
#define SH_NEG_E2 13
#define SH_NEG_E2_YNM1 8
#define SH_HS 14
#define SH_D1(k) (15 + (k))
#define SH_ARR 21
#define SH_EJ(j) (SH_ARR + (j))
#define SH_EINV(j) (SH_ARR + BPP_MAX_ROUNDS + (j))
#define SH_ESQ(j) (SH_ARR + 2 * BPP_MAX_ROUNDS + (j))
#define SH_ESQINV(j) (SH_ARR + 3 * BPP_MAX_ROUNDS + (j))
#define SH_YINVPOW(b) (SH_ARR + 4 * BPP_MAX_ROUNDS + (b))
#define SH_STRIDE (SH_ARR + 5 * BPP_MAX_ROUNDS)

// Table block of one proof in HBM (entries of 32 bytes, packed Montgomery scalars), written with one lane per proof and
// read back by k_scalars_lanes with lanes over entries (coalesced):
//   glo[8] yn2lo[8] hlo[8] | ghi[nhi_max] y2hi[nhi_max] shi[nhi_max] | e2z        (meaning: see k_scalars_lanes)
// The generator index splits as i = (hi << LB) | lo with LB = min(3, log2 n_bits) low bits, so that the party of a generator
// (i / n_bits) is a function of hi alone; nhi_max = 2^(largest round count - LB).
BPP_HD uint32_t lanes_lb(uint32_t n_bits) {
  const uint32_t l2 = n_bits ? (uint32_t)__builtin_ctz(n_bits) : 0u;
  return l2 < 3u ? l2 : 3u;
}
BPP_HD constexpr uint32_t lanes_tab_stride(uint32_t nhi_max) { return 24u + 3u * nhi_max + 1u; }
// shapes the three kernels skip alike (rejected on the host before PASS 2)
BPP_HD bool lanes_shape_ok(uint32_t rounds, uint32_t m, uint32_t nhi_max, uint32_t lb) {
  return rounds <= BPP_MAX_ROUNDS - 1 && m >= 1 && m <= 32 && rounds >= lb && (1u << (rounds - lb)) <= nhi_max;
}
// this table entry's share of k = i mod n_bits for i = (hi << LB) | lo (n_bits and nlo are powers of two): d[i] carries 2^k
BPP_HD uint32_t lanes_e2k(bool is_hi, uint32_t v, uint32_t nlo, uint32_t n_bits) {
  const uint32_t e = nlo >= n_bits ? (is_hi ? 0u : (v & (n_bits - 1u))) : (is_hi ? nlo * (v & (n_bits / nlo - 1u)) : v);
  return e & 63u;
}
__device__ __forceinline__ void sc_mul_pow2(sc &r, const sc &a, uint32_t e) {  // a * 2^e, e < 64
  sc9 a9, p2;
  sc9_from(a9, a);
#pragma unroll
  for (int q = 0; q < 9; q++) p2.l[q] = SC_POW2_R29[e][q];
  sc9_montmul(r, a9, p2);
}
// s[] over `nbits` bits of the index starting at bit b0: entry 0 = product of the inverse challenges, entry v + 2^b = entry v
// times e_j^2 (j = r - 1 - b, src/range_proof.rs:972-990 read as a product over the bits of i): nbits - 1 + 2^nbits - 1 products
__device__ __forceinline__ void lanes_s_tree(sc *dst, const sc *o, uint32_t r, uint32_t b0, uint32_t nbits, const sc &one) {
  sc a = one;
  for (uint32_t bb = 0; bb < nbits; bb++) {
    const sc f = o[SH_EINV(r - 1 - (b0 + bb))];
    if (bb == 0) a = f;
    else sc_montmul(a, a, f);
  }
  dst[0] = a;
  for (uint32_t bb = 0; bb < nbits; bb++) {
    const sc esq = o[SH_ESQ(r - 1 - (b0 + bb))];
    for (uint32_t v = 0; v < (1u << bb); v++) {
      sc x = dst[v];
      sc_montmul(x, x, esq);
      dst[v + (1u << bb)] = x;
    }
  }
}

__device__ __forceinline__ void sc_load_mont(sc &r, const uint8_t *p) {
  sc a;
  sc_load_words(a, p);
  sc_to_mont(r, a);
}

template <bool GEMM>  // GEMM: hi_t != null
__global__ void __launch_bounds__(64) k_scalars_shared(const uint8_t *__restrict__ bytes, const ProofDesc *__restrict__ desc,
                                                       const uint64_t *__restrict__ minvals, const sc *__restrict__ chal,
                                                       uint32_t n_bits, uint32_t t, uint32_t cs, uint32_t B,
                                                       sc *__restrict__ shr, uint32_t nhi_max,
                                                       sc *tab /* lanes_tab_stride(nhi_max) entries per proof */,
                                                       int8_t *__restrict__ hi_t /* null, or the high tables once more as digit
                                                       tables for k_static_gemm: ghi | shiR | y2hi, nhi_max entries each */,
                                                       uint32_t nblk) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= B) return;
  const ProofDesc d = desc[p];
  const uint32_t r = d.rounds, m = d.m, mn = m * n_bits;
  if (r > BPP_MAX_ROUNDS - 1) return;  // rejected on the host before PASS 2
  const sc *c = chal + (size_t)p * cs;
  sc *o = shr + (size_t)p * SH_STRIDE;
  const uint8_t *pr = bytes + d.proof_off;
  const uint8_t *pd1 = pr + 1;
  const uint8_t *pr1 = pr + 1 + 32 * t + 96, *ps1 = pr1 + 32;
  sc one, y = c[0], z = c[1], ef = c[2 + r], r1, s1;
  sc_mont_one(one);
  sc_load_mont(r1, pr1);
  sc_load_mont(s1, ps1);
  // batch inversion of [e_0..e_{r-1}, y, y-1] (:897-905): prefix products parked in the output slots
  sc ym1, acc = one;
  sc_sub(ym1, y, one);
  for (uint32_t j = 0; j < r + 2; j++) {
    sc x;  // (plain copies: `cond ? c[..] : y` selects between ADDRESSES and parks y and ym1 in scratch memory)
    if (j < r) x = c[2 + j];
    else if (j == r) x = y;
    else x = ym1;
    if (j < r) o[SH_EINV(j)] = acc;
    else if (j == r) o[SH_YINV] = acc;
    else o[SH_YNM] = acc;
    sc_montmul(acc, acc, x);
  }
  sc run, y_1_inverse, y_inverse;
  sc_mont_invert_vartime(run, acc);
  for (int j = (int)r + 1; j >= 0; j--) {
    sc x;
    if ((uint32_t)j < r) x = c[2 + j];
    else if ((uint32_t)j == r) x = y;
    else x = ym1;
    const sc pre = ((uint32_t)j < r) ? o[SH_EINV(j)] : ((uint32_t)j == r ? o[SH_YINV] : o[SH_YNM]);
    sc inv_j;
    sc_montmul(inv_j, run, pre);
    sc_montmul(run, run, x);
    if ((uint32_t)j == r + 1) y_1_inverse = inv_j;
    else if ((uint32_t)j == r) y_inverse = inv_j;
    else {
      sc sq;
      o[SH_EJ(j)] = x;
      o[SH_EINV(j)] = inv_j;
      sc_montsq(sq, x);
      o[SH_ESQ(j)] = sq;
      sc_montsq(sq, inv_j);
      o[SH_ESQINV(j)] = sq;
    }
  }
  {
    sc pw = y_inverse;
    for (uint32_t b = 0; b < r; b++) {
      o[SH_YINVPOW(b)] = pw;
      sc_montsq(pw, pw);
    }
  }
  sc z_square, e_square, y_nm, y_nm_1, y_sum, tmp;
  sc_montsq(z_square, z);
  sc_montsq(e_square, ef);
  sc_mont_pow_u32(y_nm, y, mn);
  sc_montmul(y_nm_1, y_nm, y);
  sc_sub(tmp, y_nm, one);
  sc_montmul(tmp, tmp, y);
  sc_montmul(y_sum, tmp, y_1_inverse);  // :916
  sc d_sum = z_square, d_tmp = z_square;  // :932-938
  for (uint32_t mm = m; mm > 1; mm >>= 1) {
    sc_montmul(tmp, d_sum, d_tmp);
    sc_add(d_sum, d_sum, tmp);
    sc_montsq(d_tmp, d_tmp);
  }
  {
    sc tn;  // 2^n - 1
    const uint64_t v = (n_bits >= 64) ? ~0ULL : ((1ULL << n_bits) - 1ULL);
    sc_mont_from_u64(tn, v);
    sc_montmul(d_sum, d_sum, tn);
  }
  sc r1_e, s1_e, e_square_z, neg_e_square;
  sc_montmul(r1_e, r1, ef);
  sc_montmul(s1_e, s1, ef);
  sc_montmul(e_square_z, e_square, z);
  sc_neg(neg_e_square, e_square);
  o[SH_Z] = z;
  o[SH_Z2] = z_square;
  o[SH_E] = ef;
  o[SH_E2] = e_square;
  o[SH_Y] = y;
  o[SH_YINV] = y_inverse;
  o[SH_YNM] = y_nm;
  o[SH_YNM1] = y_nm_1;
  o[SH_R1E] = r1_e;
  o[SH_S1E] = s1_e;
  o[SH_E2Z] = e_square_z;
  o[SH_NEG_E2] = neg_e_square;
  {
    sc ny;
    sc_montmul(ny, neg_e_square, y_nm_1);
    o[SH_NEG_E2_YNM1] = ny;
  }
  // Pedersen h-base contribution without the weight (:1011-1017)
  sc hs;
  sc_0(hs);
  {
    sc zp = z_square;
    for (uint32_t j = 0; j < m; j++) {
      sc wv, vm;
      sc_montmul(wv, neg_e_square, zp);
      sc_montmul(wv, wv, y_nm_1);
      sc_mont_from_u64(vm, minvals[d.minval_idx + j]);
      sc_montmul(wv, wv, vm);
      sc_sub(hs, hs, wv);
      sc_montmul(zp, zp, z_square);
    }
    sc a, b2, u;
    sc_montmul(a, r1, y);
    sc_montmul(a, a, s1);
    sc_montmul(b2, y_nm_1, z);
    sc_montmul(b2, b2, d_sum);
    sc_sub(u, z_square, z);
    sc_montmul(u, u, y_sum);
    sc_add(b2, b2, u);
    sc_montmul(b2, b2, e_square);
    sc_add(a, a, b2);
    sc_add(hs, hs, a);
  }
  o[SH_HS] = hs;
  for (uint32_t k = 0; k < t; k++) {
    sc d1;
    sc_load_mont(d1, pd1 + 32 * k);
    o[SH_D1(k)] = d1;
  }
  // ---- the tables of k_scalars_lanes, before the weight: with i = (hi << LB) | lo, party = i / n_bits (a function of hi)
  //   hlo[lo] = s1e s_lo[lo]   glo[lo] = r1e y^-lo s_lo[lo]      yn2lo[lo] = y^mn y^-lo 2^klo
  //   shi[hi] = s_hi[hi]       ghi[hi] = y^-(hi << LB) s_hi[hi]  y2hi[hi] = y^-(hi << LB) 2^khi z^(2(party+1))
  if (!tab) return;  // small inputs: k_scalars_tables_wave builds them, one wavefront per proof
  const uint32_t LB = lanes_lb(n_bits);
  if (!lanes_shape_ok(r, m, nhi_max, LB)) return;
  const uint32_t HB = r - LB, nlo = 1u << LB, nhi = 1u << HB;
  sc *T = tab + (size_t)p * lanes_tab_stride(nhi_max);
  sc *glo = T, *yn2lo = T + 8, *hlo = T + 16, *ghi = T + 24, *y2hi = ghi + nhi_max, *shi = y2hi + nhi_max;
  lanes_s_tree(hlo, o, r, 0, LB, one);
  lanes_s_tree(shi, o, r, LB, HB, one);
  {
    sc yr = r1_e, yn = y_nm;
    for (uint32_t v = 0; v < nlo; v++) {
      const sc sv = hlo[v];
      sc x;
      sc_montmul(x, sv, s1_e);
      hlo[v] = x;
      sc_montmul(x, sv, yr);
      glo[v] = x;
      sc_mul_pow2(x, yn, lanes_e2k(false, v, nlo, n_bits));
      yn2lo[v] = x;
      if (v + 1 < nlo) {
        sc_montmul(yr, yr, y_inverse);
        sc_montmul(yn, yn, y_inverse);
      }
    }
  }
  {
    sc yh = one, step = one, zz = z_square;  // zz = z^(2(party+1)) of the current entry
    if (HB) step = o[SH_YINVPOW(LB)];
    const uint32_t per_party = n_bits >> LB;  // high indices per party
    for (uint32_t v = 0; v < nhi; v++) {
      const sc sv = shi[v];
      sc x;
      sc_montmul(x, sv, yh);
      ghi[v] = x;
      if constexpr (GEMM) {
        const size_t tb = sgemm_table_bytes(nhi_max, nblk);
        sgemm_store(hi_t, v, nblk, p, x);
        sgemm_store(hi_t + tb, (~v) & (nhi - 1u), nblk, p, sv);  // shiR[hi] = shi[~hi]: h[i] pairs s[mn-1-i] with index i
      }
      sc_mul_pow2(x, yh, lanes_e2k(true, v, nlo, n_bits));
      sc_montmul(x, x, zz);
      y2hi[v] = x;
      if constexpr (GEMM) sgemm_store(hi_t + 2 * sgemm_table_bytes(nhi_max, nblk), v, nblk, p, x);
      if (v + 1 < nhi) {
        sc_montmul(yh, yh, step);
        if (((v + 1) & (per_party - 1u)) == 0) sc_montmul(zz, zz, z_square);
      }
    }
  }
}

// The same weight-free tables for small inputs (a few hundred proofs: one lane per proof would leave the chip idle behind a
// chain of ~85 dependent products): one wavefront per proof, one lane per table entry, every entry straight from the bits
// of its index (at most 3 + 3 + 5 products deep, HB more for large aggregations).  Same values as k_scalars_shared writes.
__global__ void __launch_bounds__(64) k_scalars_tables_wave(const ProofDesc *__restrict__ desc, const sc *__restrict__ shr,
                                                            uint32_t n_bits, uint32_t B, uint32_t nhi_max, sc *__restrict__ tab) {
  const uint32_t p = blockIdx.x;
  if (p >= B) return;
  const ProofDesc d = desc[p];
  const uint32_t r = d.rounds, m = d.m;
  const uint32_t LB = lanes_lb(n_bits);
  if (!lanes_shape_ok(r, m, nhi_max, LB)) return;
  const uint32_t HB = r - LB, nlo = 1u << LB, nhi = 1u << HB;
  const sc *S = shr + (size_t)p * SH_STRIDE;
  sc *T = tab + (size_t)p * lanes_tab_stride(nhi_max);
  sc *glo = T, *yn2lo = T + 8, *hlo = T + 16, *ghi = T + 24, *y2hi = ghi + nhi_max, *shi = y2hi + nhi_max;
  sc one;
  sc_mont_one(one);
  for (uint32_t it = threadIdx.x; it < nlo + nhi; it += 64) {
    const bool is_hi = it >= nlo;
    const uint32_t v = is_hi ? it - nlo : it, b0 = is_hi ? LB : 0, nbits = is_hi ? HB : LB;
    sc sv = one, yv = one;
    for (uint32_t bb = 0; bb < nbits; bb++) {
      const uint32_t b = b0 + bb, j = r - 1 - b;
      const bool bit = (v >> bb) & 1u;
      const sc f = bit ? S[SH_EJ(j)] : S[SH_EINV(j)];
      const sc yp = bit ? S[SH_YINVPOW(b)] : one;
      sc_montmul(sv, sv, f);
      sc_montmul(yv, yv, yp);
    }
    sc x;
    if (is_hi) {
      shi[v] = sv;
      sc_montmul(x, yv, sv);
      ghi[v] = x;
      sc zz;
      sc_mont_pow_u32(zz, S[SH_Z2], (v >> (__builtin_ctz(n_bits) - LB)) + 1u);  // z^(2(party+1)), party = (v << LB) / n_bits
      sc_mul_pow2(x, yv, lanes_e2k(true, v, nlo, n_bits));
      sc_montmul(x, x, zz);
      y2hi[v] = x;
    } else {
      sc_montmul(x, sv, S[SH_S1E]);
      hlo[v] = x;
      sc_montmul(x, yv, S[SH_R1E]);
      sc_montmul(x, x, sv);
      glo[v] = x;
      sc_montmul(x, yv, S[SH_YNM]);
      sc_mul_pow2(x, x, lanes_e2k(false, v, nlo, n_bits));
      yn2lo[v] = x;
    }
  }
}

// Generator scalars.  With i = (hi << LB) | lo, s[i] = slo[lo]*shi[hi] and y^-i = ylo[lo]*yhi[hi] (products over the bits of
// i), so every per-index quantity is ONE product of a "low" and a "high" table entry:
//   g[i]                     = w r1e y^-i s[i]                 = glo[lo]   * ghi[hi]    glo = w*r1e*ylo*slo,  ghi = yhi*shi
//   -w e^2 d[i] y^(mn-i)     = -w e^2 y^mn y^-i 2^k z^(2(j+1)) = yn2lo[lo] * y2hi[hi]   d[i] = z^(2(j+1)) 2^k with party j = i / n
//                                                               and k = i mod n (src/range_proof.rs:919-929); k = klo(lo) + khi(hi),
//                                                               x 2^e is one product with the constant 2^e R mod l (SC_POW2_R29); j
//                                                               depends on hi alone (LB <= log2 n), so z^(2(j+1)) sits in y2hi
//   h[i]                     = w s1e s[mn-1-i]                 = hlo[~lo]  * shi[~hi]   hlo = w*s1e*slo
// A generator pair costs THREE products under TWO reductions: g[i] + e2z, and h[i] - w e^2 (d[i] y^(mn-i) + z) =
// (hlo*shi + yn2lo*y2hi) - e2z with both products accumulated in the same columns (sc9_montmul2); e2z = w e^2 z.  The proof's
// batch weight w (src/range_proof.rs:894) sits in the three low tables and e2z, so the rows come out WEIGHTED and the
// per-group column sums need no product at all (round 1: 5 products per pair here and 2 more in k_reduce_static).  The tables
// come from tab[] (k_scalars_shared / k_scalars_tables_wave; the weight is folded in on the way) and are unpacked ONCE per entry into nine 29-bit limbs in LDS
// (sc9) instead of once per use.  Dynamic LDS per proof:
//   sc9: glo[8] yn2lo[8] hlo[8] (weighted) | ghi[nhi_max] y2hi[nhi_max] shi[nhi_max]      sc: e2z (weighted)
BPP_HD constexpr uint32_t lanes_lds_bytes(uint32_t nhi_max) {
  return (lanes_tab_stride(nhi_max) - 1u) * (uint32_t)sizeof(sc9) + (uint32_t)sizeof(sc);
}

// One workgroup serves `ppw` consecutive proofs (a 64-bit single-commitment proof has 64 generator pairs: four proofs
// make four full passes of the 64 lanes); all phases run over flattened (proof, item) items.
//
// The WEIGHTED part of the scalar block (src/range_proof.rs:894, :1006-1032) is this kernel's prologue (until round 3 a
// launch of its own, k_scalars_weighted: one lane per proof, 45 products deep, one more kernel queueing behind the other
// steps' work): the proof's batch weight w goes into the three low tables and e^2 z while they are unpacked into LDS
// (nothing is written back: tab[] stays weight-free), and the proof's m + 3 + 2r dynamic scalars and t + 1 base columns are
// one product each, spread over the lanes:
//   C_j: (-e^2 y^{mn+1} w) z^{2(j+1)};  A1: -e w;  B: -w;  A: -e^2 w;  L_j: (-e^2 w) e_j^2;  R_j: (-e^2 w) e_j^-2
// A Montgomery product with one CANONICAL operand is the canonical product, so the dynamic scalars (which the MSM wants
// canonical) take the weight as it arrives and need no conversion.  Same products on the same operands as the old kernels.
#define BPP_LANES_MAX_PPW 16
__global__ void __launch_bounds__(64) k_scalars_lanes(const ProofDesc *__restrict__ desc, const sc *__restrict__ tab,
                                                      const sc *__restrict__ shr, const uint8_t *__restrict__ weights32,
                                                      uint32_t n_bits, uint32_t t, uint32_t max_mn, uint32_t cols, uint32_t B,
                                                      uint32_t nhi_max, uint32_t ppw,
                                                      sc *__restrict__ rows /* weighted, Montgomery */,
                                                      sc *__restrict__ dyn_out /* canonical */,
                                                      uint64_t *__restrict__ parts /* null, or [workgroup][2 max_mn][8] limb sums */,
                                                      uint32_t lazy /* with parts: one reduction per (workgroup, column) */,
                                                      // chain = 2 (host sponges, chain_dev.h): the 64 PRF bytes per proof in mapped host memory instead of
                                                      // weights32 -- reduced mod l here (Scalar::from_bytes_mod_order_wide, scalar_protocol.rs:23-30), the
                                                      // canonical weight left in weights_out for traces, a zero weight raises *zero_flag (the call then
                                                      // runs again with the chains on the host, which redraw): one launch fewer on a step's latency chain
                                                      const uint8_t *__restrict__ wide64 = nullptr, uint8_t *__restrict__ weights_out = nullptr,
                                                      uint32_t *__restrict__ zero_flag = nullptr, uint32_t test_zero = 0) {
  const uint32_t p0 = blockIdx.x * ppw;
  const uint32_t lane = threadIdx.x;
  extern __shared__ uint32_t lanes_lds_raw[];
  __shared__ uint32_t s_r[BPP_LANES_MAX_PPW], s_m[BPP_LANES_MAX_PPW], s_dyn[BPP_LANES_MAX_PPW];
  __shared__ sc s_mult[BPP_LANES_MAX_PPW][5];  // w, -w e^2 (Montgomery), -w e^2 (canonical), -w e^2 y^(mn+1) (canonical), the weight as it came
  const uint32_t ts = lanes_tab_stride(nhi_max), n9 = ts - 1u;  // entries per proof: n9 as limbs, then e2z packed
  const uint32_t per_bytes = lanes_lds_bytes(nhi_max);
  const uint32_t LB = lanes_lb(n_bits), nlo = 1u << LB;
  // A product costs the same ~250 instructions whether one lane of the wavefront needs it or all 64, so the prologue is laid
  // out for full wavefronts: (1) w = weight in Montgomery form, one lane per proof; (2) the three derived multipliers, one
  // lane per (proof, multiplier); (3) ONE flat list of product jobs over all proofs of the workgroup -- 3 * 2^LB low-table
  // entries, e^2 z, the dynamic scalars, the base columns: 43 per 64-bit proof; (4) the high tables are copied without a product.
  if (lane < ppw) {
    const uint32_t p = p0 + lane;
    uint32_t r = ~0u, m = 0, dyn_off = 0;
    if (p < B) {
      const ProofDesc d = desc[p];
      if (lanes_shape_ok(d.rounds, d.m, nhi_max, LB)) r = d.rounds;
      m = d.m;
      dyn_off = d.dyn_off;
    }
    s_r[lane] = r;  // ~0: nothing to do for this slot (past the end, or a shape rejected on the host before PASS 2)
    s_m[lane] = m;
    s_dyn[lane] = dyn_off;
    sc wc;
    if (wide64 && p < B) {  // (every proof of the batch, whatever its shape: what k_chain_finish_bytes does)
      uint32_t ww[16];
      const uint4 *src = reinterpret_cast<const uint4 *>(wide64 + (size_t)p * 64);
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const uint4 v = src[q];
        ww[4 * q] = v.x;
        ww[4 * q + 1] = v.y;
        ww[4 * q + 2] = v.z;
        ww[4 * q + 3] = v.w;
      }
      sc wm;
      sc_mont_from_wide_words(wm, ww);
      sc_from_mont(wc, wm);
      if (sc_iszero(wc) || p + 1 == test_zero) *zero_flag = 1u;  // (mapped host memory: every writer writes the same word)
      uint4 *dst = reinterpret_cast<uint4 *>(weights_out + (size_t)p * 32);
      dst[0] = make_uint4(wc.v[0], wc.v[1], wc.v[2], wc.v[3]);
      dst[1] = make_uint4(wc.v[4], wc.v[5], wc.v[6], wc.v[7]);
    }
    if (r != ~0u) {
      sc w;
      if (!wide64) sc_load_words(wc, weights32 + (size_t)p * 32);
      sc_to_mont(w, wc);
      s_mult[lane][0] = w;
      s_mult[lane][4] = wc;
    }
  }
  __syncthreads();
  for (uint32_t it = lane; it < 3u * ppw; it += 64) {
    const uint32_t sub = it / 3u, which = it - 3u * sub;
    if (s_r[sub] == ~0u) continue;
    const sc *S = shr + (size_t)(p0 + sub) * SH_STRIDE;
    const sc a = which == 2u ? S[SH_NEG_E2_YNM1] : S[SH_NEG_E2];
    const sc b2 = which == 0u ? s_mult[sub][0] : s_mult[sub][4];
    sc x;
    sc_montmul(x, a, b2);
    s_mult[sub][1 + which] = x;  // 1: -w e^2 Montgomery, 2: -w e^2 canonical, 3: -w e^2 y^(mn+1) canonical
  }
  __syncthreads();
  {
    uint32_t jmax = 0;  // product jobs per proof (the largest of the workgroup: jobs past a proof's own count are skipped)
    for (uint32_t sub = 0; sub < ppw; sub++)
      if (s_r[sub] != ~0u) jmax = max(jmax, 3u * nlo + 1u + s_m[sub] + 3u + 2u * s_r[sub] + t + 1u);
    for (uint32_t it = lane; it < ppw * jmax; it += 64) {
      const uint32_t sub = it / jmax, job = it - sub * jmax;
      const uint32_t r = s_r[sub], m = s_m[sub];
      const uint32_t j_e2z = 3u * nlo, j_c = j_e2z + 1u, j_a1 = j_c + m, j_lr = j_a1 + 3u, j_row = j_lr + 2u * r, n_jobs = j_row + t + 1u;
      if (r == ~0u || job >= n_jobs) continue;
      const size_t p = p0 + sub;
      const sc *S = shr + p * SH_STRIDE;
      const sc *T = tab + p * ts;
      sc *dyn = dyn_out + s_dyn[sub];
      // (with `parts` the generator columns never exist per proof: rows[] holds the t + 1 base columns only, stride t + 1)
      sc *row_base = parts ? rows + p * (t + 1) : rows + p * cols + 2 * (size_t)max_mn;  // the proof's t + 1 base columns
      uint8_t *base = reinterpret_cast<uint8_t *>(lanes_lds_raw) + (size_t)sub * per_bytes;
      const sc *src;
      sc *dst = nullptr;
      uint32_t lds_idx = ~0u;  // destination in LDS (sc9 entry), else `dst` in global memory
      uint32_t sel, zpow = 0;  // multiplier index into s_mult; C_j: z^(2(j+1)) from z^2 by j more products
      bool neg = false, plain_neg = false;
      if (job < nlo) src = T + job, lds_idx = job, sel = 0;                                // glo
      else if (job < 2u * nlo) src = T + 16 + (job - nlo), lds_idx = 16u + (job - nlo), sel = 0;      // hlo
      else if (job < 3u * nlo) src = T + 8 + (job - 2u * nlo), lds_idx = 8u + (job - 2u * nlo), sel = 1;  // yn2lo
      else if (job == j_e2z) src = S + SH_E2Z, lds_idx = n9, sel = 0;                      // w e^2 z
      else if (job < j_a1) src = S + SH_Z2, dst = dyn + (job - j_c), sel = 3, zpow = job - j_c;
      else if (job == j_a1) src = S + SH_E, dst = dyn + m, sel = 4, neg = true;               // A1: -e w
      else if (job == j_a1 + 1) src = S + SH_E, dst = dyn + m + 1, sel = 4, plain_neg = true;  // B: -w
      else if (job == j_a1 + 2) src = S + SH_NEG_E2, dst = dyn + m + 2, sel = 4;               // A: -e^2 w
      else if (job < j_row) {
        const uint32_t k = job - j_lr;
        src = k < r ? S + SH_ESQ(k) : S + SH_ESQINV(k - r);
        dst = dyn + m + 3 + k;
        sel = 2;
      } else {
        const uint32_t k = job - j_row;  // 0: the h base, 1 + k: g base k
        src = k == 0 ? S + SH_HS : S + SH_D1(k - 1);
        dst = row_base + (k == 0 ? t : k - 1);
        sel = 0;
      }
      sc a = *src, x;
      if (zpow) {
        const sc z2 = a;
        for (uint32_t i = 0; i < zpow; i++) sc_montmul(a, a, z2);
      }
      if (neg) sc_neg(a, a);
      const sc mult = s_mult[sub][sel];
      sc_montmul(x, a, mult);
      if (plain_neg) sc_neg(x, mult);
      if (lds_idx == ~0u) {
        *dst = x;
      } else if (lds_idx < n9) {
        sc9 o9;
        sc9_from(o9, x);
        reinterpret_cast<sc9 *>(base)[lds_idx] = o9;
      } else {
        *reinterpret_cast<sc *>(base + (size_t)n9 * sizeof(sc9)) = x;
      }
    }
  }
  // ---- the high tables -> LDS as they are (entries a proof's shape does not use are never read)
  for (uint32_t it = lane; it < ppw * (n9 - 24u); it += 64) {
    const uint32_t sub = it / (n9 - 24u), idx = 24u + (it - sub * (n9 - 24u));
    if (s_r[sub] == ~0u) continue;
    const sc v = tab[(size_t)(p0 + sub) * ts + idx];
    sc9 o9;
    sc9_from(o9, v);
    reinterpret_cast<sc9 *>(reinterpret_cast<uint8_t *>(lanes_lds_raw) + (size_t)sub * per_bytes)[idx] = o9;
  }
  __syncthreads();
  // ---- generator columns summed over the workgroup's proofs in registers (parts != null: max_mn >= 64, the workgroup lies
  // inside ONE group): lane = generator index, the proofs in sequence, limb-wise 64-bit sums of the Montgomery values -- the
  // per-proof rows (272 MB written here and read again by k_reduce_static in a 65 536-proof step) never exist; the group's
  // column sums are then taken over ppw times fewer addends (k_reduce_parts)
  if (parts) {
    // One reduction per (workgroup, column) instead of two per (proof, column): the products a_p b_p of the workgroup's proofs are
    // summed as they are (seventeen 64-bit columns of 29-bit limb products, carries pushed up every eight products) and the SUM is
    // Montgomery-reduced (scalar.h: sc9_mul_acc / sc18_redc) -- 243 multiply-adds per (proof, generator pair) instead of 405 plus
    // two carry / pack / conditional-subtraction tails.  The same residue as the sum of the reduced products, so the group's
    // columns come out bit for bit as before.  Needs every proof of the workgroup to have the same shape (then w e^2 z enters as
    // one sum per workgroup); any other workgroup, and lazy = 0, takes the per-proof products below.
    __shared__ sc s_E[2];
    __shared__ uint32_t s_uniform;
    if (lane == 0) {
      uint32_t uni = lazy, r0 = ~0u, m0 = 0;
      sc E;
      sc_0(E);
      for (uint32_t sub = 0; sub < ppw && uni; sub++) {
        if (s_r[sub] == ~0u) continue;
        if (r0 == ~0u) r0 = s_r[sub], m0 = s_m[sub];
        else if (s_r[sub] != r0 || s_m[sub] != m0) uni = 0;
        const sc e2z = *reinterpret_cast<const sc *>(reinterpret_cast<const uint8_t *>(lanes_lds_raw) + (size_t)sub * per_bytes + (size_t)n9 * sizeof(sc9));
        sc_add(E, E, e2z);
      }
      if (r0 == ~0u) uni = 0;
      s_uniform = uni;
      s_E[0] = E;
      sc_neg(E, E);
      s_E[1] = E;
    }
    __syncthreads();
    if (s_uniform) {
      uint32_t first = 0;
      while (s_r[first] == ~0u) first++;
      const uint32_t r = s_r[first], mn = s_m[first] * n_bits, nhi = 1u << (r - LB);
      const sc Ep = s_E[0], Em = s_E[1];
      for (uint32_t i = lane; i < max_mn; i += 64) {
        uint64_t *o = parts + ((size_t)blockIdx.x * 2 * max_mn + 2 * i) * 8;
        if (i >= mn) {  // zero padding of a smaller statement
#pragma unroll
          for (int q = 0; q < 16; q++) o[q] = 0;
          continue;
        }
        const uint32_t lo = i & (nlo - 1), hi_i = (i >> LB) & (nhi - 1);
        const uint32_t rlo = (~lo) & (nlo - 1), rhi = (~hi_i) & (nhi - 1);
        sc gi, hi;
        {
          uint64_t acc[17], top = 0;
#pragma unroll
          for (int k = 0; k < 17; k++) acc[k] = 0;
          uint32_t cnt = 0;
          for (uint32_t sub = 0; sub < ppw; sub++) {
            if (s_r[sub] == ~0u) continue;
            const sc9 *T = reinterpret_cast<const sc9 *>(reinterpret_cast<const uint8_t *>(lanes_lds_raw) + (size_t)sub * per_bytes);
            sc9_mul_acc(acc, T[lo], T[24 + hi_i]);  // glo x ghi
            if (++cnt == 8) {
              sc18_normalize(acc, top);
              cnt = 0;
            }
          }
          sc18_redc(gi, acc, top);
        }
        {
          uint64_t acc[17], top = 0;
#pragma unroll
          for (int k = 0; k < 17; k++) acc[k] = 0;
          uint32_t cnt = 0;
          for (uint32_t sub = 0; sub < ppw; sub++) {
            if (s_r[sub] == ~0u) continue;
            const sc9 *T = reinterpret_cast<const sc9 *>(reinterpret_cast<const uint8_t *>(lanes_lds_raw) + (size_t)sub * per_bytes);
            sc9_mul_acc(acc, T[16 + rlo], T[24 + 2 * nhi_max + rhi]);  // hlo x shi
            sc9_mul_acc(acc, T[8 + lo], T[24 + nhi_max + hi_i]);       // yn2lo x y2hi
            if (++cnt == 4) {
              sc18_normalize(acc, top);
              cnt = 0;
            }
          }
          sc18_redc(hi, acc, top);
        }
#pragma unroll
        for (int q = 0; q < 8; q++) {  // (limb sums: k_reduce_parts adds the workgroups' and reduces)
          o[q] = (uint64_t)gi.v[q] + Ep.v[q];
          o[8 + q] = (uint64_t)hi.v[q] + Em.v[q];
        }
      }
      return;
    }
    for (uint32_t i = lane; i < max_mn; i += 64) {
      uint64_t ag[8], ah[8];
#pragma unroll
      for (int q = 0; q < 8; q++) ag[q] = ah[q] = 0;
      for (uint32_t sub = 0; sub < ppw; sub++) {
        const uint32_t r = s_r[sub];
        if (r == ~0u) continue;
        const uint32_t mn = s_m[sub] * n_bits;
        if (i >= mn) continue;  // zero padding of a smaller statement
        const uint32_t nhi = 1u << (r - LB);
        const sc9 *T = reinterpret_cast<const sc9 *>(reinterpret_cast<const uint8_t *>(lanes_lds_raw) + (size_t)sub * per_bytes);
        const sc9 *glo = T, *yn2lo = T + 8, *hlo = T + 16;
        const sc9 *ghi = T + 24, *y2hi = ghi + nhi_max, *shi = y2hi + nhi_max;
        const sc e_square_z = *reinterpret_cast<const sc *>(reinterpret_cast<const uint8_t *>(T) + (size_t)n9 * sizeof(sc9));  // w e^2 z
        const uint32_t lo = i & (nlo - 1), hi_i = (i >> LB) & (nhi - 1);
        const uint32_t rlo = (~lo) & (nlo - 1), rhi = (~hi_i) & (nhi - 1);
        sc gi, hi;
        sc9_montmul(gi, glo[lo], ghi[hi_i]);
        sc_add(gi, gi, e_square_z);
        sc9_montmul2(hi, hlo[rlo], shi[rhi], yn2lo[lo], y2hi[hi_i]);  // w (s1e s[mn-1-i] - e^2 d[i] y^(mn-i))
        sc_sub(hi, hi, e_square_z);
#pragma unroll
        for (int q = 0; q < 8; q++) {
          ag[q] += gi.v[q];
          ah[q] += hi.v[q];
        }
      }
      uint64_t *o = parts + ((size_t)blockIdx.x * 2 * max_mn + 2 * i) * 8;
#pragma unroll
      for (int q = 0; q < 8; q++) {
        o[q] = ag[q];
        o[8 + q] = ah[q];
      }
    }
    return;
  }
  // ---- generator rows
  for (uint32_t it = lane; it < ppw * max_mn; it += 64) {
    const uint32_t sub = it / max_mn, i = it - sub * max_mn;
    const uint32_t r = s_r[sub];
    if (r == ~0u) continue;
    const uint32_t mn = s_m[sub] * n_bits;
    const uint32_t nhi = 1u << (r - LB);
    const sc9 *T = reinterpret_cast<const sc9 *>(reinterpret_cast<const uint8_t *>(lanes_lds_raw) + (size_t)sub * per_bytes);
    const sc9 *glo = T, *yn2lo = T + 8, *hlo = T + 16;
    const sc9 *ghi = T + 24, *y2hi = ghi + nhi_max, *shi = y2hi + nhi_max;
    sc *row = rows + (size_t)(p0 + sub) * cols;
    sc gi, hi;
    if (i < mn) {
      const sc e_square_z = *reinterpret_cast<const sc *>(reinterpret_cast<const uint8_t *>(T) + (size_t)n9 * sizeof(sc9));  // w e^2 z
      const uint32_t lo = i & (nlo - 1), hi_i = (i >> LB) & (nhi - 1);
      const uint32_t rlo = (~lo) & (nlo - 1), rhi = (~hi_i) & (nhi - 1);
      sc9_montmul(gi, glo[lo], ghi[hi_i]);
      sc_add(gi, gi, e_square_z);
      sc9_montmul2(hi, hlo[rlo], shi[rhi], yn2lo[lo], y2hi[hi_i]);  // w (s1e s[mn-1-i] - e^2 d[i] y^(mn-i))
      sc_sub(hi, hi, e_square_z);
    } else {
      sc_0(gi);
      sc_0(hi);
    }
    row[2 * i] = gi;
    row[2 * i + 1] = hi;
  }
}

// ---- the weighted part of the scalar block when the generator columns are k_static_gemm's (kernels_static_gemm.h) ----
// Same products on the same operands as k_scalars_lanes' prologue, laid out flat: no per-proof tables go to LDS, so nothing has to
// wait for a workgroup's sixteen proofs -- k_scalars_lanes' prologue alone took 145 us per 65 536 proofs (one wavefront per
// workgroup, 15 KB of LDS each: 2.5 wavefronts per SIMD running three dependent phases).
//   k_gemm_mult : one lane per (proof, multiplier): w (Montgomery), -w e^2 (Montgomery), -w e^2 and -w e^2 y^(mn+1) (canonical),
//                 the weight as it came -> mult[p][5]
//   k_gemm_jobs : one lane per (proof, job), sixteen neighbouring lanes = the sixteen proofs of one block of the digit tables,
//                 a workgroup = four jobs of one block.  Jobs of a proof (uniform shapes: the host checks): 3 * 2^LB low-table
//                 entries (-> digit tables glo | hloR | yn2lo), w e^2 z (-> rows[p][t + 1]), the m + 3 + 2r dynamic scalars
//                 (canonical, -> dyn_out), the t + 1 base columns (-> rows[p][0 .. t])
__global__ void __launch_bounds__(64) k_gemm_mult(const sc *__restrict__ shr, const uint8_t *__restrict__ weights32, uint32_t B,
                                                  sc *__restrict__ mult) {
  const uint32_t it = blockIdx.x * 64u + threadIdx.x, p = it >> 2, which = it & 3u;
  if (p >= B) return;
  sc wc;
  {
    const uint4 *wp = reinterpret_cast<const uint4 *>(weights32 + (size_t)p * 32);  // (mapped host memory: two reads, not thirty-two)
    const uint4 lo4 = wp[0], hi4 = wp[1];
    wc.v[0] = lo4.x, wc.v[1] = lo4.y, wc.v[2] = lo4.z, wc.v[3] = lo4.w;
    wc.v[4] = hi4.x, wc.v[5] = hi4.y, wc.v[6] = hi4.z, wc.v[7] = hi4.w;
  }
  const sc *S = shr + (size_t)p * SH_STRIDE;
  sc *M = mult + (size_t)p * 5;
  sc x;
  if (which == 0u) {
    sc_to_mont(x, wc);
    M[0] = x;
    M[4] = wc;
  } else {
    const sc a = which == 3u ? S[SH_NEG_E2_YNM1] : S[SH_NEG_E2];
    sc_montmul(x, a, wc);                 // canonical: a Montgomery value times a canonical one
    if (which == 1u) sc_to_mont(x, x);    // -w e^2 in Montgomery form (= montmul(-e^2 R, w R))
    M[which] = x;
  }
}

__global__ void __launch_bounds__(64) k_gemm_jobs(const ProofDesc *__restrict__ desc, const sc *__restrict__ tab, const sc *__restrict__ shr,
                                                  const sc *__restrict__ mult, uint32_t n_bits, uint32_t t, uint32_t B, uint32_t nhi_max,
                                                  uint32_t n_jobs, sc *__restrict__ rows /* [B][t + 2] */, sc *__restrict__ dyn_out,
                                                  int8_t *__restrict__ lo_t, uint32_t nblk) {
  const uint32_t lane = threadIdx.x, p = blockIdx.x * SGEMM_BLOCK + (lane & (SGEMM_BLOCK - 1u)), job = blockIdx.y * 4u + (lane >> 4);
  if (p >= B || job >= n_jobs) return;
  const ProofDesc d = desc[p];
  const uint32_t r = d.rounds, m = d.m, LB = lanes_lb(n_bits), nlo = 1u << LB;
  const uint32_t j_e2z = 3u * nlo, j_c = j_e2z + 1u, j_a1 = j_c + m, j_lr = j_a1 + 3u, j_row = j_lr + 2u * r;
  const sc *S = shr + (size_t)p * SH_STRIDE;
  const sc *T = tab + (size_t)p * lanes_tab_stride(nhi_max);
  const sc *M = mult + (size_t)p * 5;
  sc *dyn = dyn_out + d.dyn_off;
  sc *row_base = rows + (size_t)p * (t + 2);
  const sc *src;
  sc *dst = nullptr;
  uint32_t tsel = ~0u, ent = 0;  // digit table and entry, else `dst`
  uint32_t sel, zpow = 0;
  bool neg = false, plain_neg = false;
  if (job < nlo) src = T + job, tsel = 0, ent = job, sel = 0;                                                // glo
  else if (job < 2u * nlo) src = T + 16 + (job - nlo), tsel = 1, ent = (~(job - nlo)) & (nlo - 1u), sel = 0;  // hloR[lo] = hlo[~lo]
  else if (job < 3u * nlo) src = T + 8 + (job - 2u * nlo), tsel = 2, ent = job - 2u * nlo, sel = 1;           // yn2lo
  else if (job == j_e2z) src = S + SH_E2Z, dst = row_base + (t + 1), sel = 0;                                 // w e^2 z
  else if (job < j_a1) src = S + SH_Z2, dst = dyn + (job - j_c), sel = 3, zpow = job - j_c;
  else if (job == j_a1) src = S + SH_E, dst = dyn + m, sel = 4, neg = true;               // A1: -e w
  else if (job == j_a1 + 1) src = S + SH_E, dst = dyn + m + 1, sel = 4, plain_neg = true;  // B: -w
  else if (job == j_a1 + 2) src = S + SH_NEG_E2, dst = dyn + m + 2, sel = 4;               // A: -e^2 w
  else if (job < j_row) {
    const uint32_t k = job - j_lr;
    src = k < r ? S + SH_ESQ(k) : S + SH_ESQINV(k - r);
    dst = dyn + m + 3 + k;
    sel = 2;
  } else {
    const uint32_t k = job - j_row;  // 0: the h base, 1 + k: g base k
    src = k == 0 ? S + SH_HS : S + SH_D1(k - 1);
    dst = row_base + (k == 0 ? t : k - 1);
    sel = 0;
  }
  sc a = *src, x;
  if (zpow) {
    const sc z2 = a;
    for (uint32_t i = 0; i < zpow; i++) sc_montmul(a, a, z2);
  }
  if (neg) sc_neg(a, a);
  const sc mu = M[sel];
  sc_montmul(x, a, mu);
  if (plain_neg) sc_neg(x, mu);
  if (tsel == ~0u) *dst = x;
  else sgemm_store(lo_t + tsel * sgemm_table_bytes(SGEMM_LO_ENTRIES, nblk), ent, nblk, p, x);
}

// Per proof: byte offsets of its points in dynamic-slot order (C_j.., A1, B, A, L.., R..), their owners (bit 31 = statement
// commitment) and the two slot lists k_decompress walks: commitments (decoded at upload) and proof points (every verify).
// Slots of proof p start at dyn_off; its commitments are entries [minval_idx, minval_idx + m) of idx_commit (minval_idx =
// commitments of all earlier proofs), its proof points entries [dyn_off - minval_idx, ..) of idx_proof.
__global__ void __launch_bounds__(64) k_build_slots(const ProofDesc *__restrict__ desc, uint32_t B, uint32_t t,
                                                    uint32_t *__restrict__ src_off, uint32_t *__restrict__ owner,
                                                    uint32_t *__restrict__ idx_commit, uint32_t *__restrict__ idx_proof) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= B) return;
  const ProofDesc d = desc[p];
  uint32_t q = d.dyn_off, ic = d.minval_idx, ip = d.dyn_off - d.minval_idx;
  const uint32_t pA = d.proof_off + 1 + 32 * t;
  for (uint32_t j = 0; j < d.m; j++) {
    src_off[q] = d.commit_off + 32 * j;
    owner[q] = p | 0x80000000u;
    idx_commit[ic++] = q++;
  }
  const uint32_t fixed[3] = {pA + 32, pA + 64, pA};  // A1, B, A
  for (uint32_t j = 0; j < 3; j++) {
    src_off[q] = fixed[j];
    owner[q] = p;
    idx_proof[ip++] = q++;
  }
  for (uint32_t j = 0; j < d.rounds; j++) {
    src_off[q] = pA + 160 + 64 * j;
    owner[q] = p;
    idx_proof[ip++] = q++;
  }
  for (uint32_t j = 0; j < d.rounds; j++) {
    src_off[q] = pA + 160 + 64 * j + 32;
    owner[q] = p;
    idx_proof[ip++] = q++;
  }
}

// MSM term lists of a chunked batch: group g = its static columns, then the dynamic slots [dlo[g], dlo[g+1]) of its proofs.
// term_sidx -> index into scal[] (static part G*cols first, then dynamic), term_pidx -> index into the point tables
// (generator table first, then dynpts).  grid = (ceil(max terms per group / 256), G).
// goff / dlo / gfirst arrive in mapped page-locked host memory ([G + 1] words each: three host-to-device copies were three
// blit kernels per planned call); every workgroup reads the two or three words it needs from there, the first one of a
// group leaves the device copies the other kernels use.
__global__ void __launch_bounds__(256) k_layout_terms(const uint32_t *__restrict__ goff, const uint32_t *__restrict__ dlo,
                                                      const uint32_t *__restrict__ gfirst, uint32_t G,
                                                      uint32_t cols, uint32_t max_mn, uint32_t n_gen, uint32_t table_len, uint32_t split,
                                                      uint32_t *__restrict__ term_sidx, uint32_t *__restrict__ term_pidx,
                                                      uint32_t *__restrict__ goff_dev, uint32_t *__restrict__ dlo_dev,
                                                      uint32_t *__restrict__ gfirst_dev) {
  const uint32_t g = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t t0 = goff[g], t1 = goff[g + 1], ng = (t1 - t0) >> split;  // half-scalar plan: ng low-half terms, then ng high-half ones
  if (i == 0) {
    goff_dev[g] = t0;
    dlo_dev[g] = dlo[g];
    gfirst_dev[g] = gfirst[g];
    if (g + 1 == G) {
      goff_dev[G] = t1;
      dlo_dev[G] = dlo[G];
      gfirst_dev[G] = gfirst[G];
    }
  }
  if (i >= ng) return;
  uint32_t si, pi;
  if (i < cols) {
    si = g * cols + i;
    pi = i < 2 * max_mn ? i : n_gen + (i - 2 * max_mn);
  } else {
    const uint32_t q = dlo[g] + (i - cols);
    si = G * cols + q;
    pi = table_len + q;
  }
  term_sidx[t0 + i] = si;
  term_pidx[t0 + i] = pi;
  if (split) {
    term_sidx[t0 + ng + i] = si | BPP_TERM_HI;
    term_pidx[t0 + ng + i] = pi | BPP_POINT_HI;
  }
}

// Column sums per group: static[g][col] = sum_{p in g} rows[p][col], rows already weighted by k_scalars_lanes (the `+=`
// into gi/hi/g/h_base_scalars of src/range_proof.rs:785-788,999-1020).  One wavefront per (tile of 4 columns, group):
// lane = (proof slot, column in tile), so the four lanes of a proof slot read 4 x 32 = 128 CONTIGUOUS bytes of one proof's
// row (one lane per column would walk rows[] with a stride of cols x 32 = 4 160 bytes); the sixteen proof slots advance
// through the group together.  Limb-wise u64 sums of the Montgomery values, shuffle reduction over the proof slots, one
// Montgomery exit per column: no multiplication per (proof, column).
#define BPP_REDUCE_TILE 4u
__global__ void __launch_bounds__(64) k_reduce_static(const sc *__restrict__ rows, const uint32_t *__restrict__ group_first,
                                                      uint32_t cols, sc *__restrict__ out /* [G][cols] canonical */) {
  const uint32_t g = blockIdx.y, lane = threadIdx.x;
  const uint32_t col = blockIdx.x * BPP_REDUCE_TILE + (lane & (BPP_REDUCE_TILE - 1u)), slot = lane / BPP_REDUCE_TILE;
  const uint32_t p0 = group_first[g], p1 = group_first[g + 1];
  const bool live = col < cols;
  uint64_t acc[8];
#pragma unroll
  for (int i = 0; i < 8; i++) acc[i] = 0;
  for (uint32_t p = p0 + slot; p < p1; p += 64u / BPP_REDUCE_TILE) {
    if (live) {
      const sc v = rows[(size_t)p * cols + col];
#pragma unroll
      for (int i = 0; i < 8; i++) acc[i] += v.v[i];
    }
  }
#pragma unroll
  for (int i = 0; i < 8; i++) {
    for (int off = 32; off >= (int)BPP_REDUCE_TILE; off >>= 1) acc[i] += __shfl_xor(acc[i], off, 64);
  }
  if (slot == 0 && live) {
    uint32_t wds[8];
    uint64_t carry = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      carry += acc[i];
      wds[i] = (uint32_t)carry;
      carry >>= 32;
    }
    // value = lo + carry * 2^256 (Montgomery form of the true sum): from_mont(lo) + carry * 2^256 * R^-1
    sc lo, res, hi, p256;
    sc_const(lo, wds);
    sc_from_mont(res, lo);
    sc_0(hi);
    hi.v[0] = (uint32_t)carry;
    hi.v[1] = (uint32_t)(carry >> 32);
    sc_const(p256, SC_P256);
    sc_montmul(hi, hi, p256);
    sc_add(res, res, hi);
    out[(size_t)g * cols + col] = res;
  }
}

// The same column sums from k_scalars_lanes' per-workgroup partial sums: column < 2 max_mn: sum over the group's workgroups of
// parts[wg][col][8] (64-bit limb sums of up to ppw Montgomery values); the t + 1 base columns: sum over the group's proofs of
// rows[p][t + 1] as before.  group g's proofs are [group_first[g], group_first[g + 1]), its workgroups the ones that hold them
// (every group boundary is a multiple of ppw: checked on the host).
__global__ void __launch_bounds__(64) k_reduce_parts(const uint64_t *__restrict__ parts, const sc *__restrict__ rows_base,
                                                     const uint32_t *__restrict__ group_first, uint32_t cols, uint32_t max_mn, uint32_t t,
                                                     uint32_t ppw, sc *__restrict__ out /* [G][cols] canonical */) {
  const uint32_t g = blockIdx.y, lane = threadIdx.x;
  const uint32_t col = blockIdx.x * BPP_REDUCE_TILE + (lane & (BPP_REDUCE_TILE - 1u)), slot = lane / BPP_REDUCE_TILE;
  const uint32_t p0 = group_first[g], p1 = group_first[g + 1];
  const bool live = col < cols;
  uint64_t acc[8];
#pragma unroll
  for (int i = 0; i < 8; i++) acc[i] = 0;
  if (live && col < 2 * max_mn) {
    const uint32_t w0 = p0 / ppw, w1 = (p1 + ppw - 1) / ppw;
    for (uint32_t w = w0 + slot; w < w1; w += 64u / BPP_REDUCE_TILE) {
      const uint64_t *v = parts + ((size_t)w * 2 * max_mn + col) * 8;
#pragma unroll
      for (int i = 0; i < 8; i++) acc[i] += v[i];
    }
  } else if (live) {
    for (uint32_t p = p0 + slot; p < p1; p += 64u / BPP_REDUCE_TILE) {
      const sc v = rows_base[(size_t)p * (t + 1) + (col - 2 * max_mn)];
#pragma unroll
      for (int i = 0; i < 8; i++) acc[i] += v.v[i];
    }
  }
#pragma unroll
  for (int i = 0; i < 8; i++) {
    for (int off = 32; off >= (int)BPP_REDUCE_TILE; off >>= 1) acc[i] += __shfl_xor(acc[i], off, 64);
  }
  if (slot == 0 && live) {
    uint32_t wds[8];
    uint64_t carry = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {  // limb sums are < 2^32 x (proofs of the group) < 2^57: the running carry stays inside 64 bits
      carry += acc[i];
      wds[i] = (uint32_t)carry;
      carry >>= 32;
    }
    // value = lo + carry * 2^256 (Montgomery form of the true sum): from_mont(lo) + carry * 2^256 * R^-1
    sc lo, res, hi, p256;
    sc_const(lo, wds);
    sc_from_mont(res, lo);
    sc_0(hi);
    hi.v[0] = (uint32_t)carry;
    hi.v[1] = (uint32_t)(carry >> 32);
    sc_const(p256, SC_P256);
    sc_montmul(hi, hi, p256);
    sc_add(res, res, hi);
    out[(size_t)g * cols + col] = res;
  }
}

// Mask recovery (src/range_proof.rs:941-969) with nonce() (src/utils/generic.rs:30-60), one lane per proof.
__device__ __forceinline__ void dev_nonce(sc &out, const uint8_t seed32[32], const char *label, uint32_t llen, int j, int k) {
  uint64_t h[8];
  nonce_hash_words(h, seed32, label, llen, j, k);  // (blake2b.h: the key block is put together in registers)
  uint32_t w[16];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    w[2 * i] = (uint32_t)h[i];
    w[2 * i + 1] = (uint32_t)(h[i] >> 32);
  }
  sc_mont_from_wide_words(out, w);
}

// mask_k = ((d1_k - eta_k - e d_k) e^-2 - alpha_k - sum_j (e_j^2 dL_{j,k} + e_j^-2 dR_{j,k})) (z^2 y^(mn+1))^-1.
// The r + 2 inverses of a proof (e^2, z^2 y^(mn+1), every round challenge) come out of ONE inversion of their product
// (the reference inverts each where it needs it, :950,:958; until round 3 this kernel did the same: r + 2 divsteps inversions
// of ~10 k instructions each were half of its work).  The prefix products wait in LDS, word-major like the PASS-1 sponge: a
// register array indexed by the round would live in scratch memory.
#define BPP_MASK_INV (BPP_MAX_ROUNDS + 1)  // e^2, z^2 y^(mn+1), e_0 .. e_{r-1}: r + 2 <= 13 values
__global__ void __launch_bounds__(64) k_masks(const uint8_t *__restrict__ bytes, const ProofDesc *__restrict__ desc,
                                              const sc *__restrict__ chal, const uint8_t *__restrict__ seeds32,
                                              uint32_t n_bits, uint32_t t, uint32_t cs, uint32_t B,
                                              uint8_t *__restrict__ masks_out /* [B][t][32] */) {
  __shared__ uint32_t pre[BPP_MASK_INV * 8 * 64];  // word w of entry q of lane l at [(q * 8 + w) * 64 + l]
  uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= B) return;
  const ProofDesc d = desc[p];
  if (!(d.flags & 1u)) return;
  const sc *c = chal + (size_t)p * cs;
  const uint8_t *seed = seeds32 + (size_t)p * 32;
  const uint8_t *pd1 = bytes + d.proof_off + 1;
  const uint32_t r = d.rounds, mn = d.m * n_bits;
  if (r + 3 > cs || r + 2 > BPP_MASK_INV) return;  // more rounds than any statement allows: refused on the host, its challenges were not kept
  uint32_t *mine = pre + threadIdx.x;
  auto park = [&](uint32_t q, const sc &v) {
#pragma unroll
    for (int w = 0; w < 8; w++) mine[(q * 8 + w) * 64] = v.v[w];
  };
  auto fetch = [&](uint32_t q, sc &v) {
#pragma unroll
    for (int w = 0; w < 8; w++) v.v[w] = mine[(q * 8 + w) * 64];
  };
  const sc y = c[0], z = c[1], ef = c[2 + r];
  sc e_square, zy, tmp;
  sc_montsq(e_square, ef);
  sc_montsq(tmp, z);
  sc_mont_pow_u32(zy, y, mn + 1);
  sc_montmul(zy, zy, tmp);  // z^2 y^(mn+1)
  // prefix products: pre[q] = x_0 .. x_{q-1} with x = [e^2, z^2 y^(mn+1), e_0, .., e_{r-1}]
  sc acc;
  sc_mont_one(acc);
  for (uint32_t q = 0; q < r + 2; q++) {
    park(q, acc);
    sc x;  // (plain copies: a conditional between two locals and a memory operand makes the compiler take the locals' addresses)
    if (q == 0) x = e_square;
    else if (q == 1) x = zy;
    else x = c[q];  // c[2 + j] for q = 2 + j
    sc_montmul(acc, acc, x);
  }
  sc run;
  sc_mont_invert_vartime(run, acc);
  // walking back: inverse of x_q = run * pre[q]; the round challenges' inverses are squared and parked where their prefix was
  sc e_square_inv, zy_inv;
  for (int q = (int)r + 1; q >= 0; q--) {
    sc pq, inv;
    fetch((uint32_t)q, pq);
    sc_montmul(inv, run, pq);
    sc x;
    if (q == 0) x = e_square;
    else if (q == 1) x = zy;
    else x = c[q];
    sc_montmul(run, run, x);
    if (q == 0) e_square_inv = inv;
    else if (q == 1) zy_inv = inv;
    else {
      sc_montsq(inv, inv);
      park((uint32_t)q, inv);  // e_j^-2, j = q - 2
    }
  }
  for (uint32_t k = 0; k < t; k++) {
    sc mask, n1, n2;
    sc_load_mont(mask, pd1 + 32 * k);
    dev_nonce(n1, seed, "eta", 3, -1, (int)k);
    sc_sub(mask, mask, n1);
    dev_nonce(n2, seed, "d", 1, -1, (int)k);
    sc_montmul(n2, n2, ef);
    sc_sub(mask, mask, n2);
    sc_montmul(mask, mask, e_square_inv);
    dev_nonce(n1, seed, "alpha", 5, -1, (int)k);
    sc_sub(mask, mask, n1);
    for (uint32_t j = 0; j < r; j++) {
      sc ej2, ej2inv;
      sc_montsq(ej2, c[2 + j]);
      fetch(2 + j, ej2inv);
      dev_nonce(n1, seed, "dL", 2, (int)j, (int)k);
      sc_montmul(n1, n1, ej2);
      sc_sub(mask, mask, n1);
      dev_nonce(n2, seed, "dR", 2, (int)j, (int)k);
      sc_montmul(n2, n2, ej2inv);
      sc_sub(mask, mask, n2);
    }
    sc_montmul(mask, mask, zy_inv);
    sc_from_mont(mask, mask);
    uint8_t o[32];
    sc_store_words(o, mask);
    for (int i = 0; i < 32; i++) masks_out[((size_t)p * t + k) * 32 + i] = o[i];
  }
}

// challenges to canonical bytes for the parity trace
__global__ void k_chal_canonical(const sc *__restrict__ chal, uint32_t n, uint8_t *__restrict__ out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  sc v;
  sc_from_mont(v, chal[i]);
  uint8_t o[32];
  sc_store_words(o, v);
  for (int k = 0; k < 32; k++) out[(size_t)i * 32 + k] = o[k];
}

// What an upload brings onto the device arrives in ONE launch as well: up to six pieces (proof bytes of a small batch,
// descriptors, promises, transcript states, seed nonces) are read out of mapped page-locked staging, 16 bytes per lane, and up
// to two word arrays are cleared (status0 / status).  Five hipMemcpyAsync + a memset were six blit kernels in front of every
// small call; the proof bytes of a LARGE batch (tens of MB) still go by DMA.
struct IngestSeg {
  const uint8_t *src;
  uint8_t *dst;
  uint64_t bytes;
};
struct IngestArgs {
  IngestSeg seg[6];
  uint32_t n_seg;
  uint32_t *zero[2];
  uint32_t zero_words;
};
__global__ void __launch_bounds__(256) k_ingest(IngestArgs a) {
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
  for (uint32_t s = 0; s < a.n_seg; s++) {
    const IngestSeg g = a.seg[s];
    size_t done = 0;
    if ((((uintptr_t)g.src | (uintptr_t)g.dst) & 15u) == 0) {
      const size_t n16 = g.bytes / 16;
      for (size_t i = tid; i < n16; i += nth) reinterpret_cast<uint4 *>(g.dst)[i] = reinterpret_cast<const uint4 *>(g.src)[i];
      done = n16 * 16;
    }
    for (size_t i = done + tid; i < g.bytes; i += nth) g.dst[i] = g.src[i];
  }
  for (int z = 0; z < 2; z++)
    if (a.zero[z])
      for (size_t i = tid; i < a.zero_words; i += nth) a.zero[z][i] = 0;
}

// The results of a verification leave the device in ONE launch, straight into mapped page-locked host memory (no copy
// engine, no blit kernel per array): the per-proof status words, the per-group identity flags of the final check
// (src/range_proof.rs:1057-1062) and the recovered masks (:941-969).  status[] is then reset to the batch's initial status
// (statement commitments that do not decode, caller-side PASS-1 findings), so the next verification of the same resident batch
// starts clean without a device-to-device copy in front of its first kernel.
//
// Status words cross the bus only where there is something to say: every workgroup (BPP_STATUS_BLOCK proofs) writes ONE summary
// word -- the OR of its proofs' words -- and its proofs' words only when that is non-zero.  On the accept path a 65 536-proof
// step thus stores 256 + 64 words to host memory instead of 65 536 (round 4: 202 us in flight for this launch, at the tail of
// every step's chain; the host settles the blocks the kernel skipped: settle_status in engine.hip).
#define BPP_STATUS_BLOCK 256u
__global__ void __launch_bounds__(BPP_STATUS_BLOCK) k_results_out(uint32_t *__restrict__ status, const uint32_t *__restrict__ status0,
                                                     uint32_t *__restrict__ status_host, uint32_t *__restrict__ block_any_host, uint32_t B,
                                                     const uint32_t *__restrict__ is_identity, uint32_t *__restrict__ ident_host,
                                                     uint32_t G, const uint4 *__restrict__ masks, uint4 *__restrict__ masks_host,
                                                     uint32_t mask_pieces) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  __shared__ uint32_t any;
  if (threadIdx.x == 0) any = 0;
  __syncthreads();
  uint32_t st = 0;
  if (i < B) {
    st = status[i];
    status[i] = status0[i];
  }
  if (__ballot(st != 0) != 0 && (threadIdx.x & 63u) == 0) atomicOr(&any, 1u);
  __syncthreads();
  if (blockIdx.x * blockDim.x < B) {
    if (threadIdx.x == 0) block_any_host[blockIdx.x] = any;
    if (any && i < B) status_host[i] = st;
  }
  if (is_identity && i < G) ident_host[i] = is_identity[i];
  if (masks)
    for (uint32_t q = i; q < mask_pieces; q += gridDim.x * blockDim.x) masks_host[q] = masks[q];
}

}  // namespace bpp
