// Wavefront-cooperative STROBE-128 / Merlin (device only): ONE transcript per 64-lane wavefront.
//
// merlin.h runs a whole sponge in one lane (right when thousands of independent transcripts are replayed side by side,
// k_transcripts).  The prover's Fiat-Shamir step is different: a round cannot start before the previous round's
// challenge exists, so per proof it is a pure latency chain of ~15 Keccak-f per round, and one lane needs ~26 us per
// permutation.  Here the 200-byte state lives in LDS and the wavefront shares the work:
//   * absorb / overwrite / squeeze touch up to 64 state bytes per step (lane k <-> byte pos + k),
//   * Keccak-f[1600] is wkeccak.h's one-exchange form (round 6): the 25 words are read out of the LDS state as bit-interleaved
//     32-bit halves, one per lane on 50 lanes, permuted in registers with ONE LDS exchange per round (theta + rho for the source
//     word of a lane's pi destination, column parities summed by ds_xor_b32, chi's neighbours by DPP), and written back as words.
//     19 instruction slots per round plus ~80 for the two conversions, against ~60 per round for the form it replaces (a 64-bit
//     word per lane on 25 lanes, two exchanges per round: kept below as keccak_f1600_wave25 for the microbenchmark): 4.5 -> ~1.7
//     us per permutation for a lone wavefront, and a third of the instructions.
// pos / pos_begin / cur_flags are wave-uniform registers.  Byte-for-byte the same sponge as merlin.h (tests compare the
// prover's output with the oracle).  Must be called by all 64 lanes of the wavefront that owns the transcript.
#pragma once
#include "merlin.h"
#include "wkeccak.h"

namespace bpp {

// One transcript lives in ONE wavefront, whatever the size of the workgroup around it (the prover's round kernel runs two
// transcripts side by side in two wavefronts of one workgroup): a lane is its position in the wavefront, and the synchronisation
// between the steps of a sponge operation is wavefront-local.  The LDS operations of one wavefront execute in issue order, so what
// such a point has to guarantee is that the COMPILER keeps the accesses in program order and re-reads LDS afterwards: a release /
// acquire fence pair at workgroup scope (it also waits for the outstanding LDS operations) around a wavefront barrier -- what
// __syncthreads() is, minus the s_barrier that would tie the workgroup's other wavefronts to this one's step count.
__device__ __forceinline__ uint32_t ws_lane() { return threadIdx.x & 63u; }
__device__ __forceinline__ void ws_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

struct WStrobe {
  uint64_t *st;  // 25 words in LDS
  uint32_t pos, pos_begin, cur_flags;
};

// per-lane constants of the cooperative permutation (lane i <-> state word a[x + 5y], i = x + 5y)
struct KeccakLanes {
  int xm1, xp1;        // first word (y = 0) of column x-1 / x+1
  int pinv;            // pi: this lane's new word comes from lane pinv (already rotated there)
  int pn1, pn2;        // the same for the row neighbours x+1, x+2 (chi)
  uint32_t rot;        // rho offset of this lane's word
  bool lane0, owner;   // owner: lanes 0..24 (the others shadow lane 0 and never write)
  WkLanes W;           // wkeccak.h's lane constants
  uint32_t *img;       // its exchange image in LDS: WK_LDS_DWORDS_RC dwords, ONE PER WAVEFRONT that runs a sponge
};

// img: WK_LDS_DWORDS_RC dwords of LDS for this wavefront's permutations (the table of iota's constants behind the image is filled
// here: all 64 lanes call; the sponge operations that follow start with a wavefront barrier)
__device__ __forceinline__ KeccakLanes keccak_lanes(uint32_t *img) {
  const uint8_t ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
  const uint32_t l = ws_lane();
  const int i = l < 25 ? (int)l : 0;  // lanes 25..63 shadow lane 0 and never write back
  const int x = i % 5, y = i / 5;
  KeccakLanes k;
  k.xm1 = (x + 4) % 5;
  k.xp1 = (x + 1) % 5;
  // b[ys + 5 * ((2 xs + 3 ys) % 5)] = rot(a[xs + 5 ys])  ->  for destination (x, y): ys = x, xs = 3 (y - 3 x) mod 5
  auto src = [](int dx, int dy) { return (3 * ((dy + 15 - 3 * dx) % 5)) % 5 + 5 * dx; };
  k.pinv = src(x, y);
  k.pn1 = src((x + 1) % 5, y);
  k.pn2 = src((x + 2) % 5, y);
  uint32_t r = 0;
#pragma unroll
  for (int q = 0; q < 25; q++) r = (q == i) ? ROT[q] : r;
  k.rot = r;
  k.lane0 = i == 0;
  k.owner = l < 25;
  k.W = wk_lanes(img);
  k.img = img;
  wk_rc_table_init(img);
  return k;
}

__device__ __forceinline__ void keccak_f1600_wave25(uint64_t *st, const KeccakLanes &K) {
  const uint64_t RC[24] = {
      0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808AULL, 0x8000000080008000ULL,
      0x000000000000808BULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
      0x000000000000008AULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000AULL,
      0x000000008000808BULL, 0x800000000000008BULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
      0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800AULL, 0x800000008000000AULL,
      0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
  // volatile: every exchange below is a real LDS access in program order (the hardware runs one wavefront's LDS
  // operations in issue order, so a read issued before the next write still sees the previous round's words)
  typedef __attribute__((address_space(3))) uint64_t lds_u64;
  volatile lds_u64 *w = (volatile lds_u64 *)st;  // st points into a __shared__ object
  const uint32_t self = ws_lane() < 25 ? ws_lane() : 0;
  uint64_t a = w[self];  // the caller's barrier made the absorbed bytes visible; the state itself is the first exchange
#pragma unroll 1
  for (int rnd = 0; rnd < 24; rnd++) {
    if (rnd != 0 && K.owner) w[self] = a;
    const volatile lds_u64 *m = w + K.xm1, *p = w + K.xp1;  // theta
    const uint64_t cm = kk_xor3(kk_xor3(m[0], m[5], m[10]), m[15], m[20]);
    const uint64_t cp = kk_xor3(kk_xor3(p[0], p[5], p[10]), p[15], p[20]);
    a = kk_xor3(a, cm, kk_rol<1>(cp));
    const uint64_t r = (a << K.rot) | (a >> ((64u - K.rot) & 63u));  // rho (rot = 0: a | a)
    if (K.owner) w[self] = r;
    const uint64_t b = w[K.pinv], b1 = w[K.pn1], b2 = w[K.pn2];  // pi + chi
    a = kk_chi(b, b1, b2);
    if (K.lane0) a ^= RC[rnd];  // iota
  }
  if (K.owner) w[self] = a;
}

// the caller's barrier made the absorbed bytes visible; the words are back in st[] (and visible) on return
__device__ __forceinline__ void keccak_f1600_wave(uint64_t *st, const KeccakLanes &K) {
  uint32_t a = wk_load(st, K.W);
  a = wk_keccak_f1600_t(a, K.W);
  wk_store(st, K.img, a, K.W);
}

__device__ __forceinline__ void ws_run_f(WStrobe &s, const KeccakLanes &K) {
  uint8_t *b = (uint8_t *)s.st;
  ws_sync();
  if (ws_lane() == 0) {
    b[s.pos] ^= (uint8_t)s.pos_begin;
    b[s.pos + 1] ^= 0x04;
    b[BPP_STROBE_R + 1] ^= 0x80;
  }
  ws_sync();
  keccak_f1600_wave(s.st, K);
  ws_sync();
  s.pos = 0;
  s.pos_begin = 0;
}

// byte_at(k) = k-th byte of the message, evaluated by the lane that owns the state byte
template <class F>
__device__ __forceinline__ void ws_absorb(WStrobe &s, const KeccakLanes &K, F byte_at, uint32_t n) {
  uint8_t *b = (uint8_t *)s.st;
  uint32_t off = 0;
  while (off < n) {
    const uint32_t room = BPP_STROBE_R - s.pos, chunk = n - off < room ? n - off : room;
    for (uint32_t k = ws_lane(); k < chunk; k += 64) b[s.pos + k] ^= byte_at(off + k);
    s.pos += chunk;
    off += chunk;
    if (s.pos == BPP_STROBE_R) ws_run_f(s, K);
  }
}
template <class F>
__device__ __forceinline__ void ws_overwrite(WStrobe &s, const KeccakLanes &K, F byte_at, uint32_t n) {
  uint8_t *b = (uint8_t *)s.st;
  uint32_t off = 0;
  while (off < n) {
    const uint32_t room = BPP_STROBE_R - s.pos, chunk = n - off < room ? n - off : room;
    for (uint32_t k = ws_lane(); k < chunk; k += 64) b[s.pos + k] = byte_at(off + k);
    s.pos += chunk;
    off += chunk;
    if (s.pos == BPP_STROBE_R) ws_run_f(s, K);
  }
}
// out: LDS (or global) bytes, visible to every lane after the trailing barrier
__device__ __forceinline__ void ws_squeeze(WStrobe &s, const KeccakLanes &K, uint8_t *out, uint32_t n) {
  uint8_t *b = (uint8_t *)s.st;
  uint32_t off = 0;
  ws_sync();
  while (off < n) {
    const uint32_t room = BPP_STROBE_R - s.pos, chunk = n - off < room ? n - off : room;
    for (uint32_t k = ws_lane(); k < chunk; k += 64) {
      out[off + k] = b[s.pos + k];
      b[s.pos + k] = 0;
    }
    s.pos += chunk;
    off += chunk;
    if (s.pos == BPP_STROBE_R) ws_run_f(s, K);
  }
  ws_sync();
}

__device__ __forceinline__ void ws_begin_op(WStrobe &s, const KeccakLanes &K, uint32_t flags, bool more) {
  if (more) return;
  const uint32_t old_begin = s.pos_begin;
  s.pos_begin = s.pos + 1;
  s.cur_flags = flags;
  ws_absorb(s, K, [=](uint32_t k) { return (uint8_t)(k == 0 ? old_begin : flags); }, 2);
  if ((flags & (BPP_FLAG_C | BPP_FLAG_K)) && s.pos != 0) ws_run_f(s, K);
}

struct BytesAt {  // message bytes behind a pointer (global memory, constant strings)
  const uint8_t *p;
  __device__ __forceinline__ uint8_t operator()(uint32_t k) const { return p[k]; }
};
struct WordAt {  // little-endian bytes of an integer held in registers
  uint64_t v;
  __device__ __forceinline__ uint8_t operator()(uint32_t k) const { return (uint8_t)(v >> (8 * k)); }
};
struct ZeroAt {
  __device__ __forceinline__ uint8_t operator()(uint32_t) const { return 0; }
};

template <class F>
__device__ __forceinline__ void wm_append_message(WStrobe &s, const KeccakLanes &K, const char *label, uint32_t llen, F msg,
                                                  uint32_t mlen) {
  ws_begin_op(s, K, BPP_FLAG_M | BPP_FLAG_A, false);
  ws_absorb(s, K, BytesAt{(const uint8_t *)label}, llen);
  ws_absorb(s, K, WordAt{mlen}, 4);  // meta_ad(len, more = true)
  ws_begin_op(s, K, BPP_FLAG_A, false);
  ws_absorb(s, K, msg, mlen);
}
__device__ __forceinline__ void wm_append_u64(WStrobe &s, const KeccakLanes &K, const char *label, uint32_t llen, uint64_t x) {
  wm_append_message(s, K, label, llen, WordAt{x}, 8);
}
__device__ __forceinline__ void wm_challenge_bytes(WStrobe &s, const KeccakLanes &K, const char *label, uint32_t llen, uint8_t *out,
                                                   uint32_t n) {
  ws_begin_op(s, K, BPP_FLAG_M | BPP_FLAG_A, false);
  ws_absorb(s, K, BytesAt{(const uint8_t *)label}, llen);
  ws_absorb(s, K, WordAt{n}, 4);
  ws_begin_op(s, K, BPP_FLAG_I | BPP_FLAG_A | BPP_FLAG_C, false);
  ws_squeeze(s, K, out, n);
}
// TranscriptRngBuilder::rekey_with_witness_bytes / finalize, TranscriptRng::fill_bytes
template <class F>
__device__ __forceinline__ void wm_rng_rekey(WStrobe &rng, const KeccakLanes &K, const char *label, uint32_t llen, F w, uint32_t wlen) {
  ws_begin_op(rng, K, BPP_FLAG_M | BPP_FLAG_A, false);
  ws_absorb(rng, K, BytesAt{(const uint8_t *)label}, llen);
  ws_absorb(rng, K, WordAt{wlen}, 4);
  ws_begin_op(rng, K, BPP_FLAG_A | BPP_FLAG_C, false);
  ws_overwrite(rng, K, w, wlen);
}
template <class F>
__device__ __forceinline__ void wm_rng_finalize(WStrobe &rng, const KeccakLanes &K, F random32) {
  ws_begin_op(rng, K, BPP_FLAG_M | BPP_FLAG_A, false);
  ws_absorb(rng, K, BytesAt{(const uint8_t *)"rng"}, 3);
  ws_begin_op(rng, K, BPP_FLAG_A | BPP_FLAG_C, false);
  ws_overwrite(rng, K, random32, 32);
}
__device__ __forceinline__ void wm_rng_fill(WStrobe &rng, const KeccakLanes &K, uint8_t *out, uint32_t n) {
  ws_begin_op(rng, K, BPP_FLAG_M | BPP_FLAG_A, false);
  ws_absorb(rng, K, WordAt{n}, 4);
  ws_begin_op(rng, K, BPP_FLAG_I | BPP_FLAG_A | BPP_FLAG_C, false);
  ws_squeeze(rng, K, out, n);
}

// state <-> the per-proof Strobe kept in HBM between kernels
__device__ __forceinline__ void ws_load(WStrobe &s, uint64_t *lds25, const Strobe &g) {
  if (ws_lane() < 25) lds25[ws_lane()] = g.st[ws_lane()];
  s.st = lds25;
  s.pos = g.pos;
  s.pos_begin = g.pos_begin;
  s.cur_flags = g.cur_flags;
  ws_sync();
}
__device__ __forceinline__ void ws_store(Strobe &g, const WStrobe &s) {
  ws_sync();
  if (ws_lane() < 25) g.st[ws_lane()] = s.st[ws_lane()];
  if (ws_lane() == 0) {
    g.pos = s.pos;
    g.pos_begin = s.pos_begin;
    g.cur_flags = s.cur_flags;
  }
}
__device__ __forceinline__ void ws_clone(WStrobe &dst, uint64_t *lds25, const WStrobe &src) {
  ws_sync();
  if (ws_lane() < 25) lds25[ws_lane()] = src.st[ws_lane()];
  dst.st = lds25;
  dst.pos = src.pos;
  dst.pos_begin = src.pos_begin;
  dst.cur_flags = src.cur_flags;
  ws_sync();
}

}  // namespace bpp
