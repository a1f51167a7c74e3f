// Keccak-f[1600] by ONE wavefront with the lowest latency this chip gives a lone wavefront (device only).
//
// Why: the batch-weight chain (src/range_proof.rs:811,849,853,894; src/protocols/scalar_protocol.rs:23-30) is a strictly
// sequential sponge of 1.27 permutations per proof.  On the device it is one wavefront per reference batch, and a lone
// wavefront pays ~4 cycles per instruction it issues, whatever the instruction, plus ~50 cycles for every LDS round trip it
// has to wait for.  wstrobe.h's permutation (one 64-bit word per lane, 25 lanes, two LDS exchanges per round) issues ~60
// instructions per round.  This form issues ~22 and waits for ONE exchange:
//   * the state is kept BIT-INTERLEAVED (even bits of a word in one 32-bit half, odd bits in the other), one HALF per lane,
//     50 lanes: a 64-bit rotation is one v_alignbit_b32 per half (by r/2; an odd r swaps the halves, which costs nothing
//     here: the lane reads the other half's address), xor3 / chi are one v_bitop3_b32;
//   * a lane stands at a DESTINATION (x', y', h') of pi and does theta + rho for its SOURCE word: one exchange delivers
//     the source half and the ten halves of the two neighbouring columns (columns are contiguous in LDS: a ds_read_b128
//     and a ds_read_b32 each), so rho's result is born where pi wants it;
//   * the five lanes of a row (x' = 0..4 of one (y', h')) are neighbours in one 16-lane DPP row, so chi's two operands come
//     from row_shl / row_shr moves instead of a second exchange.
// Lane map: group g = 5 h' + y' (ten groups of five lanes, three groups per DPP row): lane = 16 (g / 3) + 5 (g % 3) + x'.
// LDS image of the state between rounds: dword [(2 x + h) * 8 + y] (column-major, columns padded to 8 dwords).
#pragma once
#include "merlin.h"

namespace bpp {

// ---- bit interleaving (XKCP's toBitInterleaving): low 16 bits = even bits of x, high 16 = odd bits
BPP_HD constexpr uint32_t wk_deint32(uint32_t x) {
  uint32_t t = (x ^ (x >> 1)) & 0x22222222u;
  x ^= t ^ (t << 1);
  t = (x ^ (x >> 2)) & 0x0C0C0C0Cu;
  x ^= t ^ (t << 2);
  t = (x ^ (x >> 4)) & 0x00F000F0u;
  x ^= t ^ (t << 4);
  t = (x ^ (x >> 8)) & 0x0000FF00u;
  x ^= t ^ (t << 8);
  return x;
}
BPP_HD constexpr uint32_t wk_int32(uint32_t x) {  // the inverse
  uint32_t t = (x ^ (x >> 8)) & 0x0000FF00u;
  x ^= t ^ (t << 8);
  t = (x ^ (x >> 4)) & 0x00F000F0u;
  x ^= t ^ (t << 4);
  t = (x ^ (x >> 2)) & 0x0C0C0C0Cu;
  x ^= t ^ (t << 2);
  t = (x ^ (x >> 1)) & 0x22222222u;
  x ^= t ^ (t << 1);
  return x;
}
// half h (0: even bits, 1: odd bits) of the 64-bit word w
BPP_HD constexpr uint32_t wk_half(uint64_t w, uint32_t h) {
  const uint32_t a = wk_deint32((uint32_t)w), b = wk_deint32((uint32_t)(w >> 32));
  return h ? ((a >> 16) | (b & 0xFFFF0000u)) : ((a & 0xFFFFu) | (b << 16));
}
BPP_HD constexpr uint64_t wk_word(uint32_t even, uint32_t odd) {
  const uint32_t lo = wk_int32((even & 0xFFFFu) | (odd << 16)), hi = wk_int32((even >> 16) | (odd & 0xFFFF0000u));
  return (uint64_t)lo | ((uint64_t)hi << 32);
}

#define WK_LDS_DWORDS (80 + 64)  // the state image (10 columns x 8) and one scratch dword per lane for the idle lanes

struct WkLanes {
  uint32_t wr, src, cm, cp;  // LDS byte addresses: own half, source half, base of column xs-1 (half hs), of column xs+1 (half 1-hs)
  uint32_t cp_sh, rho_sh;    // v_alignbit_b32 shift amounts (rol32(v, n) = alignbit(v, v, (32 - n) & 31))
  uint32_t wrap1, wrap2;     // all-ones where the row neighbour x'+1 / x'+2 wraps around the group of five
  uint32_t word, half;       // this lane's state word (x' + 5 y') and half; word = 0xffffffff for the 14 idle lanes
  uint32_t rc_mask;          // all-ones in the two lanes that hold word 0
};

__device__ __forceinline__ uint32_t wk_lds_addr(const uint32_t *lds) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint32_t *)lds;
}

__device__ __forceinline__ WkLanes wk_lanes(const uint32_t *lds) {
  const uint8_t ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
  const uint32_t lane = threadIdx.x & 63u, row = lane >> 4, col = lane & 15u;
  const uint32_t g = 3 * row + col / 5, x = col % 5;
  const bool owner = col < 15 && g < 10;
  const uint32_t h = owner ? g / 5 : 0, y = owner ? g % 5 : 0;
  // b[x' + 5 y'] = rol(a[xs + 5 ys], ROT[xs + 5 ys]) with ys = x', xs = 3 (y' - 3 x') mod 5
  const uint32_t ys = x, xs = (3 * ((y + 15 - 3 * x) % 5)) % 5;
  uint32_t r = 0;
#pragma unroll
  for (int q = 0; q < 25; q++) r = (q == (int)(xs + 5 * ys)) ? ROT[q] : r;
  // even rotation 2k: half h of the result = rol32(half h, k); odd 2k+1: even half = rol32(odd half, k + 1), odd half = rol32(even half, k)
  const uint32_t odd = r & 1u, hs = h ^ odd, sh = (r >> 1) + (odd & (h ^ 1u));
  const uint32_t base = wk_lds_addr(lds);
  WkLanes L;
  L.wr = base + 4u * (owner ? (2 * x + h) * 8 + y : 80 + lane);
  L.src = base + 4u * ((2 * xs + hs) * 8 + ys);
  L.cm = base + 4u * ((2 * ((xs + 4) % 5) + hs) * 8);
  L.cp = base + 4u * ((2 * ((xs + 1) % 5) + (hs ^ 1u)) * 8);
  L.cp_sh = hs == 0 ? 31u : 0u;  // D[x] = C[x-1] ^ rol64(C[x+1], 1): even half takes rol32(odd half of C[x+1], 1), odd half the even half as is
  L.rho_sh = (32u - sh) & 31u;
  L.wrap1 = x >= 4 ? 0xffffffffu : 0u;
  L.wrap2 = x >= 3 ? 0xffffffffu : 0u;
  L.word = owner ? x + 5 * y : 0xffffffffu;
  L.half = h;
  L.rc_mask = (owner && x == 0 && y == 0) ? 0xffffffffu : 0u;
  return L;
}

// iota's constants, interleaved: [round][half]
__device__ __constant__ const uint32_t WK_RC[24][2] = {
#define WK_RC_ROW(c) {wk_half(c, 0), wk_half(c, 1)}
    WK_RC_ROW(0x0000000000000001ULL), WK_RC_ROW(0x0000000000008082ULL), WK_RC_ROW(0x800000000000808AULL), WK_RC_ROW(0x8000000080008000ULL),
    WK_RC_ROW(0x000000000000808BULL), WK_RC_ROW(0x0000000080000001ULL), WK_RC_ROW(0x8000000080008081ULL), WK_RC_ROW(0x8000000000008009ULL),
    WK_RC_ROW(0x000000000000008AULL), WK_RC_ROW(0x0000000000000088ULL), WK_RC_ROW(0x0000000080008009ULL), WK_RC_ROW(0x000000008000000AULL),
    WK_RC_ROW(0x000000008000808BULL), WK_RC_ROW(0x800000000000008BULL), WK_RC_ROW(0x8000000000008089ULL), WK_RC_ROW(0x8000000000008003ULL),
    WK_RC_ROW(0x8000000000008002ULL), WK_RC_ROW(0x8000000000000080ULL), WK_RC_ROW(0x000000000000800AULL), WK_RC_ROW(0x800000008000000AULL),
    WK_RC_ROW(0x8000000080008081ULL), WK_RC_ROW(0x8000000000008080ULL), WK_RC_ROW(0x0000000080000001ULL), WK_RC_ROW(0x8000000080008008ULL)
#undef WK_RC_ROW
};

// the 24 per-lane iota words (zero outside the two lanes of word 0): 24 registers, so a round's iota is one v_xor_b32
struct WkRc {
  uint32_t v[24];
};
__device__ __forceinline__ WkRc wk_rc(const WkLanes &L) {
  WkRc R;
#pragma unroll
  for (int r = 0; r < 24; r++) R.v[r] = WK_RC[r][L.half] & L.rc_mask;
  return R;
}

typedef uint32_t wk_u32x4 __attribute__((ext_vector_type(4)));

// `a`: this lane's half of the state (any value in the idle lanes).  All 64 lanes must call it.  Uses lds[0 .. WK_LDS_DWORDS).
__device__ __forceinline__ uint32_t wk_keccak_f1600(uint32_t a, const WkLanes &L, const WkRc &R) {
#if defined(__HIP_DEVICE_COMPILE__)  // (gfx950 builtins and instructions: nothing for the host pass to parse)
#pragma unroll
  for (int rnd = 0; rnd < 24; rnd++) {
    uint32_t s, m4, p4;
    wk_u32x4 m, p;
    // one wavefront's LDS operations execute in issue order: the reads below see every lane's write of this round, and the next
    // round's write comes after them
    asm volatile(
        "ds_write_b32 %[wa], %[a]\n\t"
        "ds_read_b32 %[s], %[sa]\n\t"
        "ds_read_b128 %[m], %[ma]\n\t"
        "ds_read_b32 %[m4], %[ma] offset:16\n\t"
        "ds_read_b128 %[p], %[pa]\n\t"
        "ds_read_b32 %[p4], %[pa] offset:16\n\t"
        "s_waitcnt lgkmcnt(0)"
        : [s] "=&v"(s), [m] "=&v"(m), [m4] "=&v"(m4), [p] "=&v"(p), [p4] "=&v"(p4)
        : [wa] "v"(L.wr), [a] "v"(a), [sa] "v"(L.src), [ma] "v"(L.cm), [pa] "v"(L.cp)
        : "memory");
    const uint32_t cm = __builtin_amdgcn_bitop3_b32(__builtin_amdgcn_bitop3_b32(m.x, m.y, m.z, 0x96), m.w, m4, 0x96);
    const uint32_t cp = __builtin_amdgcn_bitop3_b32(__builtin_amdgcn_bitop3_b32(p.x, p.y, p.z, 0x96), p.w, p4, 0x96);
    const uint32_t in = __builtin_amdgcn_bitop3_b32(s, cm, __builtin_amdgcn_alignbit(cp, cp, L.cp_sh), 0x96);  // theta
    const uint32_t b = __builtin_amdgcn_alignbit(in, in, L.rho_sh);                                           // rho (pi: by position)
    // chi: b[x'+1], b[x'+2] of the same row = the next two lanes of the group of five, wrapping around
    const uint32_t n1 = __builtin_amdgcn_update_dpp(0u, b, 0x101, 0xf, 0xf, true);  // row_shl:1  (lane + 1)
    const uint32_t w1 = __builtin_amdgcn_update_dpp(0u, b, 0x114, 0xf, 0xf, true);  // row_shr:4  (lane - 4)
    const uint32_t n2 = __builtin_amdgcn_update_dpp(0u, b, 0x102, 0xf, 0xf, true);  // row_shl:2
    const uint32_t w2 = __builtin_amdgcn_update_dpp(0u, b, 0x113, 0xf, 0xf, true);  // row_shr:3
    const uint32_t b1 = L.wrap1 ? w1 : n1, b2 = L.wrap2 ? w2 : n2;
    a = __builtin_amdgcn_bitop3_b32(b, b1, b2, 0xD2) ^ R.v[rnd];  // chi, iota
  }
#endif
  return a;
}

// state words (64-bit, the sponge's byte order) in LDS <-> this lane's interleaved half
__device__ __forceinline__ uint32_t wk_load(const uint64_t *st25, const WkLanes &L) {
  const uint64_t w = st25[L.word < 25 ? L.word : 0];
  return wk_half(w, L.half);
}
// every lane passes its half; lanes of the even halves write the words back (the odd half comes over a DPP-free exchange
// through the scratch image: one LDS round trip, only at the sponge's byte-level operations)
__device__ __forceinline__ void wk_store(uint64_t *st25, uint32_t *lds, uint32_t a, const WkLanes &L) {
  if (L.word < 25) lds[L.word * 2 + L.half] = a;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  const uint32_t lane = threadIdx.x & 63u;
  uint64_t w = 0;
  if (lane < 25) w = wk_word(lds[2 * lane], lds[2 * lane + 1]);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  if (lane < 25) st25[lane] = w;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

}  // namespace bpp
