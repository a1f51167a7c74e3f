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
//     the source half and the parities of the two neighbouring columns, so rho's result is born where pi wants it;
//   * the column parities are summed by the LDS itself: every lane XORs its half into its column's accumulator with a
//     ds_xor_b32 (five lanes per accumulator, ten accumulators, a fresh zeroed set per round: six stores per permutation
//     clear all 24 sets) -- three reads per lane and round instead of eleven, and no XOR tree in the lanes;
//   * the five lanes of a row (x' = 0..4 of one (y', h')) are neighbours in one 16-lane DPP row, so chi's two operands come
//     from row_shl / row_shr moves instead of a second exchange.
// Lane map: group g = 5 h' + y' (ten groups of five lanes, three groups per DPP row): lane = 16 (g / 3) + 5 (g % 3) + x'.
// LDS image: dword [2 (x + 5 y) + h] holds a half of the state (64 .. 127: one scratch dword per lane for the 14 idle lanes);
// dword [128 + 16 r + 2 x + h] is round r's parity accumulator of column x, half h (10 .. 15 of a set: the idle lanes' dump).
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

#define WK_LDS_DWORDS (128 + 24 * 16)  // state halves, the idle lanes' scratch dwords, 24 sets of parity accumulators
// ... and, for callers that cannot spare 24 registers for iota's constants (the sponges of wstrobe.h inside the prover's round
// kernel), the constants as a table behind the image: [24][2] words for the two lanes of word 0, [24][2] zeros for everybody else
#define WK_LDS_DWORDS_RC (WK_LDS_DWORDS + 96)

struct WkLanes {
  uint32_t wr, src, acc, cm, cp;  // LDS byte addresses: own half, source half; in accumulator set 0: own column, column xs-1 (half hs), column xs+1 (half 1-hs)
  uint32_t zero;             // 4 x lane: the six clearing stores cover dwords 128 + lane + 64 k
  uint32_t rc;               // iota from the table behind the image (wk_rc_table_init): this lane's column of it
  uint32_t cp_sh, rho_sh;    // v_alignbit_b32 shift amounts (rol32(v, n) = alignbit(v, v, (32 - n) & 31))
  uint64_t nowrap1, wrap2;   // wave masks: lanes whose row neighbour x'+1 is the next lane (x' < 4) / whose x'+2 wraps around (x' >= 3)
  uint32_t word, half;       // this lane's state word (x' + 5 y') and half; word = 0xffffffff for the 14 idle lanes
  uint32_t rc_mask;          // all-ones in the two lanes that hold word 0
};

#if defined(__HIPCC__)  // (everything below is device code; the interleaving helpers above also serve the host tests)
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
  L.wr = base + 4u * (owner ? 2 * (x + 5 * y) + h : 64 + lane);
  L.src = base + 4u * (2 * (xs + 5 * ys) + hs);
  L.acc = base + 4u * (128 + (owner ? 2 * x + h : 10 + (lane & 3u)));
  L.cm = base + 4u * (128 + 2 * ((xs + 4) % 5) + hs);
  L.cp = base + 4u * (128 + 2 * ((xs + 1) % 5) + (hs ^ 1u));
  L.zero = base + 4u * (128 + lane);
  L.rc = base + 4u * (WK_LDS_DWORDS + ((owner && x == 0 && y == 0) ? h : 48u));
  L.cp_sh = hs == 0 ? 31u : 0u;  // D[x] = C[x-1] ^ rol64(C[x+1], 1): even half takes rol32(odd half of C[x+1], 1), odd half the even half as is
  L.rho_sh = (32u - sh) & 31u;
  L.nowrap1 = __ballot(x < 4);
  L.wrap2 = __ballot(x >= 3);
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

// Six rounds (R0 is a template parameter: the accumulator sets' offsets are immediates of the LDS instructions).  Written out as
// instructions: the compiler pads around an asm statement it cannot see into, and v_cndmask_b32 with a DPP operand (one
// instruction for "the next lane's value, or the lane four below at the end of the group") is not something it emits.  vcc (the
// lanes with x' < 4) is set once per six rounds.
#define WK_ROUND_ASM(K)                                                                                               \
  "ds_write_b32 %[wa], %[a]\n\t"                                                                                      \
  "ds_xor_b32 %[ca], %[a] offset:%[off" #K "]\n\t"                                                                    \
  "ds_read_b32 %[s], %[sa]\n\t"                                                                                       \
  "ds_read_b32 %[cm], %[ma] offset:%[off" #K "]\n\t"                                                                  \
  "ds_read_b32 %[cp], %[pa] offset:%[off" #K "]\n\t"                                                                  \
  "s_waitcnt lgkmcnt(0)\n\t"                                                                                          \
  "v_alignbit_b32 %[cp], %[cp], %[cp], %[cpsh]\n\t"                                                                   \
  "v_bitop3_b32 %[s], %[s], %[cm], %[cp] bitop3:0x96\n\t"                  /* theta */                                \
  "v_alignbit_b32 %[s], %[s], %[s], %[rsh]\n\t"                            /* rho (pi: by position): s = b */         \
  "s_nop 1\n\t"                                      /* a DPP read of a register needs two wait states after its write */ \
  "v_mov_b32_dpp %[cm], %[s] row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" /* b[x'+1]: lane + 1 ... */        \
  "v_mov_b32_dpp %[cp], %[s] row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" /* b[x'+2]: lane + 2 ... */        \
  "v_mov_b32_dpp %[w2], %[s] row_shr:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" /* ... or lane - 3 (x' = 3, 4) */  \
  "v_cndmask_b32_dpp %[cm], %[s], %[cm], vcc row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" /* or lane - 4 */  \
  "v_cndmask_b32_e64 %[cp], %[cp], %[w2], %[wr2]\n\t"                                                                 \
  "v_bitop3_b32 %[s], %[s], %[cm], %[cp] bitop3:0xd2\n\t"                  /* chi */                                  \
  "v_xor_b32_e32 %[a], %[s], %[rc" #K "]\n\t"                              /* iota */

template <int R0>
__device__ __forceinline__ uint32_t wk_rounds6(uint32_t a, const WkLanes &L, const WkRc &R) {
#if defined(__HIP_DEVICE_COMPILE__)  // (gfx950 instructions: nothing for the host pass to parse)
  uint32_t s, cm, cp, w2;
  asm volatile("s_mov_b64 vcc, %[nw1]\n\t" WK_ROUND_ASM(0) WK_ROUND_ASM(1) WK_ROUND_ASM(2) WK_ROUND_ASM(3) WK_ROUND_ASM(4) WK_ROUND_ASM(5)
               : [a] "+&v"(a), [s] "=&v"(s), [cm] "=&v"(cm), [cp] "=&v"(cp), [w2] "=&v"(w2)
               : [wa] "v"(L.wr), [ca] "v"(L.acc), [sa] "v"(L.src), [ma] "v"(L.cm), [pa] "v"(L.cp), [cpsh] "v"(L.cp_sh), [rsh] "v"(L.rho_sh),
                 [wr2] "s"(L.wrap2), [nw1] "s"(L.nowrap1), [off0] "n"(64 * R0), [off1] "n"(64 * (R0 + 1)), [off2] "n"(64 * (R0 + 2)),
                 [off3] "n"(64 * (R0 + 3)), [off4] "n"(64 * (R0 + 4)), [off5] "n"(64 * (R0 + 5)), [rc0] "v"(R.v[R0]), [rc1] "v"(R.v[R0 + 1]),
                 [rc2] "v"(R.v[R0 + 2]), [rc3] "v"(R.v[R0 + 3]), [rc4] "v"(R.v[R0 + 4]), [rc5] "v"(R.v[R0 + 5])
               : "memory", "vcc");
#endif
  return a;
}

// iota's constants into the table behind the image (all 64 lanes; a wavefront barrier before the first permutation)
__device__ __forceinline__ void wk_rc_table_init(uint32_t *lds) {
  const uint32_t l = threadIdx.x & 63u;
  lds[WK_LDS_DWORDS + l] = l < 48 ? WK_RC[l >> 1][l & 1u] : 0u;
  if (l < 32) lds[WK_LDS_DWORDS + 64 + l] = 0u;
}
// the same six rounds with iota's constant read from that table (one more LDS read per round, no constant registers)
#define WK_ROUND_ASM_T(K)                                                                                             \
  "ds_write_b32 %[wa], %[a]\n\t"                                                                                      \
  "ds_xor_b32 %[ca], %[a] offset:%[off" #K "]\n\t"                                                                    \
  "ds_read_b32 %[s], %[sa]\n\t"                                                                                       \
  "ds_read_b32 %[cm], %[ma] offset:%[off" #K "]\n\t"                                                                  \
  "ds_read_b32 %[cp], %[pa] offset:%[off" #K "]\n\t"                                                                  \
  "ds_read_b32 %[w2], %[ra] offset:%[ro" #K "]\n\t"                                                                   \
  "s_waitcnt lgkmcnt(0)\n\t"                                                                                          \
  "v_alignbit_b32 %[cp], %[cp], %[cp], %[cpsh]\n\t"                                                                   \
  "v_bitop3_b32 %[s], %[s], %[cm], %[cp] bitop3:0x96\n\t"                                                             \
  "v_alignbit_b32 %[s], %[s], %[s], %[rsh]\n\t"                                                                       \
  "v_mov_b32_e32 %[a], %[w2]\n\t"                               /* (iota's word; also one of the two wait states) */   \
  "s_nop 0\n\t"                                                                                                       \
  "v_mov_b32_dpp %[cm], %[s] row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                                   \
  "v_mov_b32_dpp %[cp], %[s] row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                                   \
  "v_mov_b32_dpp %[w2], %[s] row_shr:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                                   \
  "v_cndmask_b32_dpp %[cm], %[s], %[cm], vcc row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                   \
  "v_cndmask_b32_e64 %[cp], %[cp], %[w2], %[wr2]\n\t"                                                                 \
  "v_bitop3_b32 %[s], %[s], %[cm], %[cp] bitop3:0xd2\n\t"                                                             \
  "v_xor_b32_e32 %[a], %[s], %[a]\n\t"

template <int R0>
__device__ __forceinline__ uint32_t wk_rounds6_t(uint32_t a, const WkLanes &L) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t s, cm, cp, w2;
  asm volatile("s_mov_b64 vcc, %[nw1]\n\t" WK_ROUND_ASM_T(0) WK_ROUND_ASM_T(1) WK_ROUND_ASM_T(2) WK_ROUND_ASM_T(3) WK_ROUND_ASM_T(4) WK_ROUND_ASM_T(5)
               : [a] "+&v"(a), [s] "=&v"(s), [cm] "=&v"(cm), [cp] "=&v"(cp), [w2] "=&v"(w2)
               : [wa] "v"(L.wr), [ca] "v"(L.acc), [sa] "v"(L.src), [ma] "v"(L.cm), [pa] "v"(L.cp), [ra] "v"(L.rc), [cpsh] "v"(L.cp_sh),
                 [rsh] "v"(L.rho_sh), [wr2] "s"(L.wrap2), [nw1] "s"(L.nowrap1), [off0] "n"(64 * R0), [off1] "n"(64 * (R0 + 1)),
                 [off2] "n"(64 * (R0 + 2)), [off3] "n"(64 * (R0 + 3)), [off4] "n"(64 * (R0 + 4)), [off5] "n"(64 * (R0 + 5)), [ro0] "n"(8 * R0),
                 [ro1] "n"(8 * (R0 + 1)), [ro2] "n"(8 * (R0 + 2)), [ro3] "n"(8 * (R0 + 3)), [ro4] "n"(8 * (R0 + 4)), [ro5] "n"(8 * (R0 + 5))
               : "memory", "vcc");
#endif
  return a;
}
// the permutation with iota from the table: uses lds[0 .. WK_LDS_DWORDS_RC), wk_rc_table_init first
__device__ __forceinline__ uint32_t wk_keccak_f1600_t(uint32_t a, const WkLanes &L) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile(
      "ds_write_b32 %[z], %[o]\n\t"
      "ds_write_b32 %[z], %[o] offset:256\n\t"
      "ds_write_b32 %[z], %[o] offset:512\n\t"
      "ds_write_b32 %[z], %[o] offset:768\n\t"
      "ds_write_b32 %[z], %[o] offset:1024\n\t"
      "ds_write_b32 %[z], %[o] offset:1280"
      :
      : [z] "v"(L.zero), [o] "v"(0u)
      : "memory");
#endif
  a = wk_rounds6_t<0>(a, L);
  a = wk_rounds6_t<6>(a, L);
  a = wk_rounds6_t<12>(a, L);
  a = wk_rounds6_t<18>(a, L);
  return a;
}

// `a`: this lane's half of the state (any value in the idle lanes).  All 64 lanes must call it.  Uses lds[0 .. WK_LDS_DWORDS).
__device__ __forceinline__ uint32_t wk_keccak_f1600(uint32_t a, const WkLanes &L, const WkRc &R) {
#if defined(__HIP_DEVICE_COMPILE__)
  // One wavefront's LDS operations execute in issue order: a round's reads see every lane's write and XOR of that round, the
  // next round's write comes after them, and the clearing stores below come after the previous permutation's last reads.
  asm volatile(
      "ds_write_b32 %[z], %[o]\n\t"
      "ds_write_b32 %[z], %[o] offset:256\n\t"
      "ds_write_b32 %[z], %[o] offset:512\n\t"
      "ds_write_b32 %[z], %[o] offset:768\n\t"
      "ds_write_b32 %[z], %[o] offset:1024\n\t"
      "ds_write_b32 %[z], %[o] offset:1280"
      :
      : [z] "v"(L.zero), [o] "v"(0u)
      : "memory");
#endif
  a = wk_rounds6<0>(a, L, R);
  a = wk_rounds6<6>(a, L, R);
  a = wk_rounds6<12>(a, L, R);
  a = wk_rounds6<18>(a, L, R);
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

#endif  // __HIPCC__

}  // namespace bpp
