// The generator columns of a reference batch as ONE integer matrix product on the matrix cores.
//
// src/range_proof.rs:972-1020 adds, proof after proof, a weighted scalar to every one of the 2 mn generator columns of the batch:
//   g[i] += w r1e y^-i s[i] + w e^2 z          h[i] += w (s1e s[mn-1-i] - e^2 d[i] y^(mn-i)) - w e^2 z
// With i = (hi << LB) | lo every per-index factor is a product of a "low" and a "high" table entry of the proof
// (kernels_verify.h: k_scalars_shared builds them), so the column sums of a group are
//   G[lo][hi] = sum_p glo_p[lo] ghi_p[hi]      H[lo][hi] = sum_p hlo_p[~lo] shi_p[~hi] + yn2lo_p[lo] y2hi_p[hi]
// -- a matrix product over the PROOFS of the group, entries in Z_l.  Until round 4 every (proof, index) pair paid its own
// Montgomery products (3 products under 2 reductions, ~580 VALU instructions; 38 M of a 65 536-proof step's 62 M in
// k_scalars_lanes).  Here the reductions are taken out of the sum: the table entries (Montgomery residues below 2^253) are
// written as 32 balanced base-256 digits (int8), the digit products are summed over the proofs by V_MFMA_I32_32X32X32_I8 --
// one instruction = the 32 x 32 digit-product tile of one (lo, hi) pair over 32 proofs --, the tile's anti-diagonals are the
// 63 coefficients of sum_p a_p b_p in base 256, and ONE reduction mod l per column finishes the job (k_static_finish).
// Exact integer arithmetic throughout: the columns equal the per-proof form's bit for bit (tests/test_gpu_round4.py).
//
// Layout of the digit tables ("GEMM layout"): [table][entry][block of 16 proofs][digit 0..31][proof in block], int8.
// A lane of the MFMA (digit r = lane & 31, half h = lane >> 5) takes the 16 proofs of block 2 s + h of K-step s for its digit
// as one 16-byte piece, and the 64 lanes of a fragment are 1 KB of contiguous memory.  The k order inside the instruction
// does not matter: A and B fragments are filled by the same rule.
#pragma once
#include "scalar.h"

namespace bpp {

#define SGEMM_BLOCK 16u        // proofs per block of the digit tables
#define SGEMM_KBLOCKS 16u      // blocks per K chunk of one wavefront: 256 proofs (int32 headroom: 2 x 256 x 2^14 x 32 = 2^28)
#define SGEMM_HI_PER_WAVE 4u   // (lo, hi) tiles of G and of H per wavefront: 8 accumulator tiles = 128 registers
#define SGEMM_LO_ENTRIES 8u    // entries per low table in the buffer (2^LB <= 8 used)

BPP_HD constexpr size_t sgemm_table_bytes(uint32_t entries, uint32_t nblk) { return (size_t)entries * nblk * 32u * SGEMM_BLOCK; }

// x (below 2^253) as 32 balanced digits d_k in [-128, 127], sum d_k 256^k = x: add 0x80 to every byte with carries (one
// 256-bit addition of the constant 0x8080...80), then every byte minus 128 is the digit (as int8: the byte with its top bit flipped)
__device__ __forceinline__ void sgemm_digits(uint32_t y[8], const sc &x) {
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    c += (uint64_t)x.v[i] + 0x80808080u;
    y[i] = (uint32_t)c ^ 0x80808080u;
    c >>= 32;
  }
}
// one lane per proof (k_scalars_shared): sixteen neighbouring lanes fill the sixteen bytes of a digit row
__device__ __forceinline__ void sgemm_store(int8_t *table, uint32_t entry, uint32_t nblk, uint32_t p, const sc &x) {
  uint32_t y[8];
  sgemm_digits(y, x);
  int8_t *dst = table + (((size_t)entry * nblk + (p / SGEMM_BLOCK)) * 32u) * SGEMM_BLOCK + (p % SGEMM_BLOCK);
#pragma unroll
  for (int k = 0; k < 32; k++) dst[(size_t)k * SGEMM_BLOCK] = (int8_t)(y[k >> 2] >> (8 * (k & 3)));
}

typedef int sgemm_v4 __attribute__((ext_vector_type(4)));
typedef int sgemm_v16 __attribute__((ext_vector_type(16)));

// One wavefront: the tiles (lo, hi = 4 hc .. 4 hc + 3) of G and H of group g over K chunk kc (256 proofs); writes the 63
// anti-diagonal sums of every tile: gparts[((g nkc + kc) max_mn + i) 2 + which][64], i = (hi << 3) | lo.
// lo_t: tables glo | hloR | yn2lo (hloR[lo] = hlo[~lo]), 8 entries each; hi_t: ghi | shiR | y2hi (shiR[hi] = shi[~hi]), nhi_max
// entries each.  Group starts are multiples of 16 proofs (host check); a group's last block may be partly filled: the proofs past
// its end are masked out of the A fragments.  XCD-aware: workgroup ids go round the eight XCDs, all wavefronts of group g run on
// XCD g % 8, whose L2 then serves the re-reads of the group's tables (every (lo, hi chunk) wavefront reads its own fragments).
// Measured against a form with eight wavefronts per workgroup sharing the step's fragments through LDS (36 KB, each byte read 1.5
// times instead of 5): that one takes 44 us instead of 63 - 130 alone and LOSES with three steps in flight -- a 512-lane workgroup
// with 36 KB of LDS waits for a CU to drain, one-wavefront workgroups fit into whatever the other steps' kernels leave free
// (HISTORY.md 3.3).
__global__ void __launch_bounds__(64) k_static_gemm(const int8_t *__restrict__ lo_t, const int8_t *__restrict__ hi_t,
                                                         const uint32_t *__restrict__ group_first, uint32_t nblk, uint32_t nhi,
                                                         uint32_t nhi_max, uint32_t nkc, uint32_t max_mn, uint32_t n_groups,
                                                         int32_t *__restrict__ gparts) {
  const uint32_t nhc = nhi / SGEMM_HI_PER_WAVE, tiles = SGEMM_LO_ENTRIES * nhc;
  const uint32_t slot = blockIdx.x >> 3, tile_id = slot % tiles, g = (blockIdx.x & 7u) + 8u * (slot / tiles), kc = blockIdx.y;
  if (g >= n_groups) return;
  const uint32_t lo = tile_id / nhc, hc = tile_id - lo * nhc;
  const uint32_t lane = threadIdx.x, r = lane & 31u, h = lane >> 5;
  const uint32_t p0 = group_first[g], p1 = group_first[g + 1];
  const uint32_t b0 = p0 / SGEMM_BLOCK + kc * SGEMM_KBLOCKS;
  const uint32_t bend = min((p1 + SGEMM_BLOCK - 1u) / SGEMM_BLOCK, b0 + SGEMM_KBLOCKS);
  sgemm_v16 cg[SGEMM_HI_PER_WAVE], ch[SGEMM_HI_PER_WAVE];
#pragma unroll
  for (uint32_t j = 0; j < SGEMM_HI_PER_WAVE; j++)
#pragma unroll
    for (int q = 0; q < 16; q++) cg[j][q] = ch[j][q] = 0;
  const size_t lo_tab = sgemm_table_bytes(SGEMM_LO_ENTRIES, nblk), hi_tab = sgemm_table_bytes(nhi_max, nblk);
  for (uint32_t blk2 = b0; blk2 < bend; blk2 += 2) {
    const uint32_t blk = blk2 + h;
    const uint32_t blk_ld = min(blk, nblk - 1u);
    const uint32_t cnt = blk < bend ? min(SGEMM_BLOCK, p1 - blk * SGEMM_BLOCK) : 0u;
    const size_t row = ((size_t)blk_ld * 32u + r) * SGEMM_BLOCK;
    sgemm_v4 a[3];
#pragma unroll
    for (uint32_t T = 0; T < 3; T++) {
      a[T] = *reinterpret_cast<const sgemm_v4 *>(lo_t + T * lo_tab + (size_t)lo * nblk * 32u * SGEMM_BLOCK + row);
      if (cnt < SGEMM_BLOCK) {
#pragma unroll
        for (int w = 0; w < 4; w++) {
          const uint32_t nb = cnt > 4u * w ? min(4u, cnt - 4u * w) : 0u;
          const uint32_t m = nb >= 4u ? 0xffffffffu : ((1u << (8u * nb)) - 1u);
          a[T][w] &= (int)m;
        }
      }
    }
#pragma unroll
    for (uint32_t j = 0; j < SGEMM_HI_PER_WAVE; j++) {
      const int8_t *hb = hi_t + (size_t)(hc * SGEMM_HI_PER_WAVE + j) * nblk * 32u * SGEMM_BLOCK + row;
      const sgemm_v4 bg = *reinterpret_cast<const sgemm_v4 *>(hb);
      const sgemm_v4 bs = *reinterpret_cast<const sgemm_v4 *>(hb + hi_tab);
      const sgemm_v4 by = *reinterpret_cast<const sgemm_v4 *>(hb + 2 * hi_tab);
      cg[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[0], bg, cg[j], 0, 0, 0);
      ch[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[1], bs, ch[j], 0, 0, 0);
      ch[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[2], by, ch[j], 0, 0, 0);
    }
  }
  __shared__ int32_t tile[32 * 33];
#pragma unroll
  for (uint32_t j = 0; j < 2 * SGEMM_HI_PER_WAVE; j++) {
    const uint32_t hi = hc * SGEMM_HI_PER_WAVE + (j >> 1), which = j & 1u;
#pragma unroll
    for (int q = 0; q < 16; q++) {
      const uint32_t la = (uint32_t)(q & 3) + 8u * (uint32_t)(q >> 2) + 4u * h;
      tile[la * 33u + r] = which ? ch[j >> 1][q] : cg[j >> 1][q];
    }
    __syncthreads();
    int32_t sum = 0;
    for (uint32_t la = 0; la < 32; la++) {
      const uint32_t lbq = lane - la;
      if (lbq < 32u) sum += tile[la * 33u + lbq];
    }
    const uint32_t i = (hi << 3) | lo;
    gparts[((((size_t)g * nkc + kc) * max_mn + i) * 2u + which) * 64u + lane] = sum;
    __syncthreads();
  }
}

// Columns of group g from the anti-diagonal sums: S = sum_k c_k 256^k = sum_p (a_p R)(b_p R) as an integer below 2^524, then
//   sum_p a_p b_p mod l = S R^-2 = from_mont(from_mont(S0) + montmul(S1, 2^256) + montmul(S2, 2^512)),  S = S0 + S1 2^256 + S2 2^512,
// plus / minus E = sum_p w e^2 z (rows_base[p][t + 1], Montgomery; summed limb-wise like k_reduce_parts' base columns).
// One lane per column (i, which); columns of generator indices >= mn (a statement smaller than the parameters) are zero.  One more
// workgroup per group sums the t + 1 base columns.
__global__ void __launch_bounds__(64) k_static_finish(const int32_t *__restrict__ gparts, const sc *__restrict__ rows_base,
                                                      const uint32_t *__restrict__ group_first, uint32_t cols, uint32_t max_mn,
                                                      uint32_t mn, uint32_t t, uint32_t nkc, sc *__restrict__ out /* [G][cols] canonical */) {
  const uint32_t g = blockIdx.y, lane = threadIdx.x;
  const uint32_t col = blockIdx.x * 64u + lane;
  const uint32_t p0 = group_first[g], p1 = group_first[g + 1];
  if (blockIdx.x * 64u >= 2u * max_mn) {
    // the group's last workgroup: the t + 1 base columns (g bases, h), sums over the proofs of rows_base[p][0 .. t] -- lane = (proof
    // slot, column), eight columns side by side (t + 1 <= 7), limb-wise sums, one Montgomery exit per column (as k_reduce_parts)
    const uint32_t c = lane & 7u, slot = lane >> 3;
    uint64_t a8[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a8[i] = 0;
    if (c <= t) {
      for (uint32_t p = p0 + slot; p < p1; p += 8u) {
        const sc v = rows_base[(size_t)p * (t + 2u) + c];
#pragma unroll
        for (int i = 0; i < 8; i++) a8[i] += v.v[i];
      }
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
      for (int off = 32; off >= 8; off >>= 1) a8[i] += __shfl_xor(a8[i], off, 64);
    }
    if (slot == 0 && c <= t) {
      uint32_t wds[8];
      uint64_t carry = 0;
#pragma unroll
      for (int q = 0; q < 8; q++) {
        carry += a8[q];
        wds[q] = (uint32_t)carry;
        carry >>= 32;
      }
      sc lo, res, hi, p256;
      sc_const(lo, wds);
      sc_from_mont(res, lo);
      sc_0(hi);
      hi.v[0] = (uint32_t)carry;
      hi.v[1] = (uint32_t)(carry >> 32);
      sc_const(p256, SC_P256);
      sc_montmul(hi, hi, p256);
      sc_add(res, res, hi);
      out[(size_t)g * cols + 2u * max_mn + c] = res;
    }
    return;
  }
  // E: every lane ends up with the limb sums over the group's proofs
  uint64_t acc[8];
#pragma unroll
  for (int i = 0; i < 8; i++) acc[i] = 0;
  for (uint32_t p = p0 + lane; p < p1; p += 64u) {
    const sc v = rows_base[(size_t)p * (t + 2u) + (t + 1u)];
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] += v.v[i];
  }
#pragma unroll
  for (int i = 0; i < 8; i++) {
    for (int off = 32; off >= 1; off >>= 1) acc[i] += __shfl_xor(acc[i], off, 64);
  }
  if (col >= 2u * max_mn) return;
  const uint32_t i = col >> 1, which = col & 1u;
  sc res;
  if (i >= mn) {
    sc_0(res);
    out[(size_t)g * cols + col] = res;
    return;
  }
  sc E;
  {
    uint32_t wds[8];
    uint64_t carry = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) {
      carry += acc[q];
      wds[q] = (uint32_t)carry;
      carry >>= 32;
    }
    sc lo, hi, p256;
    sc_const(lo, wds);
    sc_from_mont(E, lo);
    sc_0(hi);
    hi.v[0] = (uint32_t)carry;
    hi.v[1] = (uint32_t)(carry >> 32);
    sc_const(p256, SC_P256);
    sc_montmul(hi, hi, p256);
    sc_add(E, E, hi);
  }
  // S in 32-bit words from the base-256 coefficients (signed, up to 2^30 each per K chunk), least significant first
  uint32_t w[17];
#pragma unroll
  for (int q = 0; q < 17; q++) w[q] = 0;
  int64_t run = 0;
#pragma unroll
  for (int k4 = 0; k4 < 16; k4++) {
    int64_t c[4] = {0, 0, 0, 0};
    for (uint32_t kc = 0; kc < nkc; kc++) {
      const sgemm_v4 v = *reinterpret_cast<const sgemm_v4 *>(gparts + ((((size_t)g * nkc + kc) * max_mn + i) * 2u + which) * 64u + 4 * k4);
      c[0] += v[0];
      c[1] += v[1];
      c[2] += v[2];
      c[3] += v[3];
    }
#pragma unroll
    for (int b = 0; b < 4; b++) {
      run += c[b];
      w[k4] |= (uint32_t)(run & 0xff) << (8 * b);
      run >>= 8;  // arithmetic: a negative running value borrows from the coefficients above
    }
  }
  w[16] = (uint32_t)run;  // the total is a sum of products of non-negative numbers: what is left is its top word
  sc s0, s1, s2, a, b2, c2, k256, k512;
#pragma unroll
  for (int q = 0; q < 8; q++) {
    s0.v[q] = w[q];
    s1.v[q] = w[8 + q];
  }
  sc_0(s2);
  s2.v[0] = w[16];
  sc_const(k256, SC_P256);
  sc_const(k512, SC_P512);
  sc_from_mont(a, s0);
  sc_montmul(b2, s1, k256);
  sc_montmul(c2, s2, k512);
  sc_add(a, a, b2);
  sc_add(a, a, c2);
  sc_from_mont(res, a);
  if (which) sc_sub(res, res, E);
  else sc_add(res, res, E);
  out[(size_t)g * cols + col] = res;
}

}  // namespace bpp
