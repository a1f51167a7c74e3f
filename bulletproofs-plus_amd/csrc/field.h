// GF(2^255 - 19) for gfx950.
//
// Representation: 10 unsigned limbs, radix 2^25.5 (26,25,26,25,... bits).  Products are accumulated in
// 64-bit columns, which hipcc lowers to v_mad_u64_u32 (one instruction per limb product, no carry
// chains inside the product).  "Reduced" below means every limb < 2^26 (even) / 2^25 (odd) plus a few
// units in limb 0/1.  fe_mul/fe_sq accept limbs up to 1.5 * 2^27 and always return reduced limbs.
//
// Replaces (reference boundary): curve25519-dalek FieldElement reached through
// src/range_proof.rs:1067-1109 (decompress) and every point operation of the final MSM (:1050-1057).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BPP_HD __host__ __device__ __forceinline__
#define BPP_D __device__ __forceinline__
// helpers that are called many times or sit off the critical chain: real calls keep register pressure of the callers low
#define BPP_HD_NOINLINE __host__ __device__ __attribute__((noinline))
#else
#define BPP_HD inline
#define BPP_D inline
#define BPP_HD_NOINLINE inline
#endif

#define BPP_CONST static constexpr

namespace bpp {

struct fe {
  uint32_t v[10];
};

BPP_HD constexpr uint32_t fe_bits(int i) { return (i & 1) ? 25u : 26u; }
BPP_HD constexpr uint32_t fe_mask(int i) { return (i & 1) ? 0x1ffffffu : 0x3ffffffu; }

BPP_HD void fe_0(fe &h) {
#pragma unroll
  for (int i = 0; i < 10; i++) h.v[i] = 0;
}
BPP_HD void fe_1(fe &h) {
  fe_0(h);
  h.v[0] = 1;
}
BPP_HD void fe_copy(fe &h, const fe &f) {
#pragma unroll
  for (int i = 0; i < 10; i++) h.v[i] = f.v[i];
}

// one carry sweep; input limbs < 2^32 - 2^7, output reduced
BPP_HD void fe_carry(fe &h) {
  uint32_t c;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    c = h.v[i] >> fe_bits(i);
    h.v[i] &= fe_mask(i);
    h.v[i + 1] += c;
  }
  c = h.v[9] >> 25;
  h.v[9] &= 0x1ffffffu;
  h.v[0] += 19u * c;
  c = h.v[0] >> 26;
  h.v[0] &= 0x3ffffffu;
  h.v[1] += c;
}

// h = f + g, no carry: reduced + reduced -> limbs < 2^27
BPP_HD void fe_add(fe &h, const fe &f, const fe &g) {
#pragma unroll
  for (int i = 0; i < 10; i++) h.v[i] = f.v[i] + g.v[i];
}

// h = f - g (f limbs < 2^27, g limbs < 2^27); adds 4p limb-wise, then carries -> reduced
BPP_HD void fe_sub(fe &h, const fe &f, const fe &g) {
  h.v[0] = f.v[0] + 0xfffffb4u - g.v[0];  // 4*(2^26-19)
#pragma unroll
  for (int i = 1; i < 10; i++) h.v[i] = f.v[i] + ((i & 1) ? 0x7fffffcu : 0xffffffcu) - g.v[i];
  fe_carry(h);
}

BPP_HD void fe_neg(fe &h, const fe &f) {
  fe z;
  fe_0(z);
  fe_sub(h, z, f);
}

// carry a 10-column 64-bit product into reduced limbs
BPP_HD void fe_reduce_wide(fe &h, uint64_t t[10]) {
  uint64_t c;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    c = t[i] >> fe_bits(i);
    t[i] &= fe_mask(i);
    t[i + 1] += c;
  }
  c = t[9] >> 25;
  t[9] &= 0x1ffffffu;
  t[0] += 19u * c;  // c < 2^39 -> no overflow
  c = t[0] >> 26;
  t[0] &= 0x3ffffffu;
  t[1] += c;  // t[1] < 2^25 + 2^19
#pragma unroll
  for (int i = 0; i < 10; i++) h.v[i] = (uint32_t)t[i];
}

BPP_HD void fe_mul(fe &h, const fe &f, const fe &g) {
  uint32_t g19[10], f2[10];
#pragma unroll
  for (int i = 0; i < 10; i++) {
    g19[i] = 19u * g.v[i];
    f2[i] = 2u * f.v[i];
  }
  uint64_t t[10];
#pragma unroll
  for (int k = 0; k < 10; k++) {
    uint64_t acc = 0;
#pragma unroll
    for (int i = 0; i < 10; i++) {
      const int j = (k - i + 10) % 10;
      const bool wrap = i > k;
      const bool dbl = (i & 1) && (j & 1);
      const uint32_t a = dbl ? f2[i] : f.v[i];
      const uint32_t b = wrap ? g19[j] : g.v[j];
      acc += (uint64_t)a * b;
    }
    t[k] = acc;
  }
  fe_reduce_wide(h, t);
}

BPP_HD void fe_sq(fe &h, const fe &f) {
  // symmetric products folded: 55 multiplies
  uint32_t f19[10], f2[10];
#pragma unroll
  for (int i = 0; i < 10; i++) {
    f19[i] = 19u * f.v[i];
    f2[i] = 2u * f.v[i];
  }
  uint64_t t[10];
#pragma unroll
  for (int k = 0; k < 10; k++) {
    uint64_t acc = 0;
#pragma unroll
    for (int i = 0; i < 10; i++) {
      const int j = (k - i + 10) % 10;
      if (i > j) continue;
      const bool wrap = i > k;  // i + j == k + 10
      const bool dbl = (i & 1) && (j & 1);
      // a = f_i * (2 if dbl), b = f_j * (19 if wrap); off-diagonal pairs counted twice
      uint64_t p = (uint64_t)(dbl ? f2[i] : f.v[i]) * (wrap ? f19[j] : f.v[j]);
      if (i != j) p += p;
      acc += p;
    }
    t[k] = acc;
  }
  fe_reduce_wide(h, t);
}

BPP_HD void fe_sqn(fe &h, const fe &f, int n) {
  fe_sq(h, f);
  for (int i = 1; i < n; i++) fe_sq(h, h);
}

// canonical value as 8 little-endian 32-bit words (the primitive: byte arrays cost one register per byte on the GPU)
BPP_HD void fe_towords(uint32_t w[8], const fe &f) {
  fe h;
  fe_copy(h, f);
  fe_carry(h);
  fe_carry(h);
  // h < 2^255 + small; q = 1 iff h >= p
  uint32_t q = (h.v[0] + 19u) >> 26;
#pragma unroll
  for (int i = 1; i < 10; i++) q = (h.v[i] + q) >> fe_bits(i);
  h.v[0] += 19u * q;
  uint32_t c;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    c = h.v[i] >> fe_bits(i);
    h.v[i] &= fe_mask(i);
    h.v[i + 1] += c;
  }
  h.v[9] &= 0x1ffffffu;
  // pack 26/25-bit limbs at bit offsets 0,26,51,77,102,128,153,179,204,230
  w[0] = h.v[0] | (h.v[1] << 26);
  w[1] = (h.v[1] >> 6) | (h.v[2] << 19);
  w[2] = (h.v[2] >> 13) | (h.v[3] << 13);
  w[3] = (h.v[3] >> 19) | (h.v[4] << 6);
  w[4] = h.v[5] | (h.v[6] << 25);
  w[5] = (h.v[6] >> 7) | (h.v[7] << 19);
  w[6] = (h.v[7] >> 13) | (h.v[8] << 12);
  w[7] = (h.v[8] >> 20) | (h.v[9] << 6);
}

// canonical little-endian bytes
BPP_HD void fe_tobytes(uint8_t s[32], const fe &f) {
  uint32_t w[8];
  fe_towords(w, f);
#pragma unroll
  for (int i = 0; i < 8; i++) {
    s[4 * i + 0] = (uint8_t)(w[i]);
    s[4 * i + 1] = (uint8_t)(w[i] >> 8);
    s[4 * i + 2] = (uint8_t)(w[i] >> 16);
    s[4 * i + 3] = (uint8_t)(w[i] >> 24);
  }
}

// bit 255 ignored (dalek FieldElement::from_bytes)
BPP_HD void fe_fromwords(fe &h, const uint32_t w[8]) {
  h.v[0] = w[0] & 0x3ffffffu;
  h.v[1] = ((w[0] >> 26) | (w[1] << 6)) & 0x1ffffffu;
  h.v[2] = ((w[1] >> 19) | (w[2] << 13)) & 0x3ffffffu;
  h.v[3] = ((w[2] >> 13) | (w[3] << 19)) & 0x1ffffffu;
  h.v[4] = (w[3] >> 6) & 0x3ffffffu;
  h.v[5] = w[4] & 0x1ffffffu;
  h.v[6] = ((w[4] >> 25) | (w[5] << 7)) & 0x3ffffffu;
  h.v[7] = ((w[5] >> 19) | (w[6] << 13)) & 0x1ffffffu;
  h.v[8] = ((w[6] >> 12) | (w[7] << 20)) & 0x3ffffffu;
  h.v[9] = (w[7] >> 6) & 0x1ffffffu;
}
BPP_HD void fe_frombytes(fe &h, const uint8_t s[32]) {
  uint32_t w[8];
#pragma unroll
  for (int i = 0; i < 8; i++)
    w[i] = (uint32_t)s[4 * i] | ((uint32_t)s[4 * i + 1] << 8) | ((uint32_t)s[4 * i + 2] << 16) |
           ((uint32_t)s[4 * i + 3] << 24);
  fe_fromwords(h, w);
}

BPP_HD bool fe_isnegative(const fe &f) {
  uint32_t w[8];
  fe_towords(w, f);
  return w[0] & 1u;
}

BPP_HD bool fe_iszero(const fe &f) {
  uint32_t w[8];
  fe_towords(w, f);
  uint32_t r = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) r |= w[i];
  return r == 0;
}

BPP_HD bool fe_eq(const fe &f, const fe &g) {
  fe d;
  fe_sub(d, f, g);
  return fe_iszero(d);
}

BPP_HD void fe_cmov(fe &h, const fe &g, bool b) {
#pragma unroll
  for (int i = 0; i < 10; i++) h.v[i] = b ? g.v[i] : h.v[i];
}

// |f|: negate if canonical f is odd
BPP_HD void fe_abs(fe &h, const fe &f) {
  fe n;
  fe_neg(n, f);
  bool neg = fe_isnegative(f);
  fe_copy(h, f);
  fe_cmov(h, n, neg);
}

// z^(2^252 - 3)
// Scheduling fence: the ten limbs pass through an empty asm, so the instruction scheduler cannot interleave the field
// operation before it with the one after it.  The addition chains below are strictly sequential, yet without fences
// the scheduler overlaps neighbouring operations until fe_pow22523 alone is allocated 364 VGPRs (k_decompress: 256 +
// 218 spilled); with them it is ~150 and the kernels around it fit 3-4 wavefronts per SIMD without scratch.
BPP_HD void fe_fence(fe &x) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(x.v[0]), "+v"(x.v[1]), "+v"(x.v[2]), "+v"(x.v[3]), "+v"(x.v[4]), "+v"(x.v[5]), "+v"(x.v[6]), "+v"(x.v[7]),
               "+v"(x.v[8]), "+v"(x.v[9]));
#else
  (void)x;
#endif
}

BPP_HD void fe_pow22523(fe &out, const fe &z) {
  fe t0, t1, t2;
  fe_sq(t0, z);          // 2
  fe_fence(t0);
  fe_sqn(t1, t0, 2);     // 8
  fe_fence(t1);
  fe_mul(t1, z, t1);     // 9
  fe_fence(t1);
  fe_mul(t0, t0, t1);    // 11
  fe_fence(t0);
  fe_sq(t0, t0);         // 22
  fe_fence(t0);
  fe_mul(t0, t1, t0);    // 31 = 2^5 - 1
  fe_fence(t0);
  fe_sqn(t1, t0, 5);
  fe_fence(t1);
  fe_mul(t0, t1, t0);    // 2^10 - 1
  fe_fence(t0);
  fe_sqn(t1, t0, 10);
  fe_fence(t1);
  fe_mul(t1, t1, t0);    // 2^20 - 1
  fe_fence(t1);
  fe_sqn(t2, t1, 20);
  fe_fence(t2);
  fe_mul(t1, t2, t1);    // 2^40 - 1
  fe_fence(t1);
  fe_sqn(t1, t1, 10);
  fe_fence(t1);
  fe_mul(t0, t1, t0);    // 2^50 - 1
  fe_fence(t0);
  fe_sqn(t1, t0, 50);
  fe_fence(t1);
  fe_mul(t1, t1, t0);    // 2^100 - 1
  fe_fence(t1);
  fe_sqn(t2, t1, 100);
  fe_fence(t2);
  fe_mul(t1, t2, t1);    // 2^200 - 1
  fe_fence(t1);
  fe_sqn(t1, t1, 50);
  fe_fence(t1);
  fe_mul(t0, t1, t0);    // 2^250 - 1
  fe_fence(t0);
  fe_sqn(t0, t0, 2);     // 2^252 - 4
  fe_fence(t0);
  fe_mul(out, t0, z);    // 2^252 - 3
}

// z^(p-2)
BPP_HD void fe_invert(fe &out, const fe &z) {
  fe t0, t1, t2, t3;
  fe_sq(t0, z);          // 2
  fe_fence(t0);
  fe_sqn(t1, t0, 2);     // 8
  fe_fence(t1);
  fe_mul(t1, z, t1);     // 9
  fe_fence(t1);
  fe_mul(t0, t0, t1);    // 11
  fe_fence(t0);
  fe_sq(t2, t0);         // 22
  fe_fence(t2);
  fe_mul(t1, t1, t2);    // 31
  fe_fence(t1);
  fe_sqn(t2, t1, 5);
  fe_fence(t2);
  fe_mul(t1, t2, t1);    // 2^10 - 1
  fe_fence(t1);
  fe_sqn(t2, t1, 10);
  fe_fence(t2);
  fe_mul(t2, t2, t1);    // 2^20 - 1
  fe_fence(t2);
  fe_sqn(t3, t2, 20);
  fe_fence(t3);
  fe_mul(t2, t3, t2);    // 2^40 - 1
  fe_fence(t2);
  fe_sqn(t2, t2, 10);
  fe_fence(t2);
  fe_mul(t1, t2, t1);    // 2^50 - 1
  fe_fence(t1);
  fe_sqn(t2, t1, 50);
  fe_fence(t2);
  fe_mul(t2, t2, t1);    // 2^100 - 1
  fe_fence(t2);
  fe_sqn(t3, t2, 100);
  fe_fence(t3);
  fe_mul(t2, t3, t2);    // 2^200 - 1
  fe_fence(t2);
  fe_sqn(t2, t2, 50);
  fe_fence(t2);
  fe_mul(t1, t2, t1);    // 2^250 - 1
  fe_fence(t1);
  fe_sqn(t1, t1, 5);     // 2^255 - 32
  fe_fence(t1);
  fe_mul(out, t1, t0);   // 2^255 - 21
}

}  // namespace bpp
