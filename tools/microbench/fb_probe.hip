// Microbenchmark (gfx950): the prover's fixed-base MSM (k_fb_msm) on the shape of one round of BASELINE configs[4]
// (outputs = 2 x proofs, 260 terms each, 516 generators, 11-bit windows: 24 x 1024 entries of 128 bytes = 1.6 GB of table,
// filled with pseudo-random field elements: the arithmetic does not care) against two other ways of cutting the same work:
//   A  k_fb_msm as shipped: one workgroup of 256 lanes per output, (term, window) items dealt round the lanes, 8-level tree
//      through LDS at the end (one busy wavefront, three idle ones holding their registers)
//   B  A without the tree (wrong result: the price of the tree and its barriers, a lower bound for any restructuring)
//   C  k_fb_part + k_fb_sum: one independent wavefront per (output, slice of <= PER terms), no barrier beyond its own staging,
//      64 partial sums per wavefront written to memory; a second launch sums an output's partials (one wavefront per output:
//      lane-wise over the slices at full width, then the 6-level tree)
// usage: fb_probe [proofs=1024]     build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I bulletproofs-plus_amd/csrc -I include ...
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <random>
#include <vector>
#include "kernels_prove.h"
using namespace bpp;

__global__ void k_fill_tbl(fbent *t, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  niels q;
  const uint32_t a = (uint32_t)i * 2654435761u, b = (uint32_t)(i >> 7) * 40503u;
  for (int k = 0; k < 10; k++) {
    q.yplusx.v[k] = (a + k * 40503u + b) & 0x1ffffff;
    q.yminusx.v[k] = (a * 3u + k * 2654435761u) & 0x1ffffff;
    q.xy2d.v[k] = (a ^ (b + k * 977u)) & 0x1ffffff;
  }
  t[i].q = q;
}

// ---- B: k_fb_msm without its reduction tree
__global__ void __launch_bounds__(FB_THREADS) k_fb_msm_notree(const sc *__restrict__ scal, const uint32_t *__restrict__ gidx,
                                                              const uint32_t *__restrict__ count, uint32_t stride, const fbent *__restrict__ tbl,
                                                              FbGeom geo, ge *__restrict__ out) {
  const uint32_t o = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
  const uint32_t n = count[o];
  __shared__ FbShared sh;
  ge acc;
  ge_identity(acc);
  for (uint32_t base = 0; base < n; base += FB_CHUNK) {
    const uint32_t cn = n - base < FB_CHUNK ? n - base : FB_CHUNK;
    __syncthreads();
    for (uint32_t i = tid; i < cn; i += nthr) {
      const sc s = scal[(size_t)o * stride + base + i];
      fb_recode(sh.st.dig + (size_t)i * geo.items, s, geo);
      sh.st.gi[i] = gidx[(size_t)o * stride + base + i];
    }
    __syncthreads();
    const uint32_t items = cn * geo.items;
    uint32_t it = tid;
    niels nxt;
    int nd = 0;
    if (it < items) fb_fetch(nxt, nd, sh.st, tbl, geo, it);
    while (it < items) {
      niels cur = nxt;
      const int cd = nd;
      it += nthr;
      if (it < items) fb_fetch(nxt, nd, sh.st, tbl, geo, it);
      if (cd != 0) ge_madd_swapped(acc, acc, cur, cd < 0);
    }
  }
  if (acc.X.v[0] == 0x7fffffffu) out[o] = acc;  // never true: keeps the loop alive
}

// ---- D: the shipped kernel's body bounded for four wavefronts per SIMD (<= 128 VGPRs)
__global__ void __launch_bounds__(FB_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_fb_msm_w4(const sc *__restrict__ scal, const uint32_t *__restrict__ gidx, const uint32_t *__restrict__ count, uint32_t stride,
            const fbent *__restrict__ tbl, FbGeom geo, ge *__restrict__ out) {
  const uint32_t o = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
  const uint32_t n = count[o];
  __shared__ FbShared sh;
  ge acc;
  ge_identity(acc);
  for (uint32_t base = 0; base < n; base += FB_CHUNK) {
    const uint32_t cn = n - base < FB_CHUNK ? n - base : FB_CHUNK;
    __syncthreads();
    for (uint32_t i = tid; i < cn; i += nthr) {
      const sc s = scal[(size_t)o * stride + base + i];
      fb_recode(sh.st.dig + (size_t)i * geo.items, s, geo);
      sh.st.gi[i] = gidx[(size_t)o * stride + base + i];
    }
    __syncthreads();
    const uint32_t items = cn * geo.items;
    uint32_t it = tid;
    niels nxt;
    int nd = 0;
    if (it < items) fb_fetch(nxt, nd, sh.st, tbl, geo, it);
    while (it < items) {
      niels cur = nxt;
      const int cd = nd;
      it += nthr;
      if (it < items) fb_fetch(nxt, nd, sh.st, tbl, geo, it);
      if (cd != 0) ge_madd_swapped(acc, acc, cur, cd < 0);
    }
  }
  __syncthreads();
  sh.red[tid] = acc;
  __syncthreads();
  for (uint32_t off = nthr / 2; off >= 1; off >>= 1) {
    if (tid < off) {
      ge x = sh.red[tid], y2 = sh.red[tid + off];
      ge_add(x, x, y2);
      sh.red[tid] = x;
    }
    __syncthreads();
  }
  if (tid == 0) out[o] = sh.red[0];
}

// ---- C: one wavefront per (output, slice): k_fb_part + k_fb_sum, since round 5 the product's own kernels (kernels_prove.h)

// ---- E: 64-byte table entries.  An entry is the affine point (x, y) as 2 x 8 packed words -- one 64-byte request per lookup
// instead of a 128-byte line (random 64-byte reads out of <= 2 GB come 1.6x as often: rand_lines.hip) -- and the addition
// recomputes what the 128-byte entry carried: y+x, y-x by limb additions, 2dxy as two more products (9 instead of 7).
struct fbxy {
  uint32_t x[8], y[8];
};
static_assert(sizeof(fbxy) == 64, "64-byte entry");
__global__ void k_fill_xy(fbxy *t, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t a = (uint32_t)i * 2654435761u, b = (uint32_t)(i >> 7) * 40503u;
  fbxy e;
  for (int k = 0; k < 8; k++) {
    e.x[k] = a + k * 40503u + b;
    e.y[k] = a * 3u + k * 2654435761u;
  }
  e.x[7] &= 0x7fffffffu;
  e.y[7] &= 0x7fffffffu;
  t[i] = e;
}
struct xyw {
  uint4 q[4];
};
BPP_D void fb_fetch_xy(xyw &e, int &d, const FbStage &st, const fbxy *__restrict__ tbl, const FbGeom &geo, uint32_t it) {
  const uint32_t i = it / geo.items, w = it - i * geo.items;
  d = st.dig[it];
  const uint32_t mag = (uint32_t)(d < 0 ? -d : d);
  const uint4 *p = (const uint4 *)&tbl[((size_t)st.gi[i] * geo.windows + w) * geo.entries + (mag ? mag - 1u : 0u)];
  e.q[0] = p[0];
  e.q[1] = p[1];
  e.q[2] = p[2];
  e.q[3] = p[3];
}
BPP_D void ge_madd_xy(ge &r, const ge &p, const xyw &e, bool neg) {
  const uint32_t wx[8] = {e.q[0].x, e.q[0].y, e.q[0].z, e.q[0].w, e.q[1].x, e.q[1].y, e.q[1].z, e.q[1].w};
  const uint32_t wy[8] = {e.q[2].x, e.q[2].y, e.q[2].z, e.q[2].w, e.q[3].x, e.q[3].y, e.q[3].z, e.q[3].w};
  fe x, y, yp, ym, t, a, b, c, ee, f, g, h, u, v, d2;
  fe_fromwords(x, wx);
  fe_fromwords(y, wy);
  fe_add(u, y, x);
  fe_sub_lazy(v, y, x);
#pragma unroll
  for (int i = 0; i < 10; i++) {
    yp.v[i] = neg ? v.v[i] : u.v[i];
    ym.v[i] = neg ? u.v[i] : v.v[i];
  }
  fe_mul(t, x, y);
  fe_const(d2, FE_D2);
  fe_mul(t, t, d2);
  fe_add(a, p.Y, p.X);
  fe_mul(a, a, yp);
  fe_sub_lazy(b, p.Y, p.X);
  fe_mul(b, b, ym);
  fe_mul(c, t, p.T);
  fe_sub_lazy(ee, a, b);
  fe_add(h, a, b);
  fe_dbl_add(u, p.Z, c);
  fe_dbl_sub_lazy(v, p.Z, c);
  fe_mul(r.Z, v, u);
#pragma unroll
  for (int i = 0; i < 10; i++) {
    f.v[i] = neg ? u.v[i] : v.v[i];
    g.v[i] = neg ? v.v[i] : u.v[i];
  }
  fe_mul(r.X, f, ee);
  fe_mul(r.Y, g, h);
  fe_mul(r.T, ee, h);
}
template <bool TREE>
__global__ void __launch_bounds__(FB_THREADS) k_fb_msm_xy(const sc *__restrict__ scal, const uint32_t *__restrict__ gidx,
                                                          const uint32_t *__restrict__ count, uint32_t stride, const fbxy *__restrict__ tbl,
                                                          FbGeom geo, ge *__restrict__ out) {
  const uint32_t o = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
  const uint32_t n = count[o];
  __shared__ FbShared sh;
  ge acc;
  ge_identity(acc);
  for (uint32_t base = 0; base < n; base += FB_CHUNK) {
    const uint32_t cn = n - base < FB_CHUNK ? n - base : FB_CHUNK;
    __syncthreads();
    for (uint32_t i = tid; i < cn; i += nthr) {
      const sc s = scal[(size_t)o * stride + base + i];
      fb_recode(sh.st.dig + (size_t)i * geo.items, s, geo);
      sh.st.gi[i] = gidx[(size_t)o * stride + base + i];
    }
    __syncthreads();
    const uint32_t items = cn * geo.items;
    uint32_t it = tid;
    xyw nxt;
    int nd = 0;
    if (it < items) fb_fetch_xy(nxt, nd, sh.st, tbl, geo, it);
    while (it < items) {
      xyw cur = nxt;
      const int cd = nd;
      it += nthr;
      if (it < items) fb_fetch_xy(nxt, nd, sh.st, tbl, geo, it);
      if (cd != 0) ge_madd_xy(acc, acc, cur, cd < 0);
    }
  }
  if (!TREE) {
    if (acc.X.v[0] == 0x7fffffffu) out[o] = acc;
    return;
  }
  __syncthreads();
  sh.red[tid] = acc;
  __syncthreads();
  for (uint32_t off = nthr / 2; off >= 1; off >>= 1) {
    if (tid < off) {
      ge x = sh.red[tid], y2 = sh.red[tid + off];
      ge_add(x, x, y2);
      sh.red[tid] = x;
    }
    __syncthreads();
  }
  if (tid == 0) out[o] = sh.red[0];
}

// ---- G: ONE wavefront per output.  The shipped kernel's workgroup is four wavefronts that meet at barriers (staging, an
// 8-level tree of which the last 6 levels run on one wavefront while three wait) and holds 40 KB of LDS whatever its size, which
// is what bounds it to four workgroups per CU.  Here an output is 64 lanes: 4x the additions per lane, a 6-level tree inside the
// wavefront, staging in chunks of GCH terms (LDS per workgroup: GCH x (2 x items + 4) + 64 x 160 bytes).  PF: table lines in flight
// per lane ahead of the addition.
template <int GCH, int PF>
__global__ void __launch_bounds__(64) k_fb_msm_wave(const sc *__restrict__ scal, const uint32_t *__restrict__ gidx,
                                                    const uint32_t *__restrict__ count, uint32_t stride, const fbent *__restrict__ tbl,
                                                    FbGeom geo, ge *__restrict__ out) {
  const uint32_t o = blockIdx.x, lane = threadIdx.x;
  const uint32_t n = count[o];
  __shared__ int16_t s_dig[GCH * FB_MAX_WINDOWS];
  __shared__ uint32_t s_gi[GCH];
  __shared__ ge red[64];
  ge acc;
  ge_identity(acc);
  auto fetch = [&](niels &q, int &d, uint32_t it) {
    const uint32_t i = it / geo.items, w = it - i * geo.items;
    d = s_dig[it];
    const uint32_t mag = (uint32_t)(d < 0 ? -d : d);
    niels_load_swapped(q, &tbl[((size_t)s_gi[i] * geo.windows + w) * geo.entries + (mag ? mag - 1u : 0u)].q, d < 0);
  };
  for (uint32_t base = 0; base < n; base += GCH) {
    const uint32_t cn = n - base < GCH ? n - base : GCH;
    __syncthreads();
    for (uint32_t i = lane; i < cn; i += 64) {
      const sc s = scal[(size_t)o * stride + base + i];
      fb_recode(s_dig + (size_t)i * geo.items, s, geo);
      s_gi[i] = gidx[(size_t)o * stride + base + i];
    }
    __syncthreads();
    const uint32_t items = cn * geo.items;
    uint32_t it = lane;
    if constexpr (PF == 1) {
      niels nxt;
      int nd = 0;
      if (it < items) fetch(nxt, nd, it);
      while (it < items) {
        niels cur = nxt;
        const int cd = nd;
        it += 64;
        if (it < items) fetch(nxt, nd, it);
        if (cd != 0) ge_madd_swapped(acc, acc, cur, cd < 0);
      }
    } else {
      niels n0, n1;
      int d0 = 0, d1 = 0;
      if (it < items) fetch(n0, d0, it);
      if (it + 64 < items) fetch(n1, d1, it + 64);
      while (it < items) {
        niels cur = n0;
        const int cd = d0;
        n0 = n1;
        d0 = d1;
        it += 64;
        if (it + 64 < items) fetch(n1, d1, it + 64);
        if (cd != 0) ge_madd_swapped(acc, acc, cur, cd < 0);
      }
    }
  }
  red[lane] = acc;
  __syncthreads();
  for (uint32_t off = 32; off >= 1; off >>= 1) {
    if (lane < off) {
      ge x = red[lane], y2 = red[lane + off];
      ge_add(x, x, y2);
      red[lane] = x;
    }
    __syncthreads();
  }
  if (lane == 0) out[o] = red[0];
}

int main(int argc, char **argv) {
  const uint32_t proofs = argc > 1 ? (uint32_t)atoi(argv[1]) : 1024u;
  const uint32_t n_gen = 516, terms = 260, outputs = 2 * proofs, stride = 2 * 256 + 4;
  const FbGeom geo = fb_geometry(n_gen);
  const size_t tbl_n = (size_t)n_gen * fb_stride(geo);
  printf("fb_probe: %u outputs x %u terms, %u-bit windows (%u x %u entries), table %.2f GB\n", outputs, terms, geo.wbits, geo.windows, geo.entries,
         tbl_n * 128.0 / 1e9);
  fbent *d_tbl;
  if (hipMalloc(&d_tbl, tbl_n * sizeof(fbent)) != hipSuccess) return 1;
  hipLaunchKernelGGL(k_fill_tbl, dim3((uint32_t)((tbl_n + 255) / 256)), dim3(256), 0, 0, d_tbl, tbl_n);
  std::mt19937 rng(777);
  std::vector<sc> h_s((size_t)outputs * stride);
  std::vector<uint32_t> h_g((size_t)outputs * stride), h_c(outputs, terms);
  for (size_t k = 0; k < h_s.size(); k++) {
    for (int j = 0; j < 8; j++) h_s[k].v[j] = rng();
    h_s[k].v[7] &= 0x0fffffffu;
    h_g[k] = rng() % n_gen;
  }
  sc *d_s;
  uint32_t *d_g, *d_c;
  ge *d_out, *d_part;
  const uint32_t max_parts = 8;
  (void)hipMalloc(&d_s, h_s.size() * sizeof(sc));
  (void)hipMalloc(&d_g, h_g.size() * 4);
  (void)hipMalloc(&d_c, h_c.size() * 4);
  (void)hipMalloc(&d_out, (size_t)outputs * sizeof(ge));
  (void)hipMalloc(&d_part, (size_t)outputs * max_parts * 64 * sizeof(ge));
  (void)hipMemcpy(d_s, h_s.data(), h_s.size() * sizeof(sc), hipMemcpyHostToDevice);
  (void)hipMemcpy(d_g, h_g.data(), h_g.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(d_c, h_c.data(), h_c.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1, e2;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  (void)hipEventCreate(&e2);
  auto best_of = [&](auto &&launch) {
    float best = 1e9f;
    for (int r = 0; r < 6; r++) {
      (void)hipEventRecord(e0);
      launch();
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      if (r) best = std::min(best, ms);
    }
    return best;
  };
  const double adds = (double)outputs * terms * geo.items;
  float a256 = 0;
  for (uint32_t thr : {256u, 192u, 128u}) {
    const float a = best_of([&] { hipLaunchKernelGGL(k_fb_msm, dim3(outputs), dim3(thr), 0, 0, d_s, d_g, d_c, stride, d_tbl, geo, d_out, 0u); });
    if (thr == 256u) a256 = a;
    printf("A k_fb_msm, %3u lanes per output        : %.3f ms  (%.1f G additions/s)\n", thr, a, adds / a / 1e6);
  }
  std::vector<ge> ref(outputs), got(outputs);
  (void)hipMemcpy(ref.data(), d_out, (size_t)outputs * sizeof(ge), hipMemcpyDeviceToHost);
  const float b = best_of([&] { hipLaunchKernelGGL(k_fb_msm_notree, dim3(outputs), dim3(256), 0, 0, d_s, d_g, d_c, stride, d_tbl, geo, d_out); });
  printf("B the same without the tree (256 lanes)  : %.3f ms  (%.1f G additions/s)\n", b, adds / b / 1e6);
  const float dd = best_of([&] { hipLaunchKernelGGL(k_fb_msm_w4, dim3(outputs), dim3(256), 0, 0, d_s, d_g, d_c, stride, d_tbl, geo, d_out); });
  printf("D k_fb_msm bounded for 4 wavefronts per SIMD: %.3f ms  (%.1f G additions/s)\n", dd, adds / dd / 1e6);
  for (uint32_t parts : {3u, 5u}) {
    float sum_ms = 0;
    const float c = best_of([&] {
      hipLaunchKernelGGL(k_fb_part, dim3(outputs * parts), dim3(64), 0, 0, d_s, d_g, d_c, stride, parts, d_tbl, geo, d_part, 0u);
      (void)hipEventRecord(e2);
      hipLaunchKernelGGL(k_fb_sum, dim3(outputs), dim3(64), 0, 0, d_part, parts, d_out);
    });
    (void)hipEventElapsedTime(&sum_ms, e2, e1);
    printf("C k_fb_part + k_fb_sum, %u slices (%3u terms): %.3f ms  (%.1f G additions/s; the sum %.3f ms)\n", parts, (terms + parts - 1) / parts, c,
           adds / c / 1e6, sum_ms);
  }
  {
    const float g1 = best_of([&] { hipLaunchKernelGGL((k_fb_msm_wave<128, 1>), dim3(outputs), dim3(64), 0, 0, d_s, d_g, d_c, stride, d_tbl, geo, d_out); });
    printf("G one wavefront per output, 128-term chunks, 1 line ahead : %.3f ms  (%.1f G additions/s; %.2fx the time of A)\n", g1, adds / g1 / 1e6, g1 / a256);
    (void)hipMemcpy(got.data(), d_out, (size_t)outputs * sizeof(ge), hipMemcpyDeviceToHost);
    const float g2 = best_of([&] { hipLaunchKernelGGL((k_fb_msm_wave<128, 2>), dim3(outputs), dim3(64), 0, 0, d_s, d_g, d_c, stride, d_tbl, geo, d_out); });
    printf("G one wavefront per output, 128-term chunks, 2 lines ahead: %.3f ms  (%.1f G additions/s; %.2fx the time of A)\n", g2, adds / g2 / 1e6, g2 / a256);
    const float g3 = best_of([&] { hipLaunchKernelGGL((k_fb_msm_wave<64, 1>), dim3(outputs), dim3(64), 0, 0, d_s, d_g, d_c, stride, d_tbl, geo, d_out); });
    printf("G one wavefront per output,  64-term chunks, 1 line ahead : %.3f ms  (%.1f G additions/s; %.2fx the time of A)\n", g3, adds / g3 / 1e6, g3 / a256);
    const float g4 = best_of([&] { hipLaunchKernelGGL((k_fb_msm_wave<260, 1>), dim3(outputs), dim3(64), 0, 0, d_s, d_g, d_c, stride, d_tbl, geo, d_out); });
    printf("G one wavefront per output, 260-term chunks, 1 line ahead : %.3f ms  (%.1f G additions/s; %.2fx the time of A)\n", g4, adds / g4 / 1e6, g4 / a256);
  }
  (void)hipFree(d_tbl);
  for (uint32_t wb : {11u, 12u}) {  // E: 64-byte entries, the same table footprint at one more window bit
    FbGeom g2;
    g2.wbits = wb;
    g2.windows = (254 + wb - 1) / wb;
    g2.entries = 1u << (wb - 1);
    g2.items = (253u % wb == 0u) ? 253u / wb : g2.windows;
    const size_t n2 = (size_t)n_gen * fb_stride(g2);
    fbxy *d_xy;
    if (hipMalloc(&d_xy, n2 * sizeof(fbxy)) != hipSuccess) return 1;
    hipLaunchKernelGGL(k_fill_xy, dim3((uint32_t)((n2 + 255) / 256)), dim3(256), 0, 0, d_xy, n2);
    const double adds2 = (double)outputs * terms * g2.items;
    for (uint32_t thr : {256u, 128u}) {
      const float e = best_of([&] { hipLaunchKernelGGL(k_fb_msm_xy<true>, dim3(outputs), dim3(thr), 0, 0, d_s, d_g, d_c, stride, d_xy, g2, d_out); });
      printf("E 64-byte entries, %2u-bit windows (%u additions per term, table %.2f GB), %3u lanes: %.3f ms  (%.1f G additions/s; %.2fx the time of A)\n",
             wb, g2.items, n2 * 64.0 / 1e9, thr, e, adds2 / e / 1e6, e / a256);
    }
    const float e2 = best_of([&] { hipLaunchKernelGGL(k_fb_msm_xy<false>, dim3(outputs), dim3(256), 0, 0, d_s, d_g, d_c, stride, d_xy, g2, d_out); });
    printf("E the same without the tree (256 lanes)   : %.3f ms  (%.1f G additions/s)\n", e2, adds2 / e2 / 1e6);
    (void)hipFree(d_xy);
  }
  printf("[%s]\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
