// Microbenchmark (gfx950) for VERDICT r2 item 6: "keep hot buckets on chip".  The engine's k_msm_accumulate keeps one
// accumulator per lane in registers, stores every bucket (160 B) to memory and k_msm_window_rc loads them again for the
// row / column reduction: 241 MB of stores + as much of loads per 65 536-proof step.  The on-chip form measured here: ONE
// workgroup of 512 lanes owns half a window (512 buckets x 160 B = 80 KB of LDS), accumulates with one lane per bucket exactly
// as the engine does, parks the accumulators in LDS and reduces them there (sum_j j B_j and sum_j B_j of its half; the two
// halves of a window combine with 9 doublings + 2 additions, not part of either measurement).
//   A  engine form: k_msm_accumulate (group-level size order) + k_msm_window_rc
//   B  on-chip form: k_acc_rc_lds (size order inside each half window: a workgroup has to own the buckets it reduces)
// Same synthetic lists as acc_probe mode 0 (64 groups, 23 windows x 1024 buckets, 16 514 points per group, multinomial sizes).
// Prints both times; run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE for the traffic.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I bulletproofs-plus_amd/csrc -o tools/microbench/lds_buckets tools/microbench/lds_buckets.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <random>
#include <vector>
#include "msm.h"
using namespace bpp;

__global__ void k_fill(niels *t, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  niels q;
  for (int k = 0; k < 10; k++) {
    q.yplusx.v[k] = (i * 2654435761u + k * 40503u) & 0x1ffffff;
    q.yminusx.v[k] = (i * 40503u + k * 2654435761u) & 0x1ffffff;
    q.xy2d.v[k] = (i + k * 977u) & 0x1ffffff;
  }
  t[i] = q;
}

// One workgroup per (group, window, half): lane l accumulates bucket order_half[l] (buckets of the half by descending size),
// the accumulators go to LDS indexed by bucket, the first wavefront reduces the 16 x 32 grid of the half as k_msm_window_rc
// does for a whole window.  out[wg] = (sum_j j B_j, sum_j B_j).
template <uint32_t HALF>
__global__ void __launch_bounds__(HALF, 1) k_acc_rc_lds(const uint32_t *__restrict__ sorted, const uint32_t *__restrict__ starts,
                                                        const uint32_t *__restrict__ counts, const uint32_t *__restrict__ order_half,
                                                        PointTables tabs, uint32_t G, uint32_t K, ge *__restrict__ out) {
  extern __shared__ uint32_t lds_raw[];
  ge *lb = reinterpret_cast<ge *>(lds_raw);  // [HALF]
  const uint32_t xcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
  const uint32_t per_g = K * (1024u / HALF), g = xcd + 8u * (j / per_g), kh = j % per_g;
  if (g >= G) return;
  const size_t hbase = ((size_t)g * per_g + kh) * HALF;  // first bucket of this half (global bucket id)
  const uint32_t lane = threadIdx.x;
  {
    const uint32_t bkt = order_half[hbase + lane];
    const uint32_t a = starts[bkt], n = counts[bkt];
    ge acc;
    ge_identity(acc);
    if (n) {
      uint32_t e = sorted[a];
      niels q;
      niels_load_swapped(q, point_ptr(tabs, e & 0x7fffffffu), (e >> 31) != 0);
      ge_from_niels_first(acc, q);
      fe_fence(acc.X);
      fe_fence(acc.Y);
      fe_fence(acc.T);
      if (n > 1) {
        e = sorted[a + 1];
        niels_load_swapped(q, point_ptr(tabs, e & 0x7fffffffu), (e >> 31) != 0);
      }
      uint32_t i = 1;
      while (i < n) {
        uint32_t e2 = e;
        niels q2;
        if (i + 1 < n) {
          e2 = sorted[a + i + 1];
          niels_load_swapped(q2, point_ptr(tabs, e2 & 0x7fffffffu), (e2 >> 31) != 0);
        }
        ge_madd_swapped(acc, acc, q, (e >> 31) != 0);
        if (++i >= n) break;
        if (i + 1 < n) {
          e = sorted[a + i + 1];
          niels_load_swapped(q, point_ptr(tabs, e & 0x7fffffffu), (e >> 31) != 0);
        }
        ge_madd_swapped(acc, acc, q2, (e2 >> 31) != 0);
        ++i;
      }
    }
    lb[bkt - hbase] = acc;  // an empty bucket parks the identity: the reduction needs no counts
  }
  __syncthreads();
  if (lane >= 64) return;  // the reduction is one wavefront's work (48 busy lanes), as in k_msm_window_rc
  // grid of the half: 32 rows x 16 columns, bucket j0 = 16 a + b holds digit j0 + 1
  const uint32_t Bc = 16, A = HALF / 16u, lBc = 4;
  __shared__ ge red[64];
  ge acc;
  ge_identity(acc);
  const bool is_row = lane < 32;
  const uint32_t idx = is_row ? lane : lane - 32;
  if (is_row ? (idx < A) : (idx < Bc)) {
    const uint32_t cnt = is_row ? Bc : A;
    for (uint32_t q = 0; q < cnt; q++) {
      const uint32_t j0 = is_row ? (Bc * idx + q) : (Bc * q + idx);
      const ge x = lb[j0];
      ge_add(acc, acc, x);
    }
  }
  red[lane] = acc;
  __syncthreads();
  const uint32_t half_n = is_row ? A : Bc;
  for (uint32_t off = 1; off < 32; off <<= 1) {
    ge y2;
    const bool act = idx + off < half_n;
    if (act) y2 = red[lane + off];
    __syncthreads();
    if (act) {
      ge_add(acc, acc, y2);
      red[lane] = acc;
    }
    __syncthreads();
  }
  const ge total = red[0];  // suffix_0 of the rows = sum of every bucket of the half
  if (is_row ? (idx == 0 || idx >= A) : (idx >= Bc)) ge_identity(acc);
  red[lane] = acc;
  __syncthreads();
  for (uint32_t off = 16; off >= 1; off >>= 1) {
    if (idx < off) {
      ge x = red[lane], y2 = red[lane + off];
      ge_add(x, x, y2);
      red[lane] = x;
    }
    __syncthreads();
  }
  if (lane == 0) {
    ge rows = red[0];
    const ge cols = red[32];
    ge_dbl_n(rows, rows, (int)lBc);
    ge_add(rows, rows, cols);
    out[2 * (size_t)(blockIdx.x)] = rows;
    out[2 * (size_t)(blockIdx.x) + 1] = total;
  }
}

template <uint32_t PIECE>
static float run_piece(const uint32_t *d_sorted, const uint32_t *d_starts, const uint32_t *d_counts, const std::vector<uint32_t> &counts, PointTables tabs, uint32_t G,
                       uint32_t K, ge *d_out) {
  // each piece's buckets by descending size
  std::vector<uint32_t> order(counts.size());
  for (size_t base = 0; base < counts.size(); base += PIECE) {
    std::vector<uint32_t> ids(PIECE);
    for (uint32_t i = 0; i < PIECE; i++) ids[i] = (uint32_t)base + i;
    std::stable_sort(ids.begin(), ids.end(), [&](uint32_t a, uint32_t b) { return counts[a] > counts[b]; });
    for (uint32_t i = 0; i < PIECE; i++) order[base + i] = ids[i];
  }
  uint32_t *d_order;
  (void)hipMalloc(&d_order, order.size() * 4);
  (void)hipMemcpy(d_order, order.data(), order.size() * 4, hipMemcpyHostToDevice);
  (void)hipFuncSetAttribute((const void *)k_acc_rc_lds<PIECE>, hipFuncAttributeMaxDynamicSharedMemorySize, PIECE * sizeof(ge));
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float best = 1e9f;
  for (int r = 0; r < 6; r++) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k_acc_rc_lds<PIECE>, dim3(8 * ((G + 7) / 8) * K * (1024u / PIECE)), dim3(PIECE), PIECE * sizeof(ge), 0, d_sorted, d_starts, d_counts, d_order,
                       tabs, G, K, d_out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (r) best = std::min(best, ms);
  }
  (void)hipFree(d_order);
  return best;
}

int main() {
  const uint32_t G = 64, K = 23, NB = 1024, per_group = K * NB, pts = 16514, terms = pts;
  std::mt19937 rng(12345);
  std::vector<uint32_t> counts((size_t)G * per_group), starts((size_t)G * per_group), order((size_t)G * per_group), sorted;
  sorted.reserve((size_t)G * K * terms);
  for (uint32_t g = 0; g < G; g++) {
    for (uint32_t k = 0; k < K; k++) {
      std::vector<uint32_t> cnt(NB, 0);
      for (uint32_t i = 0; i < terms; i++) cnt[rng() % NB]++;
      for (uint32_t b = 0; b < NB; b++) {
        const size_t id = ((size_t)g * K + k) * NB + b;
        counts[id] = cnt[b];
        starts[id] = (uint32_t)sorted.size();
        for (uint32_t j = 0; j < cnt[b]; j++) sorted.push_back((g * pts + rng() % pts) | ((rng() & 1u) << 31));
      }
    }
    std::vector<uint32_t> ids(per_group);
    for (uint32_t i = 0; i < per_group; i++) ids[i] = g * per_group + i;
    std::stable_sort(ids.begin(), ids.end(), [&](uint32_t a, uint32_t b) { return counts[a] > counts[b]; });
    for (uint32_t i = 0; i < per_group; i++) order[(size_t)g * per_group + i] = ids[i];
  }
  uint32_t *d_sorted, *d_starts, *d_counts, *d_order;
  niels *d_tab;
  ge *d_b, *d_w, *d_out;
  (void)hipMalloc(&d_sorted, sorted.size() * 4);
  (void)hipMalloc(&d_starts, starts.size() * 4);
  (void)hipMalloc(&d_counts, counts.size() * 4);
  (void)hipMalloc(&d_order, order.size() * 4);
  (void)hipMalloc(&d_tab, (size_t)G * pts * sizeof(niels));
  (void)hipMalloc(&d_b, counts.size() * sizeof(ge));
  (void)hipMalloc(&d_w, (size_t)G * K * sizeof(ge));
  (void)hipMalloc(&d_out, (size_t)8 * ((G + 7) / 8) * K * 4 * 2 * sizeof(ge));
  (void)hipMemcpy(d_sorted, sorted.data(), sorted.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(d_starts, starts.data(), starts.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(d_counts, counts.data(), counts.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(d_order, order.data(), order.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_fill, dim3((G * pts + 255) / 256), dim3(256), 0, 0, d_tab, G * pts);
  PointTables tabs{d_tab, d_tab, 0xffffffffu, nullptr, nullptr};
  const MsmPlan plan = msm_make_plan(11, G, G * terms);
  hipEvent_t e0, e1, e2;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  (void)hipEventCreate(&e2);
  float a_acc = 1e9f, a_rc = 1e9f;
  for (int r = 0; r < 6; r++) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k_msm_accumulate, dim3(8 * ((G + 7) / 8) * ((per_group + 63) / 64)), dim3(64), 0, 0, d_sorted, d_starts, d_counts, d_order, tabs,
                       per_group, G, d_b);
    (void)hipEventRecord(e1);
    hipLaunchKernelGGL(k_msm_window_rc, dim3(G * K), dim3(64), 0, 0, d_b, d_counts, plan, d_w);
    (void)hipEventRecord(e2);
    (void)hipEventSynchronize(e2);
    float m1, m2;
    (void)hipEventElapsedTime(&m1, e0, e1);
    (void)hipEventElapsedTime(&m2, e1, e2);
    if (r) a_acc = std::min(a_acc, m1), a_rc = std::min(a_rc, m2);
  }
  const float b512 = run_piece<512>(d_sorted, d_starts, d_counts, counts, tabs, G, K, d_out);
  const float b256 = run_piece<256>(d_sorted, d_starts, d_counts, counts, tabs, G, K, d_out);
  hipError_t err = hipGetLastError();
  printf("A engine form   : k_msm_accumulate %.3f ms + k_msm_window_rc %.3f ms = %.3f ms (buckets through memory: %.0f MB stored, as much loaded)\n", a_acc, a_rc,
         a_acc + a_rc, counts.size() * 160.0 / 1e6);
  printf("B on-chip, 512  : k_acc_rc_lds<512> %.3f ms (80 KB of LDS per workgroup of 512 lanes, %u workgroups)\n", b512, G * K * 2);
  printf("B on-chip, 256  : k_acc_rc_lds<256> %.3f ms (40 KB of LDS per workgroup of 256 lanes, %u workgroups)   [%s]\n", b256, G * K * 4, hipGetErrorString(err));
  return 0;
}
