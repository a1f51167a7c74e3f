// libbpp_hip.so -- host side of the C ABI in include/bpp.h.  No CPU fallback: every entry point needs a gfx950 device.
//
// Host responsibilities (mirroring what the reference does around its hot loops):
//   * argument / wire-format checks with the reference's error kinds and precedence
//     (src/range_proof.rs:719-734, :610-709, :875-888, :1155-1257; src/range_statement.rs:36-73)
//   * packing a batch into the HBM layout documented in kernels_verify.h
//   * the batch-weight transcript (src/range_proof.rs:811,849,853,894): a sequential sponge, run on the host
//   * SHAKE256 / SHA3-512 byte streams for generator derivation (src/generators/generators_chain.rs:23-33,
//     src/protocols/curve_point_protocol.rs:31-35); the hash-to-group map itself runs on the device
#include <hip/hip_runtime.h>
#include <errno.h>
#include <sched.h>
#include <stdio.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <map>
#include <set>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/bpp.h"
#include "chain_dev.h"
#include "chain_host.h"
#include "ct.h"
#include "kernels_prove.h"
#include "kernels_verify.h"
#include "msm.h"
#include "upload_host.h"

using namespace bpp;

namespace {

// ------------------------------------------------------------------ utilities
struct EngineError {
  int code;
  std::string msg;
};

#define HIP_CHECK(expr)                                                                          \
  do {                                                                                           \
    hipError_t _e = (expr);                                                                      \
    if (_e != hipSuccess) {                                                                      \
      char _b[256];                                                                              \
      snprintf(_b, sizeof(_b), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      throw EngineError{BPP_ERR_ENGINE, _b};                                                     \
    }                                                                                            \
  } while (0)

template <typename T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  DevBuf() {}
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  void alloc(size_t count) {
    if (count <= n && p) return;
    release();
    if (count == 0) count = 1;
    HIP_CHECK(hipMalloc((void **)&p, count * sizeof(T)));
    n = count;
  }
  void swap(DevBuf &o) {
    std::swap(p, o.p);
    std::swap(n, o.n);
  }
};

// page-locked host memory: asynchronous copies to/from it do not stall other streams or host threads.  It is also MAPPED
// into the device's address space (coherent, never cached by the GPU): the verifier's kernels write their small results --
// transcript-RNG bytes, status words, identity flags, masks -- straight into it and read the batch weights from it, so a
// verification enqueues no copy at all (each hipMemcpyAsync of a few KB ran as a blit kernel of its own that queued behind
// the other steps' 15 k-wavefront launches: five per step, 0.19 ms each with four steps in flight, two of them on the
// PASS 1 -> weight chain -> PASS 2 critical path)
template <typename T>
struct PinnedBuf {
  T *p = nullptr;
  T *d = nullptr;  // the same memory as the device sees it
  size_t n = 0;
  PinnedBuf() {}
  PinnedBuf(const PinnedBuf &) = delete;
  PinnedBuf &operator=(const PinnedBuf &) = delete;
  ~PinnedBuf() {
    if (p) (void)hipHostFree(p);
  }
  void resize(size_t count) {
    if (count <= n && p) return;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    d = nullptr;
    if (count == 0) count = 1;
    HIP_CHECK(hipHostMalloc((void **)&p, count * sizeof(T), hipHostMallocMapped | hipHostMallocPortable));
    HIP_CHECK(hipHostGetDevicePointer((void **)&d, p, 0));
    n = count;
  }
  T *dev() { return d; }
  T *data() { return p; }
  const T *data() const { return p; }
  T &operator[](size_t i) { return p[i]; }
  const T &operator[](size_t i) const { return p[i]; }
  void swap(PinnedBuf &o) {
    std::swap(p, o.p);
    std::swap(d, o.d);
    std::swap(n, o.n);
  }
};

// runs on every exit path of a scope (return, ProofErr, EngineError): used to wipe secret-bearing buffers, as the
// reference does with Zeroizing<> (src/range_proof.rs:300-301,325,438-464, src/commitment_opening.rs:14)
struct ScopeExit {
  std::function<void()> f;
  ~ScopeExit() {
    if (f) f();
  }
};
inline void wipe(void *p, size_t n) {
  if (p && n) explicit_bzero(p, n);
}

void set_err(char *errbuf, size_t len, const std::string &m) {
  if (errbuf && len) {
    snprintf(errbuf, len, "%s", m.c_str());
  }
}

// a whole decimal number in [lo, hi] from an environment variable; false (and *out untouched) for anything else -- a typo must
// not silently become 0, which for a deadline means "none" and for the small-call gate "off"
bool parse_env_long(const char *text, long lo, long hi, long *out) {
  if (!text || !*text) return false;
  char *end = nullptr;
  errno = 0;
  const long v = strtol(text, &end, 10);
  while (end && (*end == ' ' || *end == '\t')) end++;
  if (errno || end == text || (end && *end) || v < lo || v > hi) return false;
  *out = v;
  return true;
}

// ------------------------------------------------------------------ host hashing helpers (keccak from merlin.h)
void keccak_sponge(const uint8_t *in, size_t inlen, uint8_t *out, size_t outlen, uint32_t rate, uint8_t pad) {
  uint64_t st[25];
  memset(st, 0, sizeof(st));
  uint8_t *sb = (uint8_t *)st;  // little-endian host
  size_t off = 0;
  while (inlen - off >= rate) {
    for (uint32_t i = 0; i < rate; i++) sb[i] ^= in[off + i];
    keccak_f1600(st);
    off += rate;
  }
  for (size_t i = 0; i < inlen - off; i++) sb[i] ^= in[off + i];
  sb[inlen - off] ^= pad;
  sb[rate - 1] ^= 0x80;
  keccak_f1600(st);
  size_t o = 0;
  while (o < outlen) {
    size_t take = (outlen - o < rate) ? outlen - o : rate;
    memcpy(out + o, sb, take);
    o += take;
    if (o < outlen) keccak_f1600(st);
  }
}
void shake256(const uint8_t *in, size_t inlen, uint8_t *out, size_t outlen) { keccak_sponge(in, inlen, out, outlen, 136, 0x1f); }
void sha3_512(const uint8_t *in, size_t inlen, uint8_t out[64]) { keccak_sponge(in, inlen, out, 64, 72, 0x06); }

void run_weight_chains_generic(const uint8_t *h_rng, uint8_t *h_weights, const uint32_t *group_first, uint32_t G, bool wide = false);
void host_parallel_for(uint32_t n, const std::function<void(uint32_t)> &fn);  // on the persistent host pool
uint32_t host_pool_size();
uint64_t host_pool_cpu_ns();

// weight transcript (src/range_proof.rs:811,849,853,894)
void weights_from_chain_host(const uint8_t *rng32, size_t n, uint8_t *weights32) {
  static const bool generic = getenv("BPP_CHAIN_GENERIC") && atoi(getenv("BPP_CHAIN_GENERIC"));  // tests: merlin.h's sponge
  if (generic) weights_chain_generic(rng32, n, weights32);
  else weights_chain_single(rng32, n, weights32);
}

// ------------------------------------------------------------------ objects
// RangeParameters / Precomputation objects are process-wide, reference-counted and read-only once built (the reference
// shares them through Arc, `Precomputation: Send + Sync`, src/traits.rs:42, src/generators/bulletproof_gens.rs:52,103):
// any context of the same device may use a handle concurrently; only work buffers are per context.
struct Params {
  int device = 0;
  std::mutex fb_mu;  // serialises the one-off build of the prover's fixed-base table
  uint32_t n_bits, m_max, t;
  DevBuf<niels> table;  // [2*n*m_max interleaved G,H | t g_bases | h_base]
  DevBuf<niels> table_hi;  // the same points times 2^126: the half-scalar MSM plan of small calls (msm.h)
  uint32_t table_len;
  DevBuf<uint8_t> d_hg32;           // compressed H, G_0..G_{t-1}
  std::vector<uint8_t> hg32;        // host copy
  std::vector<uint8_t> gi32, hi32;  // compressed generators, party-major
  DevBuf<fbent> fb_table;           // prover's fixed-base window table, built on first use (geometry fb_geo)
  FbGeom fb_geo{}, fb_ped_geo{};
  DevBuf<fbent> fb_ped;             // same for the Pedersen bases only [G_0..G_{t-1}, H], always resident (commit)
  DevBuf<fbent> fb_ct;              // the Pedersen bases' 4-bit lines for the uniform-access form (ct.h: k_ct_fixed), [base][64][8]
};

struct Precomp {
  int device = 0;
  DevBuf<niels> table;
  uint32_t count;
};

template <typename T>
struct SharedRegistry {
  struct Entry {
    std::shared_ptr<T> obj;
    uint32_t refs;
  };
  std::mutex mu;
  std::map<uint64_t, Entry> live;
  uint64_t add(std::shared_ptr<T> o);
  std::shared_ptr<T> get(uint64_t h) {
    std::lock_guard<std::mutex> lk(mu);
    auto it = live.find(h);
    return it == live.end() ? nullptr : it->second.obj;
  }
  bool retain(uint64_t h) {
    std::lock_guard<std::mutex> lk(mu);
    auto it = live.find(h);
    if (it == live.end()) return false;
    it->second.refs++;
    return true;
  }
  // drops one reference; the object itself lives on while a call or a resident batch still holds its shared_ptr
  bool release(uint64_t h) {
    std::shared_ptr<T> dying;
    std::lock_guard<std::mutex> lk(mu);
    auto it = live.find(h);
    if (it == live.end()) return false;
    if (--it->second.refs == 0) {
      dying = std::move(it->second.obj);
      live.erase(it);
    }
    return true;
  }
};
std::atomic<uint64_t> g_next_handle{1};
template <typename T>
uint64_t SharedRegistry<T>::add(std::shared_ptr<T> o) {
  const uint64_t h = g_next_handle.fetch_add(1);
  std::lock_guard<std::mutex> lk(mu);
  live[h] = Entry{std::move(o), 1};
  return h;
}
SharedRegistry<Params> &params_registry() {
  static SharedRegistry<Params> *r = new SharedRegistry<Params>();  // leaked: contexts may outlive static destruction
  return *r;
}
SharedRegistry<Precomp> &precomp_registry() {
  static SharedRegistry<Precomp> *r = new SharedRegistry<Precomp>();
  return *r;
}

// ------------------------------------------------------------------ per-device runtime state: admission gate, bookkeeping
// A SMALL call (one reference batch of up to a few hundred proofs: what separate callers of RangeProof::verify_batch issue,
// src/range_proof.rs:73-76,712-752) is a chain of about fifteen latency-bound kernels with tiny grids.  The chip runs about
// six such chains side by side; more callers than that add nothing, and many more take it away: 8-16 callers with a context
// each reached 5.3 k calls/s, 32 callers 2.6 k, 64 callers 1.6 k (profiles/r03_v10_calls_in_flight.jsonl) -- every context
// brings a stream and a side stream, the runtime multiplexes them onto its hardware queues (GPU_MAX_HW_QUEUES, 4 unless the
// host sets more) and streams that share a queue serialise behind each other's kernels.  So the library admits at most
// `limit` small calls per device at a time (BPP_SMALL_CALLS_IN_FLIGHT, default 12; bpp_small_call_limit); the others wait
// their turn in arrival order, on the host, holding nothing.  Large calls (they fill the chip by themselves) pass freely.
struct DeviceState {
  std::mutex mu;
  struct Waiter {
    std::condition_variable cv;
    bool go = false;
  };
  std::deque<Waiter *> queue;
  uint32_t limit = 12, in_flight = 0;
  uint64_t small_calls = 0, small_calls_queued = 0;
  uint32_t contexts = 0, contexts_peak = 0;
  bool warned = false;
  std::string env_note;  // a malformed BPP_SMALL_CALLS_IN_FLIGHT: said once, where bpp_ctx_last_error finds it
};
DeviceState &device_state(int dev) {
  static std::mutex mu;
  static std::map<int, DeviceState *> *all = new std::map<int, DeviceState *>();  // leaked: contexts may outlive static destruction
  std::lock_guard<std::mutex> lk(mu);
  DeviceState *&d = (*all)[dev];
  if (!d) {
    d = new DeviceState();
    if (const char *e = getenv("BPP_SMALL_CALLS_IN_FLIGHT")) {  // 0: no gate
      long v = 0;
      if (parse_env_long(e, 0, 1 << 20, &v)) d->limit = (uint32_t)v;
      else d->env_note = std::string("note: BPP_SMALL_CALLS_IN_FLIGHT=\"") + e + "\" is not a number: the default gate (12 calls) stands";
    }
  }
  return *d;
}
// hardware queues the HIP runtime multiplexes this process' streams onto: read once at its start-up from GPU_MAX_HW_QUEUES
uint32_t runtime_hw_queues() {
  static const uint32_t q = [] {
    const char *e = getenv("GPU_MAX_HW_QUEUES");
    const int v = e ? atoi(e) : 0;
    return v > 0 ? (uint32_t)v : 4u;
  }();
  return q;
}
thread_local int tl_gate_depth = 0;  // a thread that holds the gate passes it again (upload + verify of one call)
struct GateHold {
  DeviceState *d = nullptr;
  GateHold(int dev, bool small) {
    if (!small) return;
    if (tl_gate_depth++ > 0) {
      d = nullptr;
      held_outer = true;
      return;
    }
    DeviceState &D = device_state(dev);
    std::unique_lock<std::mutex> lk(D.mu);
    D.small_calls++;
    if (D.limit == 0) {  // gate switched off: counted, never held
      tl_gate_depth--;
      return;
    }
    d = &D;
    if (D.in_flight < D.limit && D.queue.empty()) {
      D.in_flight++;
      return;
    }
    D.small_calls_queued++;
    DeviceState::Waiter w;
    D.queue.push_back(&w);
    w.cv.wait(lk, [&] { return w.go; });  // the releasing thread has already counted this call in
  }
  ~GateHold() {
    if (held_outer) {
      tl_gate_depth--;
      return;
    }
    if (!d) return;
    tl_gate_depth--;
    std::lock_guard<std::mutex> lk(d->mu);
    d->in_flight--;
    while (!d->queue.empty() && d->in_flight < d->limit) {
      DeviceState::Waiter *w = d->queue.front();
      d->queue.pop_front();
      d->in_flight++;
      w->go = true;
      w->cv.notify_one();
    }
  }
  GateHold(const GateHold &) = delete;
  GateHold &operator=(const GateHold &) = delete;

 private:
  bool held_outer = false;
};
// calls of up to this many proofs are "small" for the gate (their MSM has at most ~20 000 terms: BPP_SMALL_CALL_TERMS)
#define BPP_GATE_SMALL_PROOFS 1024u
// proof + commitment bytes of an upload up to this size travel with the small arrays in k_ingest; above it by DMA
#define BPP_INGEST_MAX_BYTES (1u << 20)

struct MsmWork {
  DevBuf<uint32_t> counts, starts, sorted, order, order_win, cls_hist;
  DevBuf<ge> buckets, Q, W, R;
  DevBuf<uint8_t> comp32;
  DevBuf<uint32_t> is_identity;
  DevBuf<uint32_t> term_sidx, term_pidx, group_off;
  MsmPlan plan{};
  uint32_t max_group_terms = 0;
  bool split = false;  // half-scalar plan (small verifier calls): every term twice, 127-bit windows
};

struct Batch {
  std::shared_ptr<Params> params;
  uint64_t params_handle = 0;
  uint32_t B = 0, rmax = 0, cs = 0, max_mn = 0, total_dyn = 0, sum_m = 0, cols = 0;
  // shape of the per-proof table block of the scalar stage (kernels_verify.h: lanes_tab_stride)
  // A proof that claims more rounds than its statement's m * n_bits has bits is refused on the host; the table blocks are
  // sized for the statements (max_mn), so that one such proof cannot inflate the allocation of the whole batch.
  uint32_t lanes_nhi_max(uint32_t n_bits) const {
    uint32_t cap = 0;
    while ((2u << cap) <= std::max<uint32_t>(max_mn, 1)) cap++;  // floor(log2(max_mn))
    const uint32_t rm = std::min({rmax, (uint32_t)BPP_MAX_ROUNDS - 1, cap}), lb = lanes_lb(n_bits);
    return 1u << (rm > lb ? rm - lb : 0);
  }
  std::vector<ProofDesc> desc;
  std::vector<uint8_t> rounds_bad;  // 0 ok, 3 InvalidLength, 5 SizeOverflow  (src/range_proof.rs:875-888)
  std::vector<uint8_t> defer;       // BPP_DEFER_* findings of verify()'s consistency loops (:637-682), raised per chunk
  bool any_seed = false, any_rounds_bad = false, any_defer = false, ext_challenges = false, uniform_rounds = true;
  std::vector<uint32_t> ext_status;
  DevBuf<uint32_t> d_ext_status;
  // device-resident inputs
  DevBuf<uint8_t> bytes, states, seeds;
  DevBuf<ProofDesc> d_desc;
  DevBuf<uint64_t> minvals;
  DevBuf<uint32_t> src_off, owner;
  DevBuf<uint32_t> idx_commit, idx_proof, status0;  // dynamic slots of the commitments / of the proof points; initial status
  // device work buffers
  DevBuf<sc> chal, rows, scal, shr, tab;
  DevBuf<uint64_t> parts;      // per-workgroup limb sums of the generator columns (k_scalars_lanes -> k_reduce_parts)
  bool fused_columns = false;  // this layout sums the generator columns inside k_scalars_lanes (layout_groups decides)
  uint32_t lanes_ppw = 1;      // proofs per workgroup of k_scalars_lanes for this batch
  // generator columns as a matrix product over the proofs of a group (kernels_static_gemm.h): digit tables, anti-diagonal sums
  DevBuf<int8_t> gemm_lo, gemm_hi;
  DevBuf<int32_t> gparts;
  DevBuf<sc> gemm_mult;        // per proof: w, -w e^2 (Montgomery), -w e^2, -w e^2 y^(mn+1) (canonical), the weight as it came
  bool static_gemm = false;    // this layout takes the generator columns from k_static_gemm (plan_lanes decides)
  uint32_t gemm_nkc = 1;       // K chunks (256 proofs each) of the largest group
  int uniform_mn = -1;         // -1 not looked at yet; 0 the proofs differ in shape; else the common m * n_bits
  bool gemm_hi_ready = false;  // the last k_scalars_shared of this batch wrote the high digit tables
  DevBuf<uint8_t> rng_out, weights, masks, chal_bytes;
  DevBuf<uint32_t> chain_wide;  // device chain: the 64 PRF bytes per proof as k_weight_chain leaves them (chain_dev.h)
  DevBuf<uint32_t> status, group_first, group_dlo;
  DevBuf<uint32_t> dec_spill;  // k_decompress parks three field elements per proof point here across its squaring chain
  DevBuf<niels> dynpts;
  DevBuf<pniels> dyn_hi;  // 2^126 x dynpts as projective Niels entries (half-scalar plan only)
  MsmWork msm;
  // layout of the last verify
  size_t last_chunk = (size_t)-1;
  uint32_t G = 0;
  std::vector<uint32_t> h_group_first;
  PinnedBuf<uint8_t> h_rng, h_weights, h_masks;  // mapped: written / read by the kernels directly (PinnedBuf)
  PinnedBuf<uint8_t> h_wide;  // chain mode 2: the host sponges' 64 bytes per proof, reduced on the device
  PinnedBuf<uint32_t> h_status, h_ident;
  // k_results_out writes one summary word per BPP_STATUS_BLOCK proofs and a block's words only when there is something in them;
  // settle_status() makes h_status whole again on the host (a block the kernel skipped is all zero: cleared here if it was not)
  PinnedBuf<uint32_t> h_status_any;
  std::vector<uint8_t> h_status_dirty;
  const uint32_t *h_status_dirty_of = nullptr;  // the allocation those flags describe (a re-allocated h_status holds anything)
  bool status_settled = true;
  bool have_trace = false, phase1_done = false;
  bool status_clean = false;     // status[] == status0[]: k_results_out resets it behind every verification
  bool dev_chain_pending = false;  // k_weight_chain of this call is on the chain stream: enqueue_phase2 waits for it
  bool weights_on_host = false;  // the last PASS 2 read its weights from h_weights (b.weights holds them only in the sharded forms)
  bool seeds_dirty = false, masks_dirty = false;  // device copies of seed nonces / recovered masks not yet wiped
};

// A destroyed batch leaves its device and pinned allocations to the next upload on the same context (a caller that
// verifies fresh host buffers call after call would otherwise pay ~30 hipMalloc/hipFree pairs, 15-25 ms, per call).
void adopt_buffers(Batch &dst, Batch &src) {
#define BPP_ADOPT(f) dst.f.swap(src.f)
  BPP_ADOPT(d_ext_status); BPP_ADOPT(bytes); BPP_ADOPT(states); BPP_ADOPT(seeds); BPP_ADOPT(d_desc); BPP_ADOPT(minvals);
  BPP_ADOPT(src_off); BPP_ADOPT(owner); BPP_ADOPT(idx_commit); BPP_ADOPT(idx_proof); BPP_ADOPT(status0); BPP_ADOPT(chal);
  BPP_ADOPT(rows); BPP_ADOPT(parts); BPP_ADOPT(gemm_lo); BPP_ADOPT(gemm_hi); BPP_ADOPT(gparts); BPP_ADOPT(gemm_mult); BPP_ADOPT(scal); BPP_ADOPT(shr); BPP_ADOPT(tab); BPP_ADOPT(rng_out); BPP_ADOPT(weights); BPP_ADOPT(chain_wide);
  BPP_ADOPT(masks); BPP_ADOPT(chal_bytes); BPP_ADOPT(status); BPP_ADOPT(group_first); BPP_ADOPT(group_dlo); BPP_ADOPT(dynpts); BPP_ADOPT(dyn_hi); BPP_ADOPT(dec_spill);
  BPP_ADOPT(msm.counts); BPP_ADOPT(msm.starts); BPP_ADOPT(msm.sorted); BPP_ADOPT(msm.order); BPP_ADOPT(msm.order_win);
  BPP_ADOPT(msm.cls_hist);
  BPP_ADOPT(msm.buckets); BPP_ADOPT(msm.Q); BPP_ADOPT(msm.W); BPP_ADOPT(msm.R); BPP_ADOPT(msm.comp32);
  BPP_ADOPT(msm.is_identity); BPP_ADOPT(msm.term_sidx); BPP_ADOPT(msm.term_pidx); BPP_ADOPT(msm.group_off);
  BPP_ADOPT(h_rng); BPP_ADOPT(h_weights); BPP_ADOPT(h_wide); BPP_ADOPT(h_status); BPP_ADOPT(h_status_any); BPP_ADOPT(h_ident); BPP_ADOPT(h_masks);
#undef BPP_ADOPT
  // what is known about the adopted status words travels with them (settle_status clears only the blocks that may hold something);
  // a verification whose results were never looked at leaves them unknown
  dst.h_status_dirty.swap(src.h_status_dirty);
  std::swap(dst.h_status_dirty_of, src.h_status_dirty_of);
  if (!src.status_settled) std::fill(dst.h_status_dirty.begin(), dst.h_status_dirty.end(), (uint8_t)1);
}

void wipe_batch_secrets(Batch &b, hipStream_t s);

}  // namespace

// pipelined host-buffers-in form (bpp_verify_submit_packed / bpp_verify_collect), implemented further down
struct PipeJob {
  uint64_t ticket = 0;
  int action = 0;
  size_t chunk = 0;
  std::shared_ptr<Params> Pp;
  uint64_t params = 0;
  UploadPlan pl;
  size_t n_items = 0;
  uint32_t t = 0;
  int rc = 0;
  std::string err;
  std::vector<uint8_t> masks, present;
  bool done = false;
};

struct PipeLane {
  bpp_ctx *child = nullptr;
  std::thread th;
  std::shared_ptr<PipeJob> job;  // posted by submit, taken by the worker
  bool busy = false;             // from the moment submit claims the lane until its job is done
};

struct Pipeline {
  std::mutex mu;                  // lanes' state, tickets
  std::condition_variable cv;
  std::mutex submit_mu;           // one submit at a time: lanes are claimed in ticket order
  std::vector<std::unique_ptr<PipeLane>> lanes;
  std::map<uint64_t, std::shared_ptr<PipeJob>> tickets;
  uint64_t next_ticket = 1;
  uint32_t next_lane = 0;
  bool quit = false;
};

void pipeline_shutdown(bpp_ctx *ctx);

// What a waiting site remembers: how long its wait took the last times (microseconds, a running mean) and for WHICH work -- a key the
// caller makes from the call's size.  A remembered time is only used for a call with the same key: a context that proved 8192
// proofs per call and then proves 1024 must not sleep through the shorter call on the longer one's memory (it did, for one
// build: 45 k proofs/s instead of 150 k in bench.py's prover leg, whose context had made the run's inputs before).
struct WaitHint {
  uint32_t us = 0;
  uint64_t key = 0;
};

struct bpp_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string err;
  std::mutex mu;
  // references this context holds on shared objects (created or retained here): released when the context dies
  std::multiset<uint64_t> held_params, held_precomps;
  std::map<uint64_t, std::unique_ptr<Batch>> batches;
  std::unique_ptr<Batch> spare_batch;  // buffers of the last destroyed batch (adopt_buffers)
  PinnedBuf<uint8_t> pin_upload, pin_upload2;  // page-locked staging of bpp_batch_upload (bytes; descriptors etc.)
  PinnedBuf<uint32_t> pin_small;       // staging for small host->device arrays (a pageable hipMemcpyAsync of a few
                                       // hundred bytes was measured at ~10 ms on this stack)
  bool profile = false;
  bool profile_light = false;  // bpp_profile_enable(ctx, 2): events around the roofline kernel (k_msm_accumulate) only
  bpp_profile prof{};
  hipEvent_t ev[16];
  bool ev_ready = false;
  hipEvent_t ev_rng;
  bool ev_rng_ready = false;
  hipEvent_t ev_wait;  // gpu_wait_stream's marker
  bool ev_wait_ready = false;
  WaitHint wait_hint_rng, wait_hint_end, wait_hint_prove;  // a verification's two waits, a prover call's wait (gpu_wait_event)
  // small inputs: decompression runs beside PASS 1 on a second stream (enqueue_phase1)
  hipStream_t side_stream = nullptr;
  hipEvent_t ev_fork, ev_join;
  // the weight chains as a kernel (chain_dev.h): their own stream beside decompression and the weight-free scalars, the
  // transcript Transcript::new(b"Bulletproofs+ verifier weights") leaves, and the "a weight was zero" word the host looks at
  // with the results
  hipStream_t chain_stream = nullptr;
  hipEvent_t ev_chain_fork, ev_chain_done;
  DevBuf<Strobe> d_chain_t0;
  PinnedBuf<uint32_t> h_chain_zero;
  uint64_t device_chain_calls = 0, device_chain_redraws = 0, wide_chain_calls = 0;
  DevBuf<uint8_t> scratch128;
  // batch prover: one device arena, page-locked staging and the sub-batch streams, all reused across calls
  DevBuf<uint8_t> prove_arena;
  PinnedBuf<uint8_t> prove_pin_in, prove_pin_out;
  std::vector<hipStream_t> prove_streams;
  // high-priority twins of the sub-batch streams: the prover's small latency-bound kernels (Fiat-Shamir step, vector fold,
  // point encoding) run on them so that they are not starved by the other sub-batch's chip-filling fixed-base MSM
  std::vector<hipStream_t> prove_lane_streams;
  hipStream_t prove_msm_stream = nullptr;  // the one stream every sub-batch's fixed-base MSM runs on, first in first out ("prove_fifo")
  std::vector<hipEvent_t> prove_sync_events;  // two per sub-batch: lane step done / fixed-base MSM done
  // the witness check of a sub-batch (commit(v_j, r_j) against the statement's commitments, :275-284) runs on a stream of its own
  // beside the call's first steps; two events per sub-batch: inputs resident / check done
  std::vector<hipStream_t> prove_aux_streams;
  std::vector<hipEvent_t> prove_aux_events;
  std::vector<hipEvent_t> prove_events;  // pairs around every k_fb_msm launch of the last bpp_prove_batch (profiling only)
  bpp_prove_profile pprof{};
  // knobs (tests, A/B timing).  -1 = the engine's own rule.  The BPP_* environment variables of the same names are read
  // ONCE, when the context is created (no getenv on any verification path: a host that calls setenv from another thread
  // would race with it); bpp_ctx_set_option changes them afterwards.
  struct Options {
    int transcripts_wave = -1, tables_wave = -1, side_decompress = -1, msm_c_bias = -1, msm_c_max = -1, msm_c_add = -1, msm_rc2 = -1, msm_quad = -1, msm_final_quad = -1,
        fb_threads = -1, prove_subs = -1, msm_split = -1, fused_columns = -1, prove_prio = -1, prove_fused = -1, static_gemm = -1, lazy_columns = -1, ct = -1, prove_parts = -1, prove_waves = -1, prove_fifo = -1, chain = -1, chain_test_zero = 0, wait = -1, ct_back = -1, chain_inline = -1, wide_in_lanes = -1;
  } opt;
  std::unique_ptr<Pipeline> pipe;  // bpp_verify_submit_packed / bpp_verify_collect: lanes, tickets (built on first submit)
  std::mutex pipe_init_mu;
  uint32_t pipe_depth = 3;
};

namespace {

// lanes per output point of k_fb_msm: enough (term, window) items per lane to outweigh the log2(lanes) reduction tree
static uint32_t fb_threads(const bpp_ctx *ctx, uint32_t terms, const FbGeom &g) {
  if (ctx->opt.fb_threads > 0) {  // any whole number of wavefronts up to the kernel's launch bound
    const uint32_t t = ((uint32_t)ctx->opt.fb_threads + 63u) & ~63u;
    return t > FB_THREADS ? FB_THREADS : t;
  }
  const uint32_t items = terms * g.windows;
  return items >= 4096 ? 256u : (items >= 1024 ? 128u : 64u);
}

struct OptionName {
  const char *name, *env;
  int bpp_ctx::Options::*field;
};
const OptionName kOptions[] = {
    {"transcripts_wave", "BPP_TRANSCRIPTS_WAVE", &bpp_ctx::Options::transcripts_wave},
    {"tables_wave", "BPP_TABLES_WAVE", &bpp_ctx::Options::tables_wave},
    {"side_decompress", "BPP_SIDE_DECOMPRESS", &bpp_ctx::Options::side_decompress},
    {"msm_c_bias", "BPP_MSM_C_BIAS", &bpp_ctx::Options::msm_c_bias},
    {"msm_c_max", "BPP_MSM_C_MAX", &bpp_ctx::Options::msm_c_max},
    {"msm_c_add", "BPP_MSM_C_ADD", &bpp_ctx::Options::msm_c_add},
    {"msm_rc2", "BPP_MSM_RC2", &bpp_ctx::Options::msm_rc2},
    {"msm_quad", "BPP_MSM_QUAD", &bpp_ctx::Options::msm_quad},
    {"msm_final_quad", "BPP_MSM_FINAL_QUAD", &bpp_ctx::Options::msm_final_quad},
    {"fb_threads", "BPP_FB_THREADS", &bpp_ctx::Options::fb_threads},
    {"prove_subs", "BPP_PROVE_SUBS", &bpp_ctx::Options::prove_subs},
    {"msm_split", "BPP_MSM_SPLIT", &bpp_ctx::Options::msm_split},
    {"fused_columns", "BPP_FUSED_COLUMNS", &bpp_ctx::Options::fused_columns},
    {"prove_prio", "BPP_PROVE_PRIO", &bpp_ctx::Options::prove_prio},
    {"prove_fused", "BPP_PROVE_FUSED", &bpp_ctx::Options::prove_fused},
    {"ct", "BPP_CT", &bpp_ctx::Options::ct},
    // "ct" = 2: how many rounds before the end the public points behind A1's folded generators are made (1..3; the engine's rule: 1)
    {"ct_back", "BPP_CT_BACK", &bpp_ctx::Options::ct_back},
    {"prove_parts", "BPP_PROVE_PARTS", &bpp_ctx::Options::prove_parts},
    {"prove_waves", "BPP_PROVE_WAVES", &bpp_ctx::Options::prove_waves},
    {"prove_fifo", "BPP_PROVE_FIFO", &bpp_ctx::Options::prove_fifo},
    {"static_gemm", "BPP_STATIC_GEMM", &bpp_ctx::Options::static_gemm},
    {"lazy_columns", "BPP_LAZY_COLUMNS", &bpp_ctx::Options::lazy_columns},
    // where the batch-weight chains run: 0 host cores (chain_host.h), 1 the device (chain_dev.h), 2 the sponges on host cores and the
    // reduction mod l on the device, -1 the engine's rule (chain_mode)
    {"chain", "BPP_CHAIN", &bpp_ctx::Options::chain},
    // tests: the device chain reports the weight of proof (value - 1) as zero, so that the redraw fall-back runs
    {"chain_test_zero", "BPP_CHAIN_TEST_ZERO", &bpp_ctx::Options::chain_test_zero},
    // how a calling thread waits for the device: 0 the runtime's wait (spins on a core until the stream is done), 1 naps between
    // looks at an event (gpu_wait), -1 the engine's rule: naps for calls of BPP_WAIT_NAP_MIN_PROOFS proofs and more
    {"wait", "BPP_WAIT", &bpp_ctx::Options::wait},
    // the device chains (chain = 1) behind PASS 1 on the call's own stream (1) instead of a stream of their own beside the decompression (0)
    {"chain_inline", "BPP_CHAIN_INLINE", &bpp_ctx::Options::chain_inline},
    // chain = 2: the reduction mod l of the host sponges' bytes in k_scalars_lanes' prologue (1, the rule) or as a launch of its own (0)
    {"wide_in_lanes", "BPP_WIDE_IN_LANES", &bpp_ctx::Options::wide_in_lanes},
};
void options_from_env(bpp_ctx *c) {
  for (const OptionName &o : kOptions)
    if (const char *e = getenv(o.env)) c->opt.*(o.field) = atoi(e);
}

int fail(bpp_ctx *ctx, int code, const std::string &m, char *errbuf = nullptr, size_t len = 0) {
  if (ctx) ctx->err = m;
  set_err(errbuf, len, m);
  return code;
}

enum Mark { M_START = 0, M_TRANSCRIPTS, M_DECOMPRESS, M_SCALARS, M_WEIGHTS_IN, M_LANES, M_REDUCE, M_DIGITS, M_SORT, M_ORDER, M_ACC,
            M_BUCKET, M_FINAL, M_MASKS0, M_MASKS, M_CHAIN, M_COUNT };

struct StageTimer {
  bpp_ctx *ctx;
  bool have[16] = {false};
  explicit StageTimer(bpp_ctx *c) : ctx(c) {
    if (ctx->profile && !ctx->ev_ready) {
      for (auto &e : ctx->ev) HIP_CHECK(hipEventCreate(&e));
      ctx->ev_ready = true;
    }
  }
  void mark(int idx) {
    if (ctx->profile && idx < 16) {
      // light form: the two events that bracket the roofline kernel and nothing else (every mark is a barrier packet with a
      // timestamp in the queue; thirteen per step read 0-4 % lower than two, inside the run-to-run spread: profiles/r06_stage_events_ab.txt)
      if (ctx->profile_light && idx != M_ORDER && idx != M_ACC) return;
      HIP_CHECK(hipEventRecord(ctx->ev[idx], ctx->stream));
      have[idx] = true;
    }
  }
  float between(int a, int b) {
    float ms = 0;
    if (ctx->profile && have[a] && have[b]) HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev[a], ctx->ev[b]));
    return ms;
  }
};

inline uint32_t cdiv(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

// ---- how a calling thread waits for the device.  hipStreamSynchronize / hipEventSynchronize spin: measured on the GPU boxes
// (tools/microbench/wait_modes.hip) a waiting thread burns 100 % of a core with either, whatever flags the event was created
// with, unless the PROCESS-WIDE device flag hipDeviceScheduleBlockingSync is set (then 25 % / 6 %) -- which is the host
// application's to set, not a library's.  A verification of tens of thousands of proofs takes milliseconds and several are in
// flight, so its caller naps instead: hipEventQuery every 50 us costs 1-2 % of a core and wakes up ~70 us late, which such a
// call does not notice (bench.py: 3.7 -> 0.3 host cores per rank for the callers of the headline).  Calls of a few
// hundred proofs (0.5-0.7 ms, one at a time) keep the runtime's spinning wait: there 70 us are 10 %.
#define BPP_WAIT_NAP_MIN_PROOFS 4096u
inline bool wait_naps(const bpp_ctx *ctx, size_t proofs) {
  return ctx->opt.wait >= 0 ? ctx->opt.wait != 0 : proofs >= BPP_WAIT_NAP_MIN_PROOFS;
}
// hint (optional): the caller's memory of how long THIS wait took the last times, microseconds (updated here).  A call of a
// steady stream of calls waits about as long as the one before it: most of that is slept in ONE piece, and the looking starts
// when the wait is nearly over -- every look is a call into the runtime and every nap two context switches, which on a busy host
// (measured: the same build 0.3 or 1.2 cores for four waiting callers, box by box) is what a rank's idle callers cost.
// tail_spin: after the piece slept ahead, look without napping (a call at a time then ends as promptly as under the runtime's
// spinning wait, for ~30 % of a core instead of all of it; with nothing remembered yet the whole wait is looked through)
inline void gpu_wait_event(hipEvent_t ev, bool nap, WaitHint *wh = nullptr, uint64_t key = 0, bool tail_spin = false) {
  if (wh && wh->key != key) {  // other work than last time: nothing is known about this wait
    wh->us = 0;
    wh->key = key;
  }
  uint32_t *hint = wh ? &wh->us : nullptr;
  if (!nap) {
    HIP_CHECK(hipEventSynchronize(ev));
    return;
  }
  // the first ~30 us by looking only: a wait for something that is (nearly) done must not cost a nap -- a nap is 60-100 us with
  // the kernel's timer slack, and a prover call has six waits in a row at its end (0.6 ms of a 6.5 ms call before this)
  const auto t0 = std::chrono::steady_clock::now();
  bool slept_ahead = false;
  for (;;) {
    const hipError_t e = hipEventQuery(ev);
    const long waited = (long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
    if (e == hipSuccess) {
      if (hint) *hint = *hint ? (uint32_t)((3ul * *hint + (unsigned long)waited) / 4ul) : (uint32_t)waited;
      return;
    }
    if (e != hipErrorNotReady) HIP_CHECK(e);
    if (waited < 30) continue;
    if (tail_spin && (slept_ahead || !hint || *hint <= 600)) continue;
    long nap_us;
    if (hint && !slept_ahead && *hint > 600 && waited < (long)*hint / 2) {
      // the bulk of an expected wait in one piece: HALF of what the last waits took when naps follow, 70 % when the rest is looked
      // through.  (70 % for the napping callers too overslept in short runs: between two synchronisations the steps in flight
      // start together and stay in step, their waits spread from 6 to 13 ms around the remembered 10, a caller that sleeps past
      // its step's end restarts its slot late and the steps in step with it all shift -- the driver's 20-step form of bench.py
      // read 25.1 M proofs/s with 70 %, 25.6 with 60 %, 25.8-26.2 with 50 %, 26.0 with 40 %, 25.9-26.0 under the runtime's
      // spinning wait; 256-step runs and the callers' CPU time do not change: profiles/r06_wait_ahead_ab.txt)
      nap_us = (long)*hint * (tail_spin ? 7 : 5) / 10 - waited;
      if (nap_us < 50) nap_us = 50;
      slept_ahead = true;
    } else if (slept_ahead && waited < (long)*hint * 5 / 4) {
      nap_us = 50;  // the expected end is near: short naps (a call at a time wakes up within one of them)
    } else {
      // naps grow with the wait: 50 us at first (a wait of a few hundred microseconds ends within a nap of its event), an eighth of
      // the time waited so far from 0.4 ms on, 400 us at most
      nap_us = waited < 400 ? 50 : (waited / 8 > 400 ? 400 : waited / 8);
    }
    struct timespec ts = {0, nap_us * 1000};
    nanosleep(&ts, nullptr);
  }
}
inline void gpu_wait_stream(bpp_ctx *ctx, hipStream_t s, bool nap, WaitHint *hint = nullptr, uint64_t key = 0, bool tail_spin = false);
inline bool gpu_wait_stream_ok(bpp_ctx *ctx, hipStream_t s, bool nap) {  // false instead of an exception (the sharded forms carry faults along)
  try {
    gpu_wait_stream(ctx, s, nap);
    return true;
  } catch (const EngineError &) {
    (void)hipGetLastError();
    return false;
  }
}
inline void gpu_wait_stream(bpp_ctx *ctx, hipStream_t s, bool nap, WaitHint *hint, uint64_t key, bool tail_spin) {
  if (!nap) {
    HIP_CHECK(hipStreamSynchronize(s));
    return;
  }
  // (NOT hipStreamQuery as a first look: measured -- tools/microbench/wait_modes.hip -- a stream query on a busy stream leaves a
  // helper thread of the runtime spinning until that work is done, 70-90 % of a core that no thread of the caller's shows; an
  // event that is already complete costs the ~30 us look of gpu_wait_event at most)
  if (!ctx->ev_wait_ready) {
    HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_wait, hipEventDisableTiming));
    ctx->ev_wait_ready = true;
  }
  HIP_CHECK(hipEventRecord(ctx->ev_wait, s));
  gpu_wait_event(ctx->ev_wait, true, hint, key, tail_spin);
}

static bool decompress_spill_enabled() {
  static const bool on = [] {
    const char *e = getenv("BPP_DECOMPRESS_SPILL");
    return e ? atoi(e) != 0 : true;
  }();
  return on;
}


// ------------------------------------------------------------------ MSM driver
// A call of up to this many MSM terms is "small": it has the chip to itself, its time is the length of its dependency chains
#define BPP_SMALL_CALL_TERMS 20000u
uint32_t choose_window(const bpp_ctx *ctx, uint32_t group_terms, uint32_t all_terms) {
  // buckets per window ~ terms / 12  (bucket lists of ~12 points keep the per-lane chains short)
  uint32_t c = 4;
  while (c < 14 && (1u << c) * 12u <= group_terms) c++;  // nb = 2^(c-1)
  // Above 11 bits the buckets of a window no longer fit the row / column reduction (2.1 additions per bucket, k_msm_window_rc)
  // and go through the bit-plane one (c / 2 per bucket): 13 bits for a 4096-proof batch (61 k terms) is 20 x 61 k + 20 x 4096
  // x 6.5 = 1.76 M additions, 11 bits 23 x 61 k + 23 x 1024 x 2.1 = 1.47 M.  The wider windows only pay from ~200 k terms on.
  const uint32_t c_max = ctx->opt.msm_c_max >= 0 ? (uint32_t)ctx->opt.msm_c_max : (group_terms < 200000u ? 11u : 14u);
  if (ctx->opt.msm_c_add > 0 && all_terms > BPP_SMALL_CALL_TERMS) c += (uint32_t)ctx->opt.msm_c_add;  // (A/B timing of wider windows)
  if (c > c_max && c_max >= 4) c = c_max;
  // A small call has the chip to itself: its time is the length of the dependency chains, not the number of additions.
  // Wider windows shorten the bucket lists (accumulation) and the Horner step (fewer windows to add) for a longer
  // row / column reduction: three more bits are worth 0.07-0.1 ms up to a few hundred proofs (one proof 0.79 -> 0.68 ms).
  const int bias = ctx->opt.msm_c_bias >= 0 ? ctx->opt.msm_c_bias : 3;  // (tests run the narrow windows as well)
  if (all_terms <= BPP_SMALL_CALL_TERMS) c = (uint32_t)std::max(4, std::min(11, (int)c + bias));
  return c;
}

// plan + work buffers for G groups with term offsets goff[0..G]; the term lists (term_sidx / term_pidx) are filled by
// the caller, from host vectors (msm_prepare) or by a kernel (layout_groups)
// half-scalar plan (msm.h: k_split_shift_quad): small verifier calls, unless the one-lane kernels are forced.  It halves the
// final Horner step (0.26 -> 0.13 ms) and doubles every bucket's list (quad accumulation: 0.034 -> 0.066 ms at 256 proofs,
// 0.07 -> 0.18 at 1024).  Measured on non-aggregated 64-bit proofs (profiles/r03_v4_bench_latency*.jsonl): 1 proof 0.63 -> 0.47
// ms, 64: 0.65 -> 0.52, 256: 0.67 -> 0.62, 512: 0.79 -> 0.73.  The two plans cross at about 900 proofs (896: 0.94 / 0.96 ms with /
// without, 1024: 1.00-1.03 / 0.98-1.00, 1280: 1.16-1.21 / 1.11-1.12; same box, alternating): up to 14 000 terms.
#define BPP_SPLIT_CALL_TERMS 14000u
bool msm_wants_split(const bpp_ctx *ctx, uint32_t n_terms) {
  if (ctx->opt.msm_split >= 0) return ctx->opt.msm_split != 0 && ctx->opt.msm_quad != 0;
  return n_terms <= BPP_SPLIT_CALL_TERMS && ctx->opt.msm_quad != 0;
}
// goff: term offsets of the groups as the kernels will see them (already doubled for a split plan)
// copy_goff: the group offsets go to the device by a copy of their own (msm_prepare); layout_groups hands them to
// k_layout_terms through mapped host memory instead
void msm_plan_alloc(bpp_ctx *ctx, MsmWork &w, const std::vector<uint32_t> &goff, bool split = false, bool copy_goff = true) {
  const uint32_t G = (uint32_t)goff.size() - 1, n = goff[G];
  uint32_t maxg = 0;
  for (uint32_t g = 0; g < G; g++) maxg = std::max(maxg, goff[g + 1] - goff[g]);
  const MsmPlan plan = msm_make_plan(choose_window(ctx, maxg, split ? n / 2 : n), G, n, split ? BPP_MSM_SPLIT_BITS : 253u);
  w.plan = plan;
  w.split = split;
  w.max_group_terms = maxg;
  const size_t nbk = (size_t)G * plan.K * plan.nb;
  w.counts.alloc(nbk);
  w.starts.alloc(nbk);
  w.sorted.alloc((size_t)n * plan.K);
  w.order.alloc(nbk);
  w.order_win.alloc(nbk);
  w.cls_hist.alloc((size_t)G * plan.K * 768);
  w.buckets.alloc(nbk);
  w.Q.alloc((size_t)G * plan.K * plan.c);
  w.W.alloc((size_t)G * plan.K);
  w.R.alloc(G);
  w.comp32.alloc((size_t)G * 32);
  w.is_identity.alloc(G);
  w.term_sidx.alloc(n);
  w.term_pidx.alloc(n);
  w.group_off.alloc(G + 1);
  HIP_CHECK(hipStreamSynchronize(ctx->stream));  // pin_small may still feed an earlier copy / kernel
  ctx->pin_small.resize(3 * (size_t)(G + 1));
  memcpy(ctx->pin_small.data(), goff.data(), (G + 1) * 4);
  if (copy_goff) HIP_CHECK(hipMemcpyAsync(w.group_off.p, ctx->pin_small.data(), (G + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
}

void msm_prepare(bpp_ctx *ctx, MsmWork &w, const std::vector<uint32_t> &sidx, const std::vector<uint32_t> &pidx,
                 const std::vector<uint32_t> &goff) {
  msm_plan_alloc(ctx, w, goff);
  const uint32_t n = (uint32_t)sidx.size(), G1 = (uint32_t)goff.size();
  HIP_CHECK(hipStreamSynchronize(ctx->stream));
  ctx->pin_small.resize(3 * (size_t)G1 + 2 * (size_t)n);  // (may move the buffer: goff's copy has completed)
  uint32_t *pin = ctx->pin_small.data() + 3 * (size_t)G1;
  memcpy(pin, sidx.data(), (size_t)n * 4);
  memcpy(pin + n, pidx.data(), (size_t)n * 4);
  HIP_CHECK(hipMemcpyAsync(w.term_sidx.p, pin, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
  HIP_CHECK(hipMemcpyAsync(w.term_pidx.p, pin + n, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
  HIP_CHECK(hipStreamSynchronize(ctx->stream));
}

void msm_run(bpp_ctx *ctx, MsmWork &w, const sc *scalars, PointTables tabs, StageTimer *tm) {
  const MsmPlan plan = w.plan;
  hipStream_t s = ctx->stream;
  // digits + counting sort + per-window size ordering in one launch (msm.h: k_msm_prelude), then the group-level order
  const uint32_t per_group = plan.K * plan.nb;
  // many small groups of a throughput call: four-wavefront workgroups (msm.h); a small call (latency) and large groups: sixteen
  // few buckets on an idle chip (one batch per call): quad forms, ~3x shorter dependency chains (tests force either form)
  const bool small = w.split || (ctx->opt.msm_quad >= 0 ? ctx->opt.msm_quad != 0 : (size_t)plan.G * per_group <= 100000);
  const bool narrow = !small && w.max_group_terms <= BPP_SORT_SMALL_GROUP_TERMS;
  if (narrow) {
    const uint32_t dig_cap = msm_prelude_dig_cap(plan, w.max_group_terms, BPP_SORT_THREADS_SMALL);
    hipLaunchKernelGGL(k_msm_prelude<BPP_SORT_THREADS_SMALL>, dim3(8 * cdiv(plan.G, 8) * plan.K), dim3(BPP_SORT_THREADS_SMALL), msm_prelude_lds(plan, dig_cap), s,
                       scalars, w.term_sidx.p, w.term_pidx.p, w.group_off.p, plan, dig_cap, w.counts.p, w.starts.p, w.sorted.p, w.order_win.p, w.cls_hist.p);
    hipLaunchKernelGGL(k_msm_order<BPP_SORT_THREADS_SMALL>, dim3(plan.G), dim3(BPP_SORT_THREADS_SMALL), 0, s, w.counts.p, w.order_win.p, w.cls_hist.p, plan,
                       w.order.p);
  } else {
    const uint32_t dig_cap = msm_prelude_dig_cap(plan, w.max_group_terms);
    hipLaunchKernelGGL(k_msm_prelude<BPP_SORT_THREADS>, dim3(8 * cdiv(plan.G, 8) * plan.K), dim3(BPP_SORT_THREADS), msm_prelude_lds(plan, dig_cap), s, scalars,
                       w.term_sidx.p, w.term_pidx.p, w.group_off.p, plan, dig_cap, w.counts.p, w.starts.p, w.sorted.p, w.order_win.p, w.cls_hist.p);
    hipLaunchKernelGGL(k_msm_order<BPP_SORT_THREADS>, dim3(plan.G), dim3(BPP_SORT_THREADS), 0, s, w.counts.p, w.order_win.p, w.cls_hist.p, plan, w.order.p);
  }
  if (tm) tm->mark(M_ORDER);  // msm_accumulate_ms brackets k_msm_accumulate alone (the roofline kernel)
  if (small)
    hipLaunchKernelGGL(k_msm_accumulate_quad, dim3(cdiv(plan.G * per_group, 16)), dim3(64), 0, s, w.sorted.p, w.starts.p,
                       w.counts.p, w.order.p, tabs, plan.G * per_group, w.buckets.p);
  else
    hipLaunchKernelGGL(k_msm_accumulate, dim3(8 * cdiv(plan.G, 8) * cdiv(per_group, 64)), dim3(64), 0, s, w.sorted.p, w.starts.p,
                       w.counts.p, w.order.p, tabs, per_group, plan.G, w.buckets.p);
  if (tm) tm->mark(M_ACC);
  if (plan.c <= 11 && small) {
    hipLaunchKernelGGL(k_msm_window_rc_quad, dim3(plan.G * plan.K), dim3(1024), 0, s, w.buckets.p, w.counts.p, plan, w.W.p);
  } else if (plan.nb <= 256 && ctx->opt.msm_rc2 != 0) {  // two windows per wavefront (msm.h)
    hipLaunchKernelGGL(k_msm_window_rc2, dim3(cdiv(plan.G * plan.K, 2)), dim3(64), 0, s, w.buckets.p, w.counts.p, plan, plan.G * plan.K, w.W.p);
  } else if (plan.c <= 11) {
    hipLaunchKernelGGL(k_msm_window_rc, dim3(plan.G * plan.K), dim3(64), 0, s, w.buckets.p, w.counts.p, plan, w.W.p);
  } else {
    hipLaunchKernelGGL(k_msm_bitsum, dim3(plan.c, plan.K, plan.G), dim3(64), 0, s, w.buckets.p, w.counts.p, plan, w.Q.p);
    hipLaunchKernelGGL(k_msm_window, dim3(cdiv(plan.G * plan.K, 64)), dim3(64), 0, s, w.Q.p, plan, w.W.p);
  }
  if (tm) tm->mark(M_BUCKET);
  {
    if (ctx->opt.msm_final_quad != 0)  // (tests force either kernel)
      hipLaunchKernelGGL(k_msm_final_quad, dim3(cdiv(plan.G, 16)), dim3(64), 0, s, w.W.p, plan, w.R.p, w.is_identity.p);
    else
      hipLaunchKernelGGL(k_msm_final, dim3(cdiv(plan.G, 64)), dim3(64), 0, s, w.W.p, plan, w.R.p, w.is_identity.p);
  }
  if (tm) tm->mark(M_FINAL);
  HIP_CHECK(hipGetLastError());
}

// decompress `n` host points into a device niels table; returns number of bad encodings
uint32_t decompress_to_device(bpp_ctx *ctx, const uint8_t *pts32, size_t n, niels *out) {
  DevBuf<uint8_t> d_in;
  DevBuf<uint32_t> d_bad;
  d_in.alloc(n * 32);
  d_bad.alloc(1);
  HIP_CHECK(hipMemcpyAsync(d_in.p, pts32, n * 32, hipMemcpyHostToDevice, ctx->stream));
  HIP_CHECK(hipMemsetAsync(d_bad.p, 0, 4, ctx->stream));
  hipLaunchKernelGGL(k_decompress_plain, dim3(cdiv((uint32_t)n, 64)), dim3(64), 0, ctx->stream, d_in.p, (uint32_t)n, out,
                     d_bad.p);
  uint32_t bad = 0;
  HIP_CHECK(hipMemcpyAsync(&bad, d_bad.p, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_CHECK(hipStreamSynchronize(ctx->stream));
  return bad;
}

// generic MSM over host scalars with a prepared device point table layout
int msm_host_entry(bpp_ctx *ctx, const niels *tab_a, uint32_t n_a, const uint8_t *static_scalars32, size_t n_static,
                   const uint8_t *dyn_scalars32, const uint8_t *dyn_points32, size_t n_dyn,
                   const uint32_t *group_off_in, size_t n_groups, uint8_t *out32) {
  // scalars: [static | dynamic], canonical check
  const size_t n = n_static + n_dyn;
  std::vector<uint8_t> sb(n * 32 + 32);
  if (n_static) memcpy(sb.data(), static_scalars32, n_static * 32);
  if (n_dyn) memcpy(sb.data() + n_static * 32, dyn_scalars32, n_dyn * 32);
  for (size_t i = 0; i < n; i++)
    if (!sc_is_canonical(sb.data() + 32 * i)) return fail(ctx, BPP_ERR_INVALID_ARGUMENT, "scalar is not canonical");
  if (n == 0 || n_groups == 0) {  // empty sum = identity
    for (size_t g = 0; g < (n_groups ? n_groups : 1); g++) memset(out32 + 32 * g, 0, 32);
    return BPP_OK;
  }
  DevBuf<sc> d_sc;
  DevBuf<niels> d_dyn;
  d_sc.alloc(n);
  d_dyn.alloc(n_dyn);
  HIP_CHECK(hipMemcpyAsync(d_sc.p, sb.data(), n * 32, hipMemcpyHostToDevice, ctx->stream));
  if (n_dyn) {
    uint32_t bad = decompress_to_device(ctx, dyn_points32, n_dyn, d_dyn.p);
    if (bad) return fail(ctx, BPP_ERR_INVALID_ARGUMENT, "point is not a canonical ristretto255 encoding");
  }
  std::vector<uint32_t> sidx(n), pidx(n), goff;
  for (size_t i = 0; i < n; i++) {
    sidx[i] = (uint32_t)i;
    pidx[i] = (i < n_static) ? (uint32_t)i : (uint32_t)(n_a + (i - n_static));
  }
  if (group_off_in) {
    goff.assign(group_off_in, group_off_in + n_groups + 1);
  } else {
    goff = {0u, (uint32_t)n};
  }
  MsmWork w;
  msm_prepare(ctx, w, sidx, pidx, goff);
  PointTables tabs{tab_a, d_dyn.p, n_a, nullptr, nullptr};
  msm_run(ctx, w, d_sc.p, tabs, nullptr);
  hipLaunchKernelGGL(k_compress_ge, dim3(cdiv(w.plan.G, 64)), dim3(64), 0, ctx->stream, w.R.p, w.plan.G, w.comp32.p);
  HIP_CHECK(hipMemcpyAsync(out32, w.comp32.p, 32 * (goff.size() - 1), hipMemcpyDeviceToHost, ctx->stream));
  HIP_CHECK(hipStreamSynchronize(ctx->stream));
  return BPP_OK;
}

}  // namespace

// =================================================================== C ABI
extern "C" {

int bpp_ctx_create_on_stream(bpp_ctx **out, int device_id, void *hip_stream) {
  if (!out) return BPP_ERR_BAD_HANDLE;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device_id < 0 || device_id >= count) return BPP_ERR_NO_DEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return BPP_ERR_NO_DEVICE;
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return BPP_ERR_NO_DEVICE;  // code object is gfx950 only
  if (hipSetDevice(device_id) != hipSuccess) return BPP_ERR_NO_DEVICE;
  bpp_ctx *c = new bpp_ctx();
  c->device = device_id;
  options_from_env(c);
  if (hip_stream) {
    c->stream = (hipStream_t)hip_stream;
  } else {
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
      delete c;
      return BPP_ERR_ENGINE;
    }
    c->own_stream = true;
  }
  {  // bookkeeping: every context brings a stream (and, for small inputs, a side stream) onto the runtime's hardware queues
    DeviceState &D = device_state(device_id);
    std::lock_guard<std::mutex> lk(D.mu);
    D.contexts++;
    D.contexts_peak = std::max(D.contexts_peak, D.contexts);
    const uint32_t q = runtime_hw_queues();
    if (D.contexts > q) {  // not an error: the call succeeds, the note sits where bpp_ctx_last_error finds it
      char note[256];
      snprintf(note, sizeof(note), "note: %u contexts on %u hardware queues (GPU_MAX_HW_QUEUES=%u, read by the HIP runtime at start-up): "
               "streams beyond that share queues and their kernels serialise; see INTEGRATION.md, runtime preconditions", D.contexts, q, q);
      c->err = note;
      D.warned = true;
    }
    if (!D.env_note.empty()) c->err = D.env_note;
  }
  *out = c;
  return BPP_OK;
}

int bpp_ctx_create(bpp_ctx **out, int device_id) { return bpp_ctx_create_on_stream(out, device_id, nullptr); }

void bpp_ctx_destroy(bpp_ctx *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  pipeline_shutdown(ctx);  // waits for the calls in flight on the lanes, joins their threads, destroys their contexts
  // a call of another thread that is still inside this context (it holds the context's lock while it waits for the device) ends
  // first: destroying a context under a running call is the caller's bug, but it must not become a use of freed events
  { std::lock_guard<std::mutex> lk(ctx->mu); }
  (void)hipStreamSynchronize(ctx->stream);
  for (auto &kv : ctx->batches) wipe_batch_secrets(*kv.second, ctx->stream);  // seed nonces / masks of batches never destroyed
  (void)hipStreamSynchronize(ctx->stream);
  ctx->batches.clear();
  ctx->spare_batch.reset();
  for (uint64_t h : ctx->held_precomps) (void)precomp_registry().release(h);
  for (uint64_t h : ctx->held_params) (void)params_registry().release(h);
  if (ctx->ev_ready)
    for (auto &e : ctx->ev) (void)hipEventDestroy(e);
  if (ctx->ev_rng_ready) (void)hipEventDestroy(ctx->ev_rng);
  if (ctx->ev_wait_ready) (void)hipEventDestroy(ctx->ev_wait);
  if (ctx->side_stream) {
    (void)hipStreamSynchronize(ctx->side_stream);
    (void)hipStreamDestroy(ctx->side_stream);
    (void)hipEventDestroy(ctx->ev_fork);
    (void)hipEventDestroy(ctx->ev_join);
  }
  if (ctx->chain_stream) {
    (void)hipStreamSynchronize(ctx->chain_stream);
    (void)hipStreamDestroy(ctx->chain_stream);
    (void)hipEventDestroy(ctx->ev_chain_fork);
    (void)hipEventDestroy(ctx->ev_chain_done);
  }
  ctx->scratch128.release();
  for (auto &ps : ctx->prove_streams) {
    (void)hipStreamSynchronize(ps);
    (void)hipStreamDestroy(ps);
  }
  for (auto &e : ctx->prove_events) (void)hipEventDestroy(e);
  for (auto &ps : ctx->prove_lane_streams) {
    (void)hipStreamSynchronize(ps);
    (void)hipStreamDestroy(ps);
  }
  for (auto &e : ctx->prove_sync_events) (void)hipEventDestroy(e);
  if (ctx->prove_msm_stream) {
    (void)hipStreamSynchronize(ctx->prove_msm_stream);
    (void)hipStreamDestroy(ctx->prove_msm_stream);
  }
  for (auto &ps : ctx->prove_aux_streams) {
    (void)hipStreamSynchronize(ps);
    (void)hipStreamDestroy(ps);
  }
  for (auto &e : ctx->prove_aux_events) (void)hipEventDestroy(e);
  ctx->prove_arena.release();
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
  {
    DeviceState &D = device_state(ctx->device);
    std::lock_guard<std::mutex> lk(D.mu);
    if (D.contexts) D.contexts--;
  }
  delete ctx;
}

int bpp_runtime_info_get(bpp_ctx *ctx, bpp_runtime_info *out) {
  if (!ctx || !out) return BPP_ERR_BAD_HANDLE;
  memset(out, 0, sizeof(*out));
  DeviceState &D = device_state(ctx->device);
  std::lock_guard<std::mutex> lk(D.mu);
  out->device = ctx->device;
  out->contexts = D.contexts;
  out->contexts_peak = D.contexts_peak;
  out->hw_queues = runtime_hw_queues();
  out->host_threads = host_pool_size();
  out->small_call_limit = D.limit;
  out->small_calls_in_flight = D.in_flight;
  out->small_calls = D.small_calls;
  out->small_calls_queued = D.small_calls_queued;
  out->oversubscribed = D.contexts > out->hw_queues ? 1u : 0u;
  return BPP_OK;
}

int bpp_device_chain_stats(bpp_ctx *ctx, uint64_t *calls, uint64_t *redraws) {
  if (!ctx) return BPP_ERR_BAD_HANDLE;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (calls) *calls = ctx->device_chain_calls + ctx->wide_chain_calls;
  if (redraws) *redraws = ctx->device_chain_redraws;
  return BPP_OK;
}

int bpp_small_call_limit(bpp_ctx *ctx, int limit) {
  if (!ctx) return BPP_ERR_BAD_HANDLE;
  DeviceState &D = device_state(ctx->device);
  std::lock_guard<std::mutex> lk(D.mu);
  const int before = (int)D.limit;
  if (limit >= 0) {
    D.limit = (uint32_t)limit;
    while (!D.queue.empty() && (D.limit == 0 || D.in_flight < D.limit)) {  // a wider (or removed) gate lets waiting calls in
      DeviceState::Waiter *w = D.queue.front();
      D.queue.pop_front();
      D.in_flight++;
      w->go = true;
      w->cv.notify_one();
    }
  }
  return before;
}

const char *bpp_ctx_last_error(bpp_ctx *ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

int bpp_ctx_set_option(bpp_ctx *ctx, const char *name, int value) {
  if (!ctx || !name) return BPP_ERR_BAD_HANDLE;
  std::lock_guard<std::mutex> lk(ctx->mu);
  for (const OptionName &o : kOptions)
    if (strcmp(name, o.name) == 0) {
      ctx->opt.*(o.field) = value;
      return BPP_OK;
    }
  return fail(ctx, BPP_ERR_INVALID_ARGUMENT, std::string("unknown option: ") + name);
}

#define BPP_ENTRY(ctx)                         \
  if (!(ctx)) return BPP_ERR_BAD_HANDLE;       \
  std::lock_guard<std::mutex> _lk((ctx)->mu);  \
  if (hipSetDevice((ctx)->device) != hipSuccess) return BPP_ERR_NO_DEVICE;
#define BPP_CATCH(ctx, errbuf, len)                                \
  catch (const EngineError &e) { return fail(ctx, e.code, e.msg, errbuf, len); } \
  catch (const ProofErr &e) { return fail(ctx, e.code, e.msg, errbuf, len); }    \
  catch (const std::exception &e) { return fail(ctx, BPP_ERR_ENGINE, e.what(), errbuf, len); }

// ---------------------------------------------------------------- B1
int bpp_precomp_create(bpp_ctx *ctx, const uint8_t *points32, size_t count, uint64_t *handle) {
  BPP_ENTRY(ctx);
  try {
    if (!handle || (!points32 && count)) return fail(ctx, BPP_ERR_INVALID_ARGUMENT, "null argument");
    auto pc = std::make_shared<Precomp>();
    pc->device = ctx->device;
    pc->count = (uint32_t)count;
    pc->table.alloc(count);
    if (count) {
      uint32_t bad = decompress_to_device(ctx, points32, count, pc->table.p);
      if (bad) return fail(ctx, BPP_ERR_INVALID_ARGUMENT, "point is not a canonical ristretto255 encoding");
    }
    const uint64_t h = precomp_registry().add(std::move(pc));
    ctx->held_precomps.insert(h);
    *handle = h;
    return BPP_OK;
  }
  BPP_CATCH(ctx, nullptr, 0)
}

int bpp_precomp_destroy(bpp_ctx *ctx, uint64_t handle) {
  BPP_ENTRY(ctx);
  auto it = ctx->held_precomps.find(handle);
  if (it == ctx->held_precomps.end()) return BPP_ERR_BAD_HANDLE;  // this context holds no reference on it
  ctx->held_precomps.erase(it);
  return precomp_registry().release(handle) ? BPP_OK : BPP_ERR_BAD_HANDLE;
}

int bpp_msm_mixed(bpp_ctx *ctx, uint64_t handle, const uint8_t *static_scalars32, size_t n_static,
                  const uint8_t *dyn_scalars32, const uint8_t *dyn_points32, size_t n_dyn, uint8_t out_point32[32]) {
  BPP_ENTRY(ctx);
  try {
    const std::shared_ptr<Precomp> pcp = precomp_registry().get(handle);
    if (!pcp || pcp->device != ctx->device) return fail(ctx, BPP_ERR_BAD_HANDLE, "unknown precomputation handle");
    Precomp &pc = *pcp;
    if (n_static > pc.count) return fail(ctx, BPP_ERR_INVALID_LENGTH, "more static scalars than precomputed points");
    return msm_host_entry(ctx, pc.table.p, pc.count, static_scalars32, n_static, dyn_scalars32, dyn_points32, n_dyn,
                          nullptr, 1, out_point32);
  }
  BPP_CATCH(ctx, nullptr, 0)
}

int bpp_msm_vartime(bpp_ctx *ctx, const uint8_t *scalars32, const uint8_t *points32, size_t n, uint8_t out_point32[32]) {
  BPP_ENTRY(ctx);
  try {
    return msm_host_entry(ctx, nullptr, 0, nullptr, 0, scalars32, points32, n, nullptr, 1, out_point32);
  }
  BPP_CATCH(ctx, nullptr, 0)
}

int bpp_msm_vartime_batched(bpp_ctx *ctx, const uint8_t *scalars32, const uint8_t *points32, const uint32_t *group_off,
                            size_t n_groups, uint8_t *out_points32) {
  BPP_ENTRY(ctx);
  try {
    if (!group_off || n_groups == 0) return fail(ctx, BPP_ERR_INVALID_ARGUMENT, "no groups");
    for (size_t g = 0; g < n_groups; g++)
      if (group_off[g + 1] < group_off[g]) return fail(ctx, BPP_ERR_INVALID_ARGUMENT, "group offsets not monotone");
    if (group_off[0] != 0) return fail(ctx, BPP_ERR_INVALID_ARGUMENT, "group offsets must start at 0");
    return msm_host_entry(ctx, nullptr, 0, nullptr, 0, scalars32, points32, group_off[n_groups], group_off, n_groups,
                          out_points32);
  }
  BPP_CATCH(ctx, nullptr, 0)
}

// ---------------------------------------------------------------- B2: parameters
int bpp_params_create(bpp_ctx *ctx, uint32_t bit_length, uint32_t max_aggregation, uint32_t extension_degree,
                      const uint8_t *h_base32, const uint8_t *g_bases32, uint64_t *params) {
  BPP_ENTRY(ctx);
  try {
    // RangeParameters::init checks (src/range_parameters.rs:37-51), ExtensionDegree::try_from (pedersen_gens.rs:68-82)
    if (!params) return fail(ctx, BPP_ERR_INVALID_ARGUMENT, "null argument");
    if (max_aggregation == 0 || (max_aggregation & (max_aggregation - 1)))
      return fail(ctx, BPP_ERR_INVALID_ARGUMENT, "Aggregation factor size must be a power of two");
    if (bit_length == 0 || (bit_length & (bit_length - 1)))
      return fail(ctx, BPP_ERR_INVALID_ARGUMENT, "Bit length must be a power of two");
    if (bit_length > 64) return fail(ctx, BPP_ERR_INVALID_ARGUMENT, "Bit length must be <= 64");
    if (extension_degree < 1 || extension_degree > 6)
      return fail(ctx, BPP_ERR_INVALID_ARGUMENT, "Extension degree not valid");
    if ((uint64_t)bit_length * max_aggregation > 2048)
      return fail(ctx, BPP_ERR_SIZE_OVERFLOW, "bit_length * max_aggregation > 2048 is not supported by this engine");
    auto P = std::make_shared<Params>();
    P->device = ctx->device;
    P->n_bits = bit_length;
    P->m_max = max_aggregation;
    P->t = extension_degree;
    const uint32_t n = bit_length, m = max_aggregation, t = extension_degree;
    const uint32_t n_gen = 2 * n * m;
    P->table_len = n_gen + t + 1;
    P->table.alloc(P->table_len);

    // uniform bytes for every derived point: interleaved G,H (party-major), then default masking points
    const uint32_t n_uni = n_gen + 6;
    std::vector<uint8_t> uni((size_t)n_uni * 64);
    for (uint32_t party = 0; party < m; party++) {
      for (int which = 0; which < 2; which++) {
        uint8_t seed[15 + 5];
        memcpy(seed, "GeneratorsChain", 15);
        seed[15] = which ? 'H' : 'G';
        seed[16] = (uint8_t)party;
        seed[17] = (uint8_t)(party >> 8);
        seed[18] = (uint8_t)(party >> 16);
        seed[19] = (uint8_t)(party >> 24);
        std::vector<uint8_t> stream((size_t)n * 64);
        shake256(seed, sizeof(seed), stream.data(), stream.size());
        for (uint32_t i = 0; i < n; i++) memcpy(&uni[(size_t)(2 * (party * n + i) + which) * 64], &stream[(size_t)i * 64], 64);
      }
    }
    for (int k = 1; k <= 6; k++) {
      char label[64];
      int ll = snprintf(label, sizeof(label), "RISTRETTO_MASKING_BASEPOINT_%d", k);
      sha3_512((const uint8_t *)label, (size_t)ll, &uni[(size_t)(n_gen + k - 1) * 64]);
    }
    DevBuf<uint8_t> d_uni, d_comp;
    DevBuf<niels> d_pts;
    d_uni.alloc(uni.size());
    d_comp.alloc((size_t)n_uni * 32);
    d_pts.alloc(n_uni);
    HIP_CHECK(hipMemcpyAsync(d_uni.p, uni.data(), uni.size(), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_from_uniform, dim3(cdiv(n_uni, 64)), dim3(64), 0, ctx->stream, d_uni.p, n_uni, d_pts.p, d_comp.p);
    std::vector<uint8_t> comp((size_t)n_uni * 32);
    HIP_CHECK(hipMemcpyAsync(comp.data(), d_comp.p, comp.size(), hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipMemcpyAsync(P->table.p, d_pts.p, (size_t)n_gen * sizeof(niels), hipMemcpyDeviceToDevice, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    P->gi32.resize((size_t)n * m * 32);
    P->hi32.resize((size_t)n * m * 32);
    for (uint32_t i = 0; i < n * m; i++) {
      memcpy(&P->gi32[(size_t)i * 32], &comp[(size_t)(2 * i) * 32], 32);
      memcpy(&P->hi32[(size_t)i * 32], &comp[(size_t)(2 * i + 1) * 32], 32);
    }
    // Pedersen bases: H then G_k (src/ristretto.rs:67-76)
    P->hg32.resize((size_t)(1 + t) * 32);
    static const uint8_t BASEPOINT32[32] = {0xe2, 0xf2, 0xae, 0x0a, 0x6a, 0xbc, 0x4e, 0x71, 0xa8, 0x84, 0xa9,
                                            0x61, 0xc5, 0x00, 0x51, 0x5f, 0x58, 0xe3, 0x0b, 0x6a, 0xa5, 0x82,
                                            0xdd, 0x8d, 0xb6, 0xa6, 0x59, 0x45, 0xe0, 0x8d, 0x2d, 0x76};
    memcpy(&P->hg32[0], h_base32 ? h_base32 : BASEPOINT32, 32);
    for (uint32_t k = 0; k < t; k++)
      memcpy(&P->hg32[(size_t)(1 + k) * 32], g_bases32 ? g_bases32 + 32 * k : &comp[(size_t)(n_gen + k) * 32], 32);
    // table tail = [G_0..G_{t-1}, H]
    std::vector<uint8_t> tail((size_t)(t + 1) * 32);
    memcpy(&tail[0], &P->hg32[32], (size_t)t * 32);
    memcpy(&tail[(size_t)t * 32], &P->hg32[0], 32);
    uint32_t bad = decompress_to_device(ctx, tail.data(), t + 1, P->table.p + n_gen);
    if (bad) return fail(ctx, BPP_ERR_INVALID_ARGUMENT, "Pedersen base point is not a canonical ristretto255 encoding");
    // validate_and_append_point(H / G) (src/transcripts.rs:72-75): identity encodings are rejected once, here
    for (uint32_t k = 0; k < 1 + t; k++) {
      bool z = true;
      for (int i = 0; i < 32; i++) z = z && P->hg32[(size_t)k * 32 + i] == 0;
      if (z) return fail(ctx, BPP_ERR_VERIFICATION_FAILED, "Identity element cannot be added to the transcript");
    }
    P->table_hi.alloc(P->table_len);
    hipLaunchKernelGGL(k_split_shift_table, dim3(cdiv(P->table_len, 64)), dim3(64), 0, ctx->stream, P->table.p, P->table_len, P->table_hi.p);
    P->fb_ped_geo = fb_geometry(t + 1);
    P->fb_ped.alloc((size_t)(t + 1) * fb_stride(P->fb_ped_geo));
    hipLaunchKernelGGL(k_fb_build,
                       dim3(cdiv((t + 1) * P->fb_ped_geo.windows * cdiv(P->fb_ped_geo.entries, FB_BUILD_BLOCK), 64)), dim3(64), 0,
                       ctx->stream, P->table.p + n_gen, t + 1, P->fb_ped_geo, P->fb_ped.p);
    {  // j * 16^w * Base, j = 1..8, w = 0..63: the lines k_ct_fixed reads (all eight of a position, every time)
      FbGeom g4;
      g4.wbits = 4;
      g4.windows = BPP_CT_DIGITS;
      g4.entries = BPP_CTF_ENTRIES;
      g4.items = BPP_CT_DIGITS;
      P->fb_ct.alloc((size_t)(t + 1) * fb_stride(g4));
      hipLaunchKernelGGL(k_fb_build, dim3(cdiv((t + 1) * g4.windows, 64)), dim3(64), 0, ctx->stream, P->table.p + n_gen, t + 1, g4, P->fb_ct.p);
    }
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    P->d_hg32.alloc(P->hg32.size());
    HIP_CHECK(hipMemcpy(P->d_hg32.p, P->hg32.data(), P->hg32.size(), hipMemcpyHostToDevice));
    const uint64_t h = params_registry().add(std::move(P));
    ctx->held_params.insert(h);
    *params = h;
    return BPP_OK;
  }
  BPP_CATCH(ctx, nullptr, 0)
}

int bpp_params_destroy(bpp_ctx *ctx, uint64_t params) {
  BPP_ENTRY(ctx);
  auto it = ctx->held_params.find(params);
  if (it == ctx->held_params.end()) return BPP_ERR_BAD_HANDLE;  // this context holds no reference on it
  ctx->held_params.erase(it);
  // resident batches and calls in flight keep their own shared_ptr: the tables are freed when the last user is gone
  return params_registry().release(params) ? BPP_OK : BPP_ERR_BAD_HANDLE;
}

int bpp_params_retain(bpp_ctx *ctx, uint64_t params) {
  BPP_ENTRY(ctx);
  const std::shared_ptr<Params> P = params_registry().get(params);
  if (!P || P->device != ctx->device) return fail(ctx, BPP_ERR_BAD_HANDLE, "unknown params handle (or another device's)");
  if (!params_registry().retain(params)) return BPP_ERR_BAD_HANDLE;
  ctx->held_params.insert(params);
  return BPP_OK;
}

int bpp_precomp_retain(bpp_ctx *ctx, uint64_t handle) {
  BPP_ENTRY(ctx);
  const std::shared_ptr<Precomp> pc = precomp_registry().get(handle);
  if (!pc || pc->device != ctx->device) return fail(ctx, BPP_ERR_BAD_HANDLE, "unknown precomputation handle (or another device's)");
  if (!precomp_registry().retain(handle)) return BPP_ERR_BAD_HANDLE;
  ctx->held_precomps.insert(handle);
  return BPP_OK;
}

int bpp_params_export(bpp_ctx *ctx, uint64_t params, uint8_t *gi_out32, uint8_t *hi_out32, uint8_t *h_out32,
                      uint8_t *g_out32) {
  BPP_ENTRY(ctx);
  const std::shared_ptr<Params> Pp = params_registry().get(params);
  if (!Pp || Pp->device != ctx->device) return fail(ctx, BPP_ERR_BAD_HANDLE, "unknown params handle");
  Params &P = *Pp;
  if (gi_out32) memcpy(gi_out32, P.gi32.data(), P.gi32.size());
  if (hi_out32) memcpy(hi_out32, P.hi32.data(), P.hi32.size());
  if (h_out32) memcpy(h_out32, P.hg32.data(), 32);
  if (g_out32) memcpy(g_out32, P.hg32.data() + 32, (size_t)P.t * 32);
  return BPP_OK;
}

int bpp_pedersen_commit(bpp_ctx *ctx, uint64_t params, const uint64_t *values, const uint8_t *blindings32,
                        uint32_t n_blind, size_t count, uint8_t *commitments32) {
  BPP_ENTRY(ctx);
  try {
    const std::shared_ptr<Params> Pp = params_registry().get(params);
    if (!Pp || Pp->device != ctx->device) return fail(ctx, BPP_ERR_BAD_HANDLE, "unknown params handle");
    Params &P = *Pp;
    if (n_blind == 0 || n_blind > P.t) return fail(ctx, BPP_ERR_INVALID_LENGTH, "blinding vector");
    if (count == 0) return BPP_OK;
    // output j = value*H + sum_k r_k G_k over the resident fixed-base table of the Pedersen bases
    const uint32_t per = 1 + n_blind;
    std::vector<uint8_t> sb(count * per * 32, 0);  // values and blinding factors: wiped on every exit path
    DevBuf<sc> d_sc;
    ScopeExit wipe_secrets{[&] {
      wipe(sb.data(), sb.size());
      if (d_sc.p) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipMemset(d_sc.p, 0, d_sc.n * sizeof(sc));
      }
    }};
    std::vector<uint32_t> gidx(count * per), cnt(count, per);
    for (size_t j = 0; j < count; j++) {
      uint8_t *v = &sb[(j * per) * 32];
      for (int k = 0; k < 8; k++) v[k] = (uint8_t)(values[j] >> (8 * k));
      memcpy(v + 32, blindings32 + j * n_blind * 32, (size_t)n_blind * 32);
      for (uint32_t k = 0; k < n_blind; k++)
        if (!sc_is_canonical(v + 32 + 32 * k)) return fail(ctx, BPP_ERR_INVALID_ARGUMENT, "scalar is not canonical");
      gidx[j * per] = P.t;  // H is the last entry of fb_ped
      for (uint32_t k = 0; k < n_blind; k++) gidx[j * per + 1 + k] = k;
    }
    // value and blinding factors are secrets and nothing else: the reference commits with dalek's CONSTANT-TIME multiscalar_mul
    // (src/generators/pedersen_gens.rs:112-122).  Default here: the uniform-access form (ct.h: every table entry read, masked
    // select, no scalar-dependent address or branch).  Option "ct" = 0 takes the fixed-base tables instead, whose addresses are
    // the scalars' digits (faster by far; for callers whose values are public, e.g. the bench's input generation).
    const bool ct = ctx->opt.ct != 0;
    const uint32_t n_gen = 2 * P.n_bits * P.m_max;
    if (ct)  // indices into the parameter set's generator table: G_k at n_gen + k, H at n_gen + t
      for (size_t j = 0; j < count; j++) {
        gidx[j * per] = n_gen + P.t;
        for (uint32_t k = 0; k < n_blind; k++) gidx[j * per + 1 + k] = n_gen + k;
      }
    DevBuf<uint32_t> d_g, d_c;
    DevBuf<uint8_t> d_out;
    d_sc.alloc(count * per);
    d_g.alloc(count * per);
    d_c.alloc(count);
    d_out.alloc(count * 32);
    hipStream_t s = ctx->stream;
    HIP_CHECK(hipMemcpyAsync(d_sc.p, sb.data(), sb.size(), hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync(d_g.p, gidx.data(), gidx.size() * 4, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync(d_c.p, cnt.data(), cnt.size() * 4, hipMemcpyHostToDevice, s));
    DevBuf<ge> d_ge;
    d_ge.alloc(count);
    if (ct)
      hipLaunchKernelGGL(k_ct_fixed, dim3((uint32_t)count), dim3(64), 0, s, d_sc.p, d_g.p, d_c.p, per, n_gen, (const niels *)P.fb_ct.p, d_ge.p);
    else
      hipLaunchKernelGGL(k_fb_msm, dim3((uint32_t)count), dim3(fb_threads(ctx, per, P.fb_ped_geo)), 0, s, d_sc.p, d_g.p, d_c.p, per, P.fb_ped.p,
                         P.fb_ped_geo, d_ge.p, 0u);
    hipLaunchKernelGGL(k_compress_ge, dim3(cdiv((uint32_t)count, 64)), dim3(64), 0, s, d_ge.p, (uint32_t)count, d_out.p);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(commitments32, d_out.p, count * 32, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    return BPP_OK;
  }
  BPP_CATCH(ctx, nullptr, 0)
}

int bpp_transcript_new(const uint8_t *label, size_t label_len, uint8_t state203[203]) {
  if (!state203 || (!label && label_len)) return BPP_ERR_INVALID_ARGUMENT;
  Strobe s;
  merlin_new(s, label, (uint32_t)label_len);
  strobe_to_bytes(state203, s);
  return BPP_OK;
}

int bpp_weights_from_chains(const uint8_t *rng32_all, size_t n_groups, size_t n_per_group, uint8_t *weights32_out) {
  if ((!rng32_all || !weights32_out) && n_groups * n_per_group) return BPP_ERR_INVALID_ARGUMENT;
  if (n_groups == 0 || n_per_group == 0) return BPP_OK;
  if (n_groups * n_per_group > (1u << 30)) return BPP_ERR_SIZE_OVERFLOW;
  std::vector<uint32_t> first(n_groups + 1);
  for (size_t g = 0; g <= n_groups; g++) first[g] = (uint32_t)(g * n_per_group);
  run_weight_chains_generic(rng32_all, weights32_out, first.data(), (uint32_t)n_groups);
  return BPP_OK;
}

int bpp_weights_from_chain(const uint8_t *rng32_all, size_t n_total, uint8_t *weights32_out) {
  if ((!rng32_all || !weights32_out) && n_total) return BPP_ERR_INVALID_ARGUMENT;
  weights_from_chain_host(rng32_all, n_total, weights32_out);
  return BPP_OK;
}

// ---------------------------------------------------------------- B2: batches
}  // extern "C"

namespace {

// Secrets a resident batch holds on the device: the statements' seed nonces and the recovered masks (the reference wipes
// both on drop: src/range_statement.rs:76-81 `Zeroize for RangeStatement`, src/extended_mask.rs:14 `ZeroizeOnDrop`).
// Enqueued on `s`; the caller synchronises before the buffers change hands.
void wipe_batch_secrets(Batch &b, hipStream_t s) {
  if (b.seeds.p && b.seeds_dirty) (void)hipMemsetAsync(b.seeds.p, 0, b.seeds.n, s);
  if (b.masks.p && b.masks_dirty) (void)hipMemsetAsync(b.masks.p, 0, b.masks.n, s);
  b.seeds_dirty = b.masks_dirty = false;
}

// ---- host half of an upload: everything that reads the caller's buffers.  Validates and packs into ctx->pin_upload
// (proof and commitment bytes) and `pl` (descriptors, promises, seed nonces, transcript states).  Pure host work: the
// pipelined entry runs it on the submitting thread while the context's lanes are busy with earlier calls.
void upload_host_pack(bpp_ctx *ctx, const Params &P, const bpp_verify_item *items, size_t n_items, const bpp_packed_batch *packed,
                      UploadPlan &pl) {
  const ParallelFor pf = [](uint32_t n, const std::function<void(uint32_t)> &fn) { host_parallel_for(n, fn); };
  const ParamShape shape{P.n_bits, P.m_max, P.t};
  std::vector<bpp_verify_item> synth;  // packed input that is not uniform after all (hostile / mixed): the item form
  const bool arithmetic = packed && upload_pass_a_packed(*packed, pl, pf);
  if (packed && !arithmetic) {
    upload_packed_as_items(*packed, synth);
    items = synth.data();
    n_items = synth.size();
  }
  if (!arithmetic) upload_pass_a(items, n_items, pl);
  // Proof and commitment bytes are assembled directly in page-locked staging: a pageable source makes hipMemcpyAsync
  // return early and the 40 MB transfer trickle on at ~2.4 GB/s behind the call (it showed up as 23 ms in the first
  // verification of every freshly uploaded batch)
  ctx->pin_upload.resize(pl.bytes_total + BPP_BYTES_SLACK);
  uint8_t *bytes = ctx->pin_upload.data();
  memset(bytes + pl.bytes_total, 0, BPP_BYTES_SLACK);
  if (arithmetic) upload_pass_b_packed(*packed, shape, pl, bytes, pf);
  else upload_pass_b(items, shape, pl, bytes, pf);
}

// ---- device half: staging -> HBM, slot lists, statement commitments decoded; consumes `pl`
uint64_t upload_device(bpp_ctx *ctx, const std::shared_ptr<Params> &Pp, uint64_t params, UploadPlan &pl,
                       const uint8_t *const *challenges32, const uint8_t *rng_out32) {
    Params &P = *Pp;
    const size_t n_items = pl.n_items;
    auto B = std::make_unique<Batch>();
    if (ctx->spare_batch) {
      adopt_buffers(*B, *ctx->spare_batch);
      ctx->spare_batch.reset();
    }
    hipStream_t s = ctx->stream;
    // seed nonces pass through page-locked staging and reach the device: both are wiped on EVERY way out of here
    SmallStaging L;
    bool registered = false;
    ScopeExit wipe_secrets{[&] {
      if (L.n_seed) {
        (void)hipStreamSynchronize(s);  // the copy out of the staging may still be running
        upload_wipe_small(ctx->pin_upload2.data(), L);
      }
      secure_wipe(pl.seeds.data(), pl.seeds.size());
      if (!registered && B) {  // an error exit: the half-built batch dies here and must not leave nonces behind
        wipe_batch_secrets(*B, s);
        (void)hipStreamSynchronize(s);
      }
    }};
    B->params = Pp;
    B->params_handle = params;
    B->B = (uint32_t)n_items;
    const uint8_t *bytes = ctx->pin_upload.data();
    const size_t bytes_len = pl.bytes_total + BPP_BYTES_SLACK;
    B->desc.swap(pl.desc);
    B->rounds_bad.swap(pl.rounds_bad);
    B->defer.swap(pl.defer);
    B->any_seed = pl.any_seed;
    B->any_rounds_bad = pl.any_rounds_bad;
    B->any_defer = pl.any_defer;
    B->uniform_rounds = pl.uniform_rounds;
    B->rmax = pl.rmax;
    B->max_mn = pl.max_mn;
    const std::vector<uint8_t> &seeds = pl.seeds, &states = pl.states;
    const std::vector<uint64_t> &minvals = pl.minvals;
    const size_t sum_m = pl.sum_m;
    const uint32_t dyn = pl.total_dyn;
    B->total_dyn = dyn;
    B->sum_m = (uint32_t)sum_m;
    // challenge slots per proof: y, z, e_0.., e_final.  A proof claiming more rounds than any statement can have (mn <= 2048)
    // is refused on the host; its transcript is still replayed for the error precedence, its challenges are not kept, so
    // one such proof cannot inflate the whole batch's buffers.
    B->cs = std::min(B->rmax, (uint32_t)BPP_MAX_ROUNDS - 1) + 3;
    B->cols = 2 * B->max_mn + P.t + 1;
    // device copies (all sources page-locked: the copies are real stream-ordered DMA)
    B->bytes.alloc(bytes_len);
    B->states.alloc(states.size());
    B->seeds.alloc(B->any_seed ? seeds.size() : 0);
    B->d_desc.alloc(n_items);
    B->minvals.alloc(minvals.size());
    B->src_off.alloc(dyn);
    B->owner.alloc(dyn);
    B->idx_commit.alloc(sum_m);
    B->idx_proof.alloc(dyn - sum_m);
    {
      const SmallStaging Lp = upload_small_layout(pl, n_items);
      ctx->pin_upload2.resize(Lp.total + 64);
      uint8_t *st = ctx->pin_upload2.data();
      L = Lp;  // from here on the staging holds nonces (if any)
      upload_fill_small(pl, B->desc.data(), n_items, st, L);
      // ONE launch reads the pieces out of the mapped staging (k_ingest) and clears the two status arrays; only the proof
      // bytes of a large batch (tens of MB) go by DMA.  (Six copies / memsets were six blit kernels in front of every call.)
      B->status0.alloc(n_items);
      B->status.alloc(n_items);
      uint8_t *st_dev = ctx->pin_upload2.dev();
      IngestArgs ia;
      memset(&ia, 0, sizeof(ia));
      auto seg = [&](const uint8_t *src, void *dst, size_t nbytes) {
        if (nbytes) ia.seg[ia.n_seg++] = IngestSeg{src, (uint8_t *)dst, nbytes};
      };
      const bool bytes_by_dma = bytes_len > BPP_INGEST_MAX_BYTES;
      if (bytes_by_dma) HIP_CHECK(hipMemcpyAsync(B->bytes.p, bytes, bytes_len, hipMemcpyHostToDevice, s));
      else seg(ctx->pin_upload.dev(), B->bytes.p, bytes_len);
      seg(st_dev + L.o_desc, B->d_desc.p, n_items * sizeof(ProofDesc));
      seg(st_dev + L.o_min, B->minvals.p, minvals.size() * 8);
      seg(st_dev + L.o_state, B->states.p, states.size());
      if (L.n_seed) {
        B->seeds_dirty = true;
        seg(st_dev + L.o_seed, B->seeds.p, L.n_seed);
      }
      ia.zero[0] = B->status0.p;
      ia.zero[1] = B->status.p;
      ia.zero_words = (uint32_t)n_items;
      size_t most = n_items * 4;
      for (uint32_t q = 0; q < ia.n_seg; q++) most = std::max<size_t>(most, ia.seg[q].bytes);
      hipLaunchKernelGGL(k_ingest, dim3((uint32_t)std::min<size_t>(2048, std::max<size_t>(1, most / (16 * 256)))), dim3(256), 0, s, ia);
      HIP_CHECK(hipGetLastError());
    }
    // point sources in dynamic-slot order C_j.., A1, B, A, L.., R.. and the two slot lists, written by one lane per proof
    hipLaunchKernelGGL(k_build_slots, dim3(cdiv((uint32_t)n_items, 64)), dim3(64), 0, s, B->d_desc.p, (uint32_t)n_items, P.t,
                       B->src_off.p, B->owner.p, B->idx_commit.p, B->idx_proof.p);
    HIP_CHECK(hipGetLastError());
    // work buffers
    B->chal.alloc((size_t)n_items * B->cs);
    B->rng_out.alloc(n_items * 32);
    B->weights.alloc(n_items * 32);
    B->dynpts.alloc(dyn);
    if (decompress_spill_enabled()) B->dec_spill.alloc((size_t)30 * (dyn - B->sum_m));
    // (rows / parts: sized by layout_groups, which knows whether the generator columns are summed inside k_scalars_lanes)
    B->shr.alloc((size_t)n_items * SH_STRIDE);
    B->tab.alloc((size_t)n_items * lanes_tab_stride(B->lanes_nhi_max(P.n_bits)));
    B->masks.alloc(n_items * P.t * 32);
    B->h_rng.resize(n_items * 32);
    B->h_weights.resize(n_items * 32);
    B->h_status.resize(n_items);
    if (challenges32) {
      // caller-side Fiat-Shamir: challenges arrive canonical, are kept in Montgomery form like k_transcripts' output
      if (!rng_out32) throw ProofErr{BPP_ERR_INVALID_ARGUMENT, "rng_out32 is required with external challenges"};
      B->ext_challenges = true;
      B->ext_status.assign(n_items, 0);
      std::vector<sc> hc((size_t)n_items * B->cs);
      memset(hc.data(), 0, hc.size() * sizeof(sc));
      for (size_t i = 0; i < n_items; i++) {
        const ProofDesc &d = B->desc[i];
        if (!challenges32[i]) throw ProofErr{BPP_ERR_INVALID_ARGUMENT, "missing challenges for a proof"};
        for (uint32_t k = 0; k < d.rounds + 3; k++) {
          const uint8_t *c = challenges32[i] + 32 * (size_t)k;
          if (!sc_is_canonical(c)) throw ProofErr{BPP_ERR_INVALID_ARGUMENT, "challenge is not canonical"};
          sc v;
          sc_load_words(v, c);
          if (sc_iszero(v)) B->ext_status[i] |= BPP_ST_TRANSCRIPT_FAIL;  // transcript_protocol.rs:71-77
          sc_to_mont(v, v);
          if (k < B->cs) hc[i * B->cs + k] = v;
        }
        // validate_and_append_point (transcript_protocol.rs:48-61): A, A1, B, L_j, R_j must not be the identity encoding
        if (B->defer[i] & BPP_DEFER_DEGREE) continue;  // other layout: its chunk fails before PASS 1 is looked at
        const uint8_t *pA = bytes + d.proof_off + 1 + 32 * P.t;  // the staged copy of the proof
        auto zero32 = [](const uint8_t *p) {
          uint8_t r = 0;
          for (int k = 0; k < 32; k++) r |= p[k];
          return r == 0;
        };
        bool ident = zero32(pA) || zero32(pA + 32) || zero32(pA + 64);
        for (uint32_t j = 0; j < 2 * d.rounds; j++) ident = ident || zero32(pA + 160 + 32 * j);
        if (ident) B->ext_status[i] |= BPP_ST_TRANSCRIPT_FAIL;
      }
      HIP_CHECK(hipMemcpyAsync(B->chal.p, hc.data(), hc.size() * sizeof(sc), hipMemcpyHostToDevice, s));
      memcpy(B->h_rng.data(), rng_out32, n_items * 32);
      HIP_CHECK(hipMemcpyAsync(B->rng_out.p, rng_out32, n_items * 32, hipMemcpyHostToDevice, s));
      B->d_ext_status.alloc(n_items);
      HIP_CHECK(hipMemcpyAsync(B->d_ext_status.p, B->ext_status.data(), n_items * 4, hipMemcpyHostToDevice, s));
    }
    // initial per-proof status of every verification of this batch: caller-side PASS-1 findings (if any) and the
    // statement's commitments, decoded here once (the reference's RangeStatement holds decompressed points)
    // (k_ingest cleared status0[] and status[]; commitments that do not decode are recorded in both: every verification finds
    // status[] == status0[] and leaves it so (k_results_out), no copy in front of PASS 1)
    if (B->ext_challenges) {
      HIP_CHECK(hipMemcpyAsync(B->status0.p, B->d_ext_status.p, n_items * 4, hipMemcpyDeviceToDevice, s));
      HIP_CHECK(hipMemcpyAsync(B->status.p, B->d_ext_status.p, n_items * 4, hipMemcpyDeviceToDevice, s));
    }
    if (B->sum_m)
      hipLaunchKernelGGL(k_decompress, dim3(cdiv(B->sum_m, 64)), dim3(64), 0, s, B->bytes.p, B->src_off.p, B->owner.p,
                         B->idx_commit.p, B->sum_m, B->dynpts.p, B->status0.p, (uint32_t *)nullptr, B->status.p);
    HIP_CHECK(hipGetLastError());
    B->status_clean = true;
    HIP_CHECK(hipStreamSynchronize(s));
    const uint64_t h = g_next_handle.fetch_add(1);
    ctx->batches[h] = std::move(B);
    registered = true;
    return h;
}

int upload_impl(bpp_ctx *ctx, uint64_t params, const bpp_verify_item *items, size_t n_items, const bpp_packed_batch *packed,
                uint64_t *batch, const uint8_t *const *challenges32, const uint8_t *rng_out32, char *errbuf, size_t errbuf_len) {
  try {
    const std::shared_ptr<Params> Pp = params_registry().get(params);
    if (!Pp || Pp->device != ctx->device) return fail(ctx, BPP_ERR_BAD_HANDLE, "unknown params handle", errbuf, errbuf_len);
    if (packed) n_items = packed->n_items;
    // verify_batch: by definition an empty batch fails (src/range_proof.rs:719-723)
    if ((!items && !packed) || n_items == 0 || !batch)
      return fail(ctx, BPP_ERR_INVALID_ARGUMENT, "Range statements or proofs length empty", errbuf, errbuf_len);
    if (n_items > (1u << 24)) return fail(ctx, BPP_ERR_SIZE_OVERFLOW, "batch too large", errbuf, errbuf_len);
    UploadPlan pl;
    PlanWipe wipe_plan{pl};
    upload_host_pack(ctx, *Pp, items, n_items, packed, pl);
    *batch = upload_device(ctx, Pp, params, pl, challenges32, rng_out32);
    return BPP_OK;
  }
  BPP_CATCH(ctx, errbuf, errbuf_len)
}
}  // namespace

extern "C" {

int bpp_batch_upload(bpp_ctx *ctx, uint64_t params, const bpp_verify_item *items, size_t n_items, uint64_t *batch,
                     char *errbuf, size_t errbuf_len) {
  BPP_ENTRY(ctx);
  return upload_impl(ctx, params, items, n_items, nullptr, batch, nullptr, nullptr, errbuf, errbuf_len);
}

int bpp_batch_upload_packed(bpp_ctx *ctx, uint64_t params, const bpp_packed_batch *in, uint64_t *batch, char *errbuf,
                            size_t errbuf_len) {
  BPP_ENTRY(ctx);
  if (!in) return fail(ctx, BPP_ERR_INVALID_ARGUMENT, "Range statements or proofs length empty", errbuf, errbuf_len);
  return upload_impl(ctx, params, nullptr, 0, in, batch, nullptr, nullptr, errbuf, errbuf_len);
}

int bpp_host_threads(void) { return (int)host_pool_size(); }
uint64_t bpp_host_pool_cpu_ns(void) { return host_pool_cpu_ns(); }

int bpp_shader_clock(bpp_ctx *ctx, uint32_t window_us, double *ghz) {
  BPP_ENTRY(ctx);
  try {
    if (!ghz) return fail(ctx, BPP_ERR_INVALID_ARGUMENT, "null argument");
    DevBuf<uint64_t> d;
    d.alloc(2);
    // s_sleep 127 naps for 127 x 64 clocks: ~3.5 us at 2.3 GHz
    const uint32_t naps = std::max<uint32_t>(1, std::min<uint32_t>(window_us, 2000000u) * 2 / 7);
    hipLaunchKernelGGL(k_shader_clock, dim3(1), dim3(64), 0, ctx->stream, d.p, naps);
    uint64_t h[2] = {0, 0};
    gpu_wait_stream(ctx, ctx->stream, true);  // (naps: a sampling thread that spins for the whole window shows up as a busy host core)
    HIP_CHECK(hipMemcpy(h, d.p, 16, hipMemcpyDeviceToHost));
    *ghz = h[1] ? (double)h[0] / (double)h[1] * 0.1 : 0.0;
    return BPP_OK;
  }
  BPP_CATCH(ctx, nullptr, 0)
}

int bpp_batch_destroy(bpp_ctx *ctx, uint64_t batch) {
  BPP_ENTRY(ctx);
  (void)hipStreamSynchronize(ctx->stream);
  auto it = ctx->batches.find(batch);
  if (it == ctx->batches.end()) return BPP_ERR_BAD_HANDLE;
  wipe_batch_secrets(*it->second, ctx->stream);  // seed nonces, recovered masks: gone before the buffers change hands
  (void)hipStreamSynchronize(ctx->stream);
  ctx->spare_batch = std::move(it->second);  // nothing of it is in flight any more; its allocations serve the next upload
  ctx->batches.erase(it);
  return BPP_OK;
}

}  // extern "C"

namespace {

// Weight-independent device work for the whole resident batch: PASS 1, decompression and -- unless `pass1_only` --
// the per-proof scalar block (k_scalars_shared).  Returns after the transcript-RNG bytes have reached the host (h_rng); the rest is
// still running on the stream (it overlaps the host weight chain).
// rng_dev_dst != nullptr (the sharded form): the transcript-RNG bytes are copied device -> device right behind PASS 1 and
// ev_rng is recorded there; nothing comes to the host and the function does not wait
void plan_lanes(bpp_ctx *ctx, Batch &b, bool for_phase2 = false);

// Where a call's weight chains run (option "chain" / BPP_CHAIN: 0 host, 1 device, -1 this rule).  The host runs a chain five
// to eight times faster than a wavefront does (chain_host.h: 0.27 us per proof, chain_dev.h: ~2), so a call that WAITS for its
// chains -- one call at a time, few groups -- keeps them on the host.  The device form is for callers that keep several
// calls in flight and want the host left alone (a rank of an 8-GPU node: bench.py): it is chosen explicitly.
enum ChainMode { CHAIN_HOST = 0, CHAIN_DEVICE = 1, CHAIN_HOST_WIDE = 2 };
// CHAIN_HOST_WIDE: the sponges on host cores, Scalar::from_bytes_mod_order_wide (half of a lock-step bundle's CPU time, and
// perfectly parallel) on the device: the default for calls of BPP_WAIT_NAP_MIN_PROOFS proofs and more -- their callers keep
// several calls in flight, and there a host core counts for more than the extra launch.
ChainMode chain_mode(const bpp_ctx *ctx, const Batch &b, bool allow_device_work) {
  if (!allow_device_work) return CHAIN_HOST;  // the second run after a zero weight: everything as the reference does it
  if (ctx->opt.chain >= 0 && ctx->opt.chain <= 2) return (ChainMode)ctx->opt.chain;
  return b.B >= BPP_WAIT_NAP_MIN_PROOFS ? CHAIN_HOST_WIDE : CHAIN_HOST;
}
// the wide bytes of the host sponges (b.h_wide, mapped) -> b.weights.  launch = false: k_scalars_lanes does it in its prologue
// (enqueue_phase2), this only arms the zero flag
void enqueue_finish_wide(bpp_ctx *ctx, Batch &b, hipStream_t st, bool launch = true) {
  if (!ctx->h_chain_zero.p) ctx->h_chain_zero.resize(1);
  ctx->h_chain_zero[0] = 0;
  const uint32_t test_zero = ctx->opt.chain_test_zero > 0 ? (uint32_t)ctx->opt.chain_test_zero : 0u;
  if (launch)
    hipLaunchKernelGGL(k_chain_finish_bytes, dim3(cdiv(b.B, 64)), dim3(64), 0, st, b.h_wide.dev(), b.B, b.weights.p, ctx->h_chain_zero.dev(), test_zero);
  ctx->wide_chain_calls++;
}

// k_weight_chain + k_chain_finish on `st`: b.rng_out -> b.weights (canonical bytes), one wavefront per group
void enqueue_device_chain(bpp_ctx *ctx, Batch &b, hipStream_t st) {
  if (!ctx->d_chain_t0.p) {  // once per context: the weight transcript as Transcript::new leaves it (src/range_proof.rs:811)
    Strobe t0;
    const char *lbl = "Bulletproofs+ verifier weights";
    merlin_new(t0, (const uint8_t *)lbl, (uint32_t)strlen(lbl));
    ctx->d_chain_t0.alloc(1);
    HIP_CHECK(hipMemcpy(ctx->d_chain_t0.p, &t0, sizeof(t0), hipMemcpyHostToDevice));
    ctx->h_chain_zero.resize(1);
  }
  b.chain_wide.alloc((size_t)b.B * 16);
  ctx->h_chain_zero[0] = 0;
  const uint32_t test_zero = ctx->opt.chain_test_zero > 0 ? (uint32_t)ctx->opt.chain_test_zero : 0u;
  hipLaunchKernelGGL(k_weight_chain, dim3(b.G), dim3(64), 0, st, b.rng_out.p, b.group_first.p, b.G, ctx->d_chain_t0.p, b.chain_wide.p);
  hipLaunchKernelGGL(k_chain_finish, dim3(cdiv(b.B, 64)), dim3(64), 0, st, b.chain_wide.p, b.B, b.weights.p, ctx->h_chain_zero.dev(), test_zero);
  ctx->device_chain_calls++;
}

// dev_chain: the weight chains follow PASS 1 as kernels (option "chain_inline" = 0: on a stream of their own beside the decompression
// and the weight-free scalars, joined by enqueue_phase2): nothing comes to the host and the function does not wait
void enqueue_phase1(bpp_ctx *ctx, Batch &b, StageTimer &tm, bool pass1_only, uint8_t *rng_dev_dst = nullptr, size_t rng_row_bytes = 0,
                    size_t rng_dst_pitch = 0, bool dev_chain = false) {
  const bool fetch_rng = rng_dev_dst == nullptr && !dev_chain;
  b.dev_chain_pending = false;
  Params &P = *b.params;
  hipStream_t s = ctx->stream;
  if (!pass1_only) plan_lanes(ctx, b);  // (an option may have changed since the layout was built: k_scalars_shared writes what PASS 2 will read)
  if (!ctx->ev_rng_ready) {
    HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_rng, hipEventDisableTiming));
    ctx->ev_rng_ready = true;
  }
  // status[] starts as status0[]: the previous verification's last kernel left it so (k_results_out), unless that call died
  // half way
  if (!b.status_clean) HIP_CHECK(hipMemcpyAsync(b.status.p, b.status0.p, (size_t)b.B * 4, hipMemcpyDeviceToDevice, s));
  b.status_clean = false;
  // Decompression needs nothing PASS 1 produces (both only OR bits into status[]).  A small input leaves most of the chip
  // idle, so its decompression runs beside PASS 1 and the weight-free scalars on a second stream and joins before the
  // weights are needed (one 256-proof call 0.84 -> 0.79 ms).  Large inputs fill the chip either way: one stream, less
  // bookkeeping (measured: no gain, HISTORY.md 3).  Stage profiling keeps the serial order so that its intervals mean something.
  const bool side = (ctx->opt.side_decompress >= 0 ? ctx->opt.side_decompress != 0 : b.B <= BPP_SIDE_DECOMPRESS_MAX) && !(ctx->profile && !ctx->profile_light);
  const uint32_t n_proof_pts = b.total_dyn - b.sum_m;
  auto launch_decompress = [&](hipStream_t st) {
    hipLaunchKernelGGL(k_decompress, dim3(cdiv(n_proof_pts, 64)), dim3(64), 0, st, b.bytes.p, b.src_off.p, b.owner.p,
                       b.idx_proof.p, n_proof_pts, b.dynpts.p, b.status.p, b.dec_spill.p, (uint32_t *)nullptr);
    // half-scalar plan of small calls: the 2^126 multiples of every dynamic point (the statements' commitments were decoded
    // at upload), 126 doublings each, right behind the decompression and -- for small inputs -- beside PASS 1 and the scalars.
    // A point that did not decode left an arbitrary entry: its multiple is never looked at (the call fails on the status).
    if (b.msm.split && !pass1_only)
      hipLaunchKernelGGL(k_split_shift_quad, dim3(cdiv(b.total_dyn, 16)), dim3(64), 0, st, b.dynpts.p, b.total_dyn, b.dyn_hi.p);
  };
  if (side) {
    if (!ctx->side_stream) {
      HIP_CHECK(hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking));
      HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
      HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
    }
    HIP_CHECK(hipEventRecord(ctx->ev_fork, s));
    HIP_CHECK(hipStreamWaitEvent(ctx->side_stream, ctx->ev_fork, 0));
    launch_decompress(ctx->side_stream);
    HIP_CHECK(hipEventRecord(ctx->ev_join, ctx->side_stream));
  }
  if (b.ext_challenges) {  // caller did PASS 1: challenges + rng bytes are already resident
    tm.mark(M_START);
  } else {
    if (!b.uniform_rounds) HIP_CHECK(hipMemsetAsync(b.chal.p, 0, (size_t)b.B * b.cs * sizeof(sc), s));  // trace padding only
    tm.mark(M_START);
    // small inputs: one proof per wavefront (latency); large inputs: one proof per lane (issue slots)
    const int force_wave = ctx->opt.transcripts_wave;  // (tests force either kernel)
    const bool wave = force_wave >= 0 ? force_wave != 0 : b.B <= BPP_TRANSCRIPTS_WAVE_MAX;
    uint8_t *rng_host = fetch_rng ? b.h_rng.dev() : nullptr;  // mapped page-locked memory: no device-to-host copy behind PASS 1
    if (wave)
      hipLaunchKernelGGL(k_transcripts_wave, dim3(b.B), dim3(64), 0, s, b.bytes.p, b.d_desc.p, b.minvals.p, b.states.p,
                         P.d_hg32.p, P.n_bits, P.t, b.B, b.cs, b.chal.p, b.rng_out.p, b.status.p, rng_host);
    else
      hipLaunchKernelGGL(k_transcripts, dim3(cdiv(b.B, 64)), dim3(64), 0, s, b.bytes.p, b.d_desc.p, b.minvals.p, b.states.p,
                         P.d_hg32.p, P.n_bits, P.t, b.B, b.cs, b.chal.p, b.rng_out.p, b.status.p, rng_host);
  }
  tm.mark(M_TRANSCRIPTS);
  if (dev_chain) {
    // On the call's own stream (the rule): measured against a high-priority stream of their own beside the decompression -- the same
    // rate once one more step is in flight, and the cross-stream events of that form leave a helper thread of the runtime at 80 %
    // of a core (profiles/r06_chain_inline_ab.txt); "chain_inline" = 0 brings the side stream back
    if (ctx->profile || ctx->opt.chain_inline != 0) {
      enqueue_device_chain(ctx, b, s);
      tm.mark(M_CHAIN);
    } else {
      if (!ctx->chain_stream) {
        int least = 0, greatest = 0;  // the chains are 64 lone wavefronts on the call's critical path: first in line for a slot
        HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIP_CHECK(hipStreamCreateWithPriority(&ctx->chain_stream, hipStreamNonBlocking, greatest));
        HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_chain_fork, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_chain_done, hipEventDisableTiming));
      }
      HIP_CHECK(hipEventRecord(ctx->ev_chain_fork, s));
      HIP_CHECK(hipStreamWaitEvent(ctx->chain_stream, ctx->ev_chain_fork, 0));
      enqueue_device_chain(ctx, b, ctx->chain_stream);
      HIP_CHECK(hipEventRecord(ctx->ev_chain_done, ctx->chain_stream));
      b.dev_chain_pending = true;
    }
  }
  if (fetch_rng || !rng_dev_dst) {
    // (nothing to copy: PASS 1 wrote the bytes the weight chain reads into h_rng itself -- or the chain runs on the device; the
    // event below tells the host)
  } else if (rng_dst_pitch && rng_dst_pitch != rng_row_bytes)  // rows of one group each, padded to the widest rank's shard
    HIP_CHECK(hipMemcpy2DAsync(rng_dev_dst, rng_dst_pitch, b.rng_out.p, rng_row_bytes, rng_row_bytes, ((size_t)b.B * 32) / rng_row_bytes,
                               hipMemcpyDeviceToDevice, s));
  else HIP_CHECK(hipMemcpyAsync(rng_dev_dst, b.rng_out.p, (size_t)b.B * 32, hipMemcpyDeviceToDevice, s));  // gathered over RCCL
  HIP_CHECK(hipEventRecord(ctx->ev_rng, s));
  if (!side) launch_decompress(s);
  tm.mark(M_DECOMPRESS);
  if (!pass1_only) {  // the weight-independent part of the PASS-2 scalars; the rest (k_scalars_lanes) takes the weights
    // tables of the generator-row kernel: by the same lane for large inputs, one wavefront per proof for small ones
    const bool tw = ctx->opt.tables_wave >= 0 ? ctx->opt.tables_wave != 0 : b.B <= BPP_TABLES_WAVE_MAX;  // (tests force either form)
    b.gemm_hi_ready = b.static_gemm && !tw;
    if (b.gemm_hi_ready)  // the high tables once more as digit tables (kernels_static_gemm.h)
      hipLaunchKernelGGL(k_scalars_shared<true>, dim3(cdiv(b.B, 64)), dim3(64), 0, s, b.bytes.p, b.d_desc.p, b.minvals.p, b.chal.p, P.n_bits,
                         P.t, b.cs, b.B, b.shr.p, b.lanes_nhi_max(P.n_bits), b.tab.p, b.gemm_hi.p, cdiv(b.B, SGEMM_BLOCK));
    else
      hipLaunchKernelGGL(k_scalars_shared<false>, dim3(cdiv(b.B, 64)), dim3(64), 0, s, b.bytes.p, b.d_desc.p, b.minvals.p, b.chal.p, P.n_bits,
                         P.t, b.cs, b.B, b.shr.p, b.lanes_nhi_max(P.n_bits), tw ? (sc *)nullptr : b.tab.p, (int8_t *)nullptr, 0u);
    if (tw)
      hipLaunchKernelGGL(k_scalars_tables_wave, dim3(b.B), dim3(64), 0, s, b.d_desc.p, b.shr.p, P.n_bits, b.B,
                         b.lanes_nhi_max(P.n_bits), b.tab.p);
    tm.mark(M_SCALARS);
  }
  if (side) HIP_CHECK(hipStreamWaitEvent(s, ctx->ev_join, 0));
  HIP_CHECK(hipGetLastError());
  if (fetch_rng) gpu_wait_event(ctx->ev_rng, wait_naps(ctx, b.B), &ctx->wait_hint_rng, ((uint64_t)b.B << 32) | b.G);
}

// Weight chains of all groups (src/range_proof.rs:811,849,853,894).  Chunks are independent reference batches: groups of
// equal size run W at a time in lockstep on vector Keccak (chain_host.h), bundles are spread over host threads.
void run_weight_chains(Batch &b) { run_weight_chains_generic(b.h_rng.data(), b.h_weights.data(), b.h_group_first.data(), b.G); }
// the sponges alone: 64 PRF bytes per proof into b.h_wide, reduced (and looked at for zeros) by k_chain_finish_bytes
void run_weight_chains_wide(Batch &b) {
  b.h_wide.resize((size_t)b.B * 64);
  run_weight_chains_generic(b.h_rng.data(), b.h_wide.data(), b.h_group_first.data(), b.G, true);
}
// Persistent host workers for the weight chains (spawning threads per call costs more than a 1024-proof chain).
class HostPool {
 public:
  static HostPool &get() {
    static HostPool *p = new HostPool();  // leaked on purpose: detached workers may still wait at process exit
    return *p;
  }
  uint32_t size() const { return (uint32_t)workers_.size() + 1; }
  // calls that have work queued on the pool right now (a hint for the chain scheduler, nothing is promised)
  uint32_t busy() {
    std::lock_guard<std::mutex> lk(mu_);
    return (uint32_t)jobs_.size();
  }
  // run fn(i) for i in [0, n) on the pool + the calling thread; returns when all are done
  void parallel_for(uint32_t n, const std::function<void(uint32_t)> &fn) {
    if (n == 0) return;
    if (n == 1 || workers_.empty()) {
      CpuSpan span;
      for (uint32_t i = 0; i < n; i++) fn(i);
      return;
    }
    auto job = std::make_shared<Job>();
    job->fn = &fn;
    job->n = n;
    {
      std::lock_guard<std::mutex> lk(mu_);
      jobs_.push_back(job);
    }
    cv_.notify_all();
    work_on(*job);
    std::unique_lock<std::mutex> lk(job->mu);
    job->cv.wait(lk, [&] { return job->done.load() == n; });
    std::lock_guard<std::mutex> lk2(mu_);
    for (auto it = jobs_.begin(); it != jobs_.end(); ++it)
      if (it->get() == job.get()) {
        jobs_.erase(it);
        break;
      }
  }

 private:
  struct Job {
    const std::function<void(uint32_t)> *fn;
    uint32_t n;
    std::atomic<uint32_t> next{0}, done{0};
    std::mutex mu;
    std::condition_variable cv;
  };
  // host threads this process may really run at once: the affinity mask, capped by a cgroup CPU quota when there is one (a
  // 1-GPU box of the pool reports 256 logical CPUs and schedules 16: sized by hardware_concurrency() the pool had 32 workers
  // there, and the chain scheduler, which compares the number of chains with the pool, took 64 chains of 4096 proofs for
  // "few": scalar chains, 10 ms per call instead of 2 ms as eight lock-step bundles)
  static uint32_t usable_cpus() {
    uint32_t n = std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0 && CPU_COUNT(&set) > 0) n = std::min<uint32_t>(n, (uint32_t)CPU_COUNT(&set));
    auto read2 = [](const char *path, long long &a, long long &b, bool &a_is_max) {
      FILE *f = fopen(path, "r");
      if (!f) return false;
      char w[64] = {0};
      a_is_max = false;
      bool ok = false;
      if (fscanf(f, "%63s %lld", w, &b) == 2) {
        ok = true;
        if (strcmp(w, "max") == 0) a_is_max = true;
        else a = atoll(w);
      }
      fclose(f);
      return ok;
    };
    long long q = 0, per = 0;
    bool is_max = false;
    if (read2("/sys/fs/cgroup/cpu.max", q, per, is_max)) {  // cgroup v2: "<quota|max> <period>"
      if (!is_max && q > 0 && per > 0) n = std::min<uint32_t>(n, (uint32_t)std::max<long long>(1, (q + per / 2) / per));
    } else {  // cgroup v1
      FILE *fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r"), *fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
      if (fq && fp && fscanf(fq, "%lld", &q) == 1 && fscanf(fp, "%lld", &per) == 1 && q > 0 && per > 0)
        n = std::min<uint32_t>(n, (uint32_t)std::max<long long>(1, (q + per / 2) / per));
      if (fq) fclose(fq);
      if (fp) fclose(fp);
    }
    return std::max(1u, n);
  }
  HostPool() {
    uint32_t hw = usable_cpus();
    const char *e = getenv("BPP_HOST_THREADS");
    uint32_t want = e ? (uint32_t)atoi(e) : std::min(hw, 32u);
    want = std::max(1u, std::min(want, 256u));
    for (uint32_t i = 1; i < want; i++)
      workers_.emplace_back([this] {
        pthread_setname_np(pthread_self(), "bpp-chain");  // (who is busy: bench.py's host_cores_busy_by_thread)
        loop();
      });
    for (auto &t : workers_) t.detach();
  }
 public:
  // CPU time spent inside jobs (thread clocks, so a thread that was not scheduled is not counted): what bpp_host_pool_cpu_ns
  // reports -- with several ranks on one host it tells a host-bound step from a GPU-bound one
  static uint64_t thread_cpu_ns() {
    struct timespec ts;
    clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
    return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
  }
  static std::atomic<uint64_t> &cpu_ns() {
    static std::atomic<uint64_t> v{0};
    return v;
  }
  // (the thread clock is a system call, ~1 us: read once per thread and JOB, never per item -- a 256-proof call parses 256 items)
  struct CpuSpan {
    uint64_t t0 = thread_cpu_ns();
    ~CpuSpan() { cpu_ns().fetch_add(thread_cpu_ns() - t0, std::memory_order_relaxed); }
  };

 private:
  static void work_on(Job &j) {
    if (j.next.load() >= j.n) return;  // nothing left: no clock reads either
    CpuSpan span;
    for (;;) {
      uint32_t i = j.next.fetch_add(1);
      if (i >= j.n) return;
      (*j.fn)(i);
      if (j.done.fetch_add(1) + 1 == j.n) {
        std::lock_guard<std::mutex> lk(j.mu);
        j.cv.notify_all();
      }
    }
  }
  void loop() {
    for (;;) {
      std::shared_ptr<Job> job;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] {
          for (auto &j : jobs_)
            if (j->next.load() < j->n) return true;
          return false;
        });
        for (auto &j : jobs_)
          if (j->next.load() < j->n) {
            job = j;
            break;
          }
      }
      if (job) work_on(*job);
    }
  }
  std::vector<std::thread> workers_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<std::shared_ptr<Job>> jobs_;
};

void host_parallel_for(uint32_t n, const std::function<void(uint32_t)> &fn) { HostPool::get().parallel_for(n, fn); }
uint32_t host_pool_size() { return HostPool::get().size(); }
uint64_t host_pool_cpu_ns() { return HostPool::cpu_ns().load(std::memory_order_relaxed); }

void run_weight_chains_generic(const uint8_t *h_rng, uint8_t *h_weights, const uint32_t *group_first, uint32_t G, bool wide) {
  const size_t stride = wide ? 64 : 32;  // output bytes per proof
  static const int simd = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl") ? 8
                          : (__builtin_cpu_supports("avx2") ? 4 : 1);
  HostPool &pool = HostPool::get();
  // One scalar chain per worker has the lowest latency when workers are plentiful; lock-step vector bundles
  // (chain_host.h) need ~5x less CPU time per chain and keep the GPU fed when several calls / ranks share few cores
  // (measured, 64 groups x 4 calls in flight: 4 host threads 14.9 M proofs/s lock-step vs 12.5 M scalar; 32 threads
  // 15.3 M vs 15.5 M).  Used as soon as the groups outnumber half the workers -- unless the pool is idle and two rounds of
  // scalar chains cover the call: a bundle of eight takes three times as long as one scalar chain, and a lone call of 16 or
  // 32 chains (a sharded call's share of the replay) is waited for by its caller with the GPU idle (32 chains of 4096
  // proofs on 16 workers: 3.4 ms as four bundles, 2.2 ms as scalar chains).
  static const int force = getenv("BPP_CHAIN_LOCKSTEP") ? atoi(getenv("BPP_CHAIN_LOCKSTEP")) : -1;
  const bool crowded = G > 2u * pool.size() || pool.busy() > 0;
  const uint32_t W = force >= 0 ? (force ? (uint32_t)simd : 1u)
                                : ((G >= 2u * (uint32_t)simd && 2 * G > pool.size() && crowded) ? (uint32_t)simd : 1u);
  struct Unit {
    uint32_t g0, cnt;
  };
  std::vector<Unit> units;
  for (uint32_t g = 0; g < G;) {
    const uint32_t n = group_first[g + 1] - group_first[g];
    uint32_t cnt = 1;
    while (cnt < W && g + cnt < G && group_first[g + cnt + 1] - group_first[g + cnt] == n) cnt++;
    units.push_back({g, cnt});
    g += cnt;
  }
  auto scalar_chain = [&](uint32_t g) {
    const uint32_t p0 = group_first[g], p1 = group_first[g + 1];
    if (wide) wide_chain_single(h_rng + (size_t)p0 * 32, p1 - p0, h_weights + (size_t)p0 * 64);
    else weights_from_chain_host(h_rng + (size_t)p0 * 32, p1 - p0, h_weights + (size_t)p0 * 32);
  };
  std::function<void(uint32_t)> run_unit = [&](uint32_t ui) {
    const Unit &u = units[ui];
    uint32_t g = u.g0, left = u.cnt;
    const size_t n = group_first[g + 1] - group_first[g];
    while (left) {
      uint32_t w = (left >= 8 && simd >= 8) ? 8 : ((left >= 4 && simd >= 4) ? 4 : 1);
      if (w == 1) {
        scalar_chain(g);
      } else {
        const uint8_t *in[8];
        uint8_t *out[8];
        for (uint32_t k = 0; k < w; k++) {
          in[k] = h_rng + (size_t)group_first[g + k] * 32;
          out[k] = h_weights + (size_t)group_first[g + k] * stride;
        }
        if (wide) {  // (zeros are the device's to find: k_chain_finish_bytes)
          if (w == 8) wide_chain_x8(in, n, out);
          else wide_chain_x4(in, n, out);
          g += w;
          left -= w;
          continue;
        }
        if (w == 8) weights_chain_x8(in, n, out);
        else weights_chain_x4(in, n, out);
        // Scalar::random_not_zero: a zero draw (2^-252) means the lockstep result is off from there on -> redo scalar
        for (uint32_t k = 0; k < w; k++) {
          bool zero = false;
          for (size_t i = 0; i < n && !zero; i++) zero = weight_is_zero(out[k] + 32 * i);
          if (zero) scalar_chain(g + k);
        }
      }
      g += w;
      left -= w;
    }
  };
  pool.parallel_for((uint32_t)units.size(), run_unit);
}

// status words (and, where asked for, the groups' identity flags and the recovered masks) -> mapped host memory, status[]
// back to its initial value: one launch, no copy (kernels_verify.h: k_results_out).  Valid on the host once the stream has
// been synchronised.
void fetch_results(bpp_ctx *ctx, Batch &b, bool ident, bool masks) {
  if (ident) b.h_ident.resize(b.G);
  const uint32_t pieces = masks ? (uint32_t)(((size_t)b.B * b.params->t * 32) / 16) : 0u;
  if (masks) b.h_masks.resize((size_t)pieces * 16);
  const uint32_t items = std::max(b.B, ident ? b.G : 0u);
  b.h_status_any.resize(cdiv(b.B, BPP_STATUS_BLOCK));
  b.status_settled = false;
  hipLaunchKernelGGL(k_results_out, dim3(cdiv(items, BPP_STATUS_BLOCK)), dim3(BPP_STATUS_BLOCK), 0, ctx->stream, b.status.p, b.status0.p,
                     b.h_status.dev(), b.h_status_any.dev(), b.B,
                     ident ? b.msm.is_identity.p : (const uint32_t *)nullptr, ident ? b.h_ident.dev() : (uint32_t *)nullptr, b.G,
                     masks ? (const uint4 *)b.masks.p : (const uint4 *)nullptr, masks ? (uint4 *)b.h_masks.dev() : (uint4 *)nullptr, pieces);
  HIP_CHECK(hipGetLastError());
  b.status_clean = true;  // (in stream order: whatever is enqueued behind this launch finds status0)
}
void fetch_status(bpp_ctx *ctx, Batch &b) { fetch_results(ctx, b, false, false); }
// After the stream has been synchronised behind fetch_results: h_status as every reader expects it.  The kernel wrote the words
// of the blocks that hold a finding; every other block is zero by its summary word and is cleared here only if an earlier
// verification left something in it -- on the accept path this touches one word per 256 proofs.
void settle_status(Batch &b) {
  if (b.status_settled) return;
  const uint32_t n_blk = cdiv(b.B, BPP_STATUS_BLOCK);
  if (b.h_status_dirty.size() != n_blk || b.h_status_dirty_of != b.h_status.data()) {  // fresh page-locked memory: contents unknown
    b.h_status_dirty.assign(n_blk, 1);
    b.h_status_dirty_of = b.h_status.data();
  }
  for (uint32_t k = 0; k < n_blk; k++) {
    if (b.h_status_any[k]) {
      b.h_status_dirty[k] = 1;
    } else if (b.h_status_dirty[k]) {
      // the WHOLE block of the allocation, not just this batch's proofs: the flags travel with the buffer to batches of other sizes
      const size_t lo = (size_t)k * BPP_STATUS_BLOCK, hi = std::min<size_t>(b.h_status.n, lo + BPP_STATUS_BLOCK);
      memset(b.h_status.data() + lo, 0, (hi - lo) * 4);
      b.h_status_dirty[k] = 0;
    }
  }
  b.status_settled = true;
}

// reference error precedence for proofs [p0, p1) treated as one verify() call (upload_host.h: check_chunk_errors)
void check_chunk_errors(Batch &b, uint32_t p0, uint32_t p1) {
  settle_status(b);
  bpp::check_chunk_errors(b.h_status.data(), b.rounds_bad.data(), p0, p1);
}

// Shape of k_scalars_lanes for the current group layout, and the buffers that go with it (called when a layout is built and
// again in front of every PASS 2: the option may have changed; nothing is reallocated once the sizes have been seen).
void plan_lanes(bpp_ctx *ctx, Batch &b, bool for_phase2) {
  // proofs per workgroup: enough (proof, generator pair) items for four passes of the 64 lanes.  Large inputs take sixteen
  // 64-bit proofs per workgroup (the prologue's product jobs, 43 per proof, then fill whole wavefronts); small inputs keep four
  // (more workgroups, shorter chains: the chip is idle anyway).
  const uint32_t per_wg = b.B <= BPP_TABLES_WAVE_MAX ? 256u : 1024u;
  b.lanes_ppw = std::max<uint32_t>(1, std::min<uint32_t>(BPP_LANES_MAX_PPW, per_wg / std::max<uint32_t>(1, b.max_mn)));
  // Generator columns summed inside k_scalars_lanes (no per-proof rows at all) when a lane can own a generator index across the
  // workgroup's proofs (max_mn a multiple of 64) and no workgroup straddles a group boundary
  bool aligned = b.max_mn >= 64 && b.max_mn % 64 == 0;
  for (uint32_t g = 1; g < b.G && aligned; g++) aligned = b.h_group_first[g] % b.lanes_ppw == 0;
  b.fused_columns = aligned && ctx->opt.fused_columns != 0;  // (tests force the per-proof rows with fused_columns = 0)
  // The generator columns as a matrix product over the proofs of each group on the matrix cores (kernels_static_gemm.h) when
  // every proof has the same shape, the tables are built by k_scalars_shared (large inputs) and every group starts on a block
  // of the digit tables; otherwise summed in k_scalars_lanes as before (tests run both: static_gemm = 0)
  if (b.uniform_mn < 0) {
    uint32_t mn0 = b.B ? b.desc[0].m * b.params->n_bits : 0;
    for (uint32_t p = 1; p < b.B && mn0; p++)
      if (b.desc[p].m * b.params->n_bits != mn0 || b.desc[p].rounds != b.desc[0].rounds) mn0 = 0;
    if (mn0 && (b.any_rounds_bad || (1u << b.desc[0].rounds) != mn0)) mn0 = 0;
    b.uniform_mn = (int)mn0;
  }
  const bool tables_wave = ctx->opt.tables_wave >= 0 ? ctx->opt.tables_wave != 0 : b.B <= BPP_TABLES_WAVE_MAX;
  const uint32_t lb = lanes_lb(b.params->n_bits);
  // (by itself from aggregation 8 on: measured with three steps in flight, 64 x 256 proofs per step, the matrix product gains 10 %
  // at m = 8, nothing at m = 2 and loses 3 - 5 % at m = 1 and 4 -- tools/agg_probe.py, profiles/r04_agg_probe.jsonl)
  const bool gemm_wanted = ctx->opt.static_gemm >= 0 ? ctx->opt.static_gemm != 0 : b.uniform_mn >= 512;
  bool gemm = b.fused_columns && gemm_wanted && !tables_wave && b.uniform_mn >= 64 && lb == 3 &&
              ((uint32_t)b.uniform_mn >> lb) >= SGEMM_HI_PER_WAVE && ((uint32_t)b.uniform_mn >> lb) <= b.lanes_nhi_max(b.params->n_bits);
  uint32_t max_group = 0;
  for (uint32_t g = 0; g < b.G && gemm; g++) {
    gemm = b.h_group_first[g] % SGEMM_BLOCK == 0;
    max_group = std::max(max_group, b.h_group_first[g + 1] - b.h_group_first[g]);
  }
  if (for_phase2 && !b.gemm_hi_ready) gemm = false;  // (PASS 1 of this verification ran under another setting)
  b.static_gemm = gemm;
  if (b.static_gemm) {
    b.gemm_mult.alloc((size_t)b.B * 5);
    const uint32_t nblk = cdiv(b.B, SGEMM_BLOCK);
    b.gemm_nkc = std::max<uint32_t>(1, cdiv(cdiv(max_group, SGEMM_BLOCK), SGEMM_KBLOCKS));
    b.rows.alloc((size_t)b.B * (b.params->t + 2));
    b.gemm_lo.alloc(3 * sgemm_table_bytes(SGEMM_LO_ENTRIES, nblk));
    b.gemm_hi.alloc(3 * sgemm_table_bytes(b.lanes_nhi_max(b.params->n_bits), nblk));
    b.gparts.alloc((size_t)b.G * b.gemm_nkc * b.max_mn * 2 * 64);
  } else if (b.fused_columns) {
    b.rows.alloc((size_t)b.B * (b.params->t + 1));
    b.parts.alloc((size_t)cdiv(b.B, b.lanes_ppw) * 2 * b.max_mn * 8);
  } else {
    b.rows.alloc((size_t)b.B * b.cols);
  }
}

// (re)build the group layout + MSM term lists for `chunk`
// `bounds` (optional): explicit group boundaries first[0..G] (0 = first[0] < ... < first[G] = B) instead of equal chunks --
// the reference batches of different callers pooled into one call (bpp_verify_resident_groups)
void layout_groups(bpp_ctx *ctx, Batch &b, size_t chunk, const std::vector<uint32_t> *bounds = nullptr) {
  if (!bounds && chunk == (size_t)-1) chunk = 0;  // ((size_t)-1 marks an explicit layout in last_chunk)
  if (bounds) {
    if (b.last_chunk == (size_t)-1 && b.G && b.h_group_first == *bounds) return;
  } else if (b.last_chunk == chunk && b.G) {
    return;
  }
  Params &P = *b.params;
  uint32_t G;
  if (bounds) {
    G = (uint32_t)bounds->size() - 1;
    b.h_group_first = *bounds;
  } else {
    const uint32_t cz = (chunk == 0 || chunk >= b.B) ? b.B : (uint32_t)chunk;
    G = cdiv(b.B, cz);
    b.h_group_first.resize(G + 1);
    for (uint32_t g = 0; g <= G; g++) b.h_group_first[g] = std::min(g * cz, b.B);
  }
  b.G = G;
  b.group_first.alloc(G + 1);
  b.scal.alloc((size_t)G * b.cols + b.total_dyn);
  plan_lanes(ctx, b);
  // terms of group g: static columns (first 2*max_mn generators, then g bases, then h) + its proofs' dynamic slots.
  // Only the G + 1 offsets come from the host; the ~1 M term entries are written by k_layout_terms (building them on the
  // host and copying two pageable vectors cost ~20 ms per freshly uploaded batch)
  const uint32_t n_gen = 2 * P.n_bits * P.m_max;
  std::vector<uint32_t> goff(G + 1), dlo(G + 1);
  uint32_t run = 0, maxg = 0;
  for (uint32_t g = 0; g < G; g++) {
    goff[g] = run;
    const uint32_t p0 = b.h_group_first[g], p1 = b.h_group_first[g + 1];
    const uint32_t d0 = b.desc[p0].dyn_off, d1 = (p1 < b.B) ? b.desc[p1].dyn_off : b.total_dyn;
    dlo[g] = d0;
    run += b.cols + (d1 - d0);
    maxg = std::max(maxg, b.cols + (d1 - d0));
  }
  goff[G] = run;
  dlo[G] = b.total_dyn;
  const bool split = msm_wants_split(ctx, run);
  if (split) {  // every term twice: (low half, P), (high half, 2^126 P)
    for (uint32_t g = 0; g <= G; g++) goff[g] *= 2;
    maxg *= 2;
    b.dyn_hi.alloc(b.total_dyn);
  }
  msm_plan_alloc(ctx, b.msm, goff, split, false);  // leaves goff in pin_small[0 .. G]
  b.group_dlo.alloc(G + 1);
  uint32_t *pin = ctx->pin_small.data();
  memcpy(pin + (G + 1), b.h_group_first.data(), (G + 1) * 4);
  memcpy(pin + 2 * (size_t)(G + 1), dlo.data(), (G + 1) * 4);
  // the three small arrays are read by k_layout_terms where they are (mapped host memory); it also leaves the device copies
  const uint32_t *pin_dev = ctx->pin_small.dev();
  hipLaunchKernelGGL(k_layout_terms, dim3(cdiv(split ? maxg / 2 : maxg, 256), G), dim3(256), 0, ctx->stream, pin_dev, pin_dev + 2 * (size_t)(G + 1),
                     pin_dev + (G + 1), G, b.cols, b.max_mn, n_gen, P.table_len, split ? 1u : 0u, b.msm.term_sidx.p, b.msm.term_pidx.p,
                     b.msm.group_off.p, b.group_dlo.p, b.group_first.p);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(ctx->stream));  // pin_small is reused by the next plan
  b.last_chunk = bounds ? (size_t)-1 : chunk;
}

// Weight-dependent tail: h_weights -> device, the weighted generator rows and dynamic scalars (k_scalars_lanes), the
// per-group column sums, final MSM.
// wide: chain = 2, the host sponges' 64 bytes per proof stand in b.h_wide (mapped).  When k_scalars_lanes is the kernel that takes
// the weights it reads and reduces them itself and leaves the canonical weights in b.weights (option "wide_in_lanes", on by
// rule: one launch fewer on the step's latency chain -- between two synchronisations the steps in flight run in step and every
// launch of the chain counts: the driver's 20-step form + 1-2 %, profiles/r06_wait_ahead_ab.txt); the matrix-product form of the
// generator columns keeps k_chain_finish_bytes in front
void enqueue_phase2(bpp_ctx *ctx, Batch &b, StageTimer &tm, bool weights_resident = false, bool wide = false) {
  Params &P = *b.params;
  hipStream_t s = ctx->stream;
  // (weights_resident: the grouped sharded form has put them into b.weights device -> device already)
  // otherwise k_scalars_lanes reads them where the chains wrote them: h_weights is mapped, each weight is read once
  plan_lanes(ctx, b, true);
  const bool wide_in_lanes = wide && !b.static_gemm && ctx->opt.wide_in_lanes != 0;
  if (wide) enqueue_finish_wide(ctx, b, s, !wide_in_lanes);
  if (b.dev_chain_pending) {  // the device chain of this call (enqueue_phase1) has written b.weights
    HIP_CHECK(hipStreamWaitEvent(s, ctx->ev_chain_done, 0));
    b.dev_chain_pending = false;
  }
  b.weights_on_host = !weights_resident;
  const uint8_t *weights = weights_resident ? b.weights.p : b.h_weights.dev();
  tm.mark(M_WEIGHTS_IN);
  sc *dyn_scal = b.scal.p + (size_t)b.G * b.cols;
  {
    const uint32_t nhi_max = b.lanes_nhi_max(P.n_bits);
    // The weighted part of the scalar block (dynamic scalars, base columns, w into the low tables) is this kernel's prologue;
    // proofs per workgroup and whether the generator columns are summed in it: layout_groups
    const uint32_t ppw = b.lanes_ppw;
    if (b.static_gemm) {  // the weighted part alone, flat over (proof, job): digit tables of the low entries, dynamic scalars, base columns
      const uint32_t rounds = b.desc[0].rounds, m0 = b.desc[0].m;
      const uint32_t n_jobs = 3u * (1u << lanes_lb(P.n_bits)) + 1u + m0 + 3u + 2u * rounds + P.t + 1u, nblk = cdiv(b.B, SGEMM_BLOCK);
      hipLaunchKernelGGL(k_gemm_mult, dim3(cdiv(4 * b.B, 64)), dim3(64), 0, s, b.shr.p, weights, b.B, b.gemm_mult.p);
      hipLaunchKernelGGL(k_gemm_jobs, dim3(nblk, cdiv(n_jobs, 4)), dim3(64), 0, s, b.d_desc.p, b.tab.p, b.shr.p, b.gemm_mult.p, P.n_bits, P.t,
                         b.B, nhi_max, n_jobs, b.rows.p, dyn_scal, b.gemm_lo.p, nblk);
    } else
      hipLaunchKernelGGL(k_scalars_lanes, dim3(cdiv(b.B, ppw)), dim3(64), ppw * lanes_lds_bytes(nhi_max), s, b.d_desc.p, b.tab.p, b.shr.p,
                         weights, P.n_bits, P.t, b.max_mn, b.cols, b.B, nhi_max, ppw, b.rows.p, dyn_scal,
                         b.fused_columns ? b.parts.p : (uint64_t *)nullptr, ctx->opt.lazy_columns != 0 ? 1u : 0u,
                         wide_in_lanes ? b.h_wide.dev() : (const uint8_t *)nullptr, b.weights.p, ctx->h_chain_zero.dev(),
                         ctx->opt.chain_test_zero > 0 ? (uint32_t)ctx->opt.chain_test_zero : 0u);
  }
  tm.mark(M_LANES);
  if (b.static_gemm) {
    const uint32_t lb = lanes_lb(P.n_bits), nhi = (uint32_t)b.uniform_mn >> lb, nblk = cdiv(b.B, SGEMM_BLOCK);
    const uint32_t nhi_max = b.lanes_nhi_max(P.n_bits);
    hipLaunchKernelGGL(k_static_gemm, dim3(8 * cdiv(b.G, 8) * SGEMM_LO_ENTRIES * (nhi / SGEMM_HI_PER_WAVE), b.gemm_nkc), dim3(64), 0, s, b.gemm_lo.p,
                       b.gemm_hi.p, b.group_first.p, nblk, nhi, nhi_max, b.gemm_nkc, b.max_mn, b.G, b.gparts.p);
    hipLaunchKernelGGL(k_static_finish, dim3(cdiv(2 * b.max_mn, 64) + 1, b.G), dim3(64), 0, s, b.gparts.p, b.rows.p, b.group_first.p, b.cols, b.max_mn,
                       (uint32_t)b.uniform_mn, P.t, b.gemm_nkc, b.scal.p);
  } else if (b.fused_columns)
    hipLaunchKernelGGL(k_reduce_parts, dim3(cdiv(b.cols, BPP_REDUCE_TILE), b.G), dim3(64), 0, s, b.parts.p, b.rows.p, b.group_first.p, b.cols,
                       b.max_mn, P.t, b.lanes_ppw, b.scal.p);
  else
    hipLaunchKernelGGL(k_reduce_static, dim3(cdiv(b.cols, BPP_REDUCE_TILE), b.G), dim3(64), 0, s, b.rows.p, b.group_first.p, b.cols,
                       b.scal.p);
  tm.mark(M_REDUCE);
  PointTables tabs{P.table.p, b.dynpts.p, P.table_len, P.table_hi.p, b.dyn_hi.p};
  msm_run(ctx, b.msm, b.scal.p, tabs, &tm);
  HIP_CHECK(hipGetLastError());
}

void collect_profile(bpp_ctx *ctx, Batch &b, StageTimer &tm, float chain_ms, float total_host_ms) {
  if (!ctx->profile) return;
  bpp_profile &pf = ctx->prof;
  memset(&pf, 0, sizeof(pf));
  pf.transcripts_ms = tm.between(M_START, M_TRANSCRIPTS);
  pf.chain_device_ms = tm.between(M_TRANSCRIPTS, M_CHAIN);  // k_weight_chain + k_chain_finish (0 with the chains on the host)
  pf.decompress_ms = tm.between(tm.have[M_CHAIN] ? M_CHAIN : M_TRANSCRIPTS, M_DECOMPRESS);  // includes the 32 B/proof device-to-host copy
  pf.scalars_ms = tm.between(M_DECOMPRESS, M_SCALARS) + tm.between(M_WEIGHTS_IN, M_LANES);  // shared + lanes
  pf.chain_host_ms = chain_ms;
  pf.reduce_ms = tm.between(M_LANES, M_REDUCE);
  pf.msm_digits_ms = 0;                             // (digits are part of the sort kernel since round 3)
  pf.msm_sort_ms = tm.between(M_REDUCE, M_ORDER);   // k_msm_prelude: digits + counting sort + size ordering of the buckets
  pf.msm_accumulate_ms = tm.between(M_ORDER, M_ACC);
  pf.msm_bucket_reduce_ms = tm.between(M_ACC, M_BUCKET);
  pf.msm_final_ms = tm.between(M_BUCKET, M_FINAL);
  pf.masks_ms = tm.between(M_MASKS0, M_MASKS);
  pf.total_ms = total_host_ms;
  pf.msm_terms = b.msm.plan.n_terms;
  pf.msm_window_bits = b.msm.plan.c;
  pf.msm_windows = b.msm.plan.K;
  pf.msm_groups = b.msm.plan.G;
}

}  // namespace

extern "C" {

int bpp_batch_prepare(bpp_ctx *ctx, uint64_t batch, size_t chunk) {
  BPP_ENTRY(ctx);
  try {
    auto it = ctx->batches.find(batch);
    if (it == ctx->batches.end()) return fail(ctx, BPP_ERR_BAD_HANDLE, "unknown batch handle");
    StageTimer tm(ctx);  // creates the profiling events when profiling is on
    if (!ctx->ev_rng_ready) {
      HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_rng, hipEventDisableTiming));
      ctx->ev_rng_ready = true;
    }
    layout_groups(ctx, *it->second, chunk);  // group layout, MSM plan and every work buffer of that plan
    it->second->h_ident.resize(it->second->G);
    return BPP_OK;
  }
  BPP_CATCH(ctx, nullptr, 0)
}

// proofs of a resident batch (0: unknown handle), read under the context's lock: the gate is always taken BEFORE that lock
static uint32_t batch_size_peek(bpp_ctx *ctx, uint64_t batch) {
  std::lock_guard<std::mutex> lk(ctx->mu);
  auto it = ctx->batches.find(batch);
  return it == ctx->batches.end() ? 0u : it->second->B;
}

// (the context's lock is held.)  BPP_REDRAW_ON_HOST: the device chain drew a zero weight (probability 2^-252 per proof): the
// caller runs the call again with the chains on the host, which redraw as the reference does (scalar_protocol.rs:23-30)
#define BPP_REDRAW_ON_HOST (-0x7fff0001)
static int verify_resident_locked(bpp_ctx *ctx, uint64_t batch, int action, size_t chunk, uint8_t *masks_out, uint8_t *mask_present,
                                  char *errbuf, size_t errbuf_len, bool allow_dev_chain) {
  try {
    auto it = ctx->batches.find(batch);
    if (it == ctx->batches.end()) return fail(ctx, BPP_ERR_BAD_HANDLE, "unknown batch handle", errbuf, errbuf_len);
    if (action < 0 || action > 2) return fail(ctx, BPP_ERR_INVALID_ARGUMENT, "unknown verify action", errbuf, errbuf_len);
    Batch &b = *it->second;
    Params &P = *b.params;
    auto t_begin = std::chrono::steady_clock::now();
    StageTimer tm(ctx);
    hipStream_t s = ctx->stream;
    layout_groups(ctx, b, chunk);
    // verify()'s own consistency loops (:637-682) come before anything else of the call: with one group nothing needs to run
    if (b.any_defer && b.G == 1) check_deferred(b.defer, 0, b.B);
    // a proof whose L/R count does not fit its statement makes the call fail (src/range_proof.rs:875-888).  With a
    // single group only the precedence against PASS-1 / decompression errors is still open, so PASS 2 is skipped;
    // with several groups the earlier groups' MSM verdicts still matter (the kernels tolerate the odd shapes).
    const bool pass1_only = b.any_rounds_bad && b.G == 1;
    const bool want_msm = !pass1_only && action != BPP_RECOVER_ONLY;
    const ChainMode cmode = want_msm ? chain_mode(ctx, b, allow_dev_chain) : CHAIN_HOST;
    const bool dev_chain = cmode == CHAIN_DEVICE;
    enqueue_phase1(ctx, b, tm, pass1_only || action == BPP_RECOVER_ONLY, nullptr, 0, 0, dev_chain);

    float chain_ms = 0;
    // recovered masks on their way to the caller (page-locked, written by k_results_out): wiped on every exit (src/extended_mask.rs:14)
    bool have_masks = false;
    ScopeExit wipe_masks{[&] {
      if (have_masks) wipe(b.h_masks.data(), std::min(b.h_masks.n, (size_t)b.B * P.t * 32));
    }};
    b.h_ident.resize(b.G);
    for (uint32_t g = 0; g < b.G; g++) b.h_ident[g] = 1;
    auto &h_ident = b.h_ident;
    const uint8_t *h_masks = nullptr;
    if (!pass1_only) {
      // weight chains: one per chunk (src/range_proof.rs:811,849,853,894); the device keeps working meanwhile
      auto c0 = std::chrono::steady_clock::now();
      if (want_msm && cmode == CHAIN_HOST) run_weight_chains(b);
      if (want_msm && cmode == CHAIN_HOST_WIDE) run_weight_chains_wide(b);
      chain_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - c0).count();
      if (action != BPP_VERIFY_ONLY && b.any_seed) {  // masks (:941-969)
        tm.mark(M_MASKS0);
        hipLaunchKernelGGL(k_masks, dim3(cdiv(b.B, 64)), dim3(64), 0, s, b.bytes.p, b.d_desc.p, b.chal.p, b.seeds.p,
                           P.n_bits, P.t, b.cs, b.B, b.masks.p);
        tm.mark(M_MASKS);
        b.masks_dirty = true;
        have_masks = true;
      }
      if (want_msm) {
        enqueue_phase2(ctx, b, tm, cmode != CHAIN_HOST, cmode == CHAIN_HOST_WIDE);
        b.have_trace = true;
      }
    }
    fetch_results(ctx, b, want_msm, have_masks);
    gpu_wait_stream(ctx, s, wait_naps(ctx, b.B), &ctx->wait_hint_end, ((uint64_t)b.B << 32) | ((uint64_t)b.G << 2) | (uint64_t)action);
    if (want_msm && cmode != CHAIN_HOST && ctx->h_chain_zero[0]) {
      ctx->device_chain_redraws++;
      return BPP_REDRAW_ON_HOST;
    }
    if (have_masks) h_masks = b.h_masks.data();
    auto t_end = std::chrono::steady_clock::now();
    collect_profile(ctx, b, tm, chain_ms, std::chrono::duration<float, std::milli>(t_end - t_begin).count());

    // errors surface chunk by chunk, in the reference's order; a chunk's MSM verdict precedes later chunks' errors
    for (uint32_t g = 0; g < b.G; g++) {
      if (b.any_defer) check_deferred(b.defer, b.h_group_first[g], b.h_group_first[g + 1]);  // :637-682
      check_chunk_errors(b, b.h_group_first[g], b.h_group_first[g + 1]);
      if (want_msm && !h_ident[g]) throw ProofErr{BPP_ERR_VERIFICATION_FAILED, "Range proof batch not valid", BPP_TIER_MSM, b.h_group_first[g]};
    }
    // outputs: Vec<Option<ExtendedMask>>
    for (uint32_t p = 0; p < b.B; p++) {
      bool present = (action != BPP_VERIFY_ONLY) && (b.desc[p].flags & 1u);
      if (mask_present) mask_present[p] = present ? 1 : 0;
      if (masks_out) {
        if (present)
          memcpy(masks_out + (size_t)p * P.t * 32, &h_masks[(size_t)p * P.t * 32], (size_t)P.t * 32);
        else
          memset(masks_out + (size_t)p * P.t * 32, 0, (size_t)P.t * 32);
      }
    }
    return BPP_OK;
  }
  BPP_CATCH(ctx, errbuf, errbuf_len)
}

int bpp_verify_resident(bpp_ctx *ctx, uint64_t batch, int action, size_t chunk, uint8_t *masks_out, uint8_t *mask_present,
                        char *errbuf, size_t errbuf_len) {
  if (!ctx) return BPP_ERR_BAD_HANDLE;
  const uint32_t n_peek = batch_size_peek(ctx, batch);
  GateHold gate(ctx->device, n_peek && n_peek <= BPP_GATE_SMALL_PROOFS);  // small calls queue for the device (DeviceState)
  BPP_ENTRY(ctx);
  int rc = verify_resident_locked(ctx, batch, action, chunk, masks_out, mask_present, errbuf, errbuf_len, true);
  if (rc == BPP_REDRAW_ON_HOST) rc = verify_resident_locked(ctx, batch, action, chunk, masks_out, mask_present, errbuf, errbuf_len, false);
  return rc;
}

// Reference batches of DIFFERENT sizes in one call: group g = proofs [group_first[g], group_first[g + 1]) of the resident
// batch, every group verified as its own verify() call by the same kernel launches, with its own outcome and its own
// VerifyAction (src/range_proof.rs:46-54) -- where bpp_verify_resident cuts equal chunks and stops at the first failing one.
// What a pool of small calls needs (bpp_batcher below).  actions == nullptr: VerifyOnly for every group.
// Masks (:941-969): a group whose action recovers them and whose verify() would have returned Ok gets the masks of its items
// that carry a seed nonce; every other item's slot is zero / absent (an Err returns no masks).  A RecoverOnly group never
// looks at the final check (:1040-1043); when EVERY group is RecoverOnly the weight chains and PASS 2 are not run at all.
}  // extern "C"
namespace {
int verify_groups_core_once(bpp_ctx *ctx, uint64_t batch, const uint32_t *group_first, size_t n_groups, const int *actions,
                            bpp_shard_result *results, uint8_t *masks_out, uint8_t *mask_present, bool allow_dev_chain) {
  try {
    auto it = ctx->batches.find(batch);
    if (it == ctx->batches.end()) return fail(ctx, BPP_ERR_BAD_HANDLE, "unknown batch handle");
    if (!group_first || !results || n_groups == 0) return fail(ctx, BPP_ERR_INVALID_ARGUMENT, "null argument");
    Batch &b = *it->second;
    Params &P = *b.params;
    std::vector<uint32_t> bounds(group_first, group_first + n_groups + 1);
    if (bounds.front() != 0 || bounds.back() != b.B) return fail(ctx, BPP_ERR_INVALID_ARGUMENT, "group boundaries must run from 0 to the batch size");
    for (size_t g = 0; g < n_groups; g++)
      if (bounds[g] >= bounds[g + 1]) return fail(ctx, BPP_ERR_INVALID_ARGUMENT, "empty or unordered group");
    bool want_msm = false, want_masks = false;
    for (size_t g = 0; g < n_groups; g++) {
      const int a = actions ? actions[g] : BPP_VERIFY_ONLY;
      if (a < 0 || a > 2) return fail(ctx, BPP_ERR_INVALID_ARGUMENT, "unknown verify action");
      want_msm = want_msm || a != BPP_RECOVER_ONLY;
      want_masks = want_masks || a != BPP_VERIFY_ONLY;
    }
    want_masks = want_masks && b.any_seed;
    auto t_begin = std::chrono::steady_clock::now();
    StageTimer tm(ctx);
    hipStream_t s = ctx->stream;
    layout_groups(ctx, b, 0, &bounds);
    // the kernels tolerate odd shapes and run on every item (as bpp_verify_resident does with several chunks); findings are
    // raised per group afterwards, in the reference's order
    const ChainMode cmode = want_msm ? chain_mode(ctx, b, allow_dev_chain) : CHAIN_HOST;
    const bool dev_chain = cmode == CHAIN_DEVICE;
    enqueue_phase1(ctx, b, tm, !want_msm, nullptr, 0, 0, dev_chain);
    b.h_ident.resize(b.G);
    for (uint32_t g = 0; g < b.G; g++) b.h_ident[g] = 1;
    ScopeExit wipe_masks{[&] {
      if (want_masks) wipe(b.h_masks.data(), std::min(b.h_masks.n, (size_t)b.B * P.t * 32));
    }};
    float chain_ms = 0;
    if (want_msm && !dev_chain) {
      auto c0 = std::chrono::steady_clock::now();
      if (cmode == CHAIN_HOST_WIDE) run_weight_chains_wide(b);
      else run_weight_chains(b);
      chain_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - c0).count();
    }
    if (want_masks) {
      hipLaunchKernelGGL(k_masks, dim3(cdiv(b.B, 64)), dim3(64), 0, s, b.bytes.p, b.d_desc.p, b.chal.p, b.seeds.p, P.n_bits, P.t, b.cs, b.B,
                         b.masks.p);
      b.masks_dirty = true;
    }
    if (want_msm) {
      enqueue_phase2(ctx, b, tm, cmode != CHAIN_HOST, cmode == CHAIN_HOST_WIDE);
      b.have_trace = true;
    }
    fetch_results(ctx, b, want_msm, want_masks);
    gpu_wait_stream(ctx, s, wait_naps(ctx, b.B), &ctx->wait_hint_end, ((uint64_t)b.B << 32) | ((uint64_t)b.G << 2) | (uint64_t)((want_msm ? 1 : 0) | (want_masks ? 2 : 0)));
    if (want_msm && cmode != CHAIN_HOST && ctx->h_chain_zero[0]) {
      ctx->device_chain_redraws++;
      return BPP_REDRAW_ON_HOST;
    }
    collect_profile(ctx, b, tm, chain_ms, std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
    for (uint32_t g = 0; g < b.G; g++) {
      bpp_shard_result &r = results[g];
      memset(&r, 0, sizeof(r));
      r.rank = -1;
      const int a = actions ? actions[g] : BPP_VERIFY_ONLY;
      const uint32_t p0 = b.h_group_first[g], p1 = b.h_group_first[g + 1];
      try {
        if (b.any_defer) check_deferred(b.defer, p0, p1);
        check_chunk_errors(b, p0, p1);
        if (a != BPP_RECOVER_ONLY && !b.h_ident[g]) throw ProofErr{BPP_ERR_VERIFICATION_FAILED, "Range proof batch not valid", BPP_TIER_MSM, p0};
      } catch (const ProofErr &e) {
        r.code = e.code;
        r.tier = e.tier;
        r.index = e.index - p0;  // position inside the group's own batch
        snprintf(r.msg, sizeof(r.msg), "%s", e.msg.c_str());
      }
      const bool give = a != BPP_VERIFY_ONLY && r.code == BPP_OK;
      for (uint32_t p = p0; p < p1; p++) {
        const bool present = give && (b.desc[p].flags & 1u);
        if (mask_present) mask_present[p] = present ? 1 : 0;
        if (masks_out) {
          if (present) memcpy(masks_out + (size_t)p * P.t * 32, b.h_masks.data() + (size_t)p * P.t * 32, (size_t)P.t * 32);
          else memset(masks_out + (size_t)p * P.t * 32, 0, (size_t)P.t * 32);
        }
      }
    }
    return BPP_OK;
  }
  BPP_CATCH(ctx, nullptr, 0)
}
int verify_groups_core(bpp_ctx *ctx, uint64_t batch, const uint32_t *group_first, size_t n_groups, const int *actions,
                       bpp_shard_result *results, uint8_t *masks_out, uint8_t *mask_present) {
  int rc = verify_groups_core_once(ctx, batch, group_first, n_groups, actions, results, masks_out, mask_present, true);
  if (rc == BPP_REDRAW_ON_HOST) rc = verify_groups_core_once(ctx, batch, group_first, n_groups, actions, results, masks_out, mask_present, false);
  return rc;
}
}  // namespace
extern "C" {

int bpp_verify_resident_groups_actions(bpp_ctx *ctx, uint64_t batch, const uint32_t *group_first, size_t n_groups, const int *actions,
                                       bpp_shard_result *results, uint8_t *masks_out, uint8_t *mask_present) {
  if (!ctx) return BPP_ERR_BAD_HANDLE;
  const uint32_t n_peek = batch_size_peek(ctx, batch);
  GateHold gate(ctx->device, n_peek && n_peek <= BPP_GATE_SMALL_PROOFS);  // small calls queue for the device (DeviceState)
  BPP_ENTRY(ctx);
  return verify_groups_core(ctx, batch, group_first, n_groups, actions, results, masks_out, mask_present);
}

int bpp_verify_resident_groups(bpp_ctx *ctx, uint64_t batch, const uint32_t *group_first, size_t n_groups, bpp_shard_result *results) {
  return bpp_verify_resident_groups_actions(ctx, batch, group_first, n_groups, nullptr, results, nullptr, nullptr);
}

int bpp_verify_batch_with_challenges(bpp_ctx *ctx, uint64_t params, const bpp_verify_item *items, size_t n_items,
                                     const uint8_t *const *challenges32, const uint8_t *rng_out32, int action, size_t chunk,
                                     uint8_t *masks_out, uint8_t *mask_present, char *errbuf, size_t errbuf_len) {
  if (!ctx) return BPP_ERR_BAD_HANDLE;
  GateHold gate(ctx->device, n_items <= BPP_GATE_SMALL_PROOFS);  // held across upload and verification of a small call
  uint64_t h = 0;
  int rc;
  {
    BPP_ENTRY(ctx);
    if (!challenges32 || !rng_out32) return fail(ctx, BPP_ERR_INVALID_ARGUMENT, "null argument", errbuf, errbuf_len);
    rc = upload_impl(ctx, params, items, n_items, nullptr, &h, challenges32, rng_out32, errbuf, errbuf_len);
  }
  if (rc != BPP_OK) return rc;
  rc = bpp_verify_resident(ctx, h, action, chunk, masks_out, mask_present, errbuf, errbuf_len);
  (void)bpp_batch_destroy(ctx, h);
  return rc;
}

int bpp_verify_batch(bpp_ctx *ctx, uint64_t params, const bpp_verify_item *items, size_t n_items, int action,
                     size_t chunk, uint8_t *masks_out, uint8_t *mask_present, char *errbuf, size_t errbuf_len) {
  if (!ctx) return BPP_ERR_BAD_HANDLE;
  GateHold gate(ctx->device, n_items <= BPP_GATE_SMALL_PROOFS);
  uint64_t h = 0;
  int rc = bpp_batch_upload(ctx, params, items, n_items, &h, errbuf, errbuf_len);
  if (rc != BPP_OK) return rc;
  rc = bpp_verify_resident(ctx, h, action, chunk, masks_out, mask_present, errbuf, errbuf_len);
  (void)bpp_batch_destroy(ctx, h);
  return rc;
}

int bpp_verify_batch_packed(bpp_ctx *ctx, uint64_t params, const bpp_packed_batch *in, int action, size_t chunk,
                            uint8_t *masks_out, uint8_t *mask_present, char *errbuf, size_t errbuf_len) {
  if (!ctx) return BPP_ERR_BAD_HANDLE;
  GateHold gate(ctx->device, in && in->n_items <= BPP_GATE_SMALL_PROOFS);
  uint64_t h = 0;
  int rc = bpp_batch_upload_packed(ctx, params, in, &h, errbuf, errbuf_len);
  if (rc != BPP_OK) return rc;
  rc = bpp_verify_resident(ctx, h, action, chunk, masks_out, mask_present, errbuf, errbuf_len);
  (void)bpp_batch_destroy(ctx, h);
  return rc;
}

}  // extern "C"

// ---------------------------------------------------------------- pipelined host-buffers-in form
// One context, `depth` lanes.  A lane is a private child context (own stream, page-locked staging, recycled work buffers)
// plus a worker thread.  submit packs the caller's buffers into the next lane's staging ON THE CALLING THREAD -- the lanes
// of earlier tickets are busy with DMA, kernels and weight chains meanwhile -- and hands the plan to the lane's worker,
// which runs the device half of the upload, the verification and the release exactly as bpp_verify_batch_packed does.
namespace {

void pipe_worker(bpp_ctx *owner, Pipeline *pp, PipeLane *lane) {
  (void)hipSetDevice(owner->device);
  for (;;) {
    std::shared_ptr<PipeJob> job;
    {
      std::unique_lock<std::mutex> lk(pp->mu);
      pp->cv.wait(lk, [&] { return pp->quit || lane->job; });
      if (!lane->job) return;  // quit with nothing posted
      job = std::move(lane->job);
      lane->job.reset();
    }
    bpp_ctx *c = lane->child;
    char err[256];
    err[0] = 0;
    uint64_t h = 0;
    int rc = BPP_OK;
    GateHold gate(owner->device, job->n_items <= BPP_GATE_SMALL_PROOFS);
    {
      std::lock_guard<std::mutex> lk(c->mu);
      try {
        h = upload_device(c, job->Pp, job->params, job->pl, nullptr, nullptr);
      } catch (const EngineError &e) {
        rc = fail(c, e.code, e.msg, err, sizeof(err));
      } catch (const ProofErr &e) {
        rc = fail(c, e.code, e.msg, err, sizeof(err));
      } catch (const std::exception &e) {
        rc = fail(c, BPP_ERR_ENGINE, e.what(), err, sizeof(err));
      }
    }
    if (rc == BPP_OK) {
      const bool want_masks = job->action != BPP_VERIFY_ONLY;
      if (want_masks) {
        job->masks.assign(job->n_items * job->t * 32, 0);
        job->present.assign(job->n_items, 0);
      }
      rc = bpp_verify_resident(c, h, job->action, job->chunk, want_masks ? job->masks.data() : nullptr,
                               want_masks ? job->present.data() : nullptr, err, sizeof(err));
      (void)bpp_batch_destroy(c, h);
    }
    {
      std::lock_guard<std::mutex> lk(pp->mu);
      job->rc = rc;
      job->err = err;
      job->done = true;
      lane->busy = false;
    }
    pp->cv.notify_all();
  }
}

Pipeline *pipeline_get(bpp_ctx *ctx) {
  std::lock_guard<std::mutex> lk(ctx->pipe_init_mu);
  if (ctx->pipe) return ctx->pipe.get();
  auto pp = std::make_unique<Pipeline>();
  for (uint32_t i = 0; i < ctx->pipe_depth; i++) {
    auto lane = std::make_unique<PipeLane>();
    if (bpp_ctx_create(&lane->child, ctx->device) != BPP_OK) throw EngineError{BPP_ERR_ENGINE, "pipeline lane: context creation failed"};
    lane->child->opt = ctx->opt;
    pp->lanes.push_back(std::move(lane));
  }
  for (auto &lane : pp->lanes) lane->th = std::thread(pipe_worker, ctx, pp.get(), lane.get());
  ctx->pipe = std::move(pp);
  return ctx->pipe.get();
}

}  // namespace

void pipeline_shutdown(bpp_ctx *ctx) {
  std::unique_ptr<Pipeline> pp;
  {
    std::lock_guard<std::mutex> lk(ctx->pipe_init_mu);
    pp = std::move(ctx->pipe);
  }
  if (!pp) return;
  {
    std::unique_lock<std::mutex> lk(pp->mu);
    pp->cv.wait(lk, [&] {  // calls in flight finish first (their results are simply dropped)
      for (auto &l : pp->lanes)
        if (l->busy) return false;
      return true;
    });
    pp->quit = true;
  }
  pp->cv.notify_all();
  for (auto &l : pp->lanes) {
    if (l->th.joinable()) l->th.join();
    bpp_ctx_destroy(l->child);
  }
  for (auto &kv : pp->tickets) wipe(kv.second->masks.data(), kv.second->masks.size());
}

extern "C" {

int bpp_ctx_pipeline_depth(bpp_ctx *ctx, uint32_t depth) {
  if (!ctx) return BPP_ERR_BAD_HANDLE;
  std::lock_guard<std::mutex> lk(ctx->pipe_init_mu);
  if (ctx->pipe) return fail(ctx, BPP_ERR_INVALID_ARGUMENT, "the pipeline is already running");
  if (depth < 1 || depth > 16) return fail(ctx, BPP_ERR_INVALID_ARGUMENT, "pipeline depth must be 1..16");
  ctx->pipe_depth = depth;
  return BPP_OK;
}

int bpp_verify_submit_packed(bpp_ctx *ctx, uint64_t params, const bpp_packed_batch *in, int action, size_t chunk,
                             uint64_t *ticket, char *errbuf, size_t errbuf_len) {
  if (!ctx) return BPP_ERR_BAD_HANDLE;
  if (hipSetDevice(ctx->device) != hipSuccess) return BPP_ERR_NO_DEVICE;
  try {
    if (!in || !ticket || in->n_items == 0)
      return fail(nullptr, BPP_ERR_INVALID_ARGUMENT, "Range statements or proofs length empty", errbuf, errbuf_len);
    if (action < 0 || action > 2) return fail(nullptr, BPP_ERR_INVALID_ARGUMENT, "unknown verify action", errbuf, errbuf_len);
    if (in->n_items > (1u << 24)) return fail(nullptr, BPP_ERR_SIZE_OVERFLOW, "batch too large", errbuf, errbuf_len);
    const std::shared_ptr<Params> Pp = params_registry().get(params);
    if (!Pp || Pp->device != ctx->device) return fail(nullptr, BPP_ERR_BAD_HANDLE, "unknown params handle", errbuf, errbuf_len);
    Pipeline *pp = pipeline_get(ctx);
    std::lock_guard<std::mutex> submit_lock(pp->submit_mu);
    PipeLane *lane;
    {
      std::unique_lock<std::mutex> lk(pp->mu);
      lane = pp->lanes[pp->next_lane].get();
      pp->cv.wait(lk, [&] { return !lane->busy; });  // its previous call is done: staging and work buffers are free
      lane->busy = true;
      pp->next_lane = (pp->next_lane + 1) % (uint32_t)pp->lanes.size();
    }
    auto job = std::make_shared<PipeJob>();
    job->action = action;
    job->chunk = chunk;
    job->Pp = Pp;
    job->params = params;
    job->n_items = in->n_items;
    job->t = Pp->t;
    try {
      upload_host_pack(lane->child, *Pp, nullptr, 0, in, job->pl);  // into the lane's staging; throws construction errors
    } catch (...) {
      secure_wipe(job->pl.seeds.data(), job->pl.seeds.size());
      {
        std::lock_guard<std::mutex> lk(pp->mu);
        lane->busy = false;
      }
      pp->cv.notify_all();
      throw;
    }
    {
      std::lock_guard<std::mutex> lk(pp->mu);
      job->ticket = pp->next_ticket++;
      pp->tickets[job->ticket] = job;
      lane->job = job;
      *ticket = job->ticket;
    }
    pp->cv.notify_all();
    return BPP_OK;
  }
  BPP_CATCH(nullptr, errbuf, errbuf_len)
}

int bpp_verify_collect(bpp_ctx *ctx, uint64_t ticket, uint8_t *masks_out, uint8_t *mask_present, char *errbuf,
                       size_t errbuf_len) {
  if (!ctx) return BPP_ERR_BAD_HANDLE;
  Pipeline *pp;
  {
    std::lock_guard<std::mutex> lk(ctx->pipe_init_mu);
    pp = ctx->pipe.get();
  }
  if (!pp) return fail(nullptr, BPP_ERR_BAD_HANDLE, "unknown ticket", errbuf, errbuf_len);
  std::shared_ptr<PipeJob> job;
  {
    std::unique_lock<std::mutex> lk(pp->mu);
    auto it = pp->tickets.find(ticket);
    if (it == pp->tickets.end()) return fail(nullptr, BPP_ERR_BAD_HANDLE, "unknown ticket", errbuf, errbuf_len);
    job = it->second;
    pp->cv.wait(lk, [&] { return job->done; });
    pp->tickets.erase(it);
  }
  ScopeExit wipe_masks{[&] { wipe(job->masks.data(), job->masks.size()); }};
  if (job->rc != BPP_OK) {
    set_err(errbuf, errbuf_len, job->err);
    return job->rc;
  }
  const size_t n = job->n_items, t = job->t;
  const bool have = job->action != BPP_VERIFY_ONLY;
  if (mask_present) {
    if (have) memcpy(mask_present, job->present.data(), n);
    else memset(mask_present, 0, n);
  }
  if (masks_out) {
    if (have) memcpy(masks_out, job->masks.data(), n * t * 32);
    else memset(masks_out, 0, n * t * 32);
  }
  return BPP_OK;
}

// ---------------------------------------------------------------- phased form
int bpp_verify_phase1(bpp_ctx *ctx, uint64_t batch, uint8_t *rng_out32, char *errbuf, size_t errbuf_len) {
  BPP_ENTRY(ctx);
  try {
    auto it = ctx->batches.find(batch);
    if (it == ctx->batches.end()) return fail(ctx, BPP_ERR_BAD_HANDLE, "unknown batch handle", errbuf, errbuf_len);
    Batch &b = *it->second;
    StageTimer tm(ctx);
    layout_groups(ctx, b, 0);
    if (b.any_defer) check_deferred(b.defer, 0, b.B);
    enqueue_phase1(ctx, b, tm, b.any_rounds_bad);
    fetch_status(ctx, b);
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (rng_out32) memcpy(rng_out32, b.h_rng.data(), (size_t)b.B * 32);
    check_chunk_errors(b, 0, b.B);
    b.phase1_done = true;
    return BPP_OK;
  }
  BPP_CATCH(ctx, errbuf, errbuf_len)
}

int bpp_verify_phase2(bpp_ctx *ctx, uint64_t batch, const uint8_t *weights32, uint8_t accumulator128[128], char *errbuf,
                      size_t errbuf_len) {
  BPP_ENTRY(ctx);
  try {
    auto it = ctx->batches.find(batch);
    if (it == ctx->batches.end()) return fail(ctx, BPP_ERR_BAD_HANDLE, "unknown batch handle", errbuf, errbuf_len);
    if (!weights32 || !accumulator128) return fail(ctx, BPP_ERR_INVALID_ARGUMENT, "null argument", errbuf, errbuf_len);
    Batch &b = *it->second;
    if (!b.phase1_done) return fail(ctx, BPP_ERR_INVALID_ARGUMENT, "bpp_verify_phase1 must succeed first", errbuf, errbuf_len);
    for (uint32_t p = 0; p < b.B; p++)
      if (!sc_is_canonical(weights32 + 32 * (size_t)p))
        return fail(ctx, BPP_ERR_INVALID_ARGUMENT, "weight is not canonical", errbuf, errbuf_len);
    StageTimer tm(ctx);
    hipStream_t s = ctx->stream;
    layout_groups(ctx, b, 0);
    memcpy(b.h_weights.data(), weights32, (size_t)b.B * 32);
    enqueue_phase2(ctx, b, tm);
    if (!ctx->scratch128.p) ctx->scratch128.alloc(128);
    hipLaunchKernelGGL(k_ge_to_bytes, dim3(1), dim3(64), 0, s, b.msm.R.p, 1u, ctx->scratch128.p);
    HIP_CHECK(hipMemcpyAsync(accumulator128, ctx->scratch128.p, 128, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    b.have_trace = true;
    return BPP_OK;
  }
  BPP_CATCH(ctx, errbuf, errbuf_len)
}

int bpp_accumulators_sum_is_identity(bpp_ctx *ctx, const uint8_t *accumulators128, size_t n, int *is_identity) {
  BPP_ENTRY(ctx);
  try {
    if (!accumulators128 || !is_identity || n == 0) return fail(ctx, BPP_ERR_INVALID_ARGUMENT, "null argument");
    DevBuf<uint8_t> d_in, d_comp;
    DevBuf<uint32_t> d_flag;
    d_in.alloc(n * 128);
    d_comp.alloc(32);
    d_flag.alloc(1);
    HIP_CHECK(hipMemcpyAsync(d_in.p, accumulators128, n * 128, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_sum_accumulators, dim3(1), dim3(64), 0, ctx->stream, d_in.p, (uint32_t)n, d_comp.p, d_flag.p);
    uint32_t flag = 0;
    HIP_CHECK(hipMemcpyAsync(&flag, d_flag.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    *is_identity = flag ? 1 : 0;
    return BPP_OK;
  }
  BPP_CATCH(ctx, nullptr, 0)
}

// ---------------------------------------------------------------- trace / shape / profile
int bpp_batch_shape(bpp_ctx *ctx, uint64_t batch, uint32_t *n_items, uint32_t *max_rounds, uint32_t *max_mn,
                    uint32_t *total_dyn, uint32_t *groups) {
  BPP_ENTRY(ctx);
  auto it = ctx->batches.find(batch);
  if (it == ctx->batches.end()) return fail(ctx, BPP_ERR_BAD_HANDLE, "unknown batch handle");
  Batch &b = *it->second;
  if (n_items) *n_items = b.B;
  if (max_rounds) *max_rounds = b.cs - 3;  // rounds the challenge trace has slots for (capped at BPP_MAX_ROUNDS - 1)
  if (max_mn) *max_mn = b.max_mn;
  if (total_dyn) *total_dyn = b.total_dyn;
  if (groups) *groups = b.G;
  return BPP_OK;
}

int bpp_batch_trace(bpp_ctx *ctx, uint64_t batch, int what, uint8_t *out, size_t out_len, size_t *written) {
  BPP_ENTRY(ctx);
  try {
    auto it = ctx->batches.find(batch);
    if (it == ctx->batches.end()) return fail(ctx, BPP_ERR_BAD_HANDLE, "unknown batch handle");
    Batch &b = *it->second;
    hipStream_t s = ctx->stream;
    size_t need = 0;
    const void *src = nullptr;
    DevBuf<uint8_t> tmp;
    switch (what) {
      case BPP_TRACE_CHALLENGES: {
        const uint32_t n = b.B * b.cs;
        need = (size_t)n * 32;
        tmp.alloc(need);
        hipLaunchKernelGGL(k_chal_canonical, dim3(cdiv(n, 64)), dim3(64), 0, s, b.chal.p, n, tmp.p);
        src = tmp.p;
        break;
      }
      case BPP_TRACE_RNG_OUT:
        need = (size_t)b.B * 32;
        src = b.rng_out.p;
        break;
      case BPP_TRACE_WEIGHTS:
        need = (size_t)b.B * 32;
        if (b.weights_on_host) {  // PASS 2 read them from the mapped host buffer the chains wrote
          if (written) *written = need;
          if (!out || out_len < need) return fail(ctx, BPP_ERR_INVALID_LENGTH, "trace buffer too small");
          HIP_CHECK(hipStreamSynchronize(s));
          memcpy(out, b.h_weights.data(), need);
          return BPP_OK;
        }
        src = b.weights.p;
        break;
      case BPP_TRACE_STATIC_SCALARS:
        need = (size_t)b.G * b.cols * 32;
        src = b.scal.p;
        break;
      case BPP_TRACE_DYNAMIC_SCALARS:
        need = (size_t)b.total_dyn * 32;
        src = b.scal.p + (size_t)b.G * b.cols;
        break;
      case BPP_TRACE_MSM_RESULT:  // encoded on demand: verification itself only tests for the identity
        need = (size_t)b.G * 32;
        hipLaunchKernelGGL(k_compress_ge, dim3(cdiv(b.G, 64)), dim3(64), 0, s, b.msm.R.p, b.G, b.msm.comp32.p);
        src = b.msm.comp32.p;
        break;
      case BPP_TRACE_PLAN: {  // host-side: which form of the scalar stage the last layout / verification chose
        const uint32_t w[4] = {(b.fused_columns ? 1u : 0u) | (b.static_gemm ? 2u : 0u), b.lanes_ppw, b.gemm_nkc, b.G};
        if (written) *written = sizeof(w);
        if (!out || out_len < sizeof(w)) return fail(ctx, BPP_ERR_INVALID_LENGTH, "trace buffer too small");
        memcpy(out, w, sizeof(w));
        return BPP_OK;
      }
      default:
        return fail(ctx, BPP_ERR_INVALID_ARGUMENT, "unknown trace selector");
    }
    if (written) *written = need;
    if (!out || out_len < need) return fail(ctx, BPP_ERR_INVALID_LENGTH, "trace buffer too small");
    HIP_CHECK(hipMemcpyAsync(out, src, need, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    return BPP_OK;
  }
  BPP_CATCH(ctx, nullptr, 0)
}

int bpp_batch_secret_bytes(bpp_ctx *ctx, uint64_t batch, uint64_t *nonzero) {
  BPP_ENTRY(ctx);
  try {
    if (!nonzero) return fail(ctx, BPP_ERR_INVALID_ARGUMENT, "null argument");
    Batch *b = nullptr;
    if (batch == 0) {
      b = ctx->spare_batch.get();
    } else {
      auto it = ctx->batches.find(batch);
      if (it == ctx->batches.end()) return fail(ctx, BPP_ERR_BAD_HANDLE, "unknown batch handle");
      b = it->second.get();
    }
    uint64_t cnt = 0;
    if (b) {
      HIP_CHECK(hipStreamSynchronize(ctx->stream));
      for (DevBuf<uint8_t> *buf : {&b->seeds, &b->masks}) {
        if (!buf->p || !buf->n) continue;
        std::vector<uint8_t> h(buf->n);
        HIP_CHECK(hipMemcpy(h.data(), buf->p, buf->n, hipMemcpyDeviceToHost));
        for (uint8_t v : h) cnt += v != 0;
        wipe(h.data(), h.size());
      }
    }
    *nonzero = cnt;
    return BPP_OK;
  }
  BPP_CATCH(ctx, nullptr, 0)
}

int bpp_prove_secret_bytes(bpp_ctx *ctx, uint64_t *examined, uint64_t *nonzero) {
  BPP_ENTRY(ctx);
  try {
    if (!examined || !nonzero) return fail(ctx, BPP_ERR_INVALID_ARGUMENT, "null argument");
    uint64_t seen = 0, cnt = 0;
    if (ctx->prove_arena.p && ctx->prove_arena.n) {
      for (auto &ps : ctx->prove_streams) HIP_CHECK(hipStreamSynchronize(ps));
      std::vector<uint8_t> h(ctx->prove_arena.n);
      HIP_CHECK(hipMemcpy(h.data(), ctx->prove_arena.p, h.size(), hipMemcpyDeviceToHost));
      for (uint8_t v : h) cnt += v != 0;
      seen += h.size();
      wipe(h.data(), h.size());
    }
    // (the staging on the way out holds proofs and status words: nothing secret, not looked at)
    if (ctx->prove_pin_in.p && ctx->prove_pin_in.n) {
      for (size_t i = 0; i < ctx->prove_pin_in.n; i++) cnt += ctx->prove_pin_in.p[i] != 0;
      seen += ctx->prove_pin_in.n;
    }
    *examined = seen;
    *nonzero = cnt;
    return BPP_OK;
  }
  BPP_CATCH(ctx, nullptr, 0)
}

int bpp_profile_enable(bpp_ctx *ctx, int on) {
  if (!ctx) return BPP_ERR_BAD_HANDLE;
  ctx->profile = on != 0;
  ctx->profile_light = on == 2;
  return BPP_OK;
}

int bpp_prove_profile_get(bpp_ctx *ctx, bpp_prove_profile *out) {
  if (!ctx || !out) return BPP_ERR_BAD_HANDLE;
  *out = ctx->pprof;
  return BPP_OK;
}

int bpp_profile_get(bpp_ctx *ctx, bpp_profile *out) {
  if (!ctx || !out) return BPP_ERR_BAD_HANDLE;
  *out = ctx->prof;
  return BPP_OK;
}

}  // extern "C"

#include "engine_shard.h"
#include "engine_batcher.h"

#include "engine_prove.h"
