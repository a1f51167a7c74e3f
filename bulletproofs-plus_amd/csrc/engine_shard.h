// ONE reference batch sharded over the GPUs of a node, behind the C ABI (include/bpp.h: bpp_comm_*, bpp_verify_sharded,
// bpp_verify_sharded_wave: k batches on k contexts; bpp_verify_sharded_groups(_wave): a rank's shards of many batches as one
// resident batch -- one set of kernel launches --, k of those as a software pipeline of one host thread, the weight-chain
// replay shared out over the ranks).  Part of engine.hip's translation unit (it drives the same phase functions as
// bpp_verify_resident); the collectives are RCCL calls on device buffers, issued from here -- no framework in between.
//
// Reference coupling points (src/range_proof.rs): the batch weights come from ONE transcript over all proofs in order
// (:811,:849,:853,:894) and the final check is one group equation (:1050-1062).  SURVEY 8(e).
//
// RCCL is loaded lazily (dlopen) so that libbpp_hip.so itself does not depend on a 570 MB library its single-GPU callers
// never touch; the copy next to the HIP runtime in use is preferred (see rccl_api).
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

struct RcclApi {
  void *lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommAbort) CommAbort = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string err;
};

namespace {

RcclApi &rccl_api() {
  static RcclApi *api = [] {
    RcclApi *a = new RcclApi();
    // RCCL must sit on the SAME HIP / HSA runtime this library is running on: RCCL opens the HSA runtime by name on its own,
    // and a copy from another ROCm tree (say PyTorch's bundled librccl.so, already in the process, while this library was
    // loaded first and bound to /opt/rocm's libamdhip64) finds an HSA runtime nobody initialised: "no ROCm-capable device".
    // So the first candidates are the librccl files NEXT TO the libamdhip64 that hipGetDeviceCount resolves to; only then
    // the bare names (which return whatever copy the process already holds).
    // BPP_RCCL_LIB names THE library to use: when it cannot be loaded nothing else is tried (an explicit choice that silently
    // became another copy would be the mismatch above all over again) and every bpp_comm_* call returns BPP_ERR_COMM.
    std::vector<std::string> names;
    const char *forced = getenv("BPP_RCCL_LIB");
    if (forced && *forced) names.push_back(forced);
    Dl_info info;
    if (!(forced && *forced) && dladdr((void *)&hipGetDeviceCount, &info) && info.dli_fname) {
      std::string dir(info.dli_fname);
      const size_t slash = dir.rfind('/');
      if (slash != std::string::npos) {
        dir.resize(slash + 1);
        names.push_back(dir + "librccl.so.1");
        names.push_back(dir + "librccl.so");
      }
    }
    if (!(forced && *forced)) {
      names.push_back("librccl.so.1");
      names.push_back("librccl.so");
    }
    for (const std::string &n : names) {
      if (n.empty()) continue;
      a->lib = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
      if (a->lib) break;
    }
    if (!a->lib) {
      const char *why = dlerror();
      a->err = std::string("RCCL not loadable") + (forced && *forced ? std::string(" (BPP_RCCL_LIB=") + forced + ")" : std::string()) + ": " +
               (why ? why : "librccl.so.1 not found");
      return a;
    }
#define BPP_RCCL_SYM(field, name)                              \
  a->field = (decltype(a->field))dlsym(a->lib, name);          \
  if (!a->field && a->err.empty()) a->err = std::string("RCCL symbol missing: ") + name;
    BPP_RCCL_SYM(GetUniqueId, "ncclGetUniqueId");
    BPP_RCCL_SYM(CommInitRank, "ncclCommInitRank");
    BPP_RCCL_SYM(CommDestroy, "ncclCommDestroy");
    BPP_RCCL_SYM(CommAbort, "ncclCommAbort");
    BPP_RCCL_SYM(AllGather, "ncclAllGather");
    BPP_RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef BPP_RCCL_SYM
    return a;
  }();
  return *api;
}

struct CommError {
  std::string msg;
};
#define RCCL_CHECK(expr)                                                                            \
  do {                                                                                              \
    ncclResult_t _r = (expr);                                                                       \
    if (_r != ncclSuccess) {                                                                        \
      char _b[256];                                                                                 \
      snprintf(_b, sizeof(_b), "%s failed: %s", #expr, rccl_api().GetErrorString(_r));              \
      throw CommError{_b};                                                                          \
    }                                                                                               \
  } while (0)

// sum over ranks of batch i's accumulators -> identity flag.  in: [rank][per_rank bytes], batch i's 128 bytes at i * 128.
// One lane per batch (a handful of additions: the exchange is latency, not work).
__global__ void __launch_bounds__(64) k_sum_accumulators_wave(const uint8_t *__restrict__ in, uint32_t world, uint32_t per_rank, uint32_t k,
                                        uint32_t *__restrict__ is_identity) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= k) return;
  ge acc;
  ge_identity(acc);
  for (uint32_t r = 0; r < world; r++) {
    const uint8_t *src = in + (size_t)r * per_rank + (size_t)i * 128;
    ge p;
    uint8_t b[32];
    for (int c = 0; c < 4; c++) {
      for (int j = 0; j < 32; j++) b[j] = src[32 * c + j];
      fe_frombytes(c == 0 ? p.X : c == 1 ? p.Y : c == 2 ? p.Z : p.T, b);
    }
    ge_add(acc, acc, p);
  }
  is_identity[i] = ge_is_ristretto_identity(acc) ? 1u : 0u;
}

}  // namespace

// In-process stand-in for the communicator (bpp_comm_create_local): the "ranks" are threads of ONE process on ONE device,
// each with its own context; an all_gather is a rendezvous plus device-to-device copies.  It exists so that the sharded
// entry points can be driven with several ranks -- ragged shard sizes, findings on any rank, the slicing of the replayed
// weight chain -- on a box with a single GPU, through exactly the code the RCCL form runs (tests/test_gpu_round3.py).
struct LocalGroup {
  std::mutex mu;
  std::condition_variable cv;
  int world = 1, arrived = 0;
  uint64_t generation = 0;
  bool broken = false;  // a rank gave up waiting: every rendezvous of this group fails from then on (as an aborted ncclComm does)
  std::vector<const uint8_t *> send;
  // false: the other ranks did not all arrive within timeout_ms (0 = wait for ever), or the group is already broken
  bool barrier(uint32_t timeout_ms) {
    std::unique_lock<std::mutex> lk(mu);
    if (broken) return false;
    const uint64_t gen = generation;
    if (++arrived == world) {
      arrived = 0;
      generation++;
      cv.notify_all();
      return true;
    }
    auto ready = [&] { return generation != gen || broken; };
    if (timeout_ms == 0) cv.wait(lk, ready);
    else if (!cv.wait_for(lk, std::chrono::milliseconds(timeout_ms), ready)) {
      broken = true;
      cv.notify_all();
    }
    return !broken;
  }
};

struct bpp_comm {
  int device = 0, rank = 0, world = 1;
  std::shared_ptr<LocalGroup> local;  // set: in-process transport instead of RCCL
  bpp_all_gather_fn cb = nullptr;     // set: the caller's own transport (bpp_comm_create_callbacks), host buffers through cb_send / cb_recv
  void *cb_user = nullptr;
  PinnedBuf<uint8_t> cb_send, cb_recv;
  ncclComm_t comm = nullptr;
  bool own_comm = false;
  hipStream_t stream = nullptr;  // collectives and their staging copies
  hipEvent_t ev_wait = nullptr;  // comm_wait's marker (an EVENT is polled, not the stream: a stream query on a busy stream leaves a
                                 // helper thread of the runtime spinning, tools/microbench/wait_modes.hip)
  DevBuf<uint8_t> send1, recv1, send2, recv2;
  struct GroupSlot {  // exchange buffers of one batch of bpp_verify_sharded_groups_wave's pipeline
    DevBuf<uint8_t> send1, recv1, send2, recv2, send3, recv3;
    DevBuf<uint32_t> d_flags;
    PinnedBuf<uint8_t> h_tr, h_recv1, h_recv2;
    PinnedBuf<uint32_t> h_flags;
    std::vector<uint8_t> rng_all, weights_all;
  };
  std::vector<std::unique_ptr<GroupSlot>> slots;
  DevBuf<uint32_t> d_flags;
  PinnedBuf<uint8_t> h_tr, h_recv1, h_recv2;
  PinnedBuf<uint32_t> h_flags;
  std::vector<uint8_t> rng_all, weights_all;
  std::mutex mu;
  std::string err;
  bpp_shard_timing timing{};  // host wall-clock split of the last wave
  // Every wait for a collective has a deadline (bpp_comm_set_timeout; BPP_COMM_TIMEOUT_MS; default 60 s; 0 = none): a peer that
  // died or never called leaves an all_gather kernel spinning on this rank's stream for ever.  When the deadline passes the
  // communicator is marked dead and the call -- like every later call on it -- returns BPP_ERR_COMM: an RCCL failure maps to a C
  // error code on every surviving rank.  A communicator this library CREATED (bpp_comm_create) is aborted as well
  // (ncclCommAbort: its kernels exit).  One it merely ADOPTED (bpp_comm_adopt: say a framework's process-group communicator) is
  // NOT: the ncclComm_t belongs to the caller, who may be using it elsewhere and will destroy it himself -- an abort behind his
  // back would turn that into a use-after-free.  The owner aborts it (which also lets this rank's stream drain).
  uint32_t timeout_ms = 60000;
  bool dead = false;
  bool stream_stuck = false;  // an adopted communicator's collective is still spinning on `stream`: never wait for it unbounded
};

namespace {

int comm_fail(bpp_comm *c, int code, const std::string &m, char *errbuf = nullptr, size_t len = 0) {
  if (c) c->err = m;
  set_err(errbuf, len, m);
  return code;
}

// Waits for everything enqueued on the communicator's stream -- a collective and the copies around it -- with the
// communicator's deadline.  Polls (the wait must be interruptible: hipStreamSynchronize is not).
void comm_wait(bpp_comm *c, hipStream_t cs) {
  if (c->timeout_ms == 0) {  // no deadline asked for: the plain blocking wait
    HIP_CHECK(hipStreamSynchronize(cs));
    return;
  }
  const auto t0 = std::chrono::steady_clock::now();
  HIP_CHECK(hipEventRecord(c->ev_wait, cs));
  for (uint32_t spins = 0;; spins++) {
    const hipError_t q = hipEventQuery(c->ev_wait);
    if (q == hipSuccess) return;
    if (q != hipErrorNotReady) {
      (void)hipGetLastError();
      throw EngineError{BPP_ERR_ENGINE, std::string("the communicator's stream failed: ") + hipGetErrorString(q)};
    }
    const auto waited = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
    if (c->timeout_ms && waited >= (long long)c->timeout_ms) {
      c->dead = true;
      const bool adopted = !c->local && c->comm && !c->own_comm;
      if (!c->local && c->comm && c->own_comm) {
        (void)rccl_api().CommAbort(c->comm);  // the collective's kernel exits; the handle is gone with it
        c->comm = nullptr;
        c->own_comm = false;
      }
      // let the stream drain (bounded: nothing may hang here either).  An adopted communicator's kernel keeps spinning until
      // its owner aborts it: one short look, then the stream is remembered as stuck.
      const auto t1 = std::chrono::steady_clock::now();
      const auto patience = adopted ? std::chrono::milliseconds(50) : std::chrono::milliseconds(5000);
      while (hipStreamQuery(cs) == hipErrorNotReady && std::chrono::steady_clock::now() - t1 < patience)
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
      if (adopted && hipStreamQuery(cs) == hipErrorNotReady) c->stream_stuck = true;
      (void)hipGetLastError();
      throw CommError{"a collective did not complete within " + std::to_string(c->timeout_ms) + " ms (a peer is missing or dead); " +
                      (adopted ? "this handle is dead and must be destroyed; the adopted ncclComm_t was left alone: its owner aborts it"
                               : "the communicator was aborted and must be destroyed")};
    }
    if (spins < 2000) std::this_thread::yield();  // the exchanges are tens of microseconds when every rank is there
    else std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
}

int comm_new(bpp_ctx *ctx, ncclComm_t nc, bool own, int rank, int world, bpp_comm **out) {
  auto c = std::make_unique<bpp_comm>();
  // (a malformed value keeps the default and says so: `atoi` made a typo a silent "no deadline")
  if (const char *e = getenv("BPP_COMM_TIMEOUT_MS")) {
    long v = 0;
    if (parse_env_long(e, 0, 86400000L, &v)) c->timeout_ms = (uint32_t)v;
    else ctx->err = std::string("BPP_COMM_TIMEOUT_MS=\"") + e + "\" is not a number of milliseconds: the default deadline (60000 ms) stands";
  }
  c->device = ctx->device;
  c->rank = rank;
  c->world = world;
  c->comm = nc;
  c->own_comm = own;
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return BPP_ERR_ENGINE;
  if (hipEventCreateWithFlags(&c->ev_wait, hipEventDisableTiming) != hipSuccess) return BPP_ERR_ENGINE;
  *out = c.release();
  return BPP_OK;
}

// all_gather of `bytes` per rank on the communicator's stream: RCCL, or the in-process rendezvous
void comm_allgather(bpp_comm *c, const uint8_t *send, uint8_t *recv, size_t bytes, hipStream_t cs) {
  if (c->cb) {
    // the caller's transport moves HOST bytes (the exchanges are 32 bytes per proof and 256 bytes per batch: staging them is
    // nothing); it blocks until every rank's block is in `recv`, and its own deadline is its own business
    c->cb_send.resize(bytes);
    c->cb_recv.resize(bytes * (size_t)c->world);
    HIP_CHECK(hipMemcpyAsync(c->cb_send.p, send, bytes, hipMemcpyDeviceToHost, cs));
    HIP_CHECK(hipStreamSynchronize(cs));
    const int rc = c->cb(c->cb_user, c->cb_send.p, c->cb_recv.p, bytes);
    if (rc != 0) {
      c->dead = true;
      throw CommError{"the caller's all_gather callback failed (" + std::to_string(rc) + "); this communicator is dead and must be destroyed"};
    }
    HIP_CHECK(hipMemcpyAsync(recv, c->cb_recv.p, bytes * (size_t)c->world, hipMemcpyHostToDevice, cs));
    return;
  }
  if (!c->local) {
    RCCL_CHECK(rccl_api().AllGather(send, recv, bytes, ncclUint8, c->comm, cs));
    return;
  }
  // Whatever goes wrong on this rank between its first and its second rendezvous -- a deadline, a HIP error -- breaks the GROUP at
  // once (every peer's wait ends now, with BPP_ERR_COMM, instead of after its own full deadline), and this rank's copies out of the
  // peers' send buffers have completed or are waited for before it returns: the peers may release those buffers afterwards.  (A
  // peer that is still copying out of THIS rank's buffer when the buffer is released is covered by hipFree itself, which waits
  // for the device.)
  auto break_group = [&] {
    std::lock_guard<std::mutex> lk(c->local->mu);
    c->local->broken = true;
    c->local->cv.notify_all();
  };
  auto rendezvous = [&] {
    if (c->local->barrier(c->timeout_ms)) return;
    c->dead = true;
    (void)hipStreamSynchronize(cs);
    throw CommError{"a collective did not complete within " + std::to_string(c->timeout_ms) +
                    " ms (a peer is missing or dead); the communicator was aborted and must be destroyed"};
  };
  try {
    HIP_CHECK(hipStreamSynchronize(cs));  // everything this rank sends is in place
    {
      std::lock_guard<std::mutex> lk(c->local->mu);
      c->local->send[c->rank] = send;
    }
    rendezvous();
    for (int r = 0; r < c->world; r++)
      HIP_CHECK(hipMemcpyAsync(recv + (size_t)r * bytes, c->local->send[r], bytes, hipMemcpyDeviceToDevice, cs));
    HIP_CHECK(hipStreamSynchronize(cs));
    rendezvous();  // nobody reuses a send buffer before every rank has read it
  } catch (...) {
    c->dead = true;
    break_group();
    (void)hipStreamSynchronize(cs);
    (void)hipGetLastError();
    throw;
  }
}

std::mutex g_local_groups_mu;
std::map<uint64_t, std::weak_ptr<LocalGroup>> g_local_groups;

void shard_result_set(bpp_shard_result &r, int code, int tier, int rank, uint32_t index, const std::string &msg) {
  r.code = code;
  r.tier = tier;
  r.rank = rank;
  r.index = index;
  snprintf(r.msg, sizeof(r.msg), "%s", msg.c_str());
}

}  // namespace

extern "C" {

int bpp_comm_unique_id(uint8_t id128[128]) {
  if (!id128) return BPP_ERR_INVALID_ARGUMENT;
  RcclApi &R = rccl_api();
  if (!R.err.empty()) return BPP_ERR_COMM;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  if (R.GetUniqueId(&id) != ncclSuccess) return BPP_ERR_COMM;
  memcpy(id128, &id, 128);
  return BPP_OK;
}

int bpp_comm_create(bpp_ctx *ctx, const uint8_t id128[128], int rank, int world, bpp_comm **out) {
  BPP_ENTRY(ctx);
  if (!out || !id128 || world < 1 || rank < 0 || rank >= world) return fail(ctx, BPP_ERR_INVALID_ARGUMENT, "bad communicator arguments");
  *out = nullptr;
  RcclApi &R = rccl_api();
  if (!R.err.empty()) return fail(ctx, BPP_ERR_COMM, R.err);
  ncclUniqueId id;
  memcpy(&id, id128, 128);
  ncclComm_t nc = nullptr;
  const ncclResult_t r = R.CommInitRank(&nc, world, id, rank);  // collective over the ranks; binds to the current device
  if (r != ncclSuccess) return fail(ctx, BPP_ERR_COMM, std::string("ncclCommInitRank failed: ") + R.GetErrorString(r));
  const int rc = comm_new(ctx, nc, true, rank, world, out);
  if (rc != BPP_OK) (void)R.CommDestroy(nc);
  return rc;
}

int bpp_comm_adopt(bpp_ctx *ctx, void *nccl_comm, int rank, int world, bpp_comm **out) {
  BPP_ENTRY(ctx);
  if (!out || !nccl_comm || world < 1 || rank < 0 || rank >= world) return fail(ctx, BPP_ERR_INVALID_ARGUMENT, "bad communicator arguments");
  *out = nullptr;
  RcclApi &R = rccl_api();
  if (!R.err.empty()) return fail(ctx, BPP_ERR_COMM, R.err);
  return comm_new(ctx, (ncclComm_t)nccl_comm, false, rank, world, out);
}

int bpp_comm_create_callbacks(bpp_ctx *ctx, int rank, int world, bpp_all_gather_fn all_gather, void *user, bpp_comm **out) {
  BPP_ENTRY(ctx);
  if (!out || !all_gather || world < 1 || rank < 0 || rank >= world) return fail(ctx, BPP_ERR_INVALID_ARGUMENT, "bad communicator arguments");
  *out = nullptr;
  const int rc = comm_new(ctx, nullptr, false, rank, world, out);
  if (rc == BPP_OK) {
    (*out)->cb = all_gather;
    (*out)->cb_user = user;
  }
  return rc;
}

int bpp_comm_create_local(bpp_ctx *ctx, uint64_t group_id, int rank, int world, bpp_comm **out) {
  BPP_ENTRY(ctx);
  if (!out || world < 1 || rank < 0 || rank >= world) return fail(ctx, BPP_ERR_INVALID_ARGUMENT, "bad communicator arguments");
  *out = nullptr;
  std::shared_ptr<LocalGroup> g;
  {
    std::lock_guard<std::mutex> lk(g_local_groups_mu);
    g = g_local_groups[group_id].lock();
    if (!g) {
      g = std::make_shared<LocalGroup>();
      g->world = world;
      g->send.assign(world, nullptr);
      g_local_groups[group_id] = g;
    }
  }
  if (g->world != world) return fail(ctx, BPP_ERR_INVALID_ARGUMENT, "this group id exists with another world size");
  const int rc = comm_new(ctx, nullptr, false, rank, world, out);
  if (rc == BPP_OK) (*out)->local = g;
  return rc;
}

void bpp_comm_destroy(bpp_comm *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  {
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->stream_stuck) {
      // a timed-out collective of an ADOPTED communicator may still be spinning here (its owner has not aborted it yet): wait a
      // bounded time; if it is still there, the stream, the exchange buffers and the page-locked staging are deliberately LEFT
      // ALLOCATED (a few hundred KB) -- releasing memory a running kernel and the copies queued behind it still address would be
      // worse, and every release call of the runtime would wait for that kernel
      const auto t0 = std::chrono::steady_clock::now();
      while (hipStreamQuery(c->stream) == hipErrorNotReady && std::chrono::steady_clock::now() - t0 < std::chrono::seconds(2))
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
      const bool drained = hipStreamQuery(c->stream) != hipErrorNotReady;
      (void)hipGetLastError();
      if (!drained) return;  // (the bpp_comm object itself stays too: its destructors would free those buffers)
    } else {
      (void)hipStreamSynchronize(c->stream);
    }
    if (c->own_comm && c->comm) (void)rccl_api().CommDestroy(c->comm);
    (void)hipStreamDestroy(c->stream);
  }
  if (c->ev_wait) (void)hipEventDestroy(c->ev_wait);
  delete c;
}

const char *bpp_comm_last_error(bpp_comm *c) { return c ? c->err.c_str() : "null comm"; }

int bpp_comm_set_timeout(bpp_comm *c, uint32_t timeout_ms) {
  if (!c) return BPP_ERR_BAD_HANDLE;
  std::lock_guard<std::mutex> lk(c->mu);
  c->timeout_ms = timeout_ms;
  return BPP_OK;
}

int bpp_comm_last_timing(bpp_comm *c, bpp_shard_timing *out) {
  if (!c || !out) return BPP_ERR_BAD_HANDLE;
  std::lock_guard<std::mutex> lk(c->mu);
  *out = c->timing;
  return BPP_OK;
}

int bpp_shard_trailer(int tier, int code, uint32_t index, const char *msg, uint8_t trailer_out[BPP_SHARD_TRAILER_BYTES]) {
  if (!trailer_out || tier < 0 || tier > 255) return BPP_ERR_INVALID_ARGUMENT;
  shard_trailer_encode(trailer_out, tier, code, index, msg);
  return BPP_OK;
}

int bpp_shard_local_trailer(const uint8_t *defer, const uint32_t *status, const uint8_t *rounds_bad, uint32_t n,
                            uint32_t first_index, uint8_t trailer_out[BPP_SHARD_TRAILER_BYTES]) {
  if (!trailer_out || ((!status || !rounds_bad) && n)) return BPP_ERR_INVALID_ARGUMENT;
  shard_local_trailer(defer, status, rounds_bad, n, first_index, trailer_out);
  return BPP_OK;
}

int bpp_shard_resolve(const uint8_t *trailers, size_t stride, int world, int *tier_out, int *rank_out, uint32_t *index_out,
                      char *errbuf, size_t errbuf_len) {
  if (!trailers || world < 1 || stride < BPP_SHARD_TRAILER_BYTES) return BPP_ERR_INVALID_ARGUMENT;
  const ShardFinding f = shard_resolve(trailers, stride, world);
  if (tier_out) *tier_out = f.tier;
  if (rank_out) *rank_out = f.rank;
  if (index_out) *index_out = f.index;
  if (f.tier == BPP_TIER_NONE) return BPP_OK;
  set_err(errbuf, errbuf_len, f.msg + " (rank " + std::to_string(f.rank) + ")");
  return f.code;
}

int bpp_verify_sharded_wave(bpp_comm *comm, bpp_ctx *const *ctxs, const uint64_t *batches, size_t k_in, const uint32_t *counts,
                            bpp_shard_result *results) {
  if (!comm) return BPP_ERR_BAD_HANDLE;
  if (!ctxs || !batches || !counts || !results || k_in == 0 || k_in > 64) return comm_fail(comm, BPP_ERR_INVALID_ARGUMENT, "bad wave arguments");
  if (hipSetDevice(comm->device) != hipSuccess) return BPP_ERR_NO_DEVICE;
  const uint32_t K = (uint32_t)k_in, world = (uint32_t)comm->world, rank = (uint32_t)comm->rank;
  std::lock_guard<std::mutex> comm_lock(comm->mu);
  if (comm->dead) return comm_fail(comm, BPP_ERR_COMM, "this communicator was aborted after a collective timed out: destroy it");
  std::vector<std::unique_lock<std::mutex>> ctx_locks;
  for (uint32_t i = 0; i < K; i++) {
    if (!ctxs[i] || ctxs[i]->device != comm->device) return comm_fail(comm, BPP_ERR_BAD_HANDLE, "context of another device (or null) in the wave");
    for (uint32_t j = 0; j < i; j++)
      if (ctxs[j] == ctxs[i]) return comm_fail(comm, BPP_ERR_INVALID_ARGUMENT, "every batch of a wave needs its own context (stream)");
    ctx_locks.emplace_back(ctxs[i]->mu);
  }
  uint32_t maxc = 0, first_index = 0;
  uint64_t n_total = 0;
  for (uint32_t r = 0; r < world; r++) {
    maxc = std::max(maxc, counts[r]);
    if (r < rank) first_index += counts[r];
    n_total += counts[r];
  }
  if (n_total == 0 || n_total > (1u << 24)) return comm_fail(comm, BPP_ERR_INVALID_ARGUMENT, "Range statements or proofs length empty");
  // buffers: first exchange per rank = K slots of maxc x 32 RNG bytes; second = K accumulators + K findings (trailers)
  const size_t slot = (size_t)maxc * 32, per1 = K * slot, per2 = (size_t)K * 128 + (size_t)K * BPP_SHARD_TRAILER_BYTES;
  std::vector<Batch *> B(K, nullptr);
  std::vector<int> fault(K, 0);           // engine fault on THIS rank, per batch
  std::vector<std::string> fault_msg(K);
  std::vector<uint8_t> skip(K, 0);        // batch decided by the first exchange: no phase 2
  auto t_mark = std::chrono::steady_clock::now();
  bpp_shard_timing &tmg = comm->timing;
  memset(&tmg, 0, sizeof(tmg));
  tmg.batches = K;
  auto lap = [&](float &slot) {
    const auto now = std::chrono::steady_clock::now();
    slot += std::chrono::duration<float, std::milli>(now - t_mark).count();
    t_mark = now;
  };
  try {
    comm->send1.alloc(per1);
    comm->recv1.alloc(per1 * world);
    comm->send2.alloc(per2);
    comm->recv2.alloc(per2 * world);
    comm->d_flags.alloc(K);
    comm->h_tr.resize((size_t)K * BPP_SHARD_TRAILER_BYTES);
    comm->h_recv1.resize(per1 * world);
    comm->h_recv2.resize(per2 * world);
    comm->h_flags.resize(K);
    // ---------------------------------------------------------------- phase 1 on every context's own stream
    // Only the transcript-RNG bytes cross before the weights exist: they leave PASS 1, the first kernel of the phase, so the
    // exchange (and the weight chains behind it) overlap the decompression and the weight-free scalars of the same wave.
    // What a rank FOUND travels with the second exchange, next to its accumulator.
    hipStream_t cs = comm->stream;
    std::vector<uint8_t> no_kernels(K, 0);  // nothing runs on this batch's items here (deferred consistency finding)
    for (uint32_t i = 0; i < K; i++) {
      try {
        auto it = ctxs[i]->batches.find(batches[i]);
        if (it == ctxs[i]->batches.end()) throw EngineError{BPP_ERR_BAD_HANDLE, "unknown batch handle"};
        Batch &b = *it->second;
        if (b.B != counts[rank]) throw EngineError{BPP_ERR_ENGINE, "counts[rank] differs from the resident shard's size"};
        B[i] = &b;
        StageTimer tm(ctxs[i]);
        hipStream_t s = ctxs[i]->stream;
        uint8_t *dst = comm->send1.p + (size_t)i * slot;
        if (!ctxs[i]->ev_rng_ready) {
          HIP_CHECK(hipEventCreateWithFlags(&ctxs[i]->ev_rng, hipEventDisableTiming));
          ctxs[i]->ev_rng_ready = true;
        }
        if (b.any_defer) {
          // verify()'s consistency loops (:637-682) fail this batch before anything is computed: nothing runs on items
          // whose layout differs from the parameters'; the rank still sends a (zero) payload and, later, its finding
          no_kernels[i] = 1;
          HIP_CHECK(hipMemsetAsync(dst, 0, slot, s));
          HIP_CHECK(hipEventRecord(ctxs[i]->ev_rng, s));
        } else {
          layout_groups(ctxs[i], b, 0);
          if (b.B < maxc) HIP_CHECK(hipMemsetAsync(dst + (size_t)b.B * 32, 0, (size_t)(maxc - b.B) * 32, s));
          enqueue_phase1(ctxs[i], b, tm, b.any_rounds_bad, dst);
        }
        HIP_CHECK(hipStreamWaitEvent(cs, ctxs[i]->ev_rng, 0));
      } catch (const EngineError &e) {
        fault[i] = e.code;
        fault_msg[i] = e.msg;
        (void)hipMemsetAsync(comm->send1.p + (size_t)i * slot, 0, slot, cs);  // this rank's slot still has defined bytes
      }
    }
    lap(tmg.enqueue1_ms);
    comm_allgather(comm, comm->send1.p, comm->recv1.p, per1, cs);
    HIP_CHECK(hipMemcpyAsync(comm->h_recv1.data(), comm->recv1.p, per1 * world, hipMemcpyDeviceToHost, cs));
    comm_wait(comm, cs);
    lap(tmg.gather1_ms);
    // ---------------------------------------------------------------- weight transcripts over ALL proofs of each batch
    {
      comm->rng_all.resize((size_t)K * n_total * 32);
      comm->weights_all.resize((size_t)K * n_total * 32);
      std::vector<uint32_t> gfirst(K + 1);
      for (uint32_t i = 0; i < K; i++) {
        gfirst[i] = (uint32_t)(i * n_total);
        uint8_t *dst = comm->rng_all.data() + (size_t)i * n_total * 32;
        for (uint32_t r = 0; r < world; r++) {
          memcpy(dst, comm->h_recv1.data() + (size_t)r * per1 + (size_t)i * slot, (size_t)counts[r] * 32);
          dst += (size_t)counts[r] * 32;
        }
      }
      gfirst[K] = (uint32_t)(K * n_total);
      run_weight_chains_generic(comm->rng_all.data(), comm->weights_all.data(), gfirst.data(), K);
      lap(tmg.chains_ms);
      // PASS 2 + this rank's share of the MSM wherever the kernels ran and the shapes allow it (a batch with an L/R count
      // that does not fit its statement has a PASS-2 finding coming and no scalars to run on)
      for (uint32_t i = 0; i < K; i++) {
        if (fault[i]) continue;
        Batch &b = *B[i];
        try {
          StageTimer tm(ctxs[i]);
          hipStream_t s = ctxs[i]->stream;
          if (no_kernels[i] || b.any_rounds_bad) {
            skip[i] = 1;
            HIP_CHECK(hipMemsetAsync(comm->send2.p + (size_t)i * 128, 0, 128, s));
            if (no_kernels[i]) HIP_CHECK(hipMemsetAsync(b.status.p, 0, (size_t)b.B * 4, s));
          } else {
            memcpy(b.h_weights.data(), comm->weights_all.data() + ((size_t)i * n_total + first_index) * 32, (size_t)b.B * 32);
            enqueue_phase2(ctxs[i], b, tm);
            hipLaunchKernelGGL(k_ge_to_bytes, dim3(1), dim3(64), 0, s, b.msm.R.p, 1u, comm->send2.p + (size_t)i * 128);
            HIP_CHECK(hipGetLastError());
            b.have_trace = true;
          }
          fetch_status(ctxs[i], b);
        } catch (const EngineError &e) {
          fault[i] = e.code;
          fault_msg[i] = e.msg;
        }
      }
    }
    lap(tmg.enqueue2_ms);
    // ---------------------------------------------------------------- findings: one trailer per batch, next to the accumulator
    for (uint32_t i = 0; i < K; i++) {
      uint8_t *tr = comm->h_tr.data() + (size_t)i * BPP_SHARD_TRAILER_BYTES;
      if (!fault[i] && !gpu_wait_stream_ok(ctxs[i], ctxs[i]->stream, true)) {
        fault[i] = BPP_ERR_ENGINE;
        fault_msg[i] = "a kernel of this rank failed on the device";
      }
      if (fault[i]) {
        shard_trailer_encode(tr, BPP_TIER_ENGINE, fault[i], first_index, fault_msg[i].c_str());
      } else {
        Batch &b = *B[i];
        settle_status(b);
        shard_local_trailer(b.any_defer ? b.defer.data() : nullptr, b.h_status.data(), b.rounds_bad.data(), b.B, first_index, tr);
      }
    }
    lap(tmg.wait2_ms);
    HIP_CHECK(hipMemcpyAsync(comm->send2.p + (size_t)K * 128, comm->h_tr.data(), (size_t)K * BPP_SHARD_TRAILER_BYTES, hipMemcpyHostToDevice, cs));
    comm_allgather(comm, comm->send2.p, comm->recv2.p, per2, cs);
    hipLaunchKernelGGL(k_sum_accumulators_wave, dim3(cdiv(K, 64)), dim3(64), 0, cs, comm->recv2.p, world, (uint32_t)per2, K, comm->d_flags.p);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(comm->h_flags.data(), comm->d_flags.p, (size_t)K * 4, hipMemcpyDeviceToHost, cs));
    HIP_CHECK(hipMemcpyAsync(comm->h_recv2.data(), comm->recv2.p, per2 * world, hipMemcpyDeviceToHost, cs));
    comm_wait(comm, cs);
    lap(tmg.gather2_ms);
    // every rank reads the same findings and decides alike: a finding of any rank (lowest tier, then lowest rank) comes
    // before the final check, exactly as in the single-process verify(); an engine fault only counts when nothing was found
    for (uint32_t i = 0; i < K; i++) {
      const ShardFinding f = shard_resolve(comm->h_recv2.data() + (size_t)K * 128 + (size_t)i * BPP_SHARD_TRAILER_BYTES, per2, (int)world);
      if (f.tier != BPP_TIER_NONE)
        shard_result_set(results[i], f.code > 0 || f.code < 0 ? f.code : BPP_ERR_ENGINE, f.tier, f.rank, f.index, f.msg + " (rank " + std::to_string(f.rank) + ")");
      else if (!comm->h_flags[i])
        shard_result_set(results[i], BPP_ERR_VERIFICATION_FAILED, BPP_TIER_MSM, -1, 0, "Range proof batch not valid");
      else
        shard_result_set(results[i], BPP_OK, BPP_TIER_NONE, -1, 0, "");
    }
    return BPP_OK;
  } catch (const CommError &e) {
    return comm_fail(comm, BPP_ERR_COMM, e.msg);
  } catch (const EngineError &e) {
    // a HIP failure around the collectives themselves: this rank cannot promise to reach the next collective
    return comm_fail(comm, e.code, e.msg);
  } catch (const std::exception &e) {
    return comm_fail(comm, BPP_ERR_ENGINE, e.what());
  }
}

// The grouped form: this rank's shards of `n_groups` reference batches live in ONE resident batch (group g = proofs
// [g c, (g + 1) c) with c = counts[rank]) on ONE context, so every kernel of the verifier is launched once for all of them,
// as the chunked single-process form does (bpp_verify_resident with chunk = c), and the exchanges carry all groups.
// What a wave of k contexts pays per batch -- a dozen launches, a stream, a host thread's attention -- is paid once here;
// it is the form for many batches with small shards (eight ranks x 512 proofs of a 4096-proof batch).
//
// k such batches (each on its own context) run as a software pipeline of ONE host thread on ONE communicator:
//   phase 1 of every slot is enqueued; then, slot by slot: first exchange (waits for that slot's PASS 1 only), weight chains
//   (while the later slots' phase 1 and the earlier slots' phase 2 keep the GPU busy), phase 2 enqueued; then, slot by slot:
//   findings, second exchange, verdicts.
// The order of the collectives is the same on every rank by construction -- what several calls from several host threads
// on several communicators cannot promise.
int bpp_verify_sharded_groups_wave(bpp_comm *comm, bpp_ctx *const *ctxs, const uint64_t *batches, size_t k_in, size_t n_groups,
                                   const uint32_t *counts, bpp_shard_result *results) {
  if (!comm) return BPP_ERR_BAD_HANDLE;
  if (!ctxs || !batches || !counts || !results || k_in == 0 || k_in > 16 || n_groups == 0 || n_groups > 4096)
    return comm_fail(comm, BPP_ERR_INVALID_ARGUMENT, "bad group arguments");
  if (hipSetDevice(comm->device) != hipSuccess) return BPP_ERR_NO_DEVICE;
  const uint32_t K = (uint32_t)k_in, G = (uint32_t)n_groups, world = (uint32_t)comm->world, rank = (uint32_t)comm->rank;
  std::lock_guard<std::mutex> comm_lock(comm->mu);
  if (comm->dead) return comm_fail(comm, BPP_ERR_COMM, "this communicator was aborted after a collective timed out: destroy it");
  std::vector<std::unique_lock<std::mutex>> ctx_locks;
  for (uint32_t i = 0; i < K; i++) {
    if (!ctxs[i] || ctxs[i]->device != comm->device) return comm_fail(comm, BPP_ERR_BAD_HANDLE, "context of another device (or null)");
    for (uint32_t j = 0; j < i; j++)
      if (ctxs[j] == ctxs[i]) return comm_fail(comm, BPP_ERR_INVALID_ARGUMENT, "every batch of a wave needs its own context (stream)");
    ctx_locks.emplace_back(ctxs[i]->mu);
  }
  uint32_t maxc = 0, first_index = 0;
  uint64_t n_total = 0;
  for (uint32_t r = 0; r < world; r++) {
    maxc = std::max(maxc, counts[r]);
    if (r < rank) first_index += counts[r];
    n_total += counts[r];
  }
  const uint32_t c = counts[rank];
  if (n_total == 0 || n_total > (1u << 24)) return comm_fail(comm, BPP_ERR_INVALID_ARGUMENT, "Range statements or proofs length empty");
  if (c == 0) return comm_fail(comm, BPP_ERR_INVALID_ARGUMENT, "the grouped form needs a non-empty shard on every rank");
  const size_t slot = (size_t)maxc * 32, per1 = G * slot, per2 = (size_t)G * 128 + (size_t)G * BPP_SHARD_TRAILER_BYTES;
  // Weight transcripts over ALL proofs of each reference batch.  One rank replays all of them.  Several ranks share them out:
  // rank r replays the chains of groups r, r + world, ... and a third all_gather hands every rank every group's weights
  // (32 B per proof: 8 MB for 64 batches of 4096) -- the replay is a sequential sponge per batch on a host core, and with
  // every rank replaying every chain it, not the GPUs, bounded the rate (64 chains: 3.4 - 4.7 ms per call on sixteen
  // workers against 3 ms of kernels for a rank's 64 shards of 512 proofs).
  const bool share_chains = world > 1;
  const uint32_t n_own = share_chains ? (G > rank ? (G - rank + world - 1) / world : 0u) : G;
  const uint32_t slots3 = cdiv(G, world);
  const size_t per3 = (size_t)slots3 * n_total * 32;
  std::vector<int> fault(K, 0);  // engine fault on THIS rank: it still reaches every collective, with zero payloads and an ENGINE finding
  std::vector<std::string> fault_msg(K);
  std::vector<Batch *> B(K, nullptr);
  auto t_mark = std::chrono::steady_clock::now();
  bpp_shard_timing &tmg = comm->timing;
  memset(&tmg, 0, sizeof(tmg));
  tmg.batches = K * G;
  auto lap = [&](float &slot_ms) {
    const auto now = std::chrono::steady_clock::now();
    slot_ms += std::chrono::duration<float, std::milli>(now - t_mark).count();
    t_mark = now;
  };
  try {
    while (comm->slots.size() < K) comm->slots.emplace_back(new bpp_comm::GroupSlot());
    hipStream_t cs = comm->stream;
    for (uint32_t i = 0; i < K; i++) {
      bpp_comm::GroupSlot &S = *comm->slots[i];
      S.send1.alloc(per1);
      S.recv1.alloc(per1 * world);
      S.send2.alloc(per2);
      S.recv2.alloc(per2 * world);
      S.d_flags.alloc(G);
      S.h_tr.resize((size_t)G * BPP_SHARD_TRAILER_BYTES);
      S.h_recv1.resize(per1 * world);
      S.h_recv2.resize(per2 * world);
      S.h_flags.resize(G);
      if (share_chains) {
        S.send3.alloc(per3);
        S.recv3.alloc(per3 * world);
      }
    }
    // ---------------------------------------------------------------- phase 1 of every slot
    for (uint32_t i = 0; i < K; i++) {
      bpp_comm::GroupSlot &S = *comm->slots[i];
      bpp_ctx *ctx = ctxs[i];
      hipStream_t s = ctx->stream;
      try {
        auto it = ctx->batches.find(batches[i]);
        if (it == ctx->batches.end()) throw EngineError{BPP_ERR_BAD_HANDLE, "unknown batch handle"};
        Batch &b = *it->second;
        if ((uint64_t)b.B != (uint64_t)G * c) throw EngineError{BPP_ERR_ENGINE, "the resident batch does not hold n_groups x counts[rank] proofs"};
        B[i] = &b;
        StageTimer tm(ctx);
        if (!ctx->ev_rng_ready) {
          HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_rng, hipEventDisableTiming));
          ctx->ev_rng_ready = true;
        }
        layout_groups(ctx, b, G == 1 ? 0 : c);
        if (b.G != G) throw EngineError{BPP_ERR_ENGINE, "group layout differs from n_groups"};
        if (c < maxc) HIP_CHECK(hipMemsetAsync(S.send1.p, 0, per1, s));
        // several groups: the kernels tolerate odd shapes and run on everything (as bpp_verify_resident does with chunks), the
        // findings are raised per group afterwards; one group: the wave form's rules (nothing runs on a deferred finding)
        if (G == 1 && b.any_defer) {
          HIP_CHECK(hipMemsetAsync(S.send1.p, 0, per1, s));
          HIP_CHECK(hipEventRecord(ctx->ev_rng, s));
        } else {
          enqueue_phase1(ctx, b, tm, G == 1 && b.any_rounds_bad, S.send1.p, (size_t)c * 32, slot);
        }
        HIP_CHECK(hipStreamWaitEvent(cs, ctx->ev_rng, 0));
      } catch (const EngineError &e) {
        fault[i] = e.code;
        fault_msg[i] = e.msg;
        (void)hipMemsetAsync(S.send1.p, 0, per1, cs);
      }
    }
    lap(tmg.enqueue1_ms);
    // ---------------------------------------------------------------- per slot: first exchange, chains, phase 2
    for (uint32_t i = 0; i < K; i++) {
      bpp_comm::GroupSlot &S = *comm->slots[i];
      bpp_ctx *ctx = ctxs[i];
      hipStream_t s = ctx->stream;
      comm_allgather(comm, S.send1.p, S.recv1.p, per1, cs);
      HIP_CHECK(hipMemcpyAsync(S.h_recv1.data(), S.recv1.p, per1 * world, hipMemcpyDeviceToHost, cs));
      comm_wait(comm, cs);
      lap(tmg.gather1_ms);
      S.rng_all.resize((size_t)std::max(n_own, 1u) * n_total * 32);
      S.weights_all.resize((size_t)std::max(n_own, 1u) * n_total * 32);
      {
        std::vector<uint32_t> gfirst(n_own + 1);
        for (uint32_t j = 0; j < n_own; j++) {
          const uint32_t g = share_chains ? rank + j * world : j;
          gfirst[j] = (uint32_t)(j * n_total);
          uint8_t *dst = S.rng_all.data() + (size_t)j * n_total * 32;
          for (uint32_t r = 0; r < world; r++) {
            memcpy(dst, S.h_recv1.data() + (size_t)r * per1 + (size_t)g * slot, (size_t)counts[r] * 32);
            dst += (size_t)counts[r] * 32;
          }
        }
        gfirst[n_own] = (uint32_t)(n_own * n_total);
        if (n_own) run_weight_chains_generic(S.rng_all.data(), S.weights_all.data(), gfirst.data(), n_own);
      }
      lap(tmg.chains_ms);
      if (share_chains) {
        if (n_own < slots3) HIP_CHECK(hipMemsetAsync(S.send3.p, 0, per3, cs));
        if (n_own) HIP_CHECK(hipMemcpyAsync(S.send3.p, S.weights_all.data(), (size_t)n_own * n_total * 32, hipMemcpyHostToDevice, cs));
        comm_allgather(comm, S.send3.p, S.recv3.p, per3, cs);
        comm_wait(comm, cs);
        lap(tmg.gather1_ms);  // (counted with the first exchange: the timing struct is part of the ABI)
      }
      if (!fault[i]) {
        Batch &b = *B[i];
        try {
          StageTimer tm(ctx);
          if (G == 1 && (b.any_defer || b.any_rounds_bad)) {
            HIP_CHECK(hipMemsetAsync(S.send2.p, 0, 128, s));
            if (b.any_defer) HIP_CHECK(hipMemsetAsync(b.status.p, 0, (size_t)b.B * 4, s));
          } else {
            if (!share_chains) {
              for (uint32_t g = 0; g < G; g++)
                memcpy(b.h_weights.data() + (size_t)g * c * 32, S.weights_all.data() + ((size_t)g * n_total + first_index) * 32, (size_t)c * 32);
              enqueue_phase2(ctx, b, tm);
            } else {
              // this rank's slice of every group's weights, device -> device: the groups replayed by rank o sit n_total * 32
              // bytes apart in o's part of recv3 and `world` groups apart in the batch
              for (uint32_t o = 0; o < world && o < G; o++)
                HIP_CHECK(hipMemcpy2DAsync(b.weights.p + (size_t)o * c * 32, (size_t)world * c * 32,
                                           S.recv3.p + (size_t)o * per3 + (size_t)first_index * 32, (size_t)n_total * 32, (size_t)c * 32,
                                           (G - o + world - 1) / world, hipMemcpyDeviceToDevice, s));
              enqueue_phase2(ctx, b, tm, true);
            }
            hipLaunchKernelGGL(k_ge_to_bytes, dim3(cdiv(G, 64)), dim3(64), 0, s, b.msm.R.p, G, S.send2.p);
            HIP_CHECK(hipGetLastError());
            b.have_trace = true;
          }
          fetch_status(ctx, b);
        } catch (const EngineError &e) {
          fault[i] = e.code;
          fault_msg[i] = e.msg;
        }
      }
      lap(tmg.enqueue2_ms);
    }
    // ---------------------------------------------------------------- per slot: findings, second exchange, verdicts
    for (uint32_t i = 0; i < K; i++) {
      bpp_comm::GroupSlot &S = *comm->slots[i];
      bpp_ctx *ctx = ctxs[i];
      if (!fault[i] && !gpu_wait_stream_ok(ctx, ctx->stream, true)) {
        fault[i] = BPP_ERR_ENGINE;
        fault_msg[i] = "a kernel of this rank failed on the device";
      }
      if (fault[i]) (void)hipMemsetAsync(S.send2.p, 0, (size_t)G * 128, cs);
      else settle_status(*B[i]);
      for (uint32_t g = 0; g < G; g++) {
        uint8_t *tr = S.h_tr.data() + (size_t)g * BPP_SHARD_TRAILER_BYTES;
        if (fault[i]) {
          shard_trailer_encode(tr, BPP_TIER_ENGINE, fault[i], first_index, fault_msg[i].c_str());
        } else {
          Batch &b = *B[i];
          const size_t o = (size_t)g * c;
          shard_local_trailer(b.any_defer ? b.defer.data() + o : nullptr, b.h_status.data() + o, b.rounds_bad.data() + o, c, first_index, tr);
        }
      }
      lap(tmg.wait2_ms);
      HIP_CHECK(hipMemcpyAsync(S.send2.p + (size_t)G * 128, S.h_tr.data(), (size_t)G * BPP_SHARD_TRAILER_BYTES, hipMemcpyHostToDevice, cs));
      comm_allgather(comm, S.send2.p, S.recv2.p, per2, cs);
      hipLaunchKernelGGL(k_sum_accumulators_wave, dim3(cdiv(G, 64)), dim3(64), 0, cs, S.recv2.p, world, (uint32_t)per2, G, S.d_flags.p);
      HIP_CHECK(hipGetLastError());
      HIP_CHECK(hipMemcpyAsync(S.h_flags.data(), S.d_flags.p, (size_t)G * 4, hipMemcpyDeviceToHost, cs));
      HIP_CHECK(hipMemcpyAsync(S.h_recv2.data(), S.recv2.p, per2 * world, hipMemcpyDeviceToHost, cs));
      comm_wait(comm, cs);
      lap(tmg.gather2_ms);
      for (uint32_t g = 0; g < G; g++) {
        bpp_shard_result &out = results[(size_t)i * G + g];
        const ShardFinding f = shard_resolve(S.h_recv2.data() + (size_t)G * 128 + (size_t)g * BPP_SHARD_TRAILER_BYTES, per2, (int)world);
        if (f.tier != BPP_TIER_NONE)
          shard_result_set(out, f.code > 0 || f.code < 0 ? f.code : BPP_ERR_ENGINE, f.tier, f.rank, f.index, f.msg + " (rank " + std::to_string(f.rank) + ")");
        else if (!S.h_flags[g])
          shard_result_set(out, BPP_ERR_VERIFICATION_FAILED, BPP_TIER_MSM, -1, 0, "Range proof batch not valid");
        else
          shard_result_set(out, BPP_OK, BPP_TIER_NONE, -1, 0, "");
      }
    }
    return BPP_OK;
  } catch (const CommError &e) {
    return comm_fail(comm, BPP_ERR_COMM, e.msg);
  } catch (const EngineError &e) {
    return comm_fail(comm, e.code, e.msg);
  } catch (const std::exception &e) {
    return comm_fail(comm, BPP_ERR_ENGINE, e.what());
  }
}

int bpp_verify_sharded_groups(bpp_comm *comm, bpp_ctx *ctx, uint64_t batch, size_t n_groups, const uint32_t *counts,
                              bpp_shard_result *results) {
  if (!ctx) return comm ? comm_fail(comm, BPP_ERR_INVALID_ARGUMENT, "bad group arguments") : BPP_ERR_BAD_HANDLE;
  bpp_ctx *ctxs[1] = {ctx};
  return bpp_verify_sharded_groups_wave(comm, ctxs, &batch, 1, n_groups, counts, results);
}

int bpp_verify_sharded(bpp_comm *comm, bpp_ctx *ctx, uint64_t batch, const uint32_t *counts, int *tier_out, int *rank_out,
                       char *errbuf, size_t errbuf_len) {
  bpp_shard_result r;
  memset(&r, 0, sizeof(r));
  bpp_ctx *ctxs[1] = {ctx};
  const int rc = bpp_verify_sharded_wave(comm, ctxs, &batch, 1, counts, &r);
  if (rc != BPP_OK) {
    set_err(errbuf, errbuf_len, comm ? comm->err : "null comm");
    return rc;
  }
  if (tier_out) *tier_out = r.tier;
  if (rank_out) *rank_out = r.rank;
  if (r.code != BPP_OK) set_err(errbuf, errbuf_len, r.msg);
  return r.code;
}

}  // extern "C"
