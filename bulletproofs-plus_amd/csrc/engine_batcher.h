// bpp_batcher: pools the small verify_batch calls of many host threads into grouped engine calls.
//
// Why: a small call (one reference batch of up to a few hundred proofs) is a chain of about fifteen latency-bound kernels
// with tiny grids; the chip runs about six such kernels side by side whatever the number of callers, which caps independent
// 256-proof calls at ~5 300 per second (1.4 M proofs/s), while the same batches as the groups of ONE call run
// at 24 M proofs/s.  The batcher gives separate callers the second form: every caller hands over its own reference batch
// (bpp_batcher_verify blocks until its verdict is there); whichever caller finds a lane free becomes the leader of the next
// pooled call, takes everything that queued up while the previous pooled calls were running (plus what arrives within
// max_wait_us, if anything was asked for), concatenates it into one packed batch, uploads it, verifies it with one group per
// caller (bpp_verify_resident_groups: ragged groups, one outcome each) and hands the outcomes back.  No thread of its own.
// Each caller gets exactly what bpp_verify_batch_packed(ctx, params, its input, BPP_VERIFY_ONLY, 0) would have returned:
// src/range_proof.rs:712-752 on its own statements and proofs.  Part of engine.hip's translation unit.
#pragma once

struct bpp_batcher {
  struct Req {
    const bpp_packed_batch *in = nullptr;
    int code = BPP_OK;
    std::string msg;
    bool taken = false;  // a leader has it in its pooled call
    bool done = false;
  };
  struct Lane {
    bpp_ctx *ctx = nullptr;  // a context of its own (stream, staging, recycled work buffers)
    bool own = false;
    bool busy = false;
    std::vector<uint8_t> proofs, commitments, min_present;
    std::vector<uint64_t> min_values;
    std::vector<uint32_t> bounds;
    std::vector<bpp_shard_result> results;
  };
  uint64_t params = 0;
  uint32_t max_wait_us = 0, max_calls = 64, max_proofs = 16384;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<Req *> pending;
  std::vector<Lane> lanes;
  // what makes inputs poolable: one packed batch has one proof length, one aggregation factor, one transcript label
  size_t key_proof_len = 0;
  uint32_t key_m = 0;
  std::string key_label;
  uint64_t pooled_calls = 0, engine_calls = 0, solo_calls = 0;  // statistics
};

namespace {

bool batcher_poolable(const bpp_batcher *b, const bpp_packed_batch *in) {
  return in->transcript_state == nullptr && in->proof_len == b->key_proof_len && in->m == b->key_m && in->label_len == b->key_label.size() &&
         memcmp(in->transcript_label, b->key_label.data(), in->label_len) == 0 && in->n_items <= b->max_proofs;
}

void batcher_run_unchecked(bpp_batcher *b, bpp_batcher::Lane &L, const std::vector<bpp_batcher::Req *> &reqs);
// one pooled engine call on `lane` over `reqs`; fills every request's code / msg (nothing may escape: callers are waiting)
void batcher_run(bpp_batcher *b, bpp_batcher::Lane &L, const std::vector<bpp_batcher::Req *> &reqs) {
  try {
    batcher_run_unchecked(b, L, reqs);
  } catch (const std::exception &e) {
    for (auto *r : reqs) {
      r->code = BPP_ERR_ENGINE;
      r->msg = std::string("batcher: ") + e.what();
    }
  } catch (...) {
    for (auto *r : reqs) {
      r->code = BPP_ERR_ENGINE;
      r->msg = "batcher: unexpected failure";
    }
  }
}
void batcher_run_unchecked(bpp_batcher *b, bpp_batcher::Lane &L, const std::vector<bpp_batcher::Req *> &reqs) {
  char err[256];
  auto solo = [&](bpp_batcher::Req *r) {
    err[0] = 0;
    r->code = bpp_verify_batch_packed(L.ctx, b->params, r->in, BPP_VERIFY_ONLY, 0, nullptr, nullptr, err, sizeof(err));
    r->msg = err;
  };
  if (reqs.size() == 1) {
    solo(reqs[0]);
    return;
  }
  const size_t plen = b->key_proof_len, m = b->key_m;
  size_t n = 0;
  for (auto *r : reqs) n += r->in->n_items;
  L.proofs.resize(n * plen);
  L.commitments.resize(n * m * 32);
  L.min_values.resize(n * m);
  L.min_present.resize(n * m);
  L.bounds.resize(reqs.size() + 1);
  size_t at = 0;
  for (size_t g = 0; g < reqs.size(); g++) {
    const bpp_packed_batch &in = *reqs[g]->in;
    L.bounds[g] = (uint32_t)at;
    if (in.proof_stride == plen) memcpy(&L.proofs[at * plen], in.proofs, in.n_items * plen);
    else
      for (size_t i = 0; i < in.n_items; i++) memcpy(&L.proofs[(at + i) * plen], in.proofs + i * in.proof_stride, plen);
    memcpy(&L.commitments[at * m * 32], in.commitments32, in.n_items * m * 32);
    if (in.min_values) memcpy(&L.min_values[at * m], in.min_values, in.n_items * m * 8);
    else memset(&L.min_values[at * m], 0, in.n_items * m * 8);
    if (in.min_present && in.min_values) memcpy(&L.min_present[at * m], in.min_present, in.n_items * m);
    else memset(&L.min_present[at * m], 0, in.n_items * m);
    at += in.n_items;
  }
  L.bounds[reqs.size()] = (uint32_t)n;
  bpp_packed_batch merged;
  memset(&merged, 0, sizeof(merged));
  merged.n_items = n;
  merged.proofs = L.proofs.data();
  merged.proof_len = merged.proof_stride = plen;
  merged.commitments32 = L.commitments.data();
  merged.m = (uint32_t)m;
  merged.min_values = L.min_values.data();
  merged.min_present = L.min_present.data();
  merged.transcript_label = (const uint8_t *)b->key_label.data();
  merged.label_len = b->key_label.size();
  uint64_t h = 0;
  err[0] = 0;
  int rc = bpp_batch_upload_packed(L.ctx, b->params, &merged, &h, err, sizeof(err));
  if (rc == BPP_OK) {
    L.results.resize(reqs.size());
    rc = bpp_verify_resident_groups(L.ctx, h, L.bounds.data(), reqs.size(), L.results.data());
    (void)bpp_batch_destroy(L.ctx, h);
    if (rc == BPP_OK) {
      for (size_t g = 0; g < reqs.size(); g++) {
        reqs[g]->code = L.results[g].code;
        reqs[g]->msg = L.results[g].msg;
      }
      return;
    }
  }
  // A construction-time finding (a proof that could not have been deserialised, a statement that could not have been built)
  // belongs to ONE caller, and an engine fault should not be pinned on all of them: everybody gets a call of its own
  for (auto *r : reqs) solo(r);
}

}  // namespace

extern "C" {

int bpp_batcher_create(bpp_ctx *ctx, uint64_t params, const bpp_packed_batch *shape, uint32_t lanes, uint32_t max_wait_us, uint32_t max_calls,
                       bpp_batcher **out) {
  if (!ctx || !out || !shape || !shape->transcript_label || shape->proof_len == 0 || shape->m == 0) return BPP_ERR_INVALID_ARGUMENT;
  if (lanes == 0) lanes = 2;  // (two pooled calls in flight keep the pools large; three are within run-to-run noise of two, four and six lose)
  if (lanes > 8) lanes = 8;
  auto b = std::make_unique<bpp_batcher>();
  b->params = params;
  b->max_wait_us = max_wait_us;
  if (max_calls) b->max_calls = max_calls;
  b->key_proof_len = shape->proof_len;
  b->key_m = shape->m;
  b->key_label.assign((const char *)shape->transcript_label, shape->label_len);
  b->lanes.resize(lanes);
  for (uint32_t i = 0; i < lanes; i++) {
    if (i == 0) {
      b->lanes[i].ctx = ctx;
    } else {
      bpp_ctx *c = nullptr;
      int rc = bpp_ctx_create(&c, ctx->device);
      if (rc == BPP_OK) rc = bpp_params_retain(c, params);
      if (rc != BPP_OK) {
        if (c) bpp_ctx_destroy(c);
        for (uint32_t j = 1; j < i; j++) bpp_ctx_destroy(b->lanes[j].ctx);
        return rc;
      }
      b->lanes[i].ctx = c;
      b->lanes[i].own = true;
    }
  }
  *out = b.release();
  return BPP_OK;
}

void bpp_batcher_destroy(bpp_batcher *b) {
  if (!b) return;
  {
    std::unique_lock<std::mutex> lk(b->mu);
    b->cv.wait(lk, [&] {
      for (auto &L : b->lanes)
        if (L.busy) return false;
      return b->pending.empty();
    });
  }
  for (auto &L : b->lanes)
    if (L.own) bpp_ctx_destroy(L.ctx);
  delete b;
}

int bpp_batcher_stats(bpp_batcher *b, uint64_t *pooled_calls, uint64_t *engine_calls, uint64_t *solo_calls) {
  if (!b) return BPP_ERR_BAD_HANDLE;
  std::lock_guard<std::mutex> lk(b->mu);
  if (pooled_calls) *pooled_calls = b->pooled_calls;
  if (engine_calls) *engine_calls = b->engine_calls;
  if (solo_calls) *solo_calls = b->solo_calls;
  return BPP_OK;
}

int bpp_batcher_verify(bpp_batcher *b, const bpp_packed_batch *in, char *errbuf, size_t errbuf_len) {
  if (!b) return BPP_ERR_BAD_HANDLE;
  if (!in || in->n_items == 0 || !in->proofs || !in->commitments32 || !in->transcript_label) {
    set_err(errbuf, errbuf_len, "Range statements or proofs length empty");
    return BPP_ERR_INVALID_ARGUMENT;
  }
  bpp_batcher::Req me;
  me.in = in;
  const bool poolable = batcher_poolable(b, in);
  std::vector<bpp_batcher::Req *> mine;
  bpp_batcher::Lane *lane = nullptr;
  {
    std::unique_lock<std::mutex> lk(b->mu);
    if (poolable) b->pending.push_back(&me);
    auto free_lane = [&]() -> bpp_batcher::Lane * {
      for (auto &L : b->lanes)
        if (!L.busy) return &L;
      return nullptr;
    };
    // wait until somebody else has dealt with this request, or -- as long as nobody has taken it -- a lane is free and this
    // thread leads the next pooled call
    b->cv.wait(lk, [&] { return me.done || (!me.taken && free_lane() != nullptr); });
    if (me.done) {
      set_err(errbuf, errbuf_len, me.msg);
      return me.code;
    }
    lane = free_lane();
    lane->busy = true;
    if (poolable) {
      if (b->max_wait_us && b->pending.size() < b->max_calls)
        b->cv.wait_for(lk, std::chrono::microseconds(b->max_wait_us), [&] { return me.taken || b->pending.size() >= b->max_calls; });
      if (me.taken) {  // another leader took this thread's request while it waited for company: let that one finish it
        lane->busy = false;
        b->cv.notify_all();
        b->cv.wait(lk, [&] { return me.done; });
        set_err(errbuf, errbuf_len, me.msg);
        return me.code;
      }
      size_t proofs = 0;
      while (!b->pending.empty() && mine.size() < b->max_calls) {
        bpp_batcher::Req *r = b->pending.front();
        if (!mine.empty() && proofs + r->in->n_items > b->max_proofs) break;
        proofs += r->in->n_items;
        r->taken = true;
        mine.push_back(r);
        b->pending.pop_front();
      }
      if (!me.taken) {  // the queue was cut short before this thread's own request: it goes first, the last one taken goes back
        for (auto it = b->pending.begin(); it != b->pending.end(); ++it)
          if (*it == &me) {
            b->pending.erase(it);
            break;
          }
        if (mine.size() >= b->max_calls) {
          mine.back()->taken = false;
          b->pending.push_front(mine.back());
          mine.pop_back();
        }
        me.taken = true;
        mine.push_back(&me);
      }
    } else {
      mine.push_back(&me);
    }
    b->engine_calls++;
    if (mine.size() > 1) b->pooled_calls += mine.size();
    else b->solo_calls++;
  }
  batcher_run(b, *lane, mine);
  {
    std::lock_guard<std::mutex> lk(b->mu);
    for (auto *r : mine)
      if (r != &me) r->done = true;  // (`me` lives on this stack and is always part of `mine`)
    lane->busy = false;
  }
  b->cv.notify_all();
  set_err(errbuf, errbuf_len, me.msg);
  return me.code;
}

}  // extern "C"
