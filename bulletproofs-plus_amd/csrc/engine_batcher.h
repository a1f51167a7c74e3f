// bpp_batcher: pools the small verify_batch calls of many host threads into grouped engine calls.
//
// Why: a small call (one reference batch of up to a few hundred proofs) is a chain of about fifteen latency-bound kernels
// with tiny grids; the chip runs about six such kernels side by side whatever the number of callers, which caps independent
// 256-proof calls at ~5 300 per second (1.4 M proofs/s), while the same batches as the groups of ONE call run
// at 24 M proofs/s.  The batcher gives separate callers the second form: every caller hands over its own reference batch
// (bpp_batcher_verify / bpp_batcher_verify_action block until the outcome is there); whichever caller finds a lane free becomes
// the leader of the next pooled call, takes what queued up while the previous pooled calls were running (plus what arrives
// within max_wait_us, if anything was asked for), uploads everybody's statements and proofs as ONE resident batch -- through
// the item form, straight out of the callers' buffers: no intermediate copy, and inputs of different shapes (proof length,
// aggregation factor, transcript) pool like any others --, verifies it with one group per caller and that caller's own
// VerifyAction (bpp_verify_resident_groups_actions: ragged groups, one outcome each, masks for the groups that recover them)
// and hands the outcomes back.  No thread of its own.
// Each caller gets exactly what bpp_verify_batch_packed(ctx, params, its input, its action, 0, ...) would have returned:
// src/range_proof.rs:712-752 on its own statements and proofs, :941-969 for its masks.  RecoverOnly callers (a wallet scanning
// outputs) pool among themselves: such a pool runs neither weight chains nor PASS 2.  Seed nonces and masks exist only in the
// leader's call frame and in the lane's context, both wiped before the lane is handed on.
// Part of engine.hip's translation unit.
#pragma once

struct bpp_batcher {
  struct Req {
    const bpp_packed_batch *in = nullptr;
    int action = BPP_VERIFY_ONLY;
    uint8_t *masks_out = nullptr, *mask_present = nullptr;
    int code = BPP_OK;
    std::string msg;
    bool taken = false;  // a leader has it in its pooled call
    bool done = false;
  };
  struct Lane {
    bpp_ctx *ctx = nullptr;  // a context of its own (stream, staging, recycled work buffers)
    bool own = false;
    bool busy = false;
    std::vector<bpp_verify_item> items;
    std::vector<uint32_t> bounds;
    std::vector<int> actions;
    std::vector<bpp_shard_result> results;
    std::vector<uint8_t> masks, present;
  };
  uint64_t params = 0;
  uint32_t t = 1;
  uint32_t max_wait_us = 0, max_calls = 64, max_proofs = 16384;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<Req *> pending;
  std::vector<Lane> lanes;
  uint64_t pooled_calls = 0, engine_calls = 0, solo_calls = 0;  // statistics
  uint32_t largest_pool_calls = 0, largest_pool_proofs = 0;
};

namespace {

// A proof of more than this many bytes cannot belong to any statement (64 (L, R) pairs is the wire cap, layout.h) and an input
// of more than max_proofs items is a large call by itself: both go through a call of their own.
bool batcher_poolable(const bpp_batcher *b, const bpp_packed_batch *in) {
  return in->n_items <= b->max_proofs && in->proof_len <= 1 + 32 * (size_t)(6 + 5 + 2 * 64) && in->proof_stride >= in->proof_len;
}
// requests of one pool either all need the final check or none does (a RecoverOnly pool skips the weight chains and PASS 2)
inline bool batcher_wants_msm(const bpp_batcher::Req *r) { return r->action != BPP_RECOVER_ONLY; }

void batcher_run_unchecked(bpp_batcher *b, bpp_batcher::Lane &L, std::vector<bpp_batcher::Req *> reqs);
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

void batcher_run_unchecked(bpp_batcher *b, bpp_batcher::Lane &L, std::vector<bpp_batcher::Req *> reqs) {
  char err[256];
  auto solo = [&](bpp_batcher::Req *r) {
    err[0] = 0;
    r->code = bpp_verify_batch_packed(L.ctx, b->params, r->in, r->action, 0, r->masks_out, r->mask_present, err, sizeof(err));
    r->msg = err;
  };
  const size_t tb = (size_t)b->t * 32;
  // the lane's copies of recovered masks are wiped on every way out (src/extended_mask.rs:14 ZeroizeOnDrop); seed nonces are
  // never copied here (the items point into the callers' own buffers)
  ScopeExit wipe_lane{[&] {
    wipe(L.masks.data(), L.masks.size());
    L.items.clear();
  }};
  while (reqs.size() > 1) {
    // ---- everybody's items, one group per caller.  A VerifyOnly caller's nonces stay at home.
    size_t n = 0;
    for (auto *r : reqs) n += r->in->n_items;
    L.items.resize(n);
    L.bounds.resize(reqs.size() + 1);
    L.actions.resize(reqs.size());
    size_t at = 0;
    bool any_recover = false;
    std::vector<bpp_verify_item> one;
    for (size_t g = 0; g < reqs.size(); g++) {
      const bpp_packed_batch &in = *reqs[g]->in;
      L.bounds[g] = (uint32_t)at;
      L.actions[g] = reqs[g]->action;
      any_recover = any_recover || reqs[g]->action != BPP_VERIFY_ONLY;
      upload_packed_as_items(in, one);
      for (size_t i = 0; i < in.n_items; i++) {
        L.items[at + i] = one[i];
        if (reqs[g]->action == BPP_VERIFY_ONLY) L.items[at + i].seed_nonce32 = nullptr;
      }
      at += in.n_items;
    }
    L.bounds[reqs.size()] = (uint32_t)n;
    // ---- one upload.  A construction-time finding (a proof that could not have been deserialised, a statement that could not
    // have been built: RangeProof::from_bytes / RangeStatement::init, before verify_batch is entered) belongs to ONE caller:
    // the lowest offending index of the pool is also the lowest one of its owner's input, so the owner has its answer; the
    // others are pooled again without it.
    uint64_t h = 0;
    size_t culprit = reqs.size();
    GateHold gate(L.ctx->device, n <= BPP_GATE_SMALL_PROOFS);  // one turn at the device's gate for upload + verification
    {
      std::lock_guard<std::mutex> lk(L.ctx->mu);
      if (hipSetDevice(L.ctx->device) != hipSuccess) throw std::runtime_error("hipSetDevice failed");
      const std::shared_ptr<Params> Pp = params_registry().get(b->params);
      if (!Pp || Pp->device != L.ctx->device) throw std::runtime_error("unknown params handle");
      UploadPlan pl;
      PlanWipe wipe_plan{pl};
      try {
        upload_host_pack(L.ctx, *Pp, L.items.data(), n, nullptr, pl);
        h = upload_device(L.ctx, Pp, b->params, pl, nullptr, nullptr);
      } catch (const ProofErr &e) {
        if (e.tier != BPP_TIER_CONSTRUCTION || e.index >= n) throw std::runtime_error(e.msg);
        culprit = (size_t)(std::upper_bound(L.bounds.begin(), L.bounds.end(), e.index) - L.bounds.begin()) - 1;
        reqs[culprit]->code = e.code;
        reqs[culprit]->msg = e.msg;
      } catch (const EngineError &e) {
        throw std::runtime_error(e.msg);
      }
    }
    if (culprit < reqs.size()) {
      reqs.erase(reqs.begin() + (ptrdiff_t)culprit);
      continue;
    }
    L.results.resize(reqs.size());
    if (any_recover) {
      L.masks.assign(n * tb, 0);
      L.present.assign(n, 0);
    }
    const int rc = bpp_verify_resident_groups_actions(L.ctx, h, L.bounds.data(), reqs.size(), L.actions.data(), L.results.data(),
                                                      any_recover ? L.masks.data() : nullptr, any_recover ? L.present.data() : nullptr);
    (void)bpp_batch_destroy(L.ctx, h);
    if (rc != BPP_OK) break;  // an engine fault should not be pinned on all of them: everybody gets a call of its own
    for (size_t g = 0; g < reqs.size(); g++) {
      bpp_batcher::Req *r = reqs[g];
      r->code = L.results[g].code;
      r->msg = L.results[g].msg;
      const size_t p0 = L.bounds[g], cnt = r->in->n_items;
      if (r->code != BPP_OK) continue;  // (an Err returns no masks: the caller's buffers stay untouched, as in a call of its own)
      if (r->mask_present) {
        if (r->action != BPP_VERIFY_ONLY) memcpy(r->mask_present, &L.present[p0], cnt);
        else memset(r->mask_present, 0, cnt);
      }
      if (r->masks_out) {
        if (r->action != BPP_VERIFY_ONLY) memcpy(r->masks_out, &L.masks[p0 * tb], cnt * tb);
        else memset(r->masks_out, 0, cnt * tb);
      }
    }
    return;
  }
  for (auto *r : reqs) solo(r);
}

}  // namespace

extern "C" {

int bpp_batcher_create(bpp_ctx *ctx, uint64_t params, const bpp_packed_batch *shape, uint32_t lanes, uint32_t max_wait_us, uint32_t max_calls,
                       bpp_batcher **out) {
  (void)shape;  // (until round 3 only inputs of this shape were pooled; every shape pools now.  Kept for the ABI; may be NULL)
  if (!ctx || !out) return BPP_ERR_INVALID_ARGUMENT;
  const std::shared_ptr<Params> Pp = params_registry().get(params);
  if (!Pp || Pp->device != ctx->device) return BPP_ERR_BAD_HANDLE;
  if (lanes == 0) lanes = 2;  // (two pooled calls in flight keep the pools large; three are within run-to-run noise of two, four and six lose)
  if (lanes > 8) lanes = 8;
  auto b = std::make_unique<bpp_batcher>();
  b->params = params;
  b->t = Pp->t;
  b->max_wait_us = max_wait_us;
  if (max_calls) b->max_calls = max_calls;
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

int bpp_batcher_set_limits(bpp_batcher *b, uint32_t max_calls, uint32_t max_proofs) {
  if (!b) return BPP_ERR_BAD_HANDLE;
  std::lock_guard<std::mutex> lk(b->mu);
  if (max_calls) b->max_calls = max_calls;
  if (max_proofs) b->max_proofs = max_proofs;
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

int bpp_batcher_largest_pool(bpp_batcher *b, uint32_t *calls, uint32_t *proofs) {
  if (!b) return BPP_ERR_BAD_HANDLE;
  std::lock_guard<std::mutex> lk(b->mu);
  if (calls) *calls = b->largest_pool_calls;
  if (proofs) *proofs = b->largest_pool_proofs;
  return BPP_OK;
}

int bpp_batcher_verify_action(bpp_batcher *b, const bpp_packed_batch *in, int action, uint8_t *masks_out, uint8_t *mask_present,
                              char *errbuf, size_t errbuf_len) {
  if (!b) return BPP_ERR_BAD_HANDLE;
  if (!in || in->n_items == 0 || !in->proofs || !in->commitments32 || (!in->transcript_label && !in->transcript_state)) {
    set_err(errbuf, errbuf_len, "Range statements or proofs length empty");
    return BPP_ERR_INVALID_ARGUMENT;
  }
  if (action < 0 || action > 2) {
    set_err(errbuf, errbuf_len, "unknown verify action");
    return BPP_ERR_INVALID_ARGUMENT;
  }
  bpp_batcher::Req me;
  me.in = in;
  me.action = action;
  me.masks_out = masks_out;
  me.mask_present = mask_present;
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
      // The leader's own request goes first (it fits by itself: poolable), then whatever is queued, oldest first, as long as the
      // pool stays within max_calls requests and max_proofs proofs and is of one kind (with or without the final check).
      // Requests that do not fit stay where they are, for the next leader.
      for (auto it = b->pending.begin(); it != b->pending.end(); ++it)
        if (*it == &me) {
          b->pending.erase(it);
          break;
        }
      me.taken = true;
      mine.push_back(&me);
      size_t proofs = in->n_items;
      const bool msm = batcher_wants_msm(&me);
      for (auto it = b->pending.begin(); it != b->pending.end() && mine.size() < b->max_calls;) {
        bpp_batcher::Req *r = *it;
        if (batcher_wants_msm(r) != msm || proofs + r->in->n_items > b->max_proofs) {
          ++it;
          continue;
        }
        proofs += r->in->n_items;
        r->taken = true;
        mine.push_back(r);
        it = b->pending.erase(it);
      }
    } else {
      mine.push_back(&me);
    }
    b->engine_calls++;
    if (mine.size() > 1) {
      b->pooled_calls += mine.size();
      size_t proofs = 0;
      for (auto *r : mine) proofs += r->in->n_items;
      b->largest_pool_calls = std::max(b->largest_pool_calls, (uint32_t)mine.size());
      b->largest_pool_proofs = std::max(b->largest_pool_proofs, (uint32_t)proofs);
    } else {
      b->solo_calls++;
    }
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

int bpp_batcher_verify(bpp_batcher *b, const bpp_packed_batch *in, char *errbuf, size_t errbuf_len) {
  return bpp_batcher_verify_action(b, in, BPP_VERIFY_ONLY, nullptr, nullptr, errbuf, errbuf_len);
}

}  // extern "C"
