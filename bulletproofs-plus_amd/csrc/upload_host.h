// Host side of bpp_batch_upload: everything that looks at untrusted proof / statement bytes before the device does.
// Pure host C++ (no HIP): engine.hip drives it with page-locked staging and its worker pool, hosttest_upload.cpp drives
// it under AddressSanitizer / UBSan in the CPU test suite.
//
// Reference lines this restates:
//   RangeStatement::init                          src/range_statement.rs:36-73      (construction errors)
//   RangeProof::from_bytes                        src/range_proof.rs:1155-1257      (construction errors)
//   verify_statements_and_generators_consistency  src/range_proof.rs:637-659        (extension degree of EVERY item first)
//                                                 src/range_proof.rs:674-682        (then the minimum-value promises)
//   structural L/R checks of PASS 2               src/range_proof.rs:875-888        (recorded here, raised at verify time)
// Error precedence: a Rust caller holds RangeStatement / RangeProof OBJECTS, so construction errors come before
// verify_batch is entered at all; inside verify() the degree loop runs over all items before the promise loop.  Hence
// three tiers: (1) lowest-index construction error fails the upload; (2) degree and (3) promise findings are recorded
// per item and raised when the item's chunk (= one reference verify() call) is verified, degree findings first.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "../../include/bpp.h"
#include "layout.h"
#include "merlin.h"
#include "scalar.h"

namespace bpp {

struct ProofErr {
  int code;
  std::string msg;
  int tier = BPP_TIER_CONSTRUCTION;  // BPP_TIER_*: where in the reference's order of checks (include/bpp.h)
  uint32_t index = 0;                // proof the finding belongs to (inside the call), for tiers checked in proof order
};

struct ParsedItem {
  uint32_t t, rounds;
};

// RangeProof::from_bytes (src/range_proof.rs:1155-1257): structure + canonical scalars; points are not validated here
inline void parse_proof(const uint8_t *p, size_t len, ParsedItem &out) {
  if (len < 1) throw ProofErr{BPP_ERR_INVALID_LENGTH, "Serialized proof is too short"};
  uint32_t t = p[0];
  if (t < 1 || t > 6) throw ProofErr{BPP_ERR_INVALID_ARGUMENT, "Extension degree not valid"};
  size_t body = len - 1, nchunks = body / 32, rem = body % 32;
  auto need = [&](size_t idx) {
    if (idx >= nchunks) throw ProofErr{BPP_ERR_INVALID_LENGTH, "Serialized proof is too short"};
  };
  auto scalar_at = [&](size_t idx) {
    need(idx);
    if (!sc_is_canonical(p + 1 + 32 * idx)) throw ProofErr{BPP_ERR_INVALID_ARGUMENT, "Invalid parsing"};
  };
  for (size_t k = 0; k < t; k++) scalar_at(k);
  need(t);
  need(t + 1);
  need(t + 2);
  scalar_at(t + 3);
  scalar_at(t + 4);
  size_t rest = nchunks - (t + 5);
  if (rest / 2 == 0) throw ProofErr{BPP_ERR_INVALID_LENGTH, "Serialized proof is too short"};
  if ((rest % 2) || rem) throw ProofErr{BPP_ERR_INVALID_LENGTH, "Unused data after deserialization"};
  if (rest / 2 > BPP_MAX_WIRE_ROUNDS) throw ProofErr{BPP_ERR_SIZE_OVERFLOW, "Internal size overflow (more than 64 (L, R) pairs)"};
  out.t = t;
  out.rounds = (uint32_t)(rest / 2);
}

struct ParamShape {
  uint32_t n_bits, m_max, t;
};

// What the per-item checks look at, whichever form the caller used: an element of a bpp_verify_item array, or item i of a
// bpp_packed_batch (pointers by arithmetic).  ONE checking routine (upload_check_item) serves both forms, so the error
// kinds and their precedence cannot drift apart.
struct ItemView {
  const uint8_t *proof;
  size_t proof_len;
  const uint8_t *commitments32;
  uint32_t m;
  const uint64_t *min_values;
  const uint8_t *min_present;
  const uint8_t *seed_nonce32;
};
inline ItemView item_view(const bpp_verify_item &it) {
  return ItemView{it.proof, it.proof_len, it.commitments32, it.m, it.min_values, it.min_present, it.seed_nonce32};
}
inline ItemView item_view(const bpp_packed_batch &pk, size_t i) {
  const bool seed = pk.seed_nonces32 && (!pk.seed_present || pk.seed_present[i]);
  return ItemView{pk.proofs ? pk.proofs + i * pk.proof_stride : nullptr,
                  pk.proof_len,
                  pk.commitments32 ? pk.commitments32 + i * (size_t)pk.m * 32 : nullptr,
                  pk.m,
                  pk.min_values ? pk.min_values + i * (size_t)pk.m : nullptr,
                  pk.min_present ? pk.min_present + i * (size_t)pk.m : nullptr,
                  seed ? pk.seed_nonces32 + i * 32 : nullptr};
}

// per-item findings of verify()'s own consistency loops, raised when the item's chunk is verified
#define BPP_DEFER_DEGREE 1u   // src/range_proof.rs:637-659  -> InvalidArgument
#define BPP_DEFER_PROMISE 2u  // src/range_proof.rs:674-682  -> InvalidLength

struct UploadPlan {
  size_t n_items = 0;
  std::vector<ProofDesc> desc;
  std::vector<uint8_t> rounds_bad;  // 0 ok, 3 InvalidLength, 5 SizeOverflow  (src/range_proof.rs:875-888)
  std::vector<uint8_t> defer;       // BPP_DEFER_* bits
  std::vector<uint8_t> seeds, states;
  std::vector<uint64_t> minvals;
  struct Pre {
    uint32_t minval_idx, dyn_off, rounds;
    size_t proof_off;
  };
  std::vector<Pre> pre;
  size_t proof_bytes = 0, sum_m = 0, bytes_total = 0, tr_err_index = 0;
  uint32_t total_dyn = 0, rmax = 0, max_mn = 0, packed_rounds = 0;
  bool any_seed = false, any_rounds_bad = false, any_defer = false, uniform_rounds = true;
};

// ---- pass A (serial, cheap, validates nothing): running offsets, the round count implied by each proof's length,
// transcript ids.  A malformed item gets harmless numbers here; pass B reports it.  The count used for the LAYOUT here
// and the count the kernels loop over (desc.rounds, pass B) are the same number for every item that passes pass B:
// both are (chunks - t - 5) / 2 of the proof's own bytes, and pass B refuses anything above BPP_MAX_WIRE_ROUNDS.
inline void upload_pass_a(const bpp_verify_item *items, size_t n_items, UploadPlan &pl) {
  pl.n_items = n_items;
  pl.desc.assign(n_items, ProofDesc{});
  pl.rounds_bad.assign(n_items, 0);
  pl.defer.assign(n_items, 0);
  pl.seeds.clear();
  pl.states.clear();
  pl.pre.resize(n_items);
  std::map<std::string, uint32_t> state_ids;
  size_t proof_bytes = 0, sum_m = 0;
  uint64_t dyn64 = 0;
  bool any_seed_ptr = false;
  pl.tr_err_index = n_items;  // first item whose explicit transcript state is unusable
  for (size_t i = 0; i < n_items; i++) {
    const bpp_verify_item &it = items[i];
    ProofDesc &d = pl.desc[i];
    uint32_t rounds = 0;
    if (it.proof && it.proof_len >= 1) {
      const size_t t0 = it.proof[0], nchunks = (it.proof_len - 1) / 32;
      if (nchunks > t0 + 5) rounds = (uint32_t)std::min<size_t>((nchunks - (t0 + 5)) / 2, BPP_MAX_WIRE_ROUNDS);
    }
    pl.pre[i] = UploadPlan::Pre{(uint32_t)sum_m, (uint32_t)dyn64, rounds, proof_bytes};
    any_seed_ptr = any_seed_ptr || it.seed_nonce32 != nullptr;
    proof_bytes += it.proof ? it.proof_len : 0;
    sum_m += it.m;
    dyn64 += (uint64_t)it.m + 3 + 2 * (uint64_t)rounds;
    // 32-bit byte offsets and slot numbers (bit 31 of a slot / point index carries a flag)
    if (proof_bytes + sum_m * 32 >= (1ull << 32) || sum_m >= (1ull << 28) || dyn64 >= (1ull << 31))
      throw ProofErr{BPP_ERR_SIZE_OVERFLOW, "batch too large for one call (4 GB of proof bytes)"};
    // transcript: explicit state wins, else Transcript::new(label).  Same source as the previous item (the common case:
    // one label for the whole batch) -> same id, no key building / map lookup
    const bpp_verify_item *prev = i ? &items[i - 1] : nullptr;
    if (prev && prev->transcript_state == it.transcript_state && prev->transcript_label == it.transcript_label &&
        prev->label_len == it.label_len) {
      d.state_idx = pl.desc[i - 1].state_idx;
      continue;
    }
    std::string key;
    if (it.transcript_state) {
      key.assign((const char *)it.transcript_state, 203);
      key.push_back('S');
    } else {
      key.assign((const char *)it.transcript_label, it.transcript_label ? it.label_len : 0);
      key.push_back('L');
    }
    auto sit = state_ids.find(key);
    if (sit == state_ids.end()) {
      uint32_t id = (uint32_t)(pl.states.size() / 203);
      pl.states.resize(pl.states.size() + 203);
      if (it.transcript_state) {
        memcpy(&pl.states[(size_t)id * 203], it.transcript_state, 203);
        if (pl.states[(size_t)id * 203 + 200] >= BPP_STROBE_R && pl.tr_err_index == n_items) pl.tr_err_index = i;
      } else {
        Strobe st;
        merlin_new(st, it.transcript_label, (uint32_t)(it.transcript_label ? it.label_len : 0));
        strobe_to_bytes(&pl.states[(size_t)id * 203], st);
      }
      sit = state_ids.emplace(key, id).first;
    }
    d.state_idx = sit->second;
  }
  pl.proof_bytes = proof_bytes;
  pl.sum_m = sum_m;
  pl.bytes_total = proof_bytes + sum_m * 32;
  pl.total_dyn = (uint32_t)dyn64;
  pl.minvals.assign(sum_m, 0);
  if (any_seed_ptr) pl.seeds.assign(n_items * 32, 0);  // (2 MB per 65 536 items: only when a nonce is present at all)
}

// ---- pass B (on `parallel_for`'s workers): per item, the checks in the reference's order + the copies into
// bytes_dst[0 .. bytes_total).  A worker stops at the first CONSTRUCTION error of its range; the lowest index over all
// ranges is thrown, as the serial loop would.  Degree / promise findings are recorded per item (pl.defer).
typedef std::function<void(uint32_t, const std::function<void(uint32_t)> &)> ParallelFor;

struct UploadPart {  // what one worker of pass B found in its range of items
  size_t err_index;
  ProofErr err;
  bool any_seed = false, any_rounds_bad = false, any_defer = false, uniform = true;
  uint32_t rmax = 0, max_mn = 0;
};

// the per-item body of pass B, for either input form
inline void upload_check_item(const ItemView &it, size_t i, const ParamShape &P, UploadPlan &pl, const UploadPlan::Pre &pre,
                              uint32_t rounds0, uint8_t *bytes_dst, UploadPart &pt) {
  ProofDesc &d = pl.desc[i];
  // RangeStatement::init (src/range_statement.rs:36-73)
  if (it.m == 0 || (it.m & (it.m - 1)))
    throw ProofErr{BPP_ERR_INVALID_ARGUMENT, "Number of commitments must be a power of two"};
  if (!it.commitments32 || (!it.min_values && it.min_present))
    throw ProofErr{BPP_ERR_INVALID_ARGUMENT, "Incorrect number of minimum value promises"};
  if (P.m_max < it.m) throw ProofErr{BPP_ERR_INVALID_ARGUMENT, "Not enough generators for this statement"};
  if (it.seed_nonce32 && it.m > 1)
    throw ProofErr{BPP_ERR_INVALID_ARGUMENT, "Mask recovery is not supported with an aggregated statement"};
  if (!it.proof) throw ProofErr{BPP_ERR_INVALID_LENGTH, "Serialized proof is too short"};
  ParsedItem pi;
  parse_proof(it.proof, it.proof_len, pi);
  if (pi.rounds != pre.rounds) throw ProofErr{BPP_ERR_ENGINE, "internal: layout and parse disagree on the round count"};
  if (i == pl.tr_err_index) throw ProofErr{BPP_ERR_INVALID_ARGUMENT, "transcript state has pos >= rate"};
  // verify_statements_and_generators_consistency, first loop (src/range_proof.rs:637-659)
  if (pi.t != P.t) {
    pl.defer[i] |= BPP_DEFER_DEGREE;
    pt.any_defer = true;
  }
  d.proof_off = (uint32_t)pre.proof_off;
  if (bytes_dst + pre.proof_off != it.proof) memcpy(bytes_dst + pre.proof_off, it.proof, it.proof_len);  // (else: already staged)
  d.rounds = pi.rounds;  // == pre.rounds: the slot ranges sized in pass A are the ones the kernels walk
  d.m = it.m;
  d.minval_idx = pre.minval_idx;
  for (uint32_t j = 0; j < it.m; j++) {
    bool present = it.min_present ? it.min_present[j] != 0 : false;
    uint64_t v = (present && it.min_values) ? it.min_values[j] : 0;
    // second loop (:674-682)
    if (present && P.n_bits < 64 && (v >> P.n_bits) > 0) {
      pl.defer[i] |= BPP_DEFER_PROMISE;
      pt.any_defer = true;
    }
    pl.minvals[pre.minval_idx + j] = v;
  }
  d.dyn_off = pre.dyn_off;
  d.flags = it.seed_nonce32 ? 1u : 0u;
  if (it.seed_nonce32) {
    memcpy(&pl.seeds[i * 32], it.seed_nonce32, 32);
    pt.any_seed = true;
  }
  // structural checks evaluated with PASS-2 precedence at verify time (:875-888)
  const uint64_t mn = (uint64_t)it.m * P.n_bits;
  if (pi.rounds >= 32)
    pl.rounds_bad[i] = BPP_ERR_SIZE_OVERFLOW;
  else if ((1ull << pi.rounds) != mn)
    pl.rounds_bad[i] = BPP_ERR_INVALID_LENGTH;
  if (pl.rounds_bad[i]) pt.any_rounds_bad = true;
  if (pi.rounds != rounds0) pt.uniform = false;
  pt.rmax = std::max(pt.rmax, pi.rounds);
  pt.max_mn = std::max(pt.max_mn, (uint32_t)mn);
  // commitments follow all proofs
  d.commit_off = (uint32_t)(pl.proof_bytes + 32 * (size_t)pre.minval_idx);
  memcpy(bytes_dst + d.commit_off, it.commitments32, (size_t)it.m * 32);
}

// runs body(i, part) over all items on the workers, then folds the parts: lowest-index construction error is thrown
template <typename Body>
inline void upload_run_parts(UploadPlan &pl, const ParallelFor &parallel_for, const Body &body) {
  const size_t n_items = pl.n_items;
  const uint32_t n_parts = (uint32_t)std::max<size_t>(1, std::min<size_t>(16, n_items / 512));
  std::vector<UploadPart> parts(n_parts);
  parallel_for(n_parts, [&](uint32_t k) {
    UploadPart &pt = parts[k];
    pt.err_index = n_items;
    const size_t lo = n_items * k / n_parts, hi = n_items * (k + 1) / n_parts;
    for (size_t i = lo; i < hi; i++) {
      try {
        body(i, pt);
      } catch (const ProofErr &e) {
        pt.err_index = i;
        pt.err = e;
        return;
      }
    }
  });
  const UploadPart *first = nullptr;
  for (const UploadPart &pt : parts)
    if (pt.err_index < n_items && (!first || pt.err_index < first->err_index)) first = &pt;
  if (first) {
    ProofErr e = first->err;
    e.index = (uint32_t)first->err_index;
    throw e;
  }
  for (const UploadPart &pt : parts) {
    pl.any_seed = pl.any_seed || pt.any_seed;
    pl.any_rounds_bad = pl.any_rounds_bad || pt.any_rounds_bad;
    pl.any_defer = pl.any_defer || pt.any_defer;
    pl.uniform_rounds = pl.uniform_rounds && pt.uniform;
    pl.rmax = std::max(pl.rmax, pt.rmax);
    pl.max_mn = std::max(pl.max_mn, pt.max_mn);
  }
}

inline void upload_pass_b(const bpp_verify_item *items, const ParamShape &P, UploadPlan &pl, uint8_t *bytes_dst,
                          const ParallelFor &parallel_for) {
  const uint32_t rounds0 = pl.pre[0].rounds;
  upload_run_parts(pl, parallel_for, [&](size_t i, UploadPart &pt) {
    upload_check_item(item_view(items[i]), i, P, pl, pl.pre[i], rounds0, bytes_dst, pt);
  });
}

// ---- the packed form (bpp_packed_batch): equal-length proofs, one aggregation factor, one transcript.  When every proof
// also claims the same extension degree in its first byte (anything else is a hostile or mixed input), the layout of pass A
// is arithmetic -- proof i at i * proof_len, its slots at i * (m + 3 + 2 rounds) -- and nothing per item is built or looked
// up before the parallel pass.  Otherwise the batch goes through the item form (same checks either way).
inline void upload_packed_as_items(const bpp_packed_batch &pk, std::vector<bpp_verify_item> &items) {
  items.resize(pk.n_items);
  for (size_t i = 0; i < pk.n_items; i++) {
    const ItemView v = item_view(pk, i);
    bpp_verify_item &it = items[i];
    it.proof = v.proof;
    it.proof_len = v.proof_len;
    it.commitments32 = v.commitments32;
    it.m = v.m;
    it.min_values = v.min_values;
    it.min_present = v.min_present;
    it.seed_nonce32 = v.seed_nonce32;
    it.transcript_state = pk.transcript_state;
    it.transcript_label = pk.transcript_label;
    it.label_len = pk.label_len;
  }
}

// true: the plan's layout was filled in arithmetically (call upload_pass_b_packed next); false: use the item form
inline bool upload_pass_a_packed(const bpp_packed_batch &pk, UploadPlan &pl, const ParallelFor &parallel_for) {
  const size_t n = pk.n_items;
  if (!pk.proofs || pk.proof_len < 1 || pk.proof_stride < pk.proof_len) return false;
  // first byte of every proof (its claimed extension degree) equal?
  const uint8_t t0 = pk.proofs[0];
  const uint32_t n_parts = (uint32_t)std::max<size_t>(1, std::min<size_t>(16, n / 4096));
  std::vector<uint8_t> differs(n_parts, 0);
  parallel_for(n_parts, [&](uint32_t k) {
    const size_t lo = n * k / n_parts, hi = n * (k + 1) / n_parts;
    uint8_t acc = 0;
    for (size_t i = lo; i < hi; i++) acc |= (uint8_t)(pk.proofs[i * pk.proof_stride] ^ t0);
    differs[k] = acc;
  });
  for (uint8_t d : differs)
    if (d) return false;
  const size_t nchunks = (pk.proof_len - 1) / 32;
  uint32_t rounds = 0;
  if (nchunks > (size_t)t0 + 5) rounds = (uint32_t)std::min<size_t>((nchunks - ((size_t)t0 + 5)) / 2, BPP_MAX_WIRE_ROUNDS);
  const uint64_t per_dyn = (uint64_t)pk.m + 3 + 2 * (uint64_t)rounds;
  const uint64_t proof_bytes = (uint64_t)n * pk.proof_len, sum_m = (uint64_t)n * pk.m, dyn64 = (uint64_t)n * per_dyn;
  if (proof_bytes + sum_m * 32 >= (1ull << 32) || sum_m >= (1ull << 28) || dyn64 >= (1ull << 31))
    throw ProofErr{BPP_ERR_SIZE_OVERFLOW, "batch too large for one call (4 GB of proof bytes)"};
  pl.n_items = n;
  pl.desc.assign(n, ProofDesc{});  // state_idx 0 everywhere: one transcript
  pl.rounds_bad.assign(n, 0);
  pl.defer.assign(n, 0);
  pl.seeds.assign(pk.seed_nonces32 ? n * 32 : 0, 0);
  pl.pre.clear();
  pl.states.assign(203, 0);
  pl.tr_err_index = n;
  if (pk.transcript_state) {
    memcpy(pl.states.data(), pk.transcript_state, 203);
    if (pl.states[200] >= BPP_STROBE_R) pl.tr_err_index = 0;
  } else {
    Strobe st;
    merlin_new(st, pk.transcript_label, (uint32_t)(pk.transcript_label ? pk.label_len : 0));
    strobe_to_bytes(pl.states.data(), st);
  }
  pl.proof_bytes = (size_t)proof_bytes;
  pl.sum_m = (size_t)sum_m;
  pl.bytes_total = (size_t)(proof_bytes + sum_m * 32);
  pl.total_dyn = (uint32_t)dyn64;
  pl.minvals.assign((size_t)sum_m, 0);
  pl.packed_rounds = rounds;
  return true;
}

inline void upload_pass_b_packed(const bpp_packed_batch &pk, const ParamShape &P, UploadPlan &pl, uint8_t *bytes_dst,
                                 const ParallelFor &parallel_for) {
  const uint32_t rounds = pl.packed_rounds;
  const uint32_t per_dyn = pk.m + 3 + 2 * rounds;
  upload_run_parts(pl, parallel_for, [&](size_t i, UploadPart &pt) {
    const UploadPlan::Pre pre{(uint32_t)(i * pk.m), (uint32_t)(i * per_dyn), rounds, i * pk.proof_len};
    upload_check_item(item_view(pk, i), i, P, pl, pre, rounds, bytes_dst, pt);
  });
}

// ---- secrets on the host side of an upload.  A statement's seed nonce (src/range_statement.rs:31) is wiped by the
// reference when the statement drops (`Zeroize for RangeStatement`, :76-81).  Here it exists in two host places: the plan
// (pl.seeds) and the page-locked staging the device copy is fed from.  Both are wiped on every way out.
inline void secure_wipe(void *p, size_t n) {
  if (p && n) explicit_bzero(p, n);
}
struct PlanWipe {  // RAII: put one next to every UploadPlan
  UploadPlan &pl;
  ~PlanWipe() { secure_wipe(pl.seeds.data(), pl.seeds.size()); }
};
// layout of the small staging buffer (descriptors, promises, transcript states, seed nonces LAST so that one range holds
// everything secret)
struct SmallStaging {
  size_t o_desc = 0, o_min = 0, o_state = 0, o_seed = 0, n_seed = 0, total = 0;
};
inline SmallStaging upload_small_layout(const UploadPlan &pl, size_t n_items) {
  SmallStaging L;
  // every piece starts on a 16-byte boundary: k_ingest moves them 16 bytes per lane out of the mapped staging
  auto up16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
  L.o_desc = 0;
  L.o_min = up16(L.o_desc + n_items * sizeof(ProofDesc));
  L.o_state = up16(L.o_min + pl.minvals.size() * 8);
  L.o_seed = up16(L.o_state + pl.states.size());
  L.n_seed = pl.any_seed ? pl.seeds.size() : 0;
  L.total = L.o_seed + L.n_seed;
  return L;
}
inline void upload_fill_small(const UploadPlan &pl, const ProofDesc *desc, size_t n_items, uint8_t *st, const SmallStaging &L) {
  memcpy(st + L.o_desc, desc, n_items * sizeof(ProofDesc));
  memcpy(st + L.o_min, pl.minvals.data(), pl.minvals.size() * 8);
  memcpy(st + L.o_state, pl.states.data(), pl.states.size());
  if (L.n_seed) memcpy(st + L.o_seed, pl.seeds.data(), L.n_seed);
}
inline void upload_wipe_small(uint8_t *st, const SmallStaging &L) { secure_wipe(st + L.o_seed, L.n_seed); }

// verify()'s consistency loops for the items [p0, p1) of one chunk, in the reference's order (degree of every item, then
// the promises); throws the finding with the lowest index of the first non-empty tier
inline void check_deferred(const std::vector<uint8_t> &defer, uint32_t p0, uint32_t p1) {
  for (uint32_t p = p0; p < p1; p++)
    if (defer[p] & BPP_DEFER_DEGREE) throw ProofErr{BPP_ERR_INVALID_ARGUMENT, "Inconsistent extension degree", BPP_TIER_DEGREE, p};
  for (uint32_t p = p0; p < p1; p++)
    if (defer[p] & BPP_DEFER_PROMISE)
      throw ProofErr{BPP_ERR_INVALID_LENGTH, "Minimum value promise exceeds bit vector capacity", BPP_TIER_PROMISE, p};
}

// the per-proof findings of one verify() call over proofs [p0, p1), in the reference's order: a statement whose commitment
// does not decode could never have been constructed (RangeStatement holds points); PASS 1 over ALL proofs
// (src/range_proof.rs:816-850); then PASS 2 in proof order (:859-888): decompression, then the L/R count.
// `status` = the kernels' BPP_ST_* bits per proof, `rounds_bad` = 0 / InvalidLength / SizeOverflow recorded at upload.
inline void check_chunk_errors(const uint32_t *status, const uint8_t *rounds_bad, uint32_t p0, uint32_t p1) {
  for (uint32_t p = p0; p < p1; p++)
    if (status[p] & BPP_ST_COMMIT_FAIL)
      throw ProofErr{BPP_ERR_INVALID_ARGUMENT, "Statement commitment is not the canonical encoding of a point", BPP_TIER_STATEMENT_POINT, p};
  for (uint32_t p = p0; p < p1; p++)
    if (status[p] & BPP_ST_TRANSCRIPT_FAIL)
      throw ProofErr{BPP_ERR_VERIFICATION_FAILED,
                     "Identity element cannot be added to the transcript / transcript challenge cannot be zero", BPP_TIER_PASS1, p};
  for (uint32_t p = p0; p < p1; p++) {
    if (status[p] & BPP_ST_DECOMPRESS_FAIL)
      throw ProofErr{BPP_ERR_INVALID_ARGUMENT, "A proof member was not the canonical encoding of a point", BPP_TIER_PASS2, p};
    if (rounds_bad[p] == BPP_ERR_SIZE_OVERFLOW) throw ProofErr{BPP_ERR_SIZE_OVERFLOW, "Internal size overflow", BPP_TIER_PASS2, p};
    if (rounds_bad[p] == BPP_ERR_INVALID_LENGTH)
      throw ProofErr{BPP_ERR_INVALID_LENGTH, "Vector L/R length not adequate", BPP_TIER_PASS2, p};
  }
}

// ---- shards of ONE reference batch (bpp_verify_sharded): every rank reports the first finding of its own proofs as a
// fixed-size trailer next to its collective payload, and every rank picks the same winner: the check the single-process
// verify() would have failed first = lowest tier, then lowest rank (shards are contiguous in proof order, and inside a
// rank the finding already is the first in proof order).  Numeric tiers (include/bpp.h BPP_TIER_*), no message matching.
//   byte 0      tier (0 = nothing found)
//   bytes 1..4  return code, i32 little-endian (ProofError kind 1..5, or a negative engine code)
//   bytes 5..8  index of the proof inside the WHOLE batch, u32 little-endian
//   byte 9      message length (<= 118), bytes 10..127 message
#define BPP_SHARD_TRAILER_BYTES 128
inline void shard_trailer_encode(uint8_t out[BPP_SHARD_TRAILER_BYTES], int tier, int code, uint32_t index, const char *msg) {
  memset(out, 0, BPP_SHARD_TRAILER_BYTES);
  if (tier == BPP_TIER_NONE) return;
  out[0] = (uint8_t)tier;
  const uint32_t c = (uint32_t)code;
  for (int i = 0; i < 4; i++) {
    out[1 + i] = (uint8_t)(c >> (8 * i));
    out[5 + i] = (uint8_t)(index >> (8 * i));
  }
  const size_t n = msg ? std::min<size_t>(strlen(msg), BPP_SHARD_TRAILER_BYTES - 10) : 0;
  out[9] = (uint8_t)n;
  if (n) memcpy(out + 10, msg, n);
}
struct ShardFinding {
  int tier = BPP_TIER_NONE, code = BPP_OK, rank = -1;
  uint32_t index = 0;
  std::string msg;
};
inline ShardFinding shard_trailer_decode(const uint8_t *t, int rank) {
  ShardFinding f;
  f.tier = t[0];
  if (f.tier == BPP_TIER_NONE) return f;
  uint32_t c = 0, idx = 0;
  for (int i = 0; i < 4; i++) {
    c |= (uint32_t)t[1 + i] << (8 * i);
    idx |= (uint32_t)t[5 + i] << (8 * i);
  }
  f.code = (int)c;
  f.index = idx;
  f.rank = rank;
  f.msg.assign((const char *)t + 10, std::min<size_t>(t[9], BPP_SHARD_TRAILER_BYTES - 10));
  return f;
}
// trailer of rank r at trailers + r * stride; tier NONE in the result = every rank is clean
inline ShardFinding shard_resolve(const uint8_t *trailers, size_t stride, int world) {
  ShardFinding best;
  for (int r = 0; r < world; r++) {
    const ShardFinding f = shard_trailer_decode(trailers + (size_t)r * stride, r);
    if (f.tier != BPP_TIER_NONE && (best.tier == BPP_TIER_NONE || f.tier < best.tier)) best = f;  // ranks ascend: ties keep the lower one
  }
  return best;
}
// the first finding of a rank's own proofs [0, n) -- deferred consistency findings, then the kernels' status -- as a trailer;
// `first_index` = position of the rank's first proof in the whole batch
inline void shard_local_trailer(const uint8_t *defer, const uint32_t *status, const uint8_t *rounds_bad, uint32_t n, uint32_t first_index,
                                uint8_t out[BPP_SHARD_TRAILER_BYTES]) {
  try {
    if (defer) {
      std::vector<uint8_t> d(defer, defer + n);
      check_deferred(d, 0, n);
    }
    check_chunk_errors(status, rounds_bad, 0, n);
    shard_trailer_encode(out, BPP_TIER_NONE, BPP_OK, 0, nullptr);
  } catch (const ProofErr &e) {
    shard_trailer_encode(out, e.tier, e.code, first_index + e.index, e.msg.c_str());
  }
}

}  // namespace bpp
