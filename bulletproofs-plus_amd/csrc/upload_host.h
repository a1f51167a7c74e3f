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
  if (rest / 2 > BPP_MAX_WIRE_ROUNDS) throw ProofErr{BPP_ERR_SIZE_OVERFLOW, "Internal size overflow (proof larger than 64 MB)"};
  out.t = t;
  out.rounds = (uint32_t)(rest / 2);
}

struct ParamShape {
  uint32_t n_bits, m_max, t;
};

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
  uint32_t total_dyn = 0, rmax = 0, max_mn = 0;
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
  pl.seeds.assign(n_items * 32, 0);
  pl.states.clear();
  pl.pre.resize(n_items);
  std::map<std::string, uint32_t> state_ids;
  size_t proof_bytes = 0, sum_m = 0;
  uint64_t dyn64 = 0;
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
}

// ---- pass B (on `parallel_for`'s workers): per item, the checks in the reference's order + the copies into
// bytes_dst[0 .. bytes_total).  A worker stops at the first CONSTRUCTION error of its range; the lowest index over all
// ranges is thrown, as the serial loop would.  Degree / promise findings are recorded per item (pl.defer).
typedef std::function<void(uint32_t, const std::function<void(uint32_t)> &)> ParallelFor;

inline void upload_pass_b(const bpp_verify_item *items, const ParamShape &P, UploadPlan &pl, uint8_t *bytes_dst,
                          const ParallelFor &parallel_for) {
  const size_t n_items = pl.n_items;
  struct Part {
    size_t err_index;
    ProofErr err;
    bool any_seed = false, any_rounds_bad = false, any_defer = false, uniform = true;
    uint32_t rmax = 0, max_mn = 0;
  };
  const uint32_t n_parts = (uint32_t)std::max<size_t>(1, std::min<size_t>(16, n_items / 512));
  std::vector<Part> parts(n_parts);
  const uint32_t rounds0 = pl.pre[0].rounds;
  auto one_item = [&](size_t i, Part &pt) {
    const bpp_verify_item &it = items[i];
    ProofDesc &d = pl.desc[i];
    const UploadPlan::Pre &pre = pl.pre[i];
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
    memcpy(bytes_dst + pre.proof_off, it.proof, it.proof_len);
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
  };
  parallel_for(n_parts, [&](uint32_t k) {
    Part &pt = parts[k];
    pt.err_index = n_items;
    const size_t lo = n_items * k / n_parts, hi = n_items * (k + 1) / n_parts;
    for (size_t i = lo; i < hi; i++) {
      try {
        one_item(i, pt);
      } catch (const ProofErr &e) {
        pt.err_index = i;
        pt.err = e;
        return;
      }
    }
  });
  const Part *first = nullptr;
  for (const Part &pt : parts)
    if (pt.err_index < n_items && (!first || pt.err_index < first->err_index)) first = &pt;
  if (first) throw first->err;
  for (const Part &pt : parts) {
    pl.any_seed = pl.any_seed || pt.any_seed;
    pl.any_rounds_bad = pl.any_rounds_bad || pt.any_rounds_bad;
    pl.any_defer = pl.any_defer || pt.any_defer;
    pl.uniform_rounds = pl.uniform_rounds && pt.uniform;
    pl.rmax = std::max(pl.rmax, pt.rmax);
    pl.max_mn = std::max(pl.max_mn, pt.max_mn);
  }
}

// verify()'s consistency loops for the items [p0, p1) of one chunk, in the reference's order (degree of every item, then
// the promises); throws the finding with the lowest index of the first non-empty tier
inline void check_deferred(const std::vector<uint8_t> &defer, uint32_t p0, uint32_t p1) {
  for (uint32_t p = p0; p < p1; p++)
    if (defer[p] & BPP_DEFER_DEGREE) throw ProofErr{BPP_ERR_INVALID_ARGUMENT, "Inconsistent extension degree"};
  for (uint32_t p = p0; p < p1; p++)
    if (defer[p] & BPP_DEFER_PROMISE)
      throw ProofErr{BPP_ERR_INVALID_LENGTH, "Minimum value promise exceeds bit vector capacity"};
}

}  // namespace bpp
