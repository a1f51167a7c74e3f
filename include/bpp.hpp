// bpp.hpp -- header-only C++17 host-side mirror of the reference's public interface for the hot path, above the
// C ABI of include/bpp.h.  Same names, argument meaning and error behaviour as tari_bulletproofs_plus 0.4.1:
//
//   reference (Rust)                                          here (namespace bpp_host)
//   ---------------------------------------------------------  --------------------------------------------
//   ristretto::create_pedersen_gens_with_extension_degree      create_pedersen_gens_with_extension_degree
//   RangeParameters::init            src/range_parameters.rs:32  RangeParameters::init
//   PedersenGens::commit             pedersen_gens.rs:112        RangeParameters::commit
//   RangeStatement::init             src/range_statement.rs:36   RangeStatement::init
//   CommitmentOpening::new / RangeWitness::init                  same
//   RangeProof::{prove_with_rng, verify_batch, to_bytes, from_bytes}   same (src/range_proof.rs:232,712,1120,1155)
//   VerifyAction, ExtendedMask, ProofError{VerificationFailed,...}     same (ProofError is an exception)
//   merlin::Transcript::new(label)                               Transcript::create(label)
//
// Scalars and points are 32-byte arrays (canonical little-endian scalar / ristretto255 encoding).
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "bpp.h"

namespace bpp_host {

using Bytes32 = std::array<uint8_t, 32>;

enum class ProofErrorKind { VerificationFailed = 1, InvalidArgument = 2, InvalidLength = 3, InvalidBlake2b = 4, SizeOverflow = 5 };

// src/errors.rs:11-28
struct ProofError : std::runtime_error {
  ProofErrorKind kind;
  ProofError(ProofErrorKind k, const std::string &m) : std::runtime_error(m), kind(k) {}
};
struct EngineFault : std::runtime_error {
  int code;
  EngineFault(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

enum class VerifyAction { VerifyOnly = 0, RecoverAndVerify = 1, RecoverOnly = 2 };  // src/range_proof.rs:46-54
enum class ExtensionDegree { DefaultPedersen = 1, AddOneBasePoint, AddTwoBasePoints, AddThreeBasePoints, AddFourBasePoints, AddFiveBasePoints };

inline void check(int rc, const char *err) {
  if (rc == 0) return;
  if (rc > 0) throw ProofError(static_cast<ProofErrorKind>(rc), err ? err : "");
  throw EngineFault(rc, err ? err : "");
}

class Engine {
 public:
  explicit Engine(int device = 0) {
    int rc = bpp_ctx_create(&ctx_, device);
    if (rc != 0) throw EngineFault(rc, "bpp_ctx_create failed: a gfx950 device is required (no CPU fallback)");
  }
  ~Engine() { bpp_ctx_destroy(ctx_); }
  Engine(const Engine &) = delete;
  Engine &operator=(const Engine &) = delete;
  bpp_ctx *ctx() const { return ctx_; }

 private:
  bpp_ctx *ctx_ = nullptr;
};

struct PedersenGens {
  ExtensionDegree extension_degree;
};
inline PedersenGens create_pedersen_gens_with_extension_degree(ExtensionDegree d) { return PedersenGens{d}; }

class Transcript {
 public:
  static Transcript create(const std::string &label) {
    Transcript t;
    t.label_ = label;
    return t;
  }
  static Transcript from_state(const uint8_t state203[203]) {
    Transcript t;
    t.state_.assign(state203, state203 + 203);
    return t;
  }
  const std::string &label() const { return label_; }
  const std::vector<uint8_t> &state() const { return state_; }

 private:
  std::string label_;
  std::vector<uint8_t> state_;
};

class RangeParameters {
 public:
  static std::shared_ptr<RangeParameters> init(Engine &eng, uint32_t bit_length, uint32_t max_aggregation_factor, PedersenGens pc) {
    auto p = std::shared_ptr<RangeParameters>(new RangeParameters(eng));
    check(bpp_params_create(eng.ctx(), bit_length, max_aggregation_factor, static_cast<uint32_t>(pc.extension_degree), nullptr, nullptr,
                            &p->handle_), bpp_ctx_last_error(eng.ctx()));
    p->n_ = bit_length;
    p->m_ = max_aggregation_factor;
    p->t_ = static_cast<uint32_t>(pc.extension_degree);
    return p;
  }
  // Arc::clone for another context of the same device (bpp_params_retain): the same device tables, usable from `eng`
  // concurrently with every other holder (`Precomputation: Send + Sync`, src/traits.rs:42)
  std::shared_ptr<RangeParameters> share(Engine &eng) const {
    check(bpp_params_retain(eng.ctx(), handle_), bpp_ctx_last_error(eng.ctx()));
    auto p = std::shared_ptr<RangeParameters>(new RangeParameters(eng));
    p->handle_ = handle_;
    p->n_ = n_;
    p->m_ = m_;
    p->t_ = t_;
    return p;
  }
  ~RangeParameters() {
    if (handle_) bpp_params_destroy(eng_.ctx(), handle_);
  }
  uint32_t bit_length() const { return n_; }
  uint32_t max_aggregation_factor() const { return m_; }
  ExtensionDegree extension_degree() const { return static_cast<ExtensionDegree>(t_); }
  uint64_t handle() const { return handle_; }
  Engine &engine() const { return eng_; }
  // PedersenGens::commit(value, blindings)
  Bytes32 commit(uint64_t value, const std::vector<Bytes32> &blindings) const {
    Bytes32 out{};
    check(bpp_pedersen_commit(eng_.ctx(), handle_, &value, blindings.empty() ? nullptr : blindings[0].data(),
                              static_cast<uint32_t>(blindings.size()), 1, out.data()), bpp_ctx_last_error(eng_.ctx()));
    return out;
  }

 private:
  explicit RangeParameters(Engine &e) : eng_(e) {}
  Engine &eng_;
  uint64_t handle_ = 0;
  uint32_t n_ = 0, m_ = 0, t_ = 0;
};

struct RangeStatement {
  std::shared_ptr<RangeParameters> generators;
  std::vector<Bytes32> commitments_compressed;
  std::vector<std::optional<uint64_t>> minimum_value_promises;
  std::optional<Bytes32> seed_nonce;
  // src/range_statement.rs:36-73
  static RangeStatement init(std::shared_ptr<RangeParameters> generators, std::vector<Bytes32> commitments,
                             std::vector<std::optional<uint64_t>> minimum_value_promises, std::optional<Bytes32> seed_nonce) {
    const size_t n = commitments.size();
    if (n == 0 || (n & (n - 1))) throw ProofError(ProofErrorKind::InvalidArgument, "Number of commitments must be a power of two");
    if (minimum_value_promises.size() != n) throw ProofError(ProofErrorKind::InvalidArgument, "Incorrect number of minimum value promises");
    if (generators->max_aggregation_factor() < n) throw ProofError(ProofErrorKind::InvalidArgument, "Not enough generators for this statement");
    if (seed_nonce && n > 1) throw ProofError(ProofErrorKind::InvalidArgument, "Mask recovery is not supported with an aggregated statement");
    return RangeStatement{std::move(generators), std::move(commitments), std::move(minimum_value_promises), seed_nonce};
  }
};

struct CommitmentOpening {
  uint64_t v;
  std::vector<Bytes32> r;
  static CommitmentOpening create(uint64_t v, std::vector<Bytes32> r) { return CommitmentOpening{v, std::move(r)}; }
};

struct RangeWitness {
  std::vector<CommitmentOpening> openings;
  uint32_t extension_degree;
  // src/range_witness.rs:24-41
  static RangeWitness init(std::vector<CommitmentOpening> openings) {
    if (openings.empty()) throw ProofError(ProofErrorKind::InvalidLength, "Vector openings cannot be empty");
    const size_t t = openings[0].r.size();
    for (const auto &o : openings) {
      if (o.r.empty()) throw ProofError(ProofErrorKind::InvalidLength, "Extended blinding factors cannot be empty");
      if (o.r.size() != t) throw ProofError(ProofErrorKind::InvalidLength, "Extended blinding factors must have consistent length");
    }
    if (t < 1 || t > 6) throw ProofError(ProofErrorKind::InvalidArgument, "Extension degree not valid");
    return RangeWitness{std::move(openings), static_cast<uint32_t>(t)};
  }
};

struct ExtendedMask {
  std::vector<Bytes32> blindings;
  bool operator==(const ExtendedMask &o) const { return blindings == o.blindings; }
};

class RangeProof {
 public:
  const std::vector<uint8_t> &to_bytes() const { return raw_; }
  bool operator==(const RangeProof &o) const { return raw_ == o.raw_; }

  // src/range_proof.rs:1155-1257 (structure only here; canonical-scalar checks are repeated by the engine on upload)
  static RangeProof from_bytes(const std::vector<uint8_t> &bytes) {
    if (bytes.empty()) throw ProofError(ProofErrorKind::InvalidLength, "Serialized proof is too short");
    const uint32_t t = bytes[0];
    if (t < 1 || t > 6) throw ProofError(ProofErrorKind::InvalidArgument, "Extension degree not valid");
    const size_t body = bytes.size() - 1, chunks = body / 32;
    if (chunks < t + 5 + 2) throw ProofError(ProofErrorKind::InvalidLength, "Serialized proof is too short");
    if ((body % 32) || ((chunks - t - 5) % 2)) throw ProofError(ProofErrorKind::InvalidLength, "Unused data after deserialization");
    RangeProof p;
    p.raw_ = bytes;
    return p;
  }

  static uint32_t rounds_for(const RangeStatement &st) {
    uint32_t mn = st.generators->bit_length() * static_cast<uint32_t>(st.commitments_compressed.size()), r = 0;
    while ((1u << r) < mn) r++;
    return r;
  }

  // n x prove_with_rng in one engine call; rng_bytes[i] = (rounds + 3) x 32 bytes from the caller's RNG
  static std::vector<RangeProof> prove_batch(const std::vector<Transcript> &transcripts, const std::vector<RangeStatement> &statements,
                                             const std::vector<RangeWitness> &witnesses, const std::vector<std::vector<uint8_t>> &rng_bytes) {
    const size_t n = statements.size();
    if (n == 0 || witnesses.size() != n || transcripts.size() != n || rng_bytes.size() != n)
      throw ProofError(ProofErrorKind::InvalidArgument, "Range statements, witnesses, transcripts length mismatch");
    auto &params = *statements[0].generators;
    std::vector<bpp_prove_item> items(n);
    std::vector<std::vector<uint64_t>> vals(n), mins(n);
    std::vector<std::vector<uint8_t>> blind(n), comm(n), pres(n);
    for (size_t i = 0; i < n; i++) {
      const auto &st = statements[i];
      const auto &w = witnesses[i];
      const size_t m = st.commitments_compressed.size();
      if (w.openings.size() != m) throw ProofError(ProofErrorKind::InvalidLength, "Witness openings and statement commitments do not match!");
      if (w.extension_degree != static_cast<uint32_t>(params.extension_degree()))
        throw ProofError(ProofErrorKind::InvalidLength, "Witness and statement extension degrees do not match!");
      for (size_t j = 0; j < m; j++) {
        vals[i].push_back(w.openings[j].v);
        for (const auto &r : w.openings[j].r) blind[i].insert(blind[i].end(), r.begin(), r.end());
        comm[i].insert(comm[i].end(), st.commitments_compressed[j].begin(), st.commitments_compressed[j].end());
        mins[i].push_back(st.minimum_value_promises[j].value_or(0));
        pres[i].push_back(st.minimum_value_promises[j] ? 1 : 0);
      }
      bpp_prove_item &it = items[i];
      memset(&it, 0, sizeof(it));
      it.values = vals[i].data();
      it.blindings32 = blind[i].data();
      it.commitments32 = comm[i].data();
      it.m = static_cast<uint32_t>(m);
      it.min_values = mins[i].data();
      it.min_present = pres[i].data();
      it.seed_nonce32 = st.seed_nonce ? st.seed_nonce->data() : nullptr;
      fill_transcript(transcripts[i], it.transcript_state, it.transcript_label, it.label_len);
      it.rng_bytes = rng_bytes[i].data();
      it.rng_len = rng_bytes[i].size();
    }
    const size_t stride = 1 + 32 * (6 + 5 + 2 * 12);
    std::vector<uint8_t> out(stride * n);
    size_t plen = 0;
    char err[256] = {0};
    check(bpp_prove_batch(params.engine().ctx(), params.handle(), items.data(), n, out.data(), stride, &plen, err, sizeof(err)), err);
    std::vector<RangeProof> proofs;
    for (size_t i = 0; i < n; i++) proofs.push_back(from_bytes(std::vector<uint8_t>(out.begin() + i * stride, out.begin() + i * stride + plen)));
    return proofs;
  }
  static RangeProof prove_with_rng(const Transcript &transcript, const RangeStatement &statement, const RangeWitness &witness,
                                   const std::vector<uint8_t> &rng_bytes) {
    return prove_batch({transcript}, {statement}, {witness}, {rng_bytes})[0];
  }

  // src/range_proof.rs:712-752; every `chunk` proofs are one reference batch (all chunks are verified)
  static std::vector<std::optional<ExtendedMask>> verify_batch(const std::vector<Transcript> &transcripts,
                                                                const std::vector<RangeStatement> &statements,
                                                                const std::vector<RangeProof> &proofs, VerifyAction action,
                                                                size_t chunk = BPP_REFERENCE_CHUNK) {
    if (statements.empty() || proofs.empty() || transcripts.empty())
      throw ProofError(ProofErrorKind::InvalidArgument, "Range statements or proofs length empty");
    if (statements.size() != proofs.size()) throw ProofError(ProofErrorKind::InvalidArgument, "Range statements and proofs length mismatch");
    if (transcripts.size() != statements.size())
      throw ProofError(ProofErrorKind::InvalidArgument, "Range statements and transcripts length mismatch");
    const size_t n = proofs.size();
    // the statement with the largest generator capacity carries the tables (src/range_proof.rs:666-673, :778);
    // generators of smaller capacities are prefixes of it (bulletproof_gens.rs:24-41)
    const RangeParameters *largest = statements[0].generators.get();
    for (const auto &st : statements)
      if (st.generators->max_aggregation_factor() > largest->max_aggregation_factor()) largest = st.generators.get();
    const RangeParameters &params = *largest;
    const uint32_t t = static_cast<uint32_t>(params.extension_degree());
    std::vector<bpp_verify_item> items(n);
    std::vector<std::vector<uint64_t>> mins(n);
    std::vector<std::vector<uint8_t>> comm(n), pres(n);
    for (size_t i = 0; i < n; i++) {
      const auto &st = statements[i];
      if (st.generators->bit_length() != params.bit_length())
        throw ProofError(ProofErrorKind::InvalidArgument, "Inconsistent bit length in batch statement");
      if (st.generators->extension_degree() != params.extension_degree())
        throw ProofError(ProofErrorKind::InvalidArgument, "Inconsistent extension degree");
      for (size_t j = 0; j < st.commitments_compressed.size(); j++) {
        comm[i].insert(comm[i].end(), st.commitments_compressed[j].begin(), st.commitments_compressed[j].end());
        mins[i].push_back(st.minimum_value_promises[j].value_or(0));
        pres[i].push_back(st.minimum_value_promises[j] ? 1 : 0);
      }
      bpp_verify_item &it = items[i];
      memset(&it, 0, sizeof(it));
      it.proof = proofs[i].raw_.data();
      it.proof_len = proofs[i].raw_.size();
      it.commitments32 = comm[i].data();
      it.m = static_cast<uint32_t>(st.commitments_compressed.size());
      it.min_values = mins[i].data();
      it.min_present = pres[i].data();
      it.seed_nonce32 = st.seed_nonce ? st.seed_nonce->data() : nullptr;
      fill_transcript(transcripts[i], it.transcript_state, it.transcript_label, it.label_len);
    }
    std::vector<uint8_t> masks(n * t * 32), present(n);
    char err[256] = {0};
    check(bpp_verify_batch(params.engine().ctx(), params.handle(), items.data(), n, static_cast<int>(action), chunk, masks.data(),
                           present.data(), err, sizeof(err)), err);
    std::vector<std::optional<ExtendedMask>> out(n);
    for (size_t i = 0; i < n; i++) {
      if (!present[i]) continue;
      ExtendedMask em;
      for (uint32_t k = 0; k < t; k++) {
        Bytes32 b;
        memcpy(b.data(), &masks[(i * t + k) * 32], 32);
        em.blindings.push_back(b);
      }
      out[i] = em;
    }
    return out;
  }

 private:
  static void fill_transcript(const Transcript &tr, const uint8_t *&state, const uint8_t *&label, size_t &len) {
    if (!tr.state().empty()) {
      state = tr.state().data();
    } else {
      label = reinterpret_cast<const uint8_t *>(tr.label().data());
      len = tr.label().size();
    }
  }
  std::vector<uint8_t> raw_;
};

}  // namespace bpp_host
