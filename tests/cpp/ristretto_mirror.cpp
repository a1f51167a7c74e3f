// C++ restatement of the reference's integration test tests/ristretto.rs (prove_and_verify, :152-373) on top of the
// host-side mirror include/bpp.hpp -> C ABI -> libbpp_hip.so.  Proofs are made by the engine's own prover.
// Built and run by tests/test_gpu_cpp_mirror.py on the GPU box.
#include <cstdio>
#include <cstdlib>

#include "bpp.hpp"

using namespace bpp_host;

struct Rng {  // splitmix64: test data only
  uint64_t s;
  uint64_t next_u64() {
    uint64_t z = (s += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
  }
  Bytes32 scalar() {  // non-zero, < 2^248 < l
    Bytes32 b{};
    for (int i = 0; i < 31; i++) b[i] = (uint8_t)next_u64();
    b[0] |= 1;
    return b;
  }
  std::vector<uint8_t> bytes(size_t n) {
    std::vector<uint8_t> v(n);
    for (auto &x : v) x = (uint8_t)next_u64();
    return v;
  }
};

#define CHECK(c)                                                        \
  do {                                                                  \
    if (!(c)) {                                                         \
      fprintf(stderr, "CHECK failed: %s (%s:%d)\n", #c, __FILE__, __LINE__); \
      exit(1);                                                          \
    }                                                                   \
  } while (0)

enum class Strategy { NoOffset, Intermediate, EqualToValue, LargerThanValue };

static void prove_and_verify(Engine &eng, const std::vector<uint32_t> &bit_lengths, const std::vector<uint32_t> &proof_batch,
                             ExtensionDegree degree, Strategy strategy) {
  Rng rng{8675309};
  const std::string label = "BatchedRangeProofTest";
  const uint32_t t = static_cast<uint32_t>(degree);
  for (uint32_t bit_length : bit_lengths) {
    std::vector<std::optional<ExtendedMask>> private_masks, public_masks;
    std::vector<RangeStatement> statements_private, statements_public;
    std::vector<RangeProof> proofs;
    std::vector<Transcript> transcripts;
    const uint64_t value_max = 1ULL << (bit_length - 1);
    for (uint32_t aggregation : proof_batch) {
      auto generators = RangeParameters::init(eng, bit_length, aggregation, create_pedersen_gens_with_extension_degree(degree));
      std::vector<CommitmentOpening> openings;
      std::vector<Bytes32> commitments;
      std::vector<std::optional<uint64_t>> minimum_values;
      std::optional<ExtendedMask> mask;
      for (uint32_t m = 0; m < aggregation; m++) {
        const uint64_t value = rng.next_u64() % value_max;
        switch (strategy) {
          case Strategy::NoOffset: minimum_values.push_back(std::nullopt); break;
          case Strategy::Intermediate: minimum_values.push_back(value / 3); break;
          case Strategy::EqualToValue: minimum_values.push_back(value); break;
          case Strategy::LargerThanValue: minimum_values.push_back(value + 1); break;
        }
        std::vector<Bytes32> blindings(t, rng.scalar());
        commitments.push_back(generators->commit(value, blindings));
        openings.push_back(CommitmentOpening::create(value, blindings));
        if (m == 0 && aggregation == 1) mask = ExtendedMask{blindings};
      }
      auto witness = RangeWitness::init(openings);
      std::optional<Bytes32> seed_nonce;
      if (aggregation == 1) seed_nonce = rng.scalar();
      auto private_statement = RangeStatement::init(generators, commitments, minimum_values, seed_nonce);
      auto public_statement = RangeStatement::init(generators, commitments, minimum_values, std::nullopt);
      auto transcript = Transcript::create(label);
      const auto ext = rng.bytes(32 * (RangeProof::rounds_for(private_statement) + 3));
      if (strategy == Strategy::LargerThanValue) {
        try {
          RangeProof::prove_with_rng(transcript, private_statement, witness, ext);
          CHECK(!"expected an error here");
        } catch (const ProofError &e) {
          CHECK(e.kind == ProofErrorKind::InvalidArgument);
        }
        continue;
      }
      proofs.push_back(RangeProof::prove_with_rng(transcript, private_statement, witness, ext));
      statements_private.push_back(private_statement);
      statements_public.push_back(public_statement);
      transcripts.push_back(transcript);
      private_masks.push_back(mask);
      public_masks.push_back(std::nullopt);
    }
    if (proofs.empty()) continue;
    // 5. verify as the commitment owner
    CHECK(RangeProof::verify_batch(transcripts, statements_private, proofs, VerifyAction::RecoverOnly) == private_masks);
    CHECK(RangeProof::verify_batch(transcripts, statements_private, proofs, VerifyAction::RecoverAndVerify) == private_masks);
    CHECK(RangeProof::verify_batch(transcripts, statements_private, proofs, VerifyAction::VerifyOnly) == public_masks);
    // 6. public entity
    CHECK(RangeProof::verify_batch(transcripts, statements_public, proofs, VerifyAction::VerifyOnly) == public_masks);
    // 7. wrong seed nonce: Ok, but different masks
    bool any_seed = false;
    auto changed = statements_private;
    for (auto &s : changed)
      if (s.seed_nonce) {
        any_seed = true;
        (*s.seed_nonce)[1] ^= 1;
      }
    if (any_seed) CHECK(!(RangeProof::verify_batch(transcripts, changed, proofs, VerifyAction::RecoverAndVerify) == private_masks));
    // 8. meddle with the minimum value promises -> VerificationFailed
    auto bumped = statements_public;
    for (auto &s : bumped)
      for (auto &p : s.minimum_value_promises) p = p ? (*p == UINT64_MAX ? *p : *p + 1) : 1;
    try {
      RangeProof::verify_batch(transcripts, bumped, proofs, VerifyAction::VerifyOnly);
      CHECK(!"range proof should not verify");
    } catch (const ProofError &e) {
      CHECK(e.kind == ProofErrorKind::VerificationFailed);
    }
    // 9. serialization round trip
    for (const auto &p : proofs) CHECK(RangeProof::from_bytes(p.to_bytes()) == p);
  }
}

static int run();
int main() {
  try {
    return run();
  } catch (const ProofError &e) {
    fprintf(stderr, "uncaught ProofError kind=%d: %s\n", (int)e.kind, e.what());
  } catch (const std::exception &e) {
    fprintf(stderr, "uncaught exception: %s\n", e.what());
  }
  return 2;
}
static int run() {
  Engine eng(0);
  const ExtensionDegree D1 = ExtensionDegree::DefaultPedersen, D2 = ExtensionDegree::AddOneBasePoint, D3 = ExtensionDegree::AddTwoBasePoints;
  struct Shape { std::vector<uint32_t> bits, batch; };
  // test_non_aggregated_single_proof_multiple_bit_lengths, test_aggregated_single_proof_multiple_bit_lengths,
  // test_non_aggregated_multiple_proofs_single_bit_length, test_mixed_aggregation_multiple_proofs_single_bit_length
  const Shape shapes[4] = {{{8, 64}, {1}}, {{4, 32}, {4}}, {{64}, {1, 1}}, {{64}, {1, 2}}};
  for (const auto &sh : shapes) {
    prove_and_verify(eng, sh.bits, sh.batch, D1, Strategy::NoOffset);
    prove_and_verify(eng, sh.bits, sh.batch, D2, Strategy::Intermediate);
    prove_and_verify(eng, sh.bits, sh.batch, D3, Strategy::EqualToValue);
    prove_and_verify(eng, sh.bits, sh.batch, D1, Strategy::LargerThanValue);
  }
  // empty / mismatched vectors (src/range_proof.rs:1759-1808)
  try {
    RangeProof::verify_batch({}, {}, {}, VerifyAction::VerifyOnly);
    CHECK(!"empty batch must fail");
  } catch (const ProofError &e) {
    CHECK(e.kind == ProofErrorKind::InvalidArgument);
  }
  printf("ristretto_mirror: all reference integration shapes passed\n");
  return 0;
}
