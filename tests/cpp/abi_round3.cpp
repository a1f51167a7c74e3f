// A compiled caller of the round-3 entry points of include/bpp.h, as a Rust host would bind them (the Python tests reach the
// library through ctypes declarations of their own, which cannot notice a wrong struct layout or prototype in the header):
//   bpp_verify_batch_packed == bpp_verify_batch on the same proofs, bpp_verify_submit_packed / bpp_verify_collect with three
//   tickets collected out of order, bpp_batch_secret_bytes, bpp_ctx_set_option, and the sharded entries over the in-process
//   communicator (bpp_comm_create_local, one rank): bpp_verify_sharded, bpp_verify_sharded_wave (two contexts),
//   bpp_verify_sharded_groups (four groups, one tampered), bpp_verify_sharded_groups_wave (two slots) with their
//   bpp_shard_result records; bpp_verify_resident_groups (ragged groups) and bpp_batcher (six threads).
// Proofs come from the engine's own prover through the C++ mirror (include/bpp.hpp).  Built and run by
// tests/test_gpu_cpp_mirror.py on the GPU box.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "bpp.hpp"

using namespace bpp_host;

#define CHECK(c)                                                             \
  do {                                                                       \
    if (!(c)) {                                                              \
      fprintf(stderr, "CHECK failed: %s (%s:%d)\n", #c, __FILE__, __LINE__); \
      exit(1);                                                               \
    }                                                                        \
  } while (0)

struct Rng {  // splitmix64: test data only
  uint64_t s;
  uint64_t next_u64() {
    uint64_t z = (s += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
  }
  Bytes32 scalar() {
    Bytes32 b{};
    for (int i = 0; i < 31; i++) b[i] = (uint8_t)next_u64();
    b[0] |= 1;
    return b;
  }
};

int main() {
  Engine eng(0);
  const std::string label = "abi round 3";
  const uint32_t N = 200, n_bits = 64;
  auto params = RangeParameters::init(eng, n_bits, 1, create_pedersen_gens_with_extension_degree(ExtensionDegree::DefaultPedersen));
  Rng rng{20260704};
  std::vector<RangeStatement> statements;
  std::vector<RangeWitness> witnesses;
  std::vector<Transcript> transcripts;
  std::vector<std::vector<uint8_t>> ext;
  std::vector<uint64_t> min_values(N);
  std::vector<uint8_t> min_present(N), commitments(32 * N), seeds(32 * N);
  for (uint32_t i = 0; i < N; i++) {
    const uint64_t v = rng.next_u64() >> 1;
    const Bytes32 r = rng.scalar(), seed = rng.scalar();
    const Bytes32 c = params->commit(v, {r});
    memcpy(&commitments[32 * i], c.data(), 32);
    memcpy(&seeds[32 * i], seed.data(), 32);
    min_present[i] = (i % 3) != 0;
    min_values[i] = min_present[i] ? v / 3 : 0;
    statements.push_back(RangeStatement::init(params, {c}, {min_present[i] ? std::optional<uint64_t>(min_values[i]) : std::nullopt}, seed));
    witnesses.push_back(RangeWitness::init({CommitmentOpening::create(v, {r})}));
    transcripts.push_back(Transcript::create(label));
    std::vector<uint8_t> e(32 * (6 + 3));
    for (auto &x : e) x = (uint8_t)rng.next_u64();
    ext.push_back(e);
  }
  const auto proofs = RangeProof::prove_batch(transcripts, statements, witnesses, ext);
  const size_t plen = proofs[0].to_bytes().size();
  std::vector<uint8_t> flat(plen * N);
  for (uint32_t i = 0; i < N; i++) {
    CHECK(proofs[i].to_bytes().size() == plen);
    memcpy(&flat[plen * i], proofs[i].to_bytes().data(), plen);
  }
  bpp_ctx *ctx = eng.ctx();
  char err[256] = {0};

  auto packed_of = [&](const uint8_t *pr, size_t first, size_t n, bool with_seeds) {
    bpp_packed_batch in;
    memset(&in, 0, sizeof(in));
    in.n_items = n;
    in.proofs = pr + plen * first;
    in.proof_len = plen;
    in.proof_stride = plen;
    in.commitments32 = &commitments[32 * first];
    in.m = 1;
    in.min_values = &min_values[first];
    in.min_present = &min_present[first];
    in.seed_nonces32 = with_seeds ? &seeds[32 * first] : nullptr;
    in.transcript_label = (const uint8_t *)label.data();
    in.label_len = label.size();
    return in;
  };

  // ---- packed == items (masks included)
  std::vector<uint8_t> masks_a(32 * N), masks_b(32 * N), pres_a(N), pres_b(N);
  {
    bpp_packed_batch in = packed_of(flat.data(), 0, N, true);
    CHECK(bpp_verify_batch_packed(ctx, params->handle(), &in, BPP_RECOVER_AND_VERIFY, 64, masks_a.data(), pres_a.data(), err, sizeof(err)) == BPP_OK);
    std::vector<bpp_verify_item> items(N);
    for (uint32_t i = 0; i < N; i++) {
      memset(&items[i], 0, sizeof(items[i]));
      items[i].proof = &flat[plen * i];
      items[i].proof_len = plen;
      items[i].commitments32 = &commitments[32 * i];
      items[i].m = 1;
      items[i].min_values = &min_values[i];
      items[i].min_present = &min_present[i];
      items[i].seed_nonce32 = &seeds[32 * i];
      items[i].transcript_label = (const uint8_t *)label.data();
      items[i].label_len = label.size();
    }
    CHECK(bpp_verify_batch(ctx, params->handle(), items.data(), N, BPP_RECOVER_AND_VERIFY, 64, masks_b.data(), pres_b.data(), err, sizeof(err)) == BPP_OK);
    CHECK(masks_a == masks_b && pres_a == pres_b);
    for (uint32_t i = 0; i < N; i++) CHECK(pres_a[i] == 1 && memcmp(&masks_a[32 * i], witnesses[i].openings[0].r[0].data(), 32) == 0);
    uint64_t nonzero = 1;
    CHECK(bpp_batch_secret_bytes(ctx, 0, &nonzero) == BPP_OK && nonzero == 0);  // no seed nonce or mask left in the buffers the context kept
  }
  // ---- a tampered proof: same kind from both forms, message in errbuf
  std::vector<uint8_t> bad = flat;
  bad[plen * 77 + 1 + 32 + 96] ^= 1;  // r1 of proof 77
  {
    bpp_packed_batch in = packed_of(bad.data(), 0, N, false);
    CHECK(bpp_verify_batch_packed(ctx, params->handle(), &in, BPP_VERIFY_ONLY, 0, nullptr, nullptr, err, sizeof(err)) == BPP_ERR_VERIFICATION_FAILED);
    CHECK(strstr(err, "not valid") != nullptr);
  }
  // ---- pipeline: three tickets, collected out of order; the middle one is the tampered input
  {
    CHECK(bpp_ctx_pipeline_depth(ctx, 3) == BPP_OK);
    uint64_t t[3];
    bpp_packed_batch in0 = packed_of(flat.data(), 0, 100, false), in1 = packed_of(bad.data(), 0, N, false), in2 = packed_of(flat.data(), 100, 100, true);
    CHECK(bpp_verify_submit_packed(ctx, params->handle(), &in0, BPP_VERIFY_ONLY, 0, &t[0], err, sizeof(err)) == BPP_OK);
    CHECK(bpp_verify_submit_packed(ctx, params->handle(), &in1, BPP_VERIFY_ONLY, 50, &t[1], err, sizeof(err)) == BPP_OK);
    CHECK(bpp_verify_submit_packed(ctx, params->handle(), &in2, BPP_RECOVER_AND_VERIFY, 0, &t[2], err, sizeof(err)) == BPP_OK);
    std::vector<uint8_t> m2(32 * 100), p2(100);
    CHECK(bpp_verify_collect(ctx, t[2], m2.data(), p2.data(), err, sizeof(err)) == BPP_OK);
    CHECK(memcmp(m2.data(), &masks_a[32 * 100], 32 * 100) == 0);
    CHECK(bpp_verify_collect(ctx, t[0], nullptr, nullptr, err, sizeof(err)) == BPP_OK);
    CHECK(bpp_verify_collect(ctx, t[1], nullptr, nullptr, err, sizeof(err)) == BPP_ERR_VERIFICATION_FAILED);
    CHECK(bpp_verify_collect(ctx, t[1], nullptr, nullptr, err, sizeof(err)) < 0);  // a ticket is collected once
  }
  // ---- options are per context and named
  CHECK(bpp_ctx_set_option(ctx, "msm_split", 0) == BPP_OK);
  CHECK(bpp_ctx_set_option(ctx, "no such option", 1) != BPP_OK);
  CHECK(bpp_ctx_set_option(ctx, "msm_split", -1) == BPP_OK);
  // ---- the sharded entries over the in-process communicator, one rank
  {
    bpp_comm *comm = nullptr;
    CHECK(bpp_comm_create_local(ctx, 77001, 0, 1, &comm) == BPP_OK);
    uint64_t h = 0;
    bpp_packed_batch in = packed_of(flat.data(), 0, N, false);
    CHECK(bpp_batch_upload_packed(ctx, params->handle(), &in, &h, err, sizeof(err)) == BPP_OK);
    const uint32_t counts[1] = {N};
    int tier = -1, rank = -2;
    CHECK(bpp_verify_sharded(comm, ctx, h, counts, &tier, &rank, err, sizeof(err)) == BPP_OK && tier == BPP_TIER_NONE);
    // four groups of 50: every one accepted
    const uint32_t c50[1] = {50};
    bpp_shard_result res[4];
    memset(res, 0xff, sizeof(res));
    CHECK(bpp_verify_sharded_groups(comm, ctx, h, 4, c50, res) == BPP_OK);
    for (int g = 0; g < 4; g++) CHECK(res[g].code == BPP_OK && res[g].tier == BPP_TIER_NONE && res[g].msg[0] == 0);
    CHECK(bpp_batch_destroy(ctx, h) == BPP_OK);
    // the tampered input: group 1 (proofs 50..99) fails in its final check, a non-canonical point in group 3 is a PASS-2 finding
    std::vector<uint8_t> bad2 = bad;
    memset(&bad2[plen * 160 + 1 + 32], 0, 32);
    bad2[plen * 160 + 1 + 32] = 1;  // A of proof 160: a negative field element, never a valid encoding
    bpp_packed_batch inb = packed_of(bad2.data(), 0, N, false);
    CHECK(bpp_batch_upload_packed(ctx, params->handle(), &inb, &h, err, sizeof(err)) == BPP_OK);
    CHECK(bpp_verify_sharded_groups(comm, ctx, h, 4, c50, res) == BPP_OK);
    CHECK(res[0].code == BPP_OK && res[2].code == BPP_OK);
    CHECK(res[1].code == BPP_ERR_VERIFICATION_FAILED && res[1].tier == BPP_TIER_MSM);
    CHECK(res[3].code == BPP_ERR_INVALID_ARGUMENT && res[3].tier == BPP_TIER_PASS2 && res[3].rank == 0 && res[3].index == 10);
    CHECK(strstr(res[3].msg, "canonical") != nullptr);
    CHECK(bpp_verify_sharded(comm, ctx, h, counts, &tier, &rank, err, sizeof(err)) == BPP_ERR_INVALID_ARGUMENT && tier == BPP_TIER_PASS2 && rank == 0);
    CHECK(bpp_batch_destroy(ctx, h) == BPP_OK);
    // a wave of two batches on two contexts
    Engine eng2(0);
    auto params2 = params->share(eng2);
    uint64_t h1 = 0, h2 = 0;
    bpp_packed_batch w1 = packed_of(flat.data(), 0, 100, false), w2 = packed_of(bad.data(), 0, 100, false);
    CHECK(bpp_batch_upload_packed(ctx, params->handle(), &w1, &h1, err, sizeof(err)) == BPP_OK);
    CHECK(bpp_batch_upload_packed(eng2.ctx(), params2->handle(), &w2, &h2, err, sizeof(err)) == BPP_OK);
    bpp_ctx *ctxs[2] = {ctx, eng2.ctx()};
    const uint64_t hs[2] = {h1, h2};
    const uint32_t c100[1] = {100};
    bpp_shard_result wres[2];
    CHECK(bpp_verify_sharded_wave(comm, ctxs, hs, 2, c100, wres) == BPP_OK);
    CHECK(wres[0].code == BPP_OK && wres[1].code == BPP_ERR_VERIFICATION_FAILED && wres[1].tier == BPP_TIER_MSM);
    bpp_shard_timing tm;
    CHECK(bpp_comm_last_timing(comm, &tm) == BPP_OK && tm.batches == 2);
    // the same two batches as two slots of one pipelined grouped call, two groups of 50 each
    bpp_shard_result gres[4];
    CHECK(bpp_verify_sharded_groups_wave(comm, ctxs, hs, 2, 2, c50, gres) == BPP_OK);
    CHECK(gres[0].code == BPP_OK && gres[1].code == BPP_OK);                                       // slot 0: clean
    CHECK(gres[2].code == BPP_OK && gres[3].code == BPP_ERR_VERIFICATION_FAILED && gres[3].tier == BPP_TIER_MSM);  // slot 1: proof 77
    CHECK(bpp_batch_destroy(ctx, h1) == BPP_OK && bpp_batch_destroy(eng2.ctx(), h2) == BPP_OK);
    bpp_comm_destroy(comm);
  }
  // ---- reference batches of different sizes as the groups of one call; the pool of many callers' small calls
  {
    uint64_t h = 0;
    bpp_packed_batch in = packed_of(bad.data(), 0, N, false);  // proof 77 tampered
    CHECK(bpp_batch_upload_packed(ctx, params->handle(), &in, &h, err, sizeof(err)) == BPP_OK);
    const uint32_t first[4] = {0, 10, 120, N};  // groups of 10, 110 and 80 proofs
    bpp_shard_result res[3];
    CHECK(bpp_verify_resident_groups(ctx, h, first, 3, res) == BPP_OK);
    CHECK(res[0].code == BPP_OK && res[2].code == BPP_OK && res[1].code == BPP_ERR_VERIFICATION_FAILED && res[1].tier == BPP_TIER_MSM);
    CHECK(bpp_batch_destroy(ctx, h) == BPP_OK);
    bpp_batcher *bat = nullptr;
    bpp_packed_batch shape = packed_of(flat.data(), 0, 1, false);
    CHECK(bpp_batcher_create(ctx, params->handle(), &shape, 2, 0, 0, &bat) == BPP_OK);
    std::atomic<int> wrong{0};
    std::vector<std::thread> th;
    for (int k = 0; k < 6; k++)
      th.emplace_back([&, k] {
        char e2[256];
        for (int i = 0; i < 25; i++) {
          const bool use_bad = ((k + i) % 3) == 0;
          // [60, 100) of `bad` holds proof 77; [100, 140) does not
          bpp_packed_batch mine = packed_of(use_bad ? bad.data() : flat.data(), use_bad ? 60 : 100, 40, false);
          const int rc = bpp_batcher_verify(bat, &mine, e2, sizeof(e2));
          if (rc != (use_bad ? BPP_ERR_VERIFICATION_FAILED : BPP_OK)) wrong++;
        }
      });
    for (auto &t : th) t.join();
    CHECK(wrong == 0);
    uint64_t pooled = 0, ecalls = 0, solo = 0;
    CHECK(bpp_batcher_stats(bat, &pooled, &ecalls, &solo) == BPP_OK && ecalls > 0 && ecalls <= 150);
    bpp_batcher_destroy(bat);
  }
  printf("abi_round3 ok\n");
  return 0;
}
