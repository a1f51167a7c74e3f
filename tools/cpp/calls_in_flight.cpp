// S host threads, each with its own context, each verifying ONE resident reference batch of N proofs per call as fast as it can:
// the rate of independent small calls without Python in the loop (tools/wide_probe.py is the same from Python threads).
//   g++ -std=c++17 -O2 -I include tools/cpp/calls_in_flight.cpp -o /tmp/calls_in_flight -L bulletproofs-plus_amd -lbpp_hip -lpthread
//   /tmp/calls_in_flight [N=256] [seconds=2] [S list: 1 4 8 16 32]
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "bpp.hpp"

using namespace bpp_host;

int main(int argc, char **argv) {
  const uint32_t N = argc > 1 ? (uint32_t)atoi(argv[1]) : 256u;
  const double seconds = argc > 2 ? atof(argv[2]) : 2.0;
  std::vector<int> S_list;
  for (int i = 3; i < argc; i++) S_list.push_back(atoi(argv[i]));
  if (S_list.empty()) S_list = {1, 4, 8, 16, 32};
  Engine eng(0);
  const std::string label = "calls in flight";
  auto params = RangeParameters::init(eng, 64, 1, create_pedersen_gens_with_extension_degree(ExtensionDegree::DefaultPedersen));
  uint64_t seed = 99;
  auto next = [&]() {
    uint64_t z = (seed += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
  };
  std::vector<RangeStatement> statements;
  std::vector<RangeWitness> witnesses;
  std::vector<Transcript> transcripts;
  std::vector<std::vector<uint8_t>> ext;
  std::vector<uint64_t> min_values(N, 0);
  std::vector<uint8_t> min_present(N, 0), commitments(32 * N);
  for (uint32_t i = 0; i < N; i++) {
    Bytes32 r{};
    for (int k = 0; k < 31; k++) r[k] = (uint8_t)next();
    r[0] |= 1;
    const uint64_t v = next() >> 1;
    const Bytes32 c = params->commit(v, {r});
    memcpy(&commitments[32 * i], c.data(), 32);
    statements.push_back(RangeStatement::init(params, {c}, {std::nullopt}, std::nullopt));
    witnesses.push_back(RangeWitness::init({CommitmentOpening::create(v, {r})}));
    transcripts.push_back(Transcript::create(label));
    std::vector<uint8_t> e(32 * 9);
    for (auto &x : e) x = (uint8_t)next();
    ext.push_back(e);
  }
  const auto proofs = RangeProof::prove_batch(transcripts, statements, witnesses, ext);
  const size_t plen = proofs[0].to_bytes().size();
  std::vector<uint8_t> flat(plen * N);
  for (uint32_t i = 0; i < N; i++) memcpy(&flat[plen * i], proofs[i].to_bytes().data(), plen);
  bpp_packed_batch in;
  memset(&in, 0, sizeof(in));
  in.n_items = N;
  in.proofs = flat.data();
  in.proof_len = in.proof_stride = plen;
  in.commitments32 = commitments.data();
  in.m = 1;
  in.min_values = min_values.data();
  in.min_present = min_present.data();
  in.transcript_label = (const uint8_t *)label.data();
  in.label_len = label.size();
  for (int host = 0; host < 3; host++)  // 0: resident batches, 1: host buffers in (bpp_verify_batch_packed), 2: the same through ONE bpp_batcher
    for (int S : S_list) {
      std::vector<std::unique_ptr<Engine>> engs;
      std::vector<std::shared_ptr<RangeParameters>> pars;
      std::vector<uint64_t> handles(S, 0);
      char err[256];
      for (int k = 0; k < S; k++) {
        engs.emplace_back(new Engine(0));
        pars.push_back(params->share(*engs[k]));
        if (!host && bpp_batch_upload_packed(engs[k]->ctx(), pars[k]->handle(), &in, &handles[k], err, sizeof(err)) != BPP_OK) return 1;
      }
      bpp_batcher *bat = nullptr;
      if (host == 2 && bpp_batcher_create(eng.ctx(), params->handle(), &in, getenv("BATCHER_LANES") ? atoi(getenv("BATCHER_LANES")) : 2, getenv("BATCHER_WAIT_US") ? atoi(getenv("BATCHER_WAIT_US")) : 0, 64, &bat) != BPP_OK) return 1;
      auto call = [&](int k) {
        char e2[256];
        const int rc = host == 2 ? bpp_batcher_verify(bat, &in, e2, sizeof(e2))
                       : host  ? bpp_verify_batch_packed(engs[k]->ctx(), pars[k]->handle(), &in, BPP_VERIFY_ONLY, 0, nullptr, nullptr, e2, sizeof(e2))
                               : bpp_verify_resident(engs[k]->ctx(), handles[k], BPP_VERIFY_ONLY, 0, nullptr, nullptr, e2, sizeof(e2));
        if (rc != BPP_OK) {
          fprintf(stderr, "call failed: %d %s\n", rc, e2);
          exit(1);
        }
      };
      for (int k = 0; k < S; k++)
        for (int i = 0; i < 5; i++) call(k);
      std::atomic<uint64_t> total{0};
      const auto t0 = std::chrono::steady_clock::now();
      const auto stop = t0 + std::chrono::duration<double>(seconds);
      std::vector<std::thread> th;
      for (int k = 0; k < S; k++)
        th.emplace_back([&, k] {
          uint64_t c = 0;
          while (std::chrono::steady_clock::now() < stop) {
            call(k);
            c++;
          }
          total += c;
        });
      for (auto &t : th) t.join();
      const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      uint64_t pooled = 0, ecalls = 0, solo = 0;
      if (bat) bpp_batcher_stats(bat, &pooled, &ecalls, &solo);
      printf("{\"proofs_per_call\": %u, \"form\": \"%s\", \"host_buffers_in\": %s, \"in_flight\": %d, \"calls_per_s\": %.1f, \"proofs_per_s\": %.0f, \"ms_per_call_per_context\": %.3f, \"engine_calls\": %llu}\n", N,
             host == 2 ? "batcher" : host ? "packed" : "resident", host ? "true" : "false", S, total / el, N * (double)total / el, 1e3 * el * S / (double)total, (unsigned long long)ecalls);
      if (bat) bpp_batcher_destroy(bat);
      fflush(stdout);
      for (int k = 0; k < S; k++)
        if (!host) bpp_batch_destroy(engs[k]->ctx(), handles[k]);
      pars.clear();
      engs.clear();
    }
  return 0;
}
