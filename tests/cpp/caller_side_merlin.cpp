// The compiled rendition of rust/bpp-gpu-shim/patch/range_proof_gpu.rs `gpu::verify_batch(transcripts: &mut [Transcript], ..)`:
// the reference's own signature (src/range_proof.rs:712-717).  The caller keeps its Merlin transcripts -- any label, context
// data appended -- runs PASS 1 of verify() (:811-850, RangeProofTranscript: src/transcripts.rs:59-179) ON THEM, and hands
// the challenges and the 32 transcript-RNG bytes per proof to the engine, which does everything else
// (bpp_verify_batch_with_challenges, SURVEY 8b option (i)).  A Rust host uses the `merlin` crate for this; here the host-side
// Merlin is the product's own merlin.h built for the host.  tests/test_gpu_caller_merlin.py writes the input (statements and
// proofs made by the ORACLE's prover on transcripts with context appended), runs this program and compares everything it
// writes -- verdict, challenges, RNG bytes, masks, and a probe of every caller transcript AFTER the call (they must have
// advanced exactly as the reference's verify() advances them, :757) -- with the oracle.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "bpp.h"
#include "../../bulletproofs-plus_amd/csrc/merlin.h"
#include "../../bulletproofs-plus_amd/csrc/scalar.h"

using namespace bpp;

#define CHECK(c)                                                             \
  do {                                                                       \
    if (!(c)) {                                                              \
      fprintf(stderr, "CHECK failed: %s (%s:%d)\n", #c, __FILE__, __LINE__); \
      exit(2);                                                               \
    }                                                                        \
  } while (0)

struct Reader {
  std::vector<uint8_t> d;
  size_t at = 0;
  uint32_t u32() {
    CHECK(at + 4 <= d.size());
    uint32_t v;
    memcpy(&v, &d[at], 4);
    at += 4;
    return v;
  }
  std::vector<uint8_t> bytes(size_t n) {
    CHECK(at + n <= d.size());
    std::vector<uint8_t> v(d.begin() + at, d.begin() + at + n);
    at += n;
    return v;
  }
};

struct Item {
  uint32_t m;
  std::vector<uint8_t> proof, commitments, min_present, seed;
  std::vector<uint64_t> min_values;
  bool has_seed;
};

static bool all_zero32(const uint8_t *p) {
  uint8_t r = 0;
  for (int i = 0; i < 32; i++) r |= p[i];
  return r == 0;
}
// TranscriptProtocol (src/protocols/transcript_protocol.rs:39-79) on a host Strobe
static void append(Strobe &s, const char *label, const uint8_t *msg, uint32_t n) { merlin_append_message(s, (const uint8_t *)label, (uint32_t)strlen(label), msg, n); }
static void append_u64(Strobe &s, const char *label, uint64_t v) { merlin_append_u64(s, (const uint8_t *)label, (uint32_t)strlen(label), v); }
static bool validate_and_append_point(Strobe &s, const char *label, const uint8_t *p32) {
  if (all_zero32(p32)) return false;  // "Identity element cannot be added to the transcript"
  append(s, label, p32, 32);
  return true;
}
static bool challenge_scalar(Strobe &s, const char *label, uint8_t out32[32]) {
  uint8_t buf[64];
  merlin_challenge_bytes(s, (const uint8_t *)label, (uint32_t)strlen(label), buf, 64);
  sc v;
  sc_mont_from_wide(v, buf);  // Scalar::from_bytes_mod_order_wide
  if (sc_iszero(v)) return false;
  sc_from_mont(v, v);
  sc_store_words(out32, v);
  return true;
}

int main(int argc, char **argv) {
  CHECK(argc == 3);
  Reader in;
  {
    FILE *f = fopen(argv[1], "rb");
    CHECK(f);
    fseek(f, 0, SEEK_END);
    in.d.resize((size_t)ftell(f));
    fseek(f, 0, SEEK_SET);
    CHECK(fread(in.d.data(), 1, in.d.size(), f) == in.d.size());
    fclose(f);
  }
  CHECK(in.u32() == 0x4d435042u);  // "BPCM"
  const uint32_t n = in.u32(), t = in.u32(), n_bits = in.u32(), m_max = in.u32();
  const int action = (int)in.u32();
  const std::vector<uint8_t> label = in.bytes(in.u32());
  const std::vector<uint8_t> ctx_label = in.bytes(in.u32());
  const std::vector<uint8_t> ctx = in.bytes(in.u32());
  const std::vector<uint8_t> h32 = in.bytes(32), g32 = in.bytes(32 * t);
  std::vector<Item> items(n);
  for (auto &it : items) {
    it.m = in.u32();
    it.proof = in.bytes(in.u32());
    it.commitments = in.bytes(32 * it.m);
    it.min_values.resize(it.m);
    for (auto &v : it.min_values) {
      const std::vector<uint8_t> b = in.bytes(8);
      memcpy(&v, b.data(), 8);
    }
    it.min_present = in.bytes(it.m);
    it.has_seed = in.bytes(1)[0] != 0;
    it.seed = in.bytes(32);
  }

  // ---- the caller's transcripts: Transcript::new(label), then whatever the application appended (here: one context message)
  std::vector<Strobe> transcripts(n);
  for (auto &s : transcripts) {
    merlin_new(s, label.data(), (uint32_t)label.size());
    if (!ctx_label.empty()) merlin_append_message(s, ctx_label.data(), (uint32_t)ctx_label.size(), ctx.data(), (uint32_t)ctx.size());
  }

  // ---- PASS 1 on the caller's side (src/range_proof.rs:811-850), advancing the caller's transcripts
  int rc = BPP_OK;
  std::string msg;
  std::vector<std::vector<uint8_t>> chal(n);
  std::vector<uint8_t> rng_out(32 * (size_t)n, 0);
  for (uint32_t i = 0; i < n && rc == BPP_OK; i++) {
    Strobe &s = transcripts[i];
    const Item &it = items[i];
    const uint8_t *pr = it.proof.data();
    const uint32_t rounds = (uint32_t)(((it.proof.size() - 1) / 32 - t - 5) / 2);
    const uint8_t *pd1 = pr + 1, *pA = pr + 1 + 32 * t, *pA1 = pA + 32, *pB = pA + 64, *pr1 = pA + 96, *ps1 = pA + 128, *pLR = pA + 160;
    bool ok = true;
    // RangeProofTranscript::new (src/transcripts.rs:59-121)
    append(s, "dom-sep", (const uint8_t *)"Bulletproofs+ Range Proof", 25);
    ok = ok && validate_and_append_point(s, "H", h32.data());
    for (uint32_t k = 0; k < t && ok; k++) ok = validate_and_append_point(s, "G", &g32[32 * k]);
    if (ok) {
      append_u64(s, "N", n_bits);
      append_u64(s, "T", t);
      append_u64(s, "M", it.m);
      for (uint32_t j = 0; j < it.m; j++) append(s, "Ci", &it.commitments[32 * j], 32);
      for (uint32_t j = 0; j < it.m; j++) append_u64(s, "vi - minimum_value", it.min_present[j] ? it.min_values[j] : 0);
    }
    chal[i].assign(32 * (size_t)(rounds + 3), 0);
    uint8_t *c = chal[i].data();
    // challenges_y_z, challenge_round_e, challenge_final_e (:123-163)
    ok = ok && validate_and_append_point(s, "A", pA) && challenge_scalar(s, "y", c) && challenge_scalar(s, "z", c + 32);
    for (uint32_t j = 0; j < rounds && ok; j++)
      ok = validate_and_append_point(s, "L", pLR + 64 * j) && validate_and_append_point(s, "R", pLR + 64 * j + 32) && challenge_scalar(s, "e", c + 32 * (2 + j));
    ok = ok && validate_and_append_point(s, "A1", pA1) && validate_and_append_point(s, "B", pB) && challenge_scalar(s, "e", c + 32 * (2 + rounds));
    if (!ok) {  // `?` in the reference: the call returns here, later transcripts stay untouched
      rc = BPP_ERR_VERIFICATION_FAILED;
      msg = "caller-side PASS 1: identity element or zero challenge";
      break;
    }
    // to_verifier_rng (:166-179) + finalize(NullRng) + 32 bytes (src/range_proof.rs:845-848)
    append(s, "r1", pr1, 32);
    append(s, "s1", ps1, 32);
    for (uint32_t k = 0; k < t; k++) append(s, "d1", pd1 + 32 * k, 32);
    Strobe rng = s;  // transcript.build_rng(): a clone
    const uint8_t zeros[32] = {0};
    merlin_rng_finalize(rng, zeros);
    merlin_rng_fill(rng, &rng_out[32 * (size_t)i], 32);
  }

  // ---- everything else of verify() on the engine
  std::vector<uint8_t> masks((size_t)n * t * 32, 0), present(n, 0);
  if (rc == BPP_OK) {
    bpp_ctx *ctx_h = nullptr;
    CHECK(bpp_ctx_create(&ctx_h, 0) == BPP_OK);
    uint64_t params = 0;
    CHECK(bpp_params_create(ctx_h, n_bits, m_max, t, h32.data(), g32.data(), &params) == BPP_OK);
    std::vector<bpp_verify_item> vi(n);
    std::vector<const uint8_t *> cp(n);
    for (uint32_t i = 0; i < n; i++) {
      memset(&vi[i], 0, sizeof(vi[i]));
      vi[i].proof = items[i].proof.data();
      vi[i].proof_len = items[i].proof.size();
      vi[i].commitments32 = items[i].commitments.data();
      vi[i].m = items[i].m;
      vi[i].min_values = items[i].min_values.data();
      vi[i].min_present = items[i].min_present.data();
      vi[i].seed_nonce32 = items[i].has_seed ? items[i].seed.data() : nullptr;
      cp[i] = chal[i].data();
    }
    char err[256] = {0};
    rc = bpp_verify_batch_with_challenges(ctx_h, params, vi.data(), n, cp.data(), rng_out.data(), action, 0, masks.data(), present.data(), err,
                                          sizeof(err));
    msg = err;
    bpp_params_destroy(ctx_h, params);
    bpp_ctx_destroy(ctx_h);
  }

  // ---- what the caller holds afterwards
  FILE *o = fopen(argv[2], "wb");
  CHECK(o);
  auto w32 = [&](uint32_t v) { fwrite(&v, 4, 1, o); };
  w32((uint32_t)rc);
  w32((uint32_t)msg.size());
  fwrite(msg.data(), 1, msg.size(), o);
  for (uint32_t i = 0; i < n; i++) {
    w32((uint32_t)chal[i].size());
    fwrite(chal[i].data(), 1, chal[i].size(), o);
  }
  fwrite(rng_out.data(), 1, rng_out.size(), o);
  fwrite(masks.data(), 1, masks.size(), o);
  fwrite(present.data(), 1, present.size(), o);
  for (uint32_t i = 0; i < n; i++) {  // a probe of every caller transcript after the call
    uint8_t probe[32];
    merlin_challenge_bytes(transcripts[i], (const uint8_t *)"probe", 5, probe, 32);
    fwrite(probe, 1, 32, o);
  }
  fclose(o);
  printf("caller_side_merlin rc=%d\n", rc);
  return 0;
}
