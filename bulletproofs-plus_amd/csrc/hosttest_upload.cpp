// Sanitizer harness (CPU test suite only): the host packer of bpp_batch_upload (upload_host.h) and the shared host/device
// arithmetic headers, built with  g++ -fsanitize=address,undefined  into an executable that tests/test_host_sanitizers.py
// runs.  Every proof / statement buffer handed to the packer is an exact-size heap allocation, so any read past the end
// of untrusted input is an ASan report (exit code != 0).  Prints one "ok <case>" line per case.
#include <stdio.h>
#include <stdlib.h>

#include <memory>
#include <random>

#include "hosttest.cpp"  // the ht_* arithmetic probes, same translation unit so they run under the sanitizers too
#include "upload_host.h"

using namespace bpp;

namespace {

std::mt19937_64 rng(8675309);

struct Item {  // owns exact-size copies of everything a bpp_verify_item points to
  std::unique_ptr<uint8_t[]> proof, commits, present, seed;
  std::unique_ptr<uint64_t[]> minv;
  size_t proof_len = 0;
  uint32_t m = 1;
  bpp_verify_item view(const char *label) const {
    bpp_verify_item v;
    memset(&v, 0, sizeof(v));
    v.proof = proof.get();
    v.proof_len = proof_len;
    v.commitments32 = commits.get();
    v.m = m;
    v.min_values = minv.get();
    v.min_present = present.get();
    v.seed_nonce32 = seed.get();
    v.transcript_label = (const uint8_t *)label;
    v.label_len = strlen(label);
    return v;
  }
};

void canonical_scalar(uint8_t *p) {
  for (int i = 0; i < 32; i++) p[i] = (uint8_t)rng();
  p[31] &= 0x0f;  // < 2^252 < l
}

// wire bytes with the structure of a proof: [t] d1[t] A A1 B r1 s1 (L R)*rounds; points are random bytes (not validated at parse)
Item make_item(uint32_t t, uint32_t rounds, uint32_t m, uint64_t promise, bool with_promise, bool with_seed = false) {
  Item it;
  it.m = m;
  it.proof_len = 1 + 32 * (size_t)(t + 5 + 2 * (size_t)rounds);
  it.proof.reset(new uint8_t[it.proof_len]);
  uint8_t *p = it.proof.get();
  p[0] = (uint8_t)t;
  for (size_t i = 1; i < it.proof_len; i++) p[i] = (uint8_t)rng();
  for (uint32_t k = 0; k < t; k++) canonical_scalar(p + 1 + 32 * k);
  canonical_scalar(p + 1 + 32 * (t + 3));
  canonical_scalar(p + 1 + 32 * (t + 4));
  it.commits.reset(new uint8_t[32 * (size_t)m]);
  for (size_t i = 0; i < 32 * (size_t)m; i++) it.commits[i] = (uint8_t)rng();
  it.minv.reset(new uint64_t[m]);
  it.present.reset(new uint8_t[m]);
  for (uint32_t j = 0; j < m; j++) {
    it.minv[j] = promise;
    it.present[j] = with_promise ? 1 : 0;
  }
  if (with_seed) {
    it.seed.reset(new uint8_t[32]);
    canonical_scalar(it.seed.get());
  }
  return it;
}

const ParallelFor serial_for = [](uint32_t n, const std::function<void(uint32_t)> &fn) {
  for (uint32_t i = 0; i < n; i++) fn(i);
};

// returns 0 or the ProofError code; `pl` holds the plan on success
int plan(const std::vector<bpp_verify_item> &v, const ParamShape &P, UploadPlan &pl) {
  try {
    upload_pass_a(v.data(), v.size(), pl);
    std::unique_ptr<uint8_t[]> bytes(new uint8_t[pl.bytes_total + BPP_BYTES_SLACK]);  // exact size: overruns are ASan reports
    upload_pass_b(v.data(), P, pl, bytes.get(), serial_for);
    // what the kernels will address: every slot range inside total_dyn, every proof inside bytes
    uint64_t slots = 0;
    for (size_t i = 0; i < v.size(); i++) {
      const ProofDesc &d = pl.desc[i];
      if (d.dyn_off != slots) return -100;
      slots += (uint64_t)d.m + 3 + 2 * (uint64_t)d.rounds;
      if (d.rounds != pl.pre[i].rounds) return -101;
      const uint64_t end = (uint64_t)d.proof_off + 1 + 32 * ((uint64_t)P.t + 5 + 2 * (uint64_t)d.rounds);
      if (end > pl.bytes_total + BPP_BYTES_SLACK) return -102;
      if ((uint64_t)d.commit_off + 32 * (uint64_t)d.m > pl.bytes_total) return -103;
    }
    if (slots != pl.total_dyn) return -104;
    return 0;
  } catch (const ProofErr &e) {
    return e.code;
  }
}

int fails = 0;
#define EXPECT(name, cond)                      \
  do {                                          \
    if (cond) {                                 \
      printf("ok %s\n", name);                  \
    } else {                                    \
      printf("FAIL %s (line %d)\n", name, __LINE__); \
      fails++;                                  \
    }                                           \
  } while (0)

}  // namespace

int main() {
  const ParamShape P64{64, 8, 1}, P32{32, 2, 1}, P64t3{64, 4, 3};
  {  // plain valid shapes, mixed aggregation
    std::vector<Item> its;
    its.push_back(make_item(1, 6, 1, 5, true, true));
    its.push_back(make_item(1, 7, 2, 0, false));
    its.push_back(make_item(1, 9, 8, 1, true));
    std::vector<bpp_verify_item> v;
    for (auto &i : its) v.push_back(i.view("harness"));
    UploadPlan pl;
    EXPECT("valid_mixed", plan(v, P64, pl) == 0 && pl.rmax == 9 && pl.max_mn == 512 && !pl.any_defer && !pl.any_rounds_bad &&
                              pl.total_dyn == (1 + 3 + 12) + (2 + 3 + 14) + (8 + 3 + 18) && pl.any_seed && !pl.uniform_rounds);
  }
  {  // two-loop precedence (src/range_proof.rs:637-682): item 0 oversized promise, item 1 other extension degree.
     // Neither fails the upload; the chunk check reports the degree first
    std::vector<Item> its;
    its.push_back(make_item(1, 5, 1, 1ull << 40, true));
    its.push_back(make_item(2, 5, 1, 0, false));
    std::vector<bpp_verify_item> v;
    for (auto &i : its) v.push_back(i.view("harness"));
    UploadPlan pl;
    const int rc = plan(v, P32, pl);
    int code = 0;
    try {
      check_deferred(pl.defer, 0, 2);
    } catch (const ProofErr &e) {
      code = e.code;
    }
    int code0 = 0, code1 = 0;
    try {
      check_deferred(pl.defer, 0, 1);
    } catch (const ProofErr &e) {
      code0 = e.code;
    }
    try {
      check_deferred(pl.defer, 1, 2);
    } catch (const ProofErr &e) {
      code1 = e.code;
    }
    EXPECT("precedence_degree_before_promise", rc == 0 && pl.any_defer && code == BPP_ERR_INVALID_ARGUMENT &&
                                                   code0 == BPP_ERR_INVALID_LENGTH && code1 == BPP_ERR_INVALID_ARGUMENT);
  }
  {  // construction errors fail the upload, lowest index wins over later items of any tier
    std::vector<Item> its;
    its.push_back(make_item(2, 5, 1, 0, false));  // degree mismatch (deferred)
    its.push_back(make_item(1, 5, 1, 0, false));
    its[1].proof[1 + 32 * 4 + 31] = 0xff;  // r1 not canonical -> from_bytes fails
    its.push_back(make_item(1, 5, 3, 0, false));  // m not a power of two
    std::vector<bpp_verify_item> v;
    for (auto &i : its) v.push_back(i.view("harness"));
    UploadPlan pl;
    EXPECT("construction_error_first", plan(v, P32, pl) == BPP_ERR_INVALID_ARGUMENT);
  }
  {  // every truncation of a proof: no read past the exact-size buffer, error kinds of from_bytes
    Item full = make_item(1, 6, 1, 0, false);
    bool all_ok = true;
    for (size_t len = 0; len <= full.proof_len; len++) {
      Item it = make_item(1, 6, 1, 0, false);
      std::unique_ptr<uint8_t[]> cut(new uint8_t[len ? len : 1]);
      memcpy(cut.get(), full.proof.get(), len);
      it.proof = std::move(cut);
      it.proof_len = len;
      std::vector<bpp_verify_item> v{it.view("harness")};
      UploadPlan pl;
      const int rc = plan(v, P64, pl);
      const bool structurally_valid = len >= 1 + 32 * 8 && (len - 1) % 64 == 0;
      if (structurally_valid ? (rc != 0) : (rc != BPP_ERR_INVALID_LENGTH && rc != BPP_ERR_INVALID_ARGUMENT)) all_ok = false;
      if (len == full.proof_len && (rc != 0 || pl.rounds_bad[0])) all_ok = false;
      if (structurally_valid && len < full.proof_len && rc == 0 && pl.rounds_bad[0] != BPP_ERR_INVALID_LENGTH) all_ok = false;
    }
    EXPECT("truncations", all_ok);
  }
  {  // random mutations of length and content
    bool all_ok = true;
    for (int iter = 0; iter < 3000; iter++) {
      const uint32_t t = 1 + (uint32_t)(rng() % 3), r = 1 + (uint32_t)(rng() % 8);
      Item it = make_item(t, r, 1u << (rng() % 3), rng(), (rng() & 1) != 0);
      size_t len = it.proof_len;
      if (rng() % 3 == 0) len = rng() % (it.proof_len + 70);
      std::unique_ptr<uint8_t[]> buf(new uint8_t[len ? len : 1]);
      for (size_t i = 0; i < len; i++) buf[i] = i < it.proof_len ? it.proof[i] : (uint8_t)rng();
      for (int k = 0; k < (int)(rng() % 4); k++)
        if (len) buf[rng() % len] = (uint8_t)rng();
      it.proof = std::move(buf);
      it.proof_len = len;
      std::vector<bpp_verify_item> v{it.view("fuzz")};
      UploadPlan pl;
      const int rc = plan(v, P64t3, pl);
      if (rc < 0) all_ok = false;  // internal inconsistency codes
    }
    EXPECT("mutations", all_ok);
  }
  {  // a proof with exactly BPP_MAX_WIRE_ROUNDS pairs is laid out with that many slots; one pair more is refused
    Item big = make_item(1, BPP_MAX_WIRE_ROUNDS, 1, 0, false);
    Item small = make_item(1, 6, 1, 0, false);
    std::vector<bpp_verify_item> v{small.view("harness"), big.view("harness"), small.view("harness")};
    UploadPlan pl;
    const int rc = plan(v, P64, pl);
    EXPECT("max_wire_rounds_layout", rc == 0 && pl.desc[1].rounds == BPP_MAX_WIRE_ROUNDS && pl.rounds_bad[1] == BPP_ERR_SIZE_OVERFLOW &&
                                         pl.desc[2].dyn_off == 16 + 4 + 2 * BPP_MAX_WIRE_ROUNDS && pl.rmax == BPP_MAX_WIRE_ROUNDS);
    Item over = make_item(1, BPP_MAX_WIRE_ROUNDS + 1, 1, 0, false);
    std::vector<bpp_verify_item> v2{small.view("harness"), over.view("harness"), small.view("harness")};
    UploadPlan pl2;
    EXPECT("over_max_wire_rounds_refused", plan(v2, P64, pl2) == BPP_ERR_SIZE_OVERFLOW);
  }
  {  // explicit transcript state with pos >= rate, null pieces
    Item it = make_item(1, 6, 1, 0, false);
    uint8_t st[203];
    memset(st, 0, sizeof(st));
    st[200] = 200;
    bpp_verify_item v0 = it.view("harness");
    v0.transcript_state = st;
    std::vector<bpp_verify_item> v{v0};
    UploadPlan pl;
    EXPECT("bad_transcript_state", plan(v, P64, pl) == BPP_ERR_INVALID_ARGUMENT);
    bpp_verify_item v1 = it.view("harness");
    v1.proof = nullptr;
    std::vector<bpp_verify_item> vv{v1};
    UploadPlan pl1;
    EXPECT("null_proof", plan(vv, P64, pl1) == BPP_ERR_INVALID_LENGTH);
    bpp_verify_item v2 = it.view("harness");
    v2.commitments32 = nullptr;
    std::vector<bpp_verify_item> vvv{v2};
    UploadPlan pl2;
    EXPECT("null_commitments", plan(vvv, P64, pl2) == BPP_ERR_INVALID_ARGUMENT);
  }
  {  // the arithmetic probes under UBSan/ASan: field, scalar, point, merlin, blake2b, weight chain, recodings
    uint8_t a[64], b[32], o[32], o2[32];
    for (int iter = 0; iter < 200; iter++) {
      for (auto &x : a) x = (uint8_t)rng();
      for (auto &x : b) x = (uint8_t)rng();
      if (iter == 0) memset(a, 0xff, sizeof(a));
      if (iter == 1) memset(a, 0, sizeof(a));
      ht_fe_mul(a, b, o);
      ht_fe_sq(a, o);
      ht_fe_addsubmul(a, b, o);
      ht_sc_mul(a, b, o);
      ht_sc_wide(a, o);
      ht_sc_addsub(a, b, o, o2);
      ht_from_uniform(a, o);
      uint8_t enc[32];
      memcpy(enc, o, 32);
      ht_decompress_compress(enc, o2);
      ht_decompress_lean(enc, o2);
      ht_decompress_lean(a, o2);
      ht_madd_swapped(enc, iter & 1, o2);
      ht_from_niels(enc, iter & 1, o2);
      ht_from_niels_first(enc, iter & 1, o2);
      if (iter < 8) {
        ht_fe_invert(a, o);
        b[31] &= 0x0f;
        ht_sc_invert(b, o);
        ht_sc_invert_vartime(b, o);
        ht_scalarmult(b, enc, o2);
        int16_t dg[64];
        uint32_t wd[64], wb;
        for (uint32_t c = 4; c <= 14; c++) ht_msm_recode(b, c, dg, wd);
        ht_fb_recode(b, 131, dg, &wb);
      }
    }
    uint8_t st[203], out[64];
    ht_merlin_kat((const uint8_t *)"test protocol", 13, (const uint8_t *)"some label", 10, (const uint8_t *)"some data", 9,
                  (const uint8_t *)"challenge", 9, out, 32, st);
    static const uint8_t want[32] = {0xd5, 0xa2, 0x19, 0x72, 0xd0, 0xd5, 0xfe, 0x32, 0x0c, 0x0d, 0x26, 0x3f, 0xac, 0x7f, 0xff, 0xb8,
                                     0x14, 0x5a, 0xa6, 0x40, 0xaf, 0x6e, 0x9b, 0xca, 0x17, 0x7c, 0x03, 0xc7, 0xef, 0xcf, 0x06, 0x15};
    EXPECT("merlin_equivalence_simple", memcmp(out, want, 32) == 0);
    ht_merlin_rng(st, a, 32, b, out, 64);
    ht_blake2b(a, 43, (const uint8_t *)"alpha", 5, out);
    std::vector<uint8_t> rin(8 * 40 * 32), rout(8 * 40 * 32), rref(40 * 32);
    for (auto &x : rin) x = (uint8_t)rng();
    ht_weight_chains(rin.data(), 40, 1, rref.data());
    const int w8 = ht_weight_chains(rin.data(), 40, 8, rout.data()), w4 = ht_weight_chains(rin.data(), 40, 4, rout.data());
    EXPECT("weight_chain_lockstep", (w4 == 0 || memcmp(rout.data(), rref.data(), 40 * 32) == 0) && w8 >= 0);
    printf("ok arithmetic_probes\n");
  }
  if (fails) {
    printf("%d case(s) failed\n", fails);
    return 1;
  }
  printf("all ok\n");
  return 0;
}
