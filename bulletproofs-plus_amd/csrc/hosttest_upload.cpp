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
  {  // packed form (bpp_packed_batch) == item form: same plan, same staged bytes; strided input; a proof whose first byte
     // differs (another extension degree claimed) sends the batch through the item form with the same deferred finding
    const uint32_t n = 37, t = 1, rounds = 6, m = 1;
    std::vector<Item> its;
    for (uint32_t i = 0; i < n; i++) its.push_back(make_item(t, rounds, m, 1000 + i, (i & 1) != 0, (i % 3) == 0));
    const size_t plen = its[0].proof_len;
    for (int variant = 0; variant < 3; variant++) {
      const size_t stride = variant == 1 ? plen + 7 : plen;
      if (variant == 2) its[20] = make_item(3, 5, m, 5, true, false);  // extension degree 3 in the SAME 577 bytes: one round less
      // exact-size packed arrays
      std::unique_ptr<uint8_t[]> proofs(new uint8_t[stride * (n - 1) + plen]), commits(new uint8_t[32 * n]), pres(new uint8_t[n]),
          seeds(new uint8_t[32 * n]), seedp(new uint8_t[n]);
      std::unique_ptr<uint64_t[]> minv(new uint64_t[n]);
      memset(proofs.get(), 0xee, stride * (n - 1) + plen);
      for (uint32_t i = 0; i < n; i++) {
        memcpy(proofs.get() + stride * i, its[i].proof.get(), plen);
        memcpy(commits.get() + 32 * i, its[i].commits.get(), 32);
        minv[i] = its[i].minv[0];
        pres[i] = its[i].present[0];
        seedp[i] = its[i].seed ? 1 : 0;
        if (its[i].seed) memcpy(seeds.get() + 32 * i, its[i].seed.get(), 32);
        else memset(seeds.get() + 32 * i, 0x55, 32);
      }
      bpp_packed_batch pk;
      memset(&pk, 0, sizeof(pk));
      pk.n_items = n;
      pk.proofs = proofs.get();
      pk.proof_len = plen;
      pk.proof_stride = stride;
      pk.commitments32 = commits.get();
      pk.m = m;
      pk.min_values = minv.get();
      pk.min_present = pres.get();
      pk.seed_nonces32 = seeds.get();
      pk.seed_present = seedp.get();
      pk.transcript_label = (const uint8_t *)"harness";
      pk.label_len = 7;
      std::vector<bpp_verify_item> v;
      for (auto &i : its) v.push_back(i.view("harness"));
      UploadPlan pa, pb;
      bool same = false, arithmetic = false;
      try {
        upload_pass_a(v.data(), v.size(), pa);
        std::unique_ptr<uint8_t[]> ba(new uint8_t[pa.bytes_total + BPP_BYTES_SLACK]), bb;
        upload_pass_b(v.data(), P64, pa, ba.get(), serial_for);
        std::vector<bpp_verify_item> synth;
        arithmetic = upload_pass_a_packed(pk, pb, serial_for);
        if (!arithmetic) {
          upload_packed_as_items(pk, synth);
          upload_pass_a(synth.data(), synth.size(), pb);
        }
        bb.reset(new uint8_t[pb.bytes_total + BPP_BYTES_SLACK]);
        if (arithmetic) upload_pass_b_packed(pk, P64, pb, bb.get(), serial_for);
        else upload_pass_b(synth.data(), P64, pb, bb.get(), serial_for);
        same = pa.bytes_total == pb.bytes_total && memcmp(ba.get(), bb.get(), pa.bytes_total) == 0 &&
               pa.desc.size() == pb.desc.size() && memcmp(pa.desc.data(), pb.desc.data(), n * sizeof(ProofDesc)) == 0 &&
               pa.minvals == pb.minvals && pa.seeds == pb.seeds && pa.states == pb.states && pa.defer == pb.defer &&
               pa.rounds_bad == pb.rounds_bad && pa.total_dyn == pb.total_dyn && pa.rmax == pb.rmax && pa.max_mn == pb.max_mn &&
               pa.any_seed == pb.any_seed && pa.any_defer == pb.any_defer && pa.any_rounds_bad == pb.any_rounds_bad &&
               pa.uniform_rounds == pb.uniform_rounds && pa.sum_m == pb.sum_m;
      } catch (const ProofErr &e) {
        same = false;
      }
      EXPECT(variant == 0 ? "packed_equals_items" : variant == 1 ? "packed_strided_equals_items" : "packed_mixed_degree_falls_back",
             same && arithmetic == (variant != 2) && (variant != 2 || pb.any_defer));
    }
    // construction errors: same kind and same (lowest) index from either form
    its[20] = make_item(t, rounds, m, 5, true, false);
    its[9].proof[1 + 32 * 4 + 31] = 0xff;   // r1 of item 9 not canonical
    its[30].proof[1 + 31] = 0xff;           // d1 of item 30 not canonical
    std::unique_ptr<uint8_t[]> proofs(new uint8_t[plen * n]), commits(new uint8_t[32 * n]);
    for (uint32_t i = 0; i < n; i++) {
      memcpy(proofs.get() + plen * i, its[i].proof.get(), plen);
      memcpy(commits.get() + 32 * i, its[i].commits.get(), 32);
    }
    bpp_packed_batch pk;
    memset(&pk, 0, sizeof(pk));
    pk.n_items = n;
    pk.proofs = proofs.get();
    pk.proof_len = pk.proof_stride = plen;
    pk.commitments32 = commits.get();
    pk.m = m;
    pk.transcript_label = (const uint8_t *)"harness";
    pk.label_len = 7;
    UploadPlan pb;
    int code = 0;
    uint32_t index = 0;
    try {
      upload_pass_a_packed(pk, pb, serial_for);
      std::unique_ptr<uint8_t[]> bb(new uint8_t[pb.bytes_total + BPP_BYTES_SLACK]);
      upload_pass_b_packed(pk, P64, pb, bb.get(), serial_for);
    } catch (const ProofErr &e) {
      code = e.code;
      index = e.index;
    }
    EXPECT("packed_construction_error_lowest_index", code == BPP_ERR_INVALID_ARGUMENT && index == 9);
  }
  {  // seed nonces on the host side are wiped on both ways out (src/range_statement.rs:76-81): after a successful
     // staging (plan + the seed range of the small staging buffer, nothing else of it) and after a failed plan
    std::vector<Item> its;
    for (int i = 0; i < 5; i++) its.push_back(make_item(1, 6, 1, 7, true, true));
    std::vector<bpp_verify_item> v;
    for (auto &i : its) v.push_back(i.view("harness"));
    UploadPlan pl;
    std::vector<uint8_t> st;
    SmallStaging L;
    bool staged_ok = false;
    {
      PlanWipe guard{pl};
      upload_pass_a(v.data(), v.size(), pl);
      std::unique_ptr<uint8_t[]> bytes(new uint8_t[pl.bytes_total + BPP_BYTES_SLACK]);
      upload_pass_b(v.data(), P64, pl, bytes.get(), serial_for);
      L = upload_small_layout(pl, v.size());
      st.assign(L.total, 0xcc);
      upload_fill_small(pl, pl.desc.data(), v.size(), st.data(), L);
      staged_ok = L.n_seed == 5 * 32 && memcmp(st.data() + L.o_seed, its[0].seed.get(), 32) == 0 &&
                  memcmp(st.data() + L.o_seed + 4 * 32, its[4].seed.get(), 32) == 0;
      upload_wipe_small(st.data(), L);
    }
    bool zero = true;
    for (size_t i = 0; i < L.n_seed; i++) zero = zero && st[L.o_seed + i] == 0;
    for (uint8_t b : pl.seeds) zero = zero && b == 0;
    EXPECT("seed_nonces_wiped_after_success", staged_ok && zero && pl.seeds.size() == 5 * 32 &&
                                                  memcmp(st.data(), pl.desc.data(), sizeof(ProofDesc)) == 0);
    // failure: item 3 does not parse; items 0..2 had their nonces copied into the plan already
    its[3].proof[1 + 32 * 5 + 31] = 0xff;
    UploadPlan pf;
    int code = 0;
    {
      PlanWipe guard{pf};
      try {
        upload_pass_a(v.data(), v.size(), pf);
        std::unique_ptr<uint8_t[]> bytes(new uint8_t[pf.bytes_total + BPP_BYTES_SLACK]);
        upload_pass_b(v.data(), P64, pf, bytes.get(), serial_for);
      } catch (const ProofErr &e) {
        code = e.code;
      }
    }
    bool fz = pf.seeds.size() == 5 * 32;
    for (uint8_t b : pf.seeds) fz = fz && b == 0;
    EXPECT("seed_nonces_wiped_after_failure", code == BPP_ERR_INVALID_ARGUMENT && fz);
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
