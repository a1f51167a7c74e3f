// Host side of the batch prover (bpp_prove_batch, include/bpp.h): witness packing and checks of RangeStatement / RangeWitness
// construction (src/range_statement.rs:43-62, src/range_proof.rs:238-311), the fixed-base tables, the round schedule of the WIP
// argument (:401-607) as kernel launches over two sub-batches, zeroization of everything witness-derived.  Included at the end of
// engine.hip (it uses the context, the parameter registry and the staging buffers defined there); kernels: kernels_prove.h.
#pragma once

// ================================================================= batch prover
extern "C" int bpp_prove_batch(bpp_ctx *ctx, uint64_t params, const bpp_prove_item *items, size_t n_items, uint8_t *proofs_out,
                               size_t proof_stride, size_t *proof_len, char *errbuf, size_t errbuf_len) {
  BPP_ENTRY(ctx);
  try {
    const std::shared_ptr<Params> Pp = params_registry().get(params);
    if (!Pp || Pp->device != ctx->device) return fail(ctx, BPP_ERR_BAD_HANDLE, "unknown params handle", errbuf, errbuf_len);
    Params &P = *Pp;
    if (!items || n_items == 0 || !proofs_out) return fail(ctx, BPP_ERR_INVALID_ARGUMENT, "null argument", errbuf, errbuf_len);
    const uint32_t n = P.n_bits, t = P.t, m = items[0].m, B = (uint32_t)n_items;
    // RangeStatement::init (src/range_statement.rs:43-62)
    if (m == 0 || (m & (m - 1))) throw ProofErr{BPP_ERR_INVALID_ARGUMENT, "Number of commitments must be a power of two"};
    if (P.m_max < m) throw ProofErr{BPP_ERR_INVALID_ARGUMENT, "Not enough generators for this statement"};
    const uint32_t mn = m * n;
    if (mn < 2) throw ProofErr{BPP_ERR_INVALID_LENGTH, "bit_length * aggregation factor must be at least 2"};
    uint32_t rounds = 0;
    while ((1u << rounds) < mn) rounds++;
    const size_t plen = 1 + 32 * (size_t)(t + 5 + 2 * rounds);
    if (proof_len) *proof_len = plen;
    if (proof_stride < plen) return fail(ctx, BPP_ERR_INVALID_LENGTH, "proof_stride too small", errbuf, errbuf_len);
    const uint32_t wit_len = m * (8 + 32 * t), ext_len = 32 * (rounds + 3);

    std::vector<ProveDesc> desc(B);
    std::vector<uint8_t> bytes, states;
    // `bytes` (values, blinding factors, seed nonces), the page-locked staging in both directions (witness bytes in,
    // ProveState out) and the device arena hold witness-derived data: wiped on EVERY exit path, including the
    // "Witness opening is invalid!" and HIP-error ones
    // (the page-locked staging on the way OUT carries proofs and status words only -- nothing secret: the per-proof states stay on
    // the device and are wiped there)
    bool arena_clean = true, staging_clean = false;
    ScopeExit wipe_secrets{[&] {
      if (!staging_clean) {
        wipe(bytes.data(), bytes.size());
        wipe(ctx->prove_pin_in.p, ctx->prove_pin_in.n);
      }
      if (!arena_clean && ctx->prove_arena.p) {
        for (auto &ps : ctx->prove_aux_streams) (void)hipStreamSynchronize(ps);
        for (auto &ps : ctx->prove_streams) (void)hipStreamSynchronize(ps);
        for (auto &ps : ctx->prove_lane_streams) (void)hipStreamSynchronize(ps);
        if (ctx->prove_msm_stream) (void)hipStreamSynchronize(ctx->prove_msm_stream);
        (void)hipMemsetAsync(ctx->prove_arena.p, 0, ctx->prove_arena.n, ctx->stream);  // (stream-ordered and waited for: the next
        (void)hipStreamSynchronize(ctx->stream);                                           // call's streams do not wait for the null stream)
      }
    }};
    std::vector<uint64_t> minvals((size_t)B * m);
    std::vector<uint8_t> minpres((size_t)B * m);
    std::map<std::string, uint32_t> state_ids;
    bytes.reserve((size_t)B * (wit_len + 32 * m + ext_len + 32));
    for (uint32_t i = 0; i < B; i++) {
      const bpp_prove_item &it = items[i];
      ProveDesc &d = desc[i];
      if (it.m != m) throw ProofErr{BPP_ERR_INVALID_ARGUMENT, "all items of one prove batch must share the aggregation factor"};
      if (!it.values || !it.blindings32 || !it.commitments32 || !it.rng_bytes || (!it.min_values && it.min_present))
        throw ProofErr{BPP_ERR_INVALID_ARGUMENT, "null witness / statement field"};
      if (it.seed_nonce32 && m > 1)
        throw ProofErr{BPP_ERR_INVALID_ARGUMENT, "Mask recovery is not supported with an aggregated statement"};
      if (it.rng_len < ext_len) throw ProofErr{BPP_ERR_INVALID_LENGTH, "not enough external randomness: need (rounds + 3) * 32 bytes"};
      d.m = m;
      d.minval_idx = i * m;
      for (uint32_t j = 0; j < m; j++) {
        // :264-271
        if (n < 64 && (it.values[j] >> n) > 0) throw ProofErr{BPP_ERR_INVALID_LENGTH, "Value exceeds bit vector capacity!"};
        const bool present = it.min_present ? it.min_present[j] != 0 : false;
        const uint64_t mv = present ? it.min_values[j] : 0;
        // :308-311
        if (present && it.values[j] < mv) throw ProofErr{BPP_ERR_INVALID_ARGUMENT, "Minimum value is larger than value"};
        minvals[(size_t)i * m + j] = mv;
        minpres[(size_t)i * m + j] = present ? 1 : 0;
      }
      for (uint32_t q = 0; q < m * t; q++)
        if (!sc_is_canonical(it.blindings32 + 32 * (size_t)q)) throw ProofErr{BPP_ERR_INVALID_ARGUMENT, "blinding factor is not canonical"};
      if (it.seed_nonce32 && !sc_is_canonical(it.seed_nonce32)) throw ProofErr{BPP_ERR_INVALID_ARGUMENT, "seed nonce is not canonical"};
      d.wit_off = (uint32_t)bytes.size();
      for (uint32_t j = 0; j < m; j++) {
        uint8_t v8[8];
        for (int k = 0; k < 8; k++) v8[k] = (uint8_t)(it.values[j] >> (8 * k));
        bytes.insert(bytes.end(), v8, v8 + 8);
        bytes.insert(bytes.end(), it.blindings32 + (size_t)j * t * 32, it.blindings32 + (size_t)(j + 1) * t * 32);
      }
      d.commit_off = (uint32_t)bytes.size();
      bytes.insert(bytes.end(), it.commitments32, it.commitments32 + (size_t)m * 32);
      d.ext_off = (uint32_t)bytes.size();
      bytes.insert(bytes.end(), it.rng_bytes, it.rng_bytes + ext_len);
      d.seed_off = (uint32_t)bytes.size();
      d.flags = it.seed_nonce32 ? 1u : 0u;
      if (it.seed_nonce32) bytes.insert(bytes.end(), it.seed_nonce32, it.seed_nonce32 + 32);
      else bytes.insert(bytes.end(), 32, 0);
      // the same transcript source as the previous item (the common case: one label for the whole call): same id, no key, no lookup
      if (i && items[i - 1].transcript_state == it.transcript_state && items[i - 1].transcript_label == it.transcript_label &&
          items[i - 1].label_len == it.label_len) {
        d.state_idx = desc[i - 1].state_idx;
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
        uint32_t id = (uint32_t)(states.size() / 203);
        states.resize(states.size() + 203);
        if (it.transcript_state) {
          memcpy(&states[(size_t)id * 203], it.transcript_state, 203);
          if (states[(size_t)id * 203 + 200] >= BPP_STROBE_R) throw ProofErr{BPP_ERR_INVALID_ARGUMENT, "transcript state has pos >= rate"};
        } else {
          Strobe st;
          merlin_new(st, it.transcript_label, (uint32_t)(it.transcript_label ? it.label_len : 0));
          strobe_to_bytes(&states[(size_t)id * 203], st);
        }
        sit = state_ids.emplace(key, id).first;
      }
      d.state_idx = sit->second;
    }

    // RangeProofTranscript::new (src/transcripts.rs:59-89) starts every proof's transcript with the same seven appends -- the
    // domain separator, H, the G bases, N, T, M: parameters of the call, not of the proof.  They are applied HERE, once per distinct
    // caller transcript; kp_init continues with the proof's own commitments and promises (two Keccak-f fewer per proof on the call's
    // first stretch, where no fixed-base MSM runs yet).
    for (size_t id = 0; id < states.size() / 203; id++) {
      Strobe st;
      strobe_from_bytes(st, &states[id * 203]);
      merlin_append_message(st, (const uint8_t *)"dom-sep", 7, (const uint8_t *)"Bulletproofs+ Range Proof", 25);
      merlin_append_message(st, (const uint8_t *)"H", 1, &P.hg32[0], 32);
      for (uint32_t k = 0; k < t; k++) merlin_append_message(st, (const uint8_t *)"G", 1, &P.hg32[(size_t)(k + 1) * 32], 32);
      merlin_append_u64(st, (const uint8_t *)"N", 1, n);
      merlin_append_u64(st, (const uint8_t *)"T", 1, t);
      merlin_append_u64(st, (const uint8_t *)"M", 1, m);
      strobe_to_bytes(&states[id * 203], st);
    }

    hipStream_t s0 = ctx->stream;
    const uint32_t n_gen = 2 * P.n_bits * P.m_max;
    {  // fixed-base window tables for every generator of these parameters (one-off; contexts sharing P serialise here)
     std::lock_guard<std::mutex> fb_lock(P.fb_mu);
     if (!P.fb_table.p) {
      P.fb_geo = fb_geometry(P.table_len);
      P.fb_table.alloc((size_t)P.table_len * fb_stride(P.fb_geo));
      hipLaunchKernelGGL(k_fb_build, dim3(cdiv(P.table_len * P.fb_geo.windows * cdiv(P.fb_geo.entries, FB_BUILD_BLOCK), 64)),
                         dim3(64), 0, s0, P.table.p, P.table_len, P.fb_geo, P.fb_table.p);
      HIP_CHECK(hipGetLastError());
      HIP_CHECK(hipStreamSynchronize(s0));
     }
    }
    // The batch runs as up to PROVE_SUBS sub-batches, each on its own stream: a round is lane step (one lane per proof,
    // Fiat-Shamir latency, a handful of wavefronts) -> wave step -> fixed-base MSM (fills the chip), so one sub-batch's
    // lane step overlaps another's MSM.  All device buffers come out of one arena allocation per call.
    uint32_t PROVE_SUBS = 2;
    if (ctx->opt.prove_subs > 0) PROVE_SUBS = (uint32_t)std::min(16, ctx->opt.prove_subs);
    const uint32_t sub_size = std::max<uint32_t>(64, cdiv(B, PROVE_SUBS));
    constexpr size_t PROVE_PART_BUDGET = (size_t)2 << 30;
    const uint32_t n_sub = cdiv(B, sub_size);
    while (ctx->prove_streams.size() < n_sub) {
      hipStream_t ns;
      HIP_CHECK(hipStreamCreateWithFlags(&ns, hipStreamNonBlocking));
      ctx->prove_streams.push_back(ns);
    }
    // A round of a sub-batch is [point encoding, Fiat-Shamir step, vector fold] -> [fixed-base MSM]: three latency-bound
    // kernels of a few wavefronts, then one that fills the chip.  While one sub-batch's MSM runs, the other's small kernels
    // queue for wave slots behind its 1024 workgroups and take 2-3x their own time (point encoding 70 -> 200 us, fold 45 -> 175:
    // profiles/r04_prover_launches.txt), the MSMs of the two sub-batches drift into each other, and every period has ~110 us
    // in which no MSM runs.  With prove_prio the small kernels go to a HIGH-priority stream of their own (the hardware hands
    // freed wave slots to that queue first), joined to the MSM stream by an event each way per round.
    // Secret-only terms through the uniform-access forms of ct.h, where the reference is constant-time:
    //   "ct" >= 1 (the default): the witness check's commit(v, r) (src/generators/pedersen_gens.rs:112-122, src/range_proof.rs:275-284)
    //   "ct" == 2: A1 and B as well (:572-584): no secret scalar of theirs reaches a fixed-base table.  The Pedersen-base terms go
    //              through the uniform-access lines (k_ct_fixed); the folded generators of the final step are written as
    //              Gf[0] = e^-1 GE + e y^-1 GO, Hf[0] = e HE + e^-1 HO over four PUBLIC points of the last round, whose fixed-base MSM
    //              and 16^w multiples (k_ct_pow16: 252 doublings) run on the side stream beside the last round and the final step,
    //              and the secrets r e^-1, r e y^-1, s e, s e^-1 meet them in k_ct_var: seven additions, a select and a tree (ct.h)
    //   "ct" == 0: everything through the fixed-base tables, whose addresses are the scalars' digits
    const bool ct_check = ctx->opt.ct != 0, ct = ctx->opt.ct == 2;
    const bool fused = ctx->opt.prove_fused != 0;
    // wavefronts per proof in the fused round kernel: 1 (the three phases in a row: the default), 2 or 4 (kernels_prove.h: kp_round:
    // each workgroup's own chain gets 35 % shorter, the call does not -- a round kernel's wavefronts need 174 registers each and
    // find no room on a SIMD beside three of the other sub-batch's MSM wavefronts, so more of them per proof only wait longer:
    // profiles/r05_prover_waves_ab.txt)
    const uint32_t kp_waves = ctx->opt.prove_waves == 2 ? 2u : (ctx->opt.prove_waves == 4 ? 4u : 1u);
    // The rounds' fixed-base MSMs as independent one-wavefront slices (k_fb_part) whose partial sums the next round kernel adds
    // up, instead of one four-wavefront workgroup per output with a reduction tree at its end (k_fb_msm).  `parts` slices per
    // output: enough workgroups for ~4 wavefronts per SIMD, never more than FBP_MAX_PER terms in a slice.  "prove_parts" = 0 keeps
    // the workgroup form (tests run both), a positive value fixes the number of slices.
    uint32_t parts = 0;
    if (ctx->opt.prove_parts != 0) {
      const uint32_t outs = 2 * sub_size;  // (a round's outputs per sub-batch as it really is cut)
      parts = ctx->opt.prove_parts > 0 ? (uint32_t)ctx->opt.prove_parts : cdiv(3072u, outs);
      parts = std::max(parts, cdiv(mn + t + 1, (uint32_t)FBP_MAX_PER));
      parts = std::min<uint32_t>(std::max<uint32_t>(parts, 1u), FBP_MAX_PARTS);
      if (cdiv(mn + t + 1, parts) > FBP_MAX_PER) parts = 0;  // (aggregations whose rounds do not fit the slices: the workgroup form)
      // the slices' partial sums are 3 x proofs x parts x 64 points per sub-batch (30 KB per proof and slice), zeroed with the rest
      // of the arena after every call: beyond PROVE_PART_BUDGET per sub-batch (a call of more than ~20 000 proofs per sub-batch) the
      // workgroup form, whose sums stay in LDS, takes over
      if (parts && (size_t)3 * sub_size * parts * 64 * sizeof(ge) > PROVE_PART_BUDGET) parts = 0;
    }  // one launch per round for encoding + Fiat-Shamir step + vector step (tests run both)
    const bool prio = ctx->opt.prove_prio > 0;  // (off by default: measured, no gain -- profiles/r04_prover_prio_ab.txt)
    // The fixed-base MSMs of ALL sub-batches on ONE stream, in the order they are enqueued (round by round, sub-batch by
    // sub-batch), each behind its own round kernel by an event: first in, first out.  On a stream per sub-batch two MSM launches
    // that are both ready SHARE the chip: the later one ends when it would have ended anyway, but the earlier one ends later by
    // the time they overlapped -- and its sub-batch's next round kernel, the chain that bounds the call, starts later by as much.
    // Measured (profiles/r05_prover_waves_ab.txt, (d)): the MSMs' own event time drops 4 %, the call gets 4 % SLOWER -- two more
    // cross-stream events per round and sub-batch cost more than the sharing did.  Off unless asked for ("prove_fifo" = 1).
    const bool fifo = !prio && n_sub > 1 && ctx->opt.prove_fifo > 0;
    if (fifo && !ctx->prove_msm_stream) HIP_CHECK(hipStreamCreateWithFlags(&ctx->prove_msm_stream, hipStreamNonBlocking));
    if (prio || fifo) {
      while (ctx->prove_sync_events.size() < 2 * (size_t)n_sub) {
        hipEvent_t e;
        HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->prove_sync_events.push_back(e);
      }
    }
    if (prio) {
      int least = 0, greatest = 0;
      HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
      while (ctx->prove_lane_streams.size() < n_sub) {
        hipStream_t ns;
        HIP_CHECK(hipStreamCreateWithPriority(&ns, hipStreamNonBlocking, greatest));
        ctx->prove_lane_streams.push_back(ns);
      }
    }
    while (ctx->prove_aux_streams.size() < n_sub) {
      hipStream_t ns;
      HIP_CHECK(hipStreamCreateWithFlags(&ns, hipStreamNonBlocking));
      ctx->prove_aux_streams.push_back(ns);
    }
    while (ctx->prove_aux_events.size() < 4 * (size_t)n_sub) {  // per sub-batch: inputs resident, witness check done; ("ct" = 2) last round's lists written, 16^w multiples made
      hipEvent_t e;
      HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      ctx->prove_aux_events.push_back(e);
    }
    // lane(q): the stream of sub-batch q's small kernels; msm(q): of its fixed-base MSMs (the same stream without prove_prio)
    auto lane_stream = [&](uint32_t q) { return prio ? ctx->prove_lane_streams[q] : ctx->prove_streams[q]; };
    auto msm_stream = [&](uint32_t q) { return fifo ? ctx->prove_msm_stream : ctx->prove_streams[q]; };
    auto to_msm = [&](uint32_t q) {  // the MSM stream continues behind everything enqueued on the lane stream so far
      if (!prio && !fifo) return;
      HIP_CHECK(hipEventRecord(ctx->prove_sync_events[2 * q], lane_stream(q)));
      HIP_CHECK(hipStreamWaitEvent(msm_stream(q), ctx->prove_sync_events[2 * q], 0));
    };
    auto to_lane = [&](uint32_t q) {  // and back
      if (!prio && !fifo) return;
      HIP_CHECK(hipEventRecord(ctx->prove_sync_events[2 * q + 1], msm_stream(q)));
      HIP_CHECK(hipStreamWaitEvent(lane_stream(q), ctx->prove_sync_events[2 * q + 1], 0));
    };
    const uint32_t stride = 2 * mn + t + 1;
    // "ct" = 2: the public points behind A1's folded generators are made ex_back rounds before the end (kernels_prove.h): their
    // fixed-base MSM, the slices' sums and the 252 doublings of their 16^w multiples then have ex_back rounds of time beside the
    // call's own chain.  2^ex_back points per side and proof; ex_parts slices of <= 128 terms per point.  One round back is the
    // rule ("ct_back" = 2, 3 for the A/B): earlier, the points' MSM doubles the load of a round whose own MSM the chain waits for,
    // and the time the final step no longer spends in an MSM is not given back (profiles/r06_ct_back_ab.txt).
    const uint32_t ex_back = std::min<uint32_t>(rounds, ctx->opt.ct_back > 0 ? std::min(3, ctx->opt.ct_back) : 1u);
    const uint32_t ex_nc = 1u << ex_back, ex_nt = 2 * ex_nc, ex_terms = mn >> ex_back, ex_parts = cdiv(ex_terms, 128u);
    struct Sub {
      uint32_t lo, nb;
      size_t bytes_lo, bytes_len, arena_lo, arena_len;
      uint8_t *d_bytes, *d_states, *d_minpres, *d_a32, *d_lr, *d_a1b, *d_proofs, *d_commit32;
      ProveDesc *d_desc;
      uint64_t *d_minvals;
      ProveState *d_ps;
      sc *d_vec, *d_ts, *d_cts;
      uint32_t *d_tg, *d_tc, *d_ctg, *d_ctc, *d_ftg, *d_ftc;
      sc *d_fts;
      ge *d_ge, *d_ge_ct, *d_part;
      // "ct" = 2: the four public points per proof of the last round (GE, GO, HE, HO: kp_wave_body) as term lists, slice sums, points,
      // and their multiples by 16^w
      sc *d_exs;
      uint32_t *d_exg, *d_exc;
      ge *d_expart, *d_expts, *d_pow, *d_ctprod;
    };
    std::vector<Sub> subs(n_sub);
    size_t arena_need = 0;
    uint8_t *arena_base = nullptr;
    auto take = [&](size_t nbytes) {
      arena_need = (arena_need + 255) & ~(size_t)255;
      uint8_t *p = arena_base ? arena_base + arena_need : nullptr;
      arena_need += nbytes;
      return p;
    };
    auto carve = [&]() {
      arena_need = 0;
      for (uint32_t q = 0; q < n_sub; q++) {
        Sub &u = subs[q];
        u.lo = q * sub_size;
        u.nb = std::min(sub_size, B - u.lo);
        u.bytes_lo = desc[u.lo].wit_off;
        u.bytes_len = (u.lo + u.nb < B ? desc[u.lo + u.nb].wit_off : bytes.size()) - u.bytes_lo;
        const size_t nb = u.nb;
        arena_need = (arena_need + 255) & ~(size_t)255;
        u.arena_lo = arena_need;
        u.d_bytes = take(u.bytes_len);
        u.d_states = take(states.size());
        u.d_minpres = take(nb * m);
        u.d_minvals = (uint64_t *)take(nb * m * 8);
        u.d_desc = (ProveDesc *)take(nb * sizeof(ProveDesc));
        u.d_ps = (ProveState *)take(nb * sizeof(ProveState));
        u.d_vec = (sc *)take(nb * (size_t)KP_VEC_LEN(mn) * sizeof(sc));
        u.d_ts = (sc *)take(nb * 2 * stride * sizeof(sc));
        u.d_tg = (uint32_t *)take(nb * 2 * stride * 4);
        u.d_tc = (uint32_t *)take(nb * 3 * 4);  // (three outputs per proof in the last launch)
        u.d_a32 = take(nb * 32);
        u.d_lr = take((size_t)rounds * nb * 64);
        u.d_a1b = take(nb * 64);
        u.d_proofs = take(nb * plen);
        u.d_commit32 = take(nb * m * 32);
        u.d_cts = (sc *)take(nb * m * (1 + t) * sizeof(sc));
        u.d_ctg = (uint32_t *)take(nb * m * (1 + t) * 4);
        u.d_ctc = (uint32_t *)take(nb * m * 4);
        u.d_ge = (ge *)take(std::max<size_t>(nb * m, 3 * nb) * sizeof(ge));
        u.d_fts = (sc *)take(nb * 2 * CT_ROW * sizeof(sc));
        u.d_ftg = (uint32_t *)take(nb * 2 * CT_ROW * 4);
        u.d_ftc = (uint32_t *)take(nb * 2 * 4);
        u.d_ge_ct = (ge *)take(2 * nb * sizeof(ge));
        u.d_part = (ge *)take(parts ? (size_t)3 * nb * parts * 64 * sizeof(ge) : 16);
        u.d_exs = (sc *)take(ct ? nb * 2 * (size_t)mn * sizeof(sc) : 16);
        u.d_exg = (uint32_t *)take(ct ? nb * 2 * (size_t)mn * 4 : 16);
        u.d_exc = (uint32_t *)take(nb * ex_nt * 4);
        u.d_expart = (ge *)take(ct ? (size_t)ex_nt * nb * ex_parts * 64 * sizeof(ge) : 16);
        u.d_expts = (ge *)take(ct ? (size_t)ex_nt * nb * sizeof(ge) : 16);
        u.d_pow = (ge *)take(ct ? (size_t)ex_nt * nb * BPP_CT_DIGITS * sizeof(ge) : 16);
        u.d_ctprod = (ge *)take(ct ? (size_t)ex_nt * nb * sizeof(ge) : 16);
        u.arena_len = arena_need - u.arena_lo;
      }
    };
    carve();
    {
      // a fresh arena starts out zero as a whole: the alignment gaps between the sub-batches' ranges and the slack at its end
      // are written by nothing and wiped by nothing, and what hipMalloc hands out is not zero -- bpp_prove_secret_bytes (and
      // anyone reading the arena) must see zeros there, not somebody's left-overs
      const uint8_t *before = ctx->prove_arena.p;
      ctx->prove_arena.alloc(arena_need + 256);
      if (ctx->prove_arena.p != before) {  // (on a stream of ours and waited for: the sub-batch streams do not wait for the null stream)
        HIP_CHECK(hipMemsetAsync(ctx->prove_arena.p, 0, ctx->prove_arena.n, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
      }
    }
    arena_base = ctx->prove_arena.p;
    carve();
    // descriptors are relative to each sub-batch's own byte block / minimum-value rows
    for (uint32_t q = 0; q < n_sub; q++)
      for (uint32_t i = 0; i < subs[q].nb; i++) {
        ProveDesc &d = desc[subs[q].lo + i];
        d.wit_off -= (uint32_t)subs[q].bytes_lo;
        d.commit_off -= (uint32_t)subs[q].bytes_lo;
        d.ext_off -= (uint32_t)subs[q].bytes_lo;
        d.seed_off -= (uint32_t)subs[q].bytes_lo;
        d.minval_idx = i * m;
      }
    // page-locked staging so that no copy stalls the enqueue of the next sub-batch
    const size_t in_need = bytes.size() + states.size() + minpres.size() + minvals.size() * 8 + (size_t)B * sizeof(ProveDesc) + 64;
    ctx->prove_pin_in.resize(in_need);
    ctx->prove_pin_out.resize((size_t)B * plen + (size_t)B * sizeof(uint32_t) + 64);
    uint8_t *pin = ctx->prove_pin_in.p;
    uint8_t *pin_bytes = pin;
    memcpy(pin_bytes, bytes.data(), bytes.size());
    uint8_t *pin_states = pin_bytes + bytes.size();
    memcpy(pin_states, states.data(), states.size());
    uint8_t *pin_minpres = pin_states + states.size();
    memcpy(pin_minpres, minpres.data(), minpres.size());
    uint8_t *pin_minvals = pin_minpres + ((minpres.size() + 7) & ~(size_t)7);
    memcpy(pin_minvals, minvals.data(), minvals.size() * 8);
    uint8_t *pin_desc = pin_minvals + minvals.size() * 8;
    memcpy(pin_desc, desc.data(), (size_t)B * sizeof(ProveDesc));
    uint8_t *pin_proofs = ctx->prove_pin_out.p;
    uint32_t *pin_status = (uint32_t *)(pin_proofs + (((size_t)B * plen + 15) & ~(size_t)15));

    const dim3 b64(64);
    // profiling: an event pair around every k_fb_msm launch (the prover's dominant kernel), summed after the call
    size_t ev_used = 0;
    auto fb_mark = [&](hipStream_t st) {
      if (!ctx->profile) return;
      if (ev_used == ctx->prove_events.size()) {
        hipEvent_t e;
        HIP_CHECK(hipEventCreate(&e));
        ctx->prove_events.push_back(e);
      }
      HIP_CHECK(hipEventRecord(ctx->prove_events[ev_used++], st));
    };
    const auto t_begin = std::chrono::steady_clock::now();
    arena_clean = false;
    // The sub-batches advance together: every phase is enqueued for all of them before the next one (enqueued one
    // sub-batch after the other, the second stream started ~60 launches late and the call ended with one stream running
    // alone: its latency-bound Fiat-Shamir kernels with nothing beside them).
    for (uint32_t q = 0; q < n_sub; q++) {
      Sub &u = subs[q];
      hipStream_t s = lane_stream(q), sm = msm_stream(q);
      const uint32_t nb = u.nb;
      HIP_CHECK(hipMemcpyAsync(u.d_bytes, pin_bytes + u.bytes_lo, u.bytes_len, hipMemcpyHostToDevice, s));
      HIP_CHECK(hipMemcpyAsync(u.d_states, pin_states, states.size(), hipMemcpyHostToDevice, s));
      HIP_CHECK(hipMemcpyAsync(u.d_minpres, pin_minpres + (size_t)u.lo * m, (size_t)nb * m, hipMemcpyHostToDevice, s));
      HIP_CHECK(hipMemcpyAsync(u.d_minvals, pin_minvals + (size_t)u.lo * m * 8, (size_t)nb * m * 8, hipMemcpyHostToDevice, s));
      HIP_CHECK(hipMemcpyAsync(u.d_desc, pin_desc + (size_t)u.lo * sizeof(ProveDesc), (size_t)nb * sizeof(ProveDesc),
                               hipMemcpyHostToDevice, s));
      // witness check (:275-284): commit(v_j, r_j) for every opening, compared with the statement's commitments.  Nothing of the
      // proof depends on it (a mismatch is a status bit read after the call), so its three kernels run on a stream of their own
      // beside kp_init / kp_A / the first round's small kernels (in line they were 0.2 ms of the call's first 0.75 ms, in which no
      // round's MSM runs yet) and are joined in front of the first round's MSM, which reuses their output buffer
      hipStream_t sx = ctx->prove_aux_streams[q];
      HIP_CHECK(hipEventRecord(ctx->prove_aux_events[4 * q], s));
      HIP_CHECK(hipStreamWaitEvent(sx, ctx->prove_aux_events[4 * q], 0));
      hipLaunchKernelGGL(kp_commit_terms, dim3(cdiv(nb * m, 64)), b64, 0, sx, u.d_bytes, u.d_desc, t, n_gen, nb, m, 1 + t, u.d_cts,
                         u.d_ctg, u.d_ctc);
      if (ct_check) {
        hipLaunchKernelGGL(k_ct_fixed, dim3(nb * m), b64, 0, sx, u.d_cts, u.d_ctg, u.d_ctc, 1 + t, n_gen, (const niels *)P.fb_ct.p, u.d_ge);
      } else {
        fb_mark(sx);
        hipLaunchKernelGGL(k_fb_msm, dim3(nb * m), dim3(fb_threads(ctx, 1 + t, P.fb_geo)), 0, sx, u.d_cts, u.d_ctg, u.d_ctc, 1 + t, P.fb_table.p,
                           P.fb_geo, u.d_ge, 0u);
        fb_mark(sx);
      }
      hipLaunchKernelGGL(k_compress_ge, dim3(cdiv(nb * m, 64)), b64, 0, sx, u.d_ge, nb * m, u.d_commit32);
      HIP_CHECK(hipEventRecord(ctx->prove_aux_events[4 * q + 1], sx));
      hipLaunchKernelGGL(kp_init, dim3(nb), b64, 0, s, u.d_bytes, u.d_desc, u.d_minvals, u.d_states, P.d_hg32.p, n, t, nb, u.d_ps);
      hipLaunchKernelGGL(kp_A, dim3(nb), b64, 0, s, u.d_bytes, u.d_desc, u.d_minvals, u.d_minpres, P.table.p, P.fb_table.p, P.fb_geo, n_gen, n,
                         t, u.d_ps, u.d_a32);
    }
    for (uint32_t j = 0; j <= rounds; j++)
      for (uint32_t q = 0; q < n_sub; q++) {
        Sub &u = subs[q];
        hipStream_t s = lane_stream(q), sm = msm_stream(q);
        const uint32_t nb = u.nb;
        uint8_t *lr_prev = j ? u.d_lr + (size_t)(j - 1) * nb * 64 : nullptr;
        if (fused) {  // the previous round's L / R are encoded by the same launch (kernels_prove.h: kp_round)
          auto launch_round = [&](auto kernel) {
            hipLaunchKernelGGL(kernel, dim3(nb), dim3(64 * kp_waves), 0, s, u.d_bytes, u.d_desc, u.d_minvals, u.d_minpres, n, t, n_gen, nb, j, rounds,
                               stride, u.d_a32, j ? (parts ? u.d_part : u.d_ge) : (const ge *)nullptr, parts, lr_prev, u.d_ps, u.d_vec, u.d_ts, u.d_tg,
                               u.d_tc, ct ? u.d_fts : (sc *)nullptr, u.d_ftg, u.d_ftc, ct ? u.d_exs : (sc *)nullptr, u.d_exg, u.d_exc, ex_back);
          };
          if (kp_waves == 1) launch_round(kp_round<1>);
          else if (kp_waves == 2) launch_round(kp_round<2>);
          else launch_round(kp_round<4>);
        } else {
          hipLaunchKernelGGL(kp_lane, dim3(nb), b64, 0, s, u.d_bytes, u.d_desc, n, t, nb, j, rounds, u.d_a32, lr_prev, u.d_ps);
          hipLaunchKernelGGL(kp_wave, dim3(nb), b64, 0, s, u.d_bytes, u.d_desc, u.d_minvals, u.d_minpres, n, t, n_gen, j, rounds,
                             stride, u.d_ps, u.d_vec, u.d_ts, u.d_tg, u.d_tc, ct ? u.d_fts : (sc *)nullptr, u.d_ftg, u.d_ftc,
                             ct ? u.d_exs : (sc *)nullptr, u.d_exg, u.d_exc, ex_back);
        }
        uint8_t *out = (j < rounds) ? u.d_lr + (size_t)j * nb * 64 : u.d_a1b;
        if (j == 0) {  // the witness check joins here: its verdict into the proof's status, its buffer free for the round's MSM
          HIP_CHECK(hipStreamWaitEvent(s, ctx->prove_aux_events[4 * q + 1], 0));
          hipLaunchKernelGGL(kp_check_commitments, dim3(cdiv(nb, 64)), b64, 0, s, u.d_bytes, u.d_desc, u.d_commit32, nb, u.d_ps);
        }
        if (ct && j == rounds) {
          // the final step has no fixed-base MSM: the Pedersen-base terms of A1 and B through the uniform-access tables, A1's two
          // folded generators as 2 x 2^ex_back digit-parallel products over the multiples made above (ct.h), the encodings
          hipLaunchKernelGGL(k_ct_fixed, dim3(2 * nb), b64, 0, s, u.d_fts, u.d_ftg, u.d_ftc, CT_ROW, n_gen, (const niels *)P.fb_ct.p, u.d_ge_ct);
          HIP_CHECK(hipStreamWaitEvent(s, ctx->prove_aux_events[4 * q + 3], 0));
          hipLaunchKernelGGL(k_ct_var, dim3(nb * ex_nt), b64, 0, s, u.d_pow, u.d_fts, CT_ROW, ex_nt, u.d_ctprod);
          hipLaunchKernelGGL(k_ct_sum, dim3(cdiv(nb, 64)), b64, 0, s, u.d_ctprod, ex_nt, nb, u.d_ge_ct, 2u);
          hipLaunchKernelGGL(k_compress_ge, dim3(cdiv(2 * nb, 64)), b64, 0, s, u.d_ge_ct, 2 * nb, u.d_a1b);
          continue;
        }
        to_msm(q);
        fb_mark(sm);
        // (the last launch: three outputs per proof in rows of mn + t + 1 terms, see kp_wave_body)
        const bool three = j == rounds;
        const uint32_t n_out = (three ? 3 : 2) * nb, row = three ? mn + t + 1 : stride;
        if (parts) {
          hipLaunchKernelGGL(k_fb_part, dim3(n_out * parts), b64, 0, sm, u.d_ts, u.d_tg, u.d_tc, row, parts, P.fb_table.p, P.fb_geo, u.d_part, 1u);
          // a plain point per output where the consumer is not the fused round kernel: the last launch, the unfused form
          if (j == rounds || !fused) hipLaunchKernelGGL(k_fb_sum, dim3(n_out), b64, 0, sm, u.d_part, parts, u.d_ge);
        } else {
          hipLaunchKernelGGL(k_fb_msm, dim3(n_out), dim3(fb_threads(ctx, mn + t + 1, P.fb_geo)), 0, sm, u.d_ts, u.d_tg, u.d_tc, row, P.fb_table.p,
                             P.fb_geo, u.d_ge, 1u);
        }
        fb_mark(sm);
        to_lane(q);
        if (ct && j + ex_back == rounds) {
          // "ct" = 2: the public points behind the final step's folded generators -- their fixed-base MSM (as much work as a round's L
          // and R), the slices' sums and the 252 doublings that make their multiples by 16^w -- on the sub-batch's side stream,
          // beside the remaining rounds: nothing of it waits for a secret, and nothing secret waits for it before k_ct_var.  Enqueued
          // BEHIND this round's own MSM (the event is recorded after its launch): side by side, the two would share the chip and the
          // round's L and R -- which the chain waits for -- would arrive late by as much as the points' MSM takes (measured: + 0.6 ms
          // per call); behind it, the points' MSM fills the chip while this sub-batch's next step is a lone round kernel, the slot the
          // final step's MSM has without "ct" = 2
          hipStream_t sx = ctx->prove_aux_streams[q];  // (no stream of its own: with several calls in flight every further stream per
                                                       // call is one more tenant of the runtime's hardware queues -- measured: 4 calls x 6
                                                       // streams fall to half the rate of 4 x 4, profiles/r06_ct_inflight.jsonl)
          HIP_CHECK(hipEventRecord(ctx->prove_aux_events[4 * q + 2], sm));
          HIP_CHECK(hipStreamWaitEvent(sx, ctx->prove_aux_events[4 * q + 2], 0));
          fb_mark(sx);
          hipLaunchKernelGGL(k_fb_part, dim3(ex_nt * nb * ex_parts), b64, 0, sx, u.d_exs, u.d_exg, u.d_exc, ex_terms, ex_parts, P.fb_table.p,
                             P.fb_geo, u.d_expart, 1u);
          fb_mark(sx);
          hipLaunchKernelGGL(k_fb_sum, dim3(ex_nt * nb), b64, 0, sx, u.d_expart, ex_parts, u.d_expts);
          hipLaunchKernelGGL(k_ct_pow16, dim3(cdiv(ex_nt * nb, 16)), b64, 0, sx, u.d_expts, ex_nt * nb, u.d_pow);
          HIP_CHECK(hipEventRecord(ctx->prove_aux_events[4 * q + 3], sx));
        }
        if (j == rounds) {  // A1 = A1g + A1h and B
          hipLaunchKernelGGL(kp_final_points, dim3(cdiv(2 * nb, 64)), b64, 0, s, u.d_ge, nb, out);
        } else if (!fused) {
          hipLaunchKernelGGL(k_compress_ge, dim3(cdiv(2 * nb, 64)), b64, 0, s, u.d_ge, 2 * nb, out);
        }
      }
    for (uint32_t q = 0; q < n_sub; q++) {
      Sub &u = subs[q];
      hipStream_t s = lane_stream(q);
      const uint32_t nb = u.nb;
      hipLaunchKernelGGL(kp_finish, dim3(nb), b64, 0, s, u.d_desc, n, t, nb, rounds, u.d_a32, u.d_lr, u.d_a1b, u.d_vec, u.d_ps,
                         u.d_proofs, (uint32_t)plen);
      HIP_CHECK(hipGetLastError());
      HIP_CHECK(hipMemcpyAsync(pin_proofs + (size_t)u.lo * plen, u.d_proofs, (size_t)nb * plen, hipMemcpyDeviceToHost, s));
      // only the status word of each (secret-bearing) ProveState leaves the device
      HIP_CHECK(hipMemcpy2DAsync(pin_status + u.lo, sizeof(uint32_t), &u.d_ps[0].status, sizeof(ProveState), sizeof(uint32_t), nb,
                                 hipMemcpyDeviceToHost, s));
      // zeroize the device copies of witness-derived data (the reference uses Zeroizing<> for these)
      HIP_CHECK(hipMemsetAsync(arena_base + u.arena_lo, 0, u.arena_len, s));
    }
    // Everything is enqueued and this thread has nothing to do for the call's ~6 ms: the host copies of the witness (the packed
    // bytes and their page-locked staging) are wiped NOW, behind the events that say the staging has been read -- not after the
    // call's last kernel, where three megabytes of explicit_bzero were 0.2 ms on the caller's clock.
    const bool nap = ctx->opt.wait >= 0 ? ctx->opt.wait != 0 : n_items >= 256;  // (a call of a few proofs is a latency chain: the runtime's spinning wait)
    for (uint32_t q = 0; q < n_sub; q++) gpu_wait_event(ctx->prove_aux_events[4 * q], nap);
    wipe(bytes.data(), bytes.size());
    wipe(ctx->prove_pin_in.p, ctx->prove_pin_in.n);
    staging_clean = true;
    for (uint32_t q = 0; q < n_sub; q++) {
      // (everything of the MSM stream lies in front of the lane stream's tail; the first of these waits is the call's: it remembers how
      // long calls of this context take, sleeps 70 % of that in one piece and -- by the engine's own rule -- looks through the rest
      // without napping: a prover call is something its caller waits FOR (+ 3 % proofs/s one call at a time against naps to the end;
      // "wait" = 1 naps to the end, 0 leaves the whole wait to the runtime's spinning)
      gpu_wait_stream(ctx, lane_stream(q), nap, q == 0 ? &ctx->wait_hint_prove : nullptr, ((uint64_t)B << 32) | ((uint64_t)m << 8) | t, ctx->opt.wait < 0);
      gpu_wait_stream(ctx, ctx->prove_streams[q], nap, nullptr, 0, ctx->opt.wait < 0);
    }
    if (fifo) gpu_wait_stream(ctx, ctx->prove_msm_stream, nap);
    arena_clean = true;  // every sub-batch's arena range was zeroed on its stream
    if (ctx->profile) {
      bpp_prove_profile &pp = ctx->pprof;
      memset(&pp, 0, sizeof(pp));
      for (size_t k = 0; k + 1 < ev_used; k += 2) {
        float ms = 0;
        HIP_CHECK(hipEventElapsedTime(&ms, ctx->prove_events[k], ctx->prove_events[k + 1]));
        pp.fb_msm_ms += ms;
      }
      pp.total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
      // terms handed to k_fb_msm: witness check m x (1 + t); per round L and R of mn + t + 1 terms each (every generator
      // lands in exactly one of the two); the final step's A1 (every generator once more: 2 mn + t + 1 terms) and B (t + 1)
      pp.fb_terms = (uint64_t)B * ((ct_check ? 0ull : (uint64_t)m * (1 + t)) + (uint64_t)rounds * 2 * (mn + t + 1) + 2 * mn + (ct ? 0u : 2 * t + 2));
      pp.fb_launches = (uint32_t)(ev_used / 2);
      pp.fb_window_bits = P.fb_geo.wbits;
      pp.fb_windows = P.fb_geo.items;  // additions per term
      pp.sub_batches = n_sub;
    }
    for (uint32_t i = 0; i < B; i++) {
      if (pin_status[i] & PV_STATUS_COMMIT_MISMATCH) throw ProofErr{BPP_ERR_INVALID_ARGUMENT, "Witness opening is invalid!"};
      if (pin_status[i] & PV_STATUS_TRANSCRIPT)
        throw ProofErr{BPP_ERR_VERIFICATION_FAILED, "Identity element cannot be added to the transcript / zero challenge"};
    }
    for (uint32_t i = 0; i < B; i++) memcpy(proofs_out + (size_t)i * proof_stride, &pin_proofs[(size_t)i * plen], plen);
    return BPP_OK;
  }
  BPP_CATCH(ctx, errbuf, errbuf_len)
}

#ifdef BPP_KP_PHASES
// measurement build only (tools/gpu_kp_phases.sh): the summed shader-clock cycles per phase of kp_round; reset != 0 clears them
extern "C" int bpp_debug_kp_phases(unsigned long long out[32], int reset) {
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(bpp::g_kp_phase), sizeof(unsigned long long) * 32) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[32] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(bpp::g_kp_phase), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#endif
