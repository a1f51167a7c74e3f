"""GPU tests of round 4: every VerifyAction through the pooled path (bpp_verify_resident_groups_actions, bpp_batcher_verify_action)
held to the CPU oracle (verdict, error kind AND recovered masks), pooling across shapes, the batcher's limits and strided inputs,
the device-wide admission gate for small calls, the runtime-precondition report.

Reference: RangeProof::verify_batch / verify with its three VerifyActions (src/range_proof.rs:46-54, :712-752, :941-969,
:1040-1043); separate callers hand over at most MAX_RANGE_PROOF_BATCH_SIZE proofs per call (:73-76)."""
import importlib
import os
import random
import threading

import numpy as np
import pytest

from tests.helpers import LABEL

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def packed():
    return importlib.import_module("bulletproofs-plus_amd.packed")


def _make(bpp, packed, engine, m, count, seed):
    import bench
    params = bpp.RangeParameters.init(64, m, bpp.create_pedersen_gens_with_extension_degree(1), engine=engine)
    d = bench.make_inputs(np, packed, params, count, seed=seed)
    params.close()
    return d


def _oracle_items(pr, d, sl, seeds, label=LABEL):
    idx = range(*sl.indices(len(d["proofs"])))
    return [{"proof": bytes(pr[k]), "commitments": [bytes(c) for c in d["commitments"][i]],
             "min_values": [int(v) if p else None for v, p in zip(d["min_values"][i], d["min_present"][i])],
             "seed_nonce": bytes(seeds[k]) if seeds is not None else None, "label": label} for k, i in enumerate(idx)]


def _masks_as_lists(masks, present):
    return [[bytes(masks[i, k]) for k in range(masks.shape[1])] if present[i] else None for i in range(masks.shape[0])]


def _cases(bpp, packed, engine, n_cases, seed):
    """(PackedInput, action, oracle rc, oracle masks) tuples: non-aggregated and 2-aggregated statements (two proof lengths),
    1...48 proofs, every VerifyAction, a third of them tampered (failing sum, non-canonical point, identity point, a scalar that
    from_bytes would refuse, a commitment that does not decode)"""
    from oracle import cport
    d1, d2 = _make(bpp, packed, engine, 1, 400, seed), _make(bpp, packed, engine, 2, 120, seed + 1)
    cp = cport.Params(64, 2, 1)
    rng = random.Random(seed)
    sizes = [1, 2, 5, 16, 33, 48]
    out = []
    for i in range(n_cases):
        d = d2 if i % 4 == 3 else d1
        n = sizes[i % len(sizes)]
        lo = rng.randrange(0, len(d["proofs"]) - n)
        sl = slice(lo, lo + n)
        pr = d["proofs"][sl].copy()
        com = d["commitments"][sl].copy()
        kind, j = i % 11, rng.randrange(n)
        if kind == 1:
            pr[j, 1 + 32 + 96] ^= 1                                                   # r1 changed: the final check fails
        elif kind == 2:
            pr[j, 1 + 32:1 + 64] = np.frombuffer(b"\x01" + bytes(31), dtype=np.uint8)  # A does not decode
        elif kind == 3:
            pr[j, 1 + 32:1 + 64] = 0                                                  # A is the identity encoding: PASS 1 refuses it
        elif kind == 4:
            pr[j, 1 + 32 + 96:1 + 32 + 128] = 0xff                                    # r1 >= l: from_bytes refuses the proof
        elif kind == 5:
            com[j, 0] = np.frombuffer(b"\x01" + bytes(31), dtype=np.uint8)             # a commitment that does not decode
        action = (0, 1, 2)[(i // 2) % 3]
        seeds = d["seeds"][sl] if d["seeds"] is not None else None
        dd = dict(d, commitments=d["commitments"].copy())
        dd["commitments"][sl] = com
        rc, masks, _ = cp.verify(_oracle_items(pr, dd, sl, seeds if action else None), action=action)
        inp = packed.PackedInput(pr, com, d["min_values"][sl], d["min_present"][sl], seeds, LABEL)
        out.append((inp, action, rc, masks if rc == 0 else None))
    cp.close()
    return out


def test_grouped_actions_equal_the_oracle(bpp, packed, engine):
    """one resident batch, six groups with different VerifyActions (bpp_verify_resident_groups_actions): per group the oracle's
    verdict and -- where it returns Ok -- the oracle's masks; a RecoverOnly group with a failing sum still returns its masks
    (src/range_proof.rs:1040-1043), a RecoverAndVerify one does not; an all-RecoverOnly call never runs PASS 2"""
    from oracle import cport
    params = bpp.RangeParameters.init(64, 1, bpp.create_pedersen_gens_with_extension_degree(1), engine=engine)
    d = _make(bpp, packed, engine, 1, 300, 4100)
    K = bpp.ProofErrorKind
    bounds = [0, 40, 41, 100, 180, 260, 300]
    actions = [0, 1, 2, 1, 2, 0]
    pr = d["proofs"].copy()
    pr[120, 1 + 32 + 96] ^= 1   # group 3 (RecoverAndVerify): failing sum -> Err, no masks
    pr[200, 1 + 32 + 96] ^= 1   # group 4 (RecoverOnly): not looked at -> Ok, masks (garbage for that one proof, as in the reference)
    pr[270, 1 + 32:1 + 64] = 0  # group 5 (VerifyOnly): identity A
    cp = cport.Params(64, 1, 1)
    rb = packed.ResidentBatch(params, pr, d["commitments"], d["min_values"], d["min_present"], d["seeds"], LABEL)
    res, masks, present = packed.verify_groups_actions(rb, bounds, actions)
    got = _masks_as_lists(masks, present)
    for g, a in enumerate(actions):
        sl = slice(bounds[g], bounds[g + 1])
        rc, want, _ = cp.verify(_oracle_items(pr[sl], d, sl, d["seeds"][sl] if a else None), action=a)
        assert res[g]["code"] == rc, (g, res[g])
        assert got[sl] == (want if rc == 0 and a else [None] * (bounds[g + 1] - bounds[g])), g
    assert [r["code"] for r in res] == [0, 0, 0, int(K.VerificationFailed), 0, int(K.VerificationFailed)]
    assert res[5]["tier"] == 5 and res[5]["index"] == 10
    # the blinding the prover was given comes back for the clean RecoverAndVerify / RecoverOnly groups
    assert got[40] == [bytes(d["blindings"][40, 0, 0])] and got[41:100] == [[bytes(d["blindings"][i, 0, 0])] for i in range(41, 100)]
    # every group RecoverOnly: no weight chains, no PASS 2 (the trace of the final check keeps the previous call's groups)
    res2, masks2, present2 = packed.verify_groups_actions(rb, [0, 150, 300], [2, 2])
    assert [r["code"] for r in res2] == [0, int(K.VerificationFailed)]  # (group 1 holds the identity A: PASS 1)
    # (r1 enters the transcript after the last challenge: proof 120's mask is the prover's blinding all the same)
    assert _masks_as_lists(masks2, present2)[:150] == [[bytes(d["blindings"][i, 0, 0])] for i in range(150)]
    assert not present2[150:].any()
    with pytest.raises(bpp.ProofError):
        packed.verify_groups_actions(rb, [0, 300], [7])
    rb.close()
    cp.close()
    params.close()


def test_batcher_mixed_actions_and_shapes_match_the_oracle(bpp, packed, engine):
    """sixteen host threads, one batcher: VerifyOnly / RecoverAndVerify / RecoverOnly calls of 1...48 proofs, two statement
    shapes (aggregation 1 with seed nonces, aggregation 2: another proof length), a third tampered in five ways.  Every call
    returns the ORACLE's verdict for that input alone and -- where that is Ok and the action recovers -- the oracle's masks;
    most calls went through pooled engine calls, construction errors did not take their pool down"""
    cases = _cases(bpp, packed, engine, 66, 4200)
    assert len({rc for _, _, rc, _ in cases}) >= 3 and {a for _, a, _, _ in cases} == {0, 1, 2}
    params = bpp.RangeParameters.init(64, 2, bpp.create_pedersen_gens_with_extension_degree(1), engine=engine)
    bat = packed.Batcher(params, cases[0][0], lanes=2)
    problems = []

    def worker(k):
        r = random.Random(k)
        try:
            for _ in range(30):
                inp, action, want_rc, want_masks = cases[r.randrange(len(cases))]
                try:
                    masks, present = bat.verify_action(inp, action)
                    got_rc, got = 0, _masks_as_lists(masks, present)
                except bpp.ProofError as e:
                    got_rc, got = int(e.kind), None
                if got_rc != want_rc:
                    problems.append((k, "rc", want_rc, got_rc))
                elif got_rc == 0 and action != 0 and got != want_masks:
                    problems.append((k, "masks", action))
                elif got_rc == 0 and action == 0 and any(g is not None for g in got):
                    problems.append((k, "masks from VerifyOnly"))
        except BaseException as e:  # noqa: BLE001
            problems.append((k, "exception", repr(e)))
    ths = [threading.Thread(target=worker, args=(k,)) for k in range(16)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=600)
    assert not any(t.is_alive() for t in ths), "a caller is stuck in the batcher"
    assert not problems, problems[:5]
    st = bat.stats()
    assert st["pooled_calls"] > 100 and st["engine_calls"] < 16 * 30
    bat.close()
    params.close()


def test_batcher_limits_and_strided_proofs(bpp, packed, engine):
    """max_proofs is a hard limit (round 3: the leader's own request was appended after the cut), proofs with a stride larger
    than their length pool like any others, nothing above the limit is pooled at all"""
    params = bpp.RangeParameters.init(64, 1, bpp.create_pedersen_gens_with_extension_degree(1), engine=engine)
    d = _make(bpp, packed, engine, 1, 256, 4300)
    plen = d["proofs"].shape[1]
    wide = np.zeros((256, plen + 47), dtype=np.uint8)
    wide[:, :plen] = d["proofs"]
    wide[:, plen:] = 0xa5  # never read
    bad = wide.copy()
    bad[7, 1 + 32 + 96] ^= 1

    class Strided(packed.PackedInput):
        def __init__(self, arr, sl):
            super().__init__(d["proofs"][sl], d["commitments"][sl], d["min_values"][sl], d["min_present"][sl], None, LABEL)
            self.keep = arr
            self.struct.proofs = arr[sl].ctypes.data
            self.struct.proof_stride = arr.strides[0]
    bat = packed.Batcher(params, packed.PackedInput(d["proofs"][:1], d["commitments"][:1], d["min_values"][:1], d["min_present"][:1], None, LABEL),
                         lanes=1)
    bat.set_limits(max_proofs=150)
    inputs = [(Strided(wide, slice(0, 64)), 0), (Strided(bad, slice(0, 64)), 1), (Strided(wide, slice(64, 124)), 0),
              (Strided(wide, slice(100, 250)), 0), (Strided(wide, slice(0, 200)), 0)]  # the last: above max_proofs, never pooled
    problems = []

    def worker(k):
        r = random.Random(k)
        for _ in range(25):
            inp, want = inputs[r.randrange(len(inputs))]
            try:
                bat.verify(inp)
                got = 0
            except bpp.ProofError as e:
                got = int(e.kind)
            if got != want:
                problems.append((k, want, got))
    ths = [threading.Thread(target=worker, args=(k,)) for k in range(12)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in ths)
    assert not problems, problems[:5]
    calls, proofs = bat.largest_pool()
    assert 2 <= calls and proofs <= 150, (calls, proofs)
    assert bat.stats()["pooled_calls"] > 20
    bat.close()
    params.close()


def test_small_call_gate_and_runtime_report(bpp, packed, engine):
    """the device admits `limit` small calls at a time, the rest queue in arrival order and all of them get their verdicts;
    bpp_runtime_info_get reports the hardware queues the HIP runtime was started with, the live contexts, the gate's counters;
    a context created beyond the hardware queues carries a note where bpp_ctx_last_error finds it"""
    params = bpp.RangeParameters.init(64, 1, bpp.create_pedersen_gens_with_extension_degree(1), engine=engine)
    d = _make(bpp, packed, engine, 1, 64, 4400)
    info0 = packed.runtime_info(engine)
    assert info0["hw_queues"] == int(os.environ.get("GPU_MAX_HW_QUEUES", "0") or 0) or info0["hw_queues"] == 4
    assert info0["contexts"] >= 1 and info0["host_threads"] == bpp.host_threads() and info0["small_call_limit"] >= 1
    before = engine.lib.bpp_small_call_limit(engine.ctx, 2)
    assert before == info0["small_call_limit"]
    engines = [bpp.Engine(0) for _ in range(8)]
    pars = [params.share(e) for e in engines]
    bad = d["proofs"].copy()
    bad[3, 1 + 32 + 96] ^= 1
    problems = []

    def worker(k):
        for i in range(20):
            pr, want = (bad, 1) if (i + k) % 3 == 0 else (d["proofs"], 0)
            inp = packed.PackedInput(pr, d["commitments"], d["min_values"], d["min_present"], None, LABEL)
            try:
                packed.verify_batch(pars[k], inp, bpp.VerifyAction.VerifyOnly, 0)
                got = 0
            except bpp.ProofError as e:
                got = int(e.kind)
            if got != want:
                problems.append((k, want, got))
    ths = [threading.Thread(target=worker, args=(k,)) for k in range(8)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in ths), "a caller is stuck at the gate"
    assert not problems, problems[:5]
    info = packed.runtime_info(engine)
    assert info["small_call_limit"] == 2 and info["small_calls_in_flight"] == 0
    assert info["small_calls"] - info0["small_calls"] >= 160 and info["small_calls_queued"] > info0["small_calls_queued"]
    assert info["contexts"] >= info0["contexts"] + 8 and info["contexts_peak"] >= info["contexts"]
    assert engine.lib.bpp_small_call_limit(engine.ctx, 0) == 2  # gate off: calls are counted, never held
    q0 = packed.runtime_info(engine)["small_calls_queued"]
    worker(0)
    assert packed.runtime_info(engine)["small_calls_queued"] == q0
    engine.lib.bpp_small_call_limit(engine.ctx, before)
    # more contexts than hardware queues: creation succeeds and says so
    extra = [bpp.Engine(0) for _ in range(max(0, info["hw_queues"] + 1 - packed.runtime_info(engine)["contexts"]))]
    probe = bpp.Engine(0)
    note = engine.lib.bpp_ctx_last_error(probe.ctx).decode()
    assert "hardware queues" in note and "GPU_MAX_HW_QUEUES" in note and packed.runtime_info(engine)["oversubscribed"] == 1
    for e in extra + [probe]:
        e.close()
    for p in pars:
        p.close()
    for e in engines:
        e.close()
    params.close()


def test_recover_only_at_bench_size(bpp, packed, engine):
    """RecoverOnly over a large resident batch in 1024-proof reference batches (bench.py's extra.recover_only leg at a quarter
    of its size): the masks are the blindings the prover was given, nothing of PASS 2 runs, the same batch then verifies"""
    params = bpp.RangeParameters.init(64, 1, bpp.create_pedersen_gens_with_extension_degree(1), engine=engine)
    d = _make(bpp, packed, engine, 1, 16384, 4500)
    rb = packed.ResidentBatch(params, d["proofs"], d["commitments"], d["min_values"], d["min_present"], d["seeds"], LABEL)
    masks = rb.verify(bpp.VerifyAction.RecoverOnly, 1024)
    got = np.frombuffer(b"".join(m.blindings()[0] for m in masks), dtype=np.uint8).reshape(-1, 32)
    assert (got == d["blindings"][:, 0, 0]).all()
    rb.verify_only(1024)
    masks = rb.verify(bpp.VerifyAction.RecoverAndVerify, 1024)
    assert all(m.blindings()[0] == bytes(d["blindings"][i, 0, 0]) for i, m in enumerate(masks))
    rb.close()
    params.close()


def test_collective_deadline_reaches_every_surviving_rank(bpp, packed, engine):
    """an RCCL failure maps to a C error code.  Three in-process ranks, the third never makes the call: the other two
    wait for their first exchange, the deadline (bpp_comm_set_timeout) passes, BOTH get BPP_ERR_COMM with a message that says
    why, within the deadline's order of magnitude; the communicator is dead afterwards (every later call fails at once) and
    the contexts are still good for ordinary calls."""
    import time
    dmod = importlib.import_module("bulletproofs-plus_amd.dist")
    params = bpp.RangeParameters.init(64, 1, bpp.create_pedersen_gens_with_extension_degree(1), engine=engine)
    d = _make(bpp, packed, engine, 1, 90, 4600)
    world, counts = 3, [30, 30, 30]
    engs = [bpp.Engine(0) for _ in range(world)]
    pars = [params.share(e) for e in engs]
    comms = [dmod.ShardComm(engs[r], r, world, local_group=4646) for r in range(world)]
    for c in comms:
        c.set_timeout(400)
    out, took = [None] * world, [None] * world

    def rank_main(r):
        sl = slice(30 * r, 30 * r + 30)
        rb = packed.ResidentBatch(pars[r], d["proofs"][sl], d["commitments"][sl], d["min_values"][sl], d["min_present"][sl], None, LABEL)
        t0 = time.perf_counter()
        try:
            comms[r].verify(rb, counts)
            out[r] = "ok"
        except bpp.EngineError as e:
            out[r] = str(e)
        took[r] = time.perf_counter() - t0
        try:  # the communicator is dead: no second wait
            t1 = time.perf_counter()
            comms[r].verify(rb, counts)
            out[r] += " | second call ok"
        except bpp.EngineError as e:
            out[r] += " | " + str(e)
            assert time.perf_counter() - t1 < 0.2
        rb.verify_only(0)  # the context itself is fine
        rb.close()
    ths = [threading.Thread(target=rank_main, args=(r,)) for r in (0, 1)]  # rank 2 never arrives
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=60)
    assert not any(t.is_alive() for t in ths), "a surviving rank is stuck in a collective"
    for r in (0, 1):
        assert "(-4)" in out[r] and "did not complete within 400 ms" in out[r] and "aborted" in out[r].split("|")[1], out[r]
        assert 0.3 < took[r] < 5.0, took
    for c in comms:
        c.close()
    for p in pars:
        p.close()
    for e in engs:
        e.close()
    params.close()


def test_eight_ranks_in_process_at_configs3_shape(bpp, packed, engine):
    """BASELINE configs[3] proper: 4096 proofs as ONE reference batch, 8 x 512, here with eight in-process ranks (threads with a
    context each on the one GPU; the transport is the in-process stand-in, everything else is the code the RCCL form runs).
    Four such batches per call through bpp_verify_sharded_groups_wave as two pipelined slots of two groups: the weight chains
    are shared out over the ranks (rank r replays groups r, r + 8, ...: here ranks 0 and 1) and a third all_gather hands every
    rank all weights.  Every rank reports, per batch, what the single-call form (chunk = 0 over the 4096 proofs) says; every
    rank's weights are its 512-proof slices of that call's chain; a tampered proof on rank 5 fails batch 2 only."""
    dmod = importlib.import_module("bulletproofs-plus_amd.dist")
    params = bpp.RangeParameters.init(64, 1, bpp.create_pedersen_gens_with_extension_degree(1), engine=engine)
    nb, world, n, S, G = 4, 8, 4096, 2, 2
    d = _make(bpp, packed, engine, 1, nb * n, 4700)
    K = bpp.ProofErrorKind
    counts = [512] * world
    pr = d["proofs"].copy()
    pr[2 * n + 5 * 512 + 77, 1 + 32 + 96] ^= 1   # batch 2, a proof on rank 5
    engs = [[bpp.Engine(0) for _ in range(S)] for _ in range(world)]
    pars = [[params.share(e) for e in row] for row in engs]
    comms = [dmod.ShardComm(engs[r][0], r, world, local_group=4747) for r in range(world)]
    out, weights = [None] * world, [None] * world

    def rank_main(r):
        rbs = []
        try:
            for sl in range(S):  # slot sl holds this rank's shards of batches 2 sl and 2 sl + 1 as ONE resident batch
                idx = np.concatenate([np.arange(b * n + 512 * r, b * n + 512 * (r + 1)) for b in (G * sl, G * sl + 1)])
                rbs.append(packed.ResidentBatch(pars[r][sl], pr[idx], d["commitments"][idx], d["min_values"][idx], d["min_present"][idx], None, LABEL))
            res = comms[r].verify_groups_wave(rbs, G, counts)
            out[r] = [(x["code"], x["tier"], x["rank"]) for part in res for x in part]
            weights[r] = [rb.trace(3) for rb in rbs]
        except BaseException as e:  # noqa: BLE001
            out[r] = ("exception", repr(e))
        finally:
            for rb in rbs:
                rb.close()
    ths = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in ths), "a rank is stuck in a collective"
    assert all(o == out[0] for o in out), out
    assert out[0] == [(0, 0, -1), (0, 0, -1), (int(K.VerificationFailed), 7, -1), (0, 0, -1)]
    for b in range(nb):  # against the single-call form on the batch's 4096 proofs
        sl = slice(b * n, (b + 1) * n)
        rb = packed.ResidentBatch(params, pr[sl], d["commitments"][sl], d["min_values"][sl], d["min_present"][sl], None, LABEL)
        try:
            rb.verify_only(0)
            want = 0
        except bpp.ProofError as e:
            want = int(e.kind)
        w_all = rb.trace(3)
        rb.close()
        assert out[0][b][0] == want
        for r in range(world):
            got = weights[r][b // G][32 * 512 * (b % G):32 * 512 * (b % G + 1)]
            assert got == w_all[32 * 512 * r:32 * 512 * (r + 1)], (b, r)
    for c in comms:
        c.close()
    for row in pars:
        for p in row:
            p.close()
    for row in engs:
        for e in row:
            e.close()
    params.close()


def test_missing_rccl_library_is_an_error_code_not_a_crash():
    """BPP_RCCL_LIB names THE library to load: a missing file makes bpp_comm_create return BPP_ERR_COMM (with the reason where
    bpp_ctx_last_error finds it) and the single-GPU paths of the same process keep working.  In a child process: the RCCL
    binding is resolved once per process."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import ctypes, importlib, sys
sys.path.insert(0, %r)
bpp = importlib.import_module("bulletproofs-plus_amd")
eng = bpp.Engine(0)
lib = eng.lib
buf = (ctypes.c_uint8 * 128)()
assert lib.bpp_comm_unique_id(buf) == -4, "bpp_comm_unique_id"
comm = ctypes.c_void_p()
rc = lib.bpp_comm_create(eng.ctx, buf, 0, 1, ctypes.byref(comm))
msg = lib.bpp_ctx_last_error(eng.ctx).decode()
assert rc == -4 and "RCCL not loadable" in msg and "/nonexistent/librccl.so" in msg, (rc, msg)
p = bpp.RangeParameters.init(64, 1, bpp.create_pedersen_gens_with_extension_degree(1), engine=eng)   # the rest still works
p.close()
eng.close()
print("ok")
''' % root
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, BPP_RCCL_LIB="/nonexistent/librccl.so"), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.returncode, r.stdout[-500:], r.stderr[-1500:])


@pytest.mark.parametrize("m,count,chunk", [(1, 4096 + 32, 1024), (1, 300, 48), (8, 70, 16), (4, 37, 0), (2, 1000, 250)])
def test_column_sums_inside_the_lanes_kernel_equal_per_proof_rows(bpp, packed, engine, opt, m, count, chunk):
    """round 4: k_scalars_lanes sums the generator columns over its workgroup's proofs in registers (k_reduce_parts adds the
    workgroups' partial sums) whenever no workgroup straddles a group boundary; otherwise -- and with fused_columns = 0 -- the
    per-proof rows + k_reduce_static of round 3.  Same static scalars (trace 4), same final point per group (trace 6), aligned
    and unaligned chunk sizes, a tampered group, aggregation 1 / 2 / 4 / 8 (max_mn 64 ... 512)"""
    params = bpp.RangeParameters.init(64, m, bpp.create_pedersen_gens_with_extension_degree(1), engine=engine)
    import bench
    d = bench.make_inputs(np, packed, params, count, seed=4800 + m)
    pr = d["proofs"].copy()
    pr[count // 2, 1 + 32 + 96] ^= 1
    got = {}
    opt("static_gemm", 0)
    for fused, lazy in ((1, 1), (1, 0), (0, 0)):  # lazy: one reduction per (workgroup, column) for the summed products (scalar.h: sc18_redc)
        opt("fused_columns", fused)
        opt("lazy_columns", lazy)
        rb = packed.ResidentBatch(params, pr, d["commitments"], d["min_values"], d["min_present"], None, LABEL)
        with pytest.raises(bpp.ProofError):
            rb.verify_only(chunk)
        got[(fused, lazy)] = (rb.trace(4), rb.trace(5), rb.trace(6))
        rb.close()
    opt("fused_columns", -1)
    opt("lazy_columns", -1)
    opt("static_gemm", -1)
    assert got[(1, 1)] == got[(0, 0)] and got[(1, 0)] == got[(0, 0)]
    rb = packed.ResidentBatch(params, d["proofs"], d["commitments"], d["min_values"], d["min_present"], None, LABEL)
    rb.verify_only(chunk)
    assert set(rb.trace(6)) == {0}
    rb.close()
    params.close()


@pytest.mark.parametrize("m,t,count", [(4, 3, 96), (1, 1, 130), (8, 1, 40)])
def test_prover_round_as_one_launch_gives_the_same_bytes(bpp, packed, engine, opt, m, t, count):
    """round 4: point encoding + Fiat-Shamir step + vector step of a round as ONE launch (kp_round) against the three kernels of
    round 3 (prove_fused = 0), and the small kernels on a high-priority stream (prove_prio = 1): byte-identical proofs for the
    same witnesses and external randomness (tests/test_gpu_prove.py holds the default form to the oracle's bytes); every proof
    verifies"""
    import bench
    params = bpp.RangeParameters.init(64, m, bpp.create_pedersen_gens_with_extension_degree(t), engine=engine)
    d = bench.make_inputs(np, packed, params, count, seed=4900 + m)
    out = {}
    for name, opts in (("fused", {}), ("three", {"prove_fused": 0}), ("prio", {"prove_prio": 1}), ("prio3", {"prove_prio": 1, "prove_fused": 0})):
        for k, v in opts.items():
            opt(k, v)
        out[name] = packed.prove(params, d["values"], d["blindings"], d["commitments"], d["min_values"], d["min_present"], d["seeds"], LABEL,
                                 d["ext"])
        for k in opts:
            opt(k, -1)
    assert (out["fused"] == d["proofs"]).all() and (out["three"] == d["proofs"]).all()
    assert (out["prio"] == d["proofs"]).all() and (out["prio3"] == d["proofs"]).all()
    rb = packed.ResidentBatch(params, out["fused"], d["commitments"], d["min_values"], d["min_present"], None, LABEL)
    rb.verify_only(0)
    rb.close()
    params.close()


@pytest.mark.parametrize("m,count,chunk", [(1, 4096 + 32, 1024), (1, 300, 48), (8, 70, 16), (4, 37, 0), (2, 1000, 256), (1, 2500, 0),
                                            (2, 1000, 250)])
def test_generator_columns_as_a_matrix_product_equal_the_per_proof_form(bpp, packed, engine, opt, m, count, chunk):
    """round 4: the generator columns of a group as ONE integer matrix product over its proofs on the matrix cores
    (kernels_static_gemm.h: digit tables, V_MFMA_I32_32X32X32_I8, anti-diagonal sums, one reduction mod l per column) against the
    per-(proof, generator) Montgomery products of k_scalars_lanes (static_gemm = 0): the same static scalars (trace 4: the oracle's,
    tests/test_gpu_batch.py), dynamic scalars and final point per group; groups that end inside a 16-proof block of the digit
    tables, a group of 2500 proofs (ten K chunks), aggregation 1 ... 8, a tampered proof; chunk 250 is not a multiple of 16:
    the engine must fall back by itself"""
    import struct
    params = bpp.RangeParameters.init(64, m, bpp.create_pedersen_gens_with_extension_degree(1), engine=engine)
    import bench
    d = bench.make_inputs(np, packed, params, count, seed=5200 + m)
    pr = d["proofs"].copy()
    pr[count // 2, 1 + 32 + 96] ^= 1
    got, plan = {}, {}
    opt("tables_wave", 0)  # the tables from k_scalars_shared whatever the size (small inputs build them one wavefront per proof)
    for gemm in (1, 0):
        opt("static_gemm", gemm)
        rb = packed.ResidentBatch(params, pr, d["commitments"], d["min_values"], d["min_present"], None, LABEL)
        with pytest.raises(bpp.ProofError):
            rb.verify_only(chunk)
        got[gemm] = (rb.trace(4), rb.trace(5), rb.trace(6))
        plan[gemm] = struct.unpack("<4I", rb.trace(7))
        rb.close()
    opt("static_gemm", -1)
    assert plan[0][0] & 2 == 0
    assert (plan[1][0] & 2 != 0) == (chunk % 16 == 0), plan
    if count == 2500:
        assert plan[1][2] == 10  # 2500 proofs = 157 blocks of 16 = ten K chunks of 16 blocks
    assert got[1][0] == got[0][0]
    assert got[1] == got[0]
    rb = packed.ResidentBatch(params, d["proofs"], d["commitments"], d["min_values"], d["min_present"], None, LABEL)
    rb.verify_only(chunk)
    assert set(rb.trace(6)) == {0}
    # left to itself the engine takes the matrix product from aggregation 8 on (include/bpp.h: "static_gemm")
    assert (struct.unpack("<4I", rb.trace(7))[0] & 2 != 0) == (chunk % 16 == 0 and m >= 8)
    rb.close()
    opt("tables_wave", -1)
    params.close()


@pytest.mark.parametrize("n_bits,m,count,chunk", [(8, 8, 100, 32), (16, 4, 75, 0), (32, 2, 130, 64), (64, 16, 40, 16), (64, 32, 20, 0), (8, 32, 48, 16)])
def test_matrix_product_columns_other_bit_lengths_and_aggregations(bpp, packed, engine, opt, n_bits, m, count, chunk):
    """the matrix-product form of the generator columns against the per-proof form over the other shapes the index split allows
    (bit lengths 8 ... 64, aggregation 2 ... 32: 8 ... 256 high-table entries, 64 ... 2048 generator pairs), a tampered proof in
    the middle; every proof of the untampered input verifies"""
    import struct
    import bench
    params = bpp.RangeParameters.init(n_bits, m, bpp.create_pedersen_gens_with_extension_degree(1), engine=engine)
    d = bench.make_inputs(np, packed, params, count, seed=5300 + n_bits + m)
    pr = d["proofs"].copy()
    pr[count // 2, 1 + 32 + 96] ^= 1
    got = {}
    opt("tables_wave", 0)
    for gemm in (1, 0):
        opt("static_gemm", gemm)
        rb = packed.ResidentBatch(params, pr, d["commitments"], d["min_values"], d["min_present"], None, LABEL)
        with pytest.raises(bpp.ProofError):
            rb.verify_only(chunk)
        got[gemm] = (rb.trace(4), rb.trace(5), rb.trace(6), struct.unpack("<4I", rb.trace(7))[0] & 2)
        rb.close()
    assert got[1][3] and not got[0][3]
    assert got[1][:3] == got[0][:3]
    opt("static_gemm", 1)
    rb = packed.ResidentBatch(params, d["proofs"], d["commitments"], d["min_values"], d["min_present"], None, LABEL)
    rb.verify_only(chunk)
    assert set(rb.trace(6)) == {0}
    rb.close()
    opt("static_gemm", -1)
    opt("tables_wave", -1)
    params.close()


def test_configs2_at_bench_size_takes_the_matrix_product_by_itself(bpp, packed, engine):
    """BASELINE configs[2] as bench.py runs it (64 reference batches of 256 aggregation-8 proofs in one call, default options): the
    engine takes the matrix-product form of the generator columns by itself, every batch's final point is the identity, and one
    flipped bit in one proof makes exactly that batch fail"""
    import struct
    import bench
    params = bpp.RangeParameters.init(64, 8, bpp.create_pedersen_gens_with_extension_degree(1), engine=engine)
    d = bench.make_inputs(np, packed, params, 64 * 256, seed=5400)
    rb = packed.ResidentBatch(params, d["proofs"], d["commitments"], d["min_values"], d["min_present"], None, LABEL)
    rb.verify_only(256)
    assert struct.unpack("<4I", rb.trace(7))[0] & 2, "aggregation 8, 16 384 proofs: the matrix-product form is the default"
    acc = rb.trace(6)
    assert len(acc) == 64 * 32 and set(acc) == {0}
    rb.close()
    pr = d["proofs"].copy()
    pr[37 * 256 + 11, 1 + 32 + 64 + 5] ^= 0x40  # a bit of A1 of proof 11 of batch 37
    rb = packed.ResidentBatch(params, pr, d["commitments"], d["min_values"], d["min_present"], None, LABEL)
    with pytest.raises(bpp.ProofError):
        rb.verify_only(256)
    acc = rb.trace(6)
    bad = [g for g in range(64) if acc[32 * g:32 * g + 32] != bytes(32)]
    assert bad == [37]
    rb.close()
    params.close()


def test_mutated_aggregated_proofs_through_the_matrix_product_end_like_the_oracle(bpp, packed, engine):
    """nine reference batches of 256 aggregation-8 proofs per call with the engine's default plan (matrix-product columns), one
    random mutation of one proof per call -- a flipped bit, a zeroed member, a low bit -- : the call's outcome (Ok, or the error
    kind) is the C oracle's for the batch that holds the mutated proof, and when the final check fails it fails in that batch only"""
    import struct
    import bench
    from oracle import cport
    from tests.helpers import Prng
    params = bpp.RangeParameters.init(64, 8, bpp.create_pedersen_gens_with_extension_degree(1), engine=engine)
    n = 9 * 256
    d = bench.make_inputs(np, packed, params, n, seed=5500)
    cp = cport.Params(64, 8, 1)
    rng = Prng(b"gemm-fuzz")
    plen = d["proofs"].shape[1]
    for it in range(10):
        pr = d["proofs"].copy()
        p = int(rng.next_u64() % n)
        member = 1 + 32 * int(rng.next_u64() % ((plen - 1) // 32))
        kind = it % 3
        if kind == 0:
            bit = int(rng.next_u64() % (8 * plen))
            pr[p, bit // 8] ^= 1 << (bit % 8)
        elif kind == 1:
            pr[p, member:member + 32] = 0
        else:
            pr[p, member] ^= 1
        g = p // 256
        items = [{"proof": pr[i].tobytes(), "commitments": [d["commitments"][i, j].tobytes() for j in range(8)],
                  "min_values": [int(v) for v in d["min_values"][i]], "seed_nonce": None, "label": LABEL} for i in range(256 * g, 256 * g + 256)]
        want, _, _ = cp.verify(items, action=0)
        rb = packed.ResidentBatch(params, pr, d["commitments"], d["min_values"], d["min_present"], None, LABEL)
        try:
            rb.verify_only(256)
            got = 0
        except bpp.ProofError as e:
            got = int(e.kind)
        assert got == want, (it, p, kind, got, want)
        if got == int(bpp.ProofErrorKind.VerificationFailed):
            assert struct.unpack("<4I", rb.trace(7))[0] & 2
            acc = rb.trace(6)
            assert [k for k in range(9) if acc[32 * k:32 * k + 32] != bytes(32)] == [g]
        rb.close()
    cp.close()
    params.close()
