"""GPU tests of round 3's ABI additions, all through the C ABI:
  * bpp_batch_upload_packed / bpp_verify_batch_packed == the item form (traces 1-6, masks, error kinds and precedence)
  * bpp_verify_submit_packed / bpp_verify_collect (upload k+1 under verify k inside one context) == the blocking call
  * verifier-side secrets (seed nonces, recovered masks) are gone from the device once a batch is destroyed
    (src/range_statement.rs:76-81, src/extended_mask.rs:14)
Inputs come from the engine's own batch prover with the recipe of benches/range_proof.rs:206-262 (its bytes are pinned to
the oracle by tests/test_gpu_prove.py); the masks are additionally compared with what the prover was given."""
import ctypes
import importlib

import numpy as np
import pytest

from tests.helpers import LABEL

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def packed():
    return importlib.import_module("bulletproofs-plus_amd.packed")


def _inputs(bpp, packed, engine, m, t, count, seed):
    import bench
    params = bpp.RangeParameters.init(64, m, bpp.create_pedersen_gens_with_extension_degree(t), engine=engine)
    return params, bench.make_inputs(np, packed, params, count, seed=seed)


def _resident(packed, params, d, form, seeds=None, sl=slice(None)):
    return packed.ResidentBatch(params, d["proofs"][sl], d["commitments"][sl], d["min_values"][sl], d["min_present"][sl],
                                None if seeds is None else seeds[sl], LABEL, form=form)


@pytest.mark.parametrize("m,t,count,chunk", [(1, 1, 300, 128), (8, 1, 20, 0), (2, 3, 33, 16)])
def test_packed_upload_equals_item_form(bpp, packed, engine, m, t, count, chunk):
    params, d = _inputs(bpp, packed, engine, m, t, count, 7000 + m)
    seeds = d["seeds"]
    ra, rb = _resident(packed, params, d, "items", seeds), _resident(packed, params, d, "packed", seeds)
    action = bpp.VerifyAction.RecoverAndVerify
    ma, mb = ra.verify(action, chunk), rb.verify(action, chunk)
    assert [x.blindings() if x else None for x in ma] == [x.blindings() if x else None for x in mb]
    if m == 1:  # the blinding the prover was given comes back (benches/range_proof.rs:74: one blinding, t times)
        assert [x.blindings() for x in mb] == [[bytes(d["blindings"][i, 0, k]) for k in range(t)] for i in range(count)]
    assert ra.shape() == rb.shape()
    for what in (1, 2, 3, 4, 5, 6):
        assert ra.trace(what) == rb.trace(what), what
    assert rb.trace(6) == bytes(32) * rb.shape()["groups"]
    ra.close()
    rb.close()
    # one blocking call: upload, verify, release
    inp = packed.PackedInput(d["proofs"], d["commitments"], d["min_values"], d["min_present"], seeds, LABEL)
    masks, present = packed.verify_batch(params, inp, action, chunk or 256)
    assert present.tolist() == [1 if m == 1 else 0] * count
    if m == 1:
        assert masks.tobytes() == d["blindings"][:, 0].tobytes()
    params.close()


def test_packed_error_kinds_and_precedence_equal_the_item_form(bpp, packed, engine):
    params, d = _inputs(bpp, packed, engine, 1, 1, 40, 7100)
    K = bpp.ProofErrorKind

    def both(mut, chunk=0, stage="verify"):
        """the same mutated input through both forms: (kind, message) must agree"""
        out = []
        for form in ("items", "packed"):
            dd = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in d.items()}
            mut(dd)
            try:
                rb = _resident(packed, params, dd, form)
                try:
                    rb.verify_only(chunk)
                    out.append(None)
                finally:
                    rb.close()
            except bpp.ProofError as e:
                out.append((int(e.kind), e.msg))
        assert out[0] == out[1], out
        return out[0]

    def r1_bit(dd):
        dd["proofs"][17, 1 + 32 + 96] ^= 1  # r1: still canonical, the MSM notices

    def bad_point(dd):
        dd["proofs"][5, 1 + 32:1 + 64] = np.frombuffer(b"\x01" + bytes(31), dtype=np.uint8)  # A does not decode

    def identity(dd):
        dd["proofs"][30, 1 + 32:1 + 64] = 0  # identity appended to the transcript: PASS 1

    def noncanonical(dd):
        dd["proofs"][9, 1 + 32 * 4:1 + 32 * 5] = 0xff  # r1 >= l: from_bytes fails (construction, lowest index wins)
        dd["proofs"][3, 1 + 32:1 + 64] = 0

    def other_degree(dd):
        dd["proofs"][12, 0] = 3  # claims extension degree 3 in the same 577 bytes: no longer a uniform batch
        dd["proofs"][12, 1:1 + 96] = 0  # three canonical d1

    def commitment(dd):
        dd["commitments"][0, 0] = np.frombuffer(b"\x01" + bytes(31), dtype=np.uint8)  # statement commitment does not decode

    assert both(lambda dd: None) is None
    assert both(r1_bit)[0] == K.VerificationFailed
    assert both(bad_point)[0] == K.InvalidArgument
    assert both(identity)[0] == K.VerificationFailed
    assert both(lambda dd: (bad_point(dd), identity(dd)))[0] == K.VerificationFailed  # PASS 1 of any proof first
    assert both(lambda dd: (r1_bit(dd), bad_point(dd)), chunk=8)[0] == K.InvalidArgument  # chunk 0 has the bad point
    assert both(noncanonical)[0] == K.InvalidArgument
    assert both(other_degree)[0] == K.InvalidArgument
    assert both(commitment)[0] == K.InvalidArgument
    params.close()


def test_pipeline_equals_blocking_calls(bpp, packed, engine):
    params, d = _inputs(bpp, packed, engine, 1, 1, 1536, 7200)
    K = bpp.ProofErrorKind
    eng2 = bpp.Engine(0)
    p2 = params.share(eng2)
    pipe = packed.Pipeline(p2, depth=3)
    # six batches of 256: two of them tampered
    views = []
    for b in range(6):
        sl = slice(256 * b, 256 * (b + 1))
        proofs = d["proofs"][sl].copy()
        if b == 2:
            proofs[100, 1 + 32 + 96] ^= 1
        if b == 4:
            proofs[7, 1 + 32:1 + 64] = np.frombuffer(b"\x01" + bytes(31), dtype=np.uint8)
        views.append(packed.PackedInput(proofs, d["commitments"][sl], d["min_values"][sl], d["min_present"][sl], d["seeds"][sl], LABEL))
    want = []
    for v in views:
        try:
            masks, present = packed.verify_batch(params, v, bpp.VerifyAction.RecoverAndVerify, 128)
            want.append((masks.tobytes(), present.tobytes()))
        except bpp.ProofError as e:
            want.append(int(e.kind))
    assert want[2] == K.VerificationFailed and want[4] == K.InvalidArgument and isinstance(want[0], tuple)
    for rounds in range(2):  # lanes are reused
        tickets = [pipe.submit(v, bpp.VerifyAction.RecoverAndVerify, 128) for v in views]  # more tickets than lanes
        got = {}
        for i in (5, 0, 3, 1, 4, 2):  # any order
            try:
                masks, present = pipe.collect(tickets[i])
                got[i] = (masks.tobytes(), present.tobytes())
            except bpp.ProofError as e:
                got[i] = int(e.kind)
        assert [got[i] for i in range(6)] == want
    # VerifyOnly tickets; a construction error is submit's own return value and leaves the lane usable
    t0 = pipe.submit(views[0], bpp.VerifyAction.VerifyOnly, 256)
    bad = d["proofs"][:256].copy()
    bad[200, 1 + 32 * 4:1 + 32 * 5] = 0xff
    vb = packed.PackedInput(bad, d["commitments"][:256], d["min_values"][:256], d["min_present"][:256], None, LABEL)
    with pytest.raises(bpp.ProofError) as e:
        pipe.submit(vb, bpp.VerifyAction.VerifyOnly, 256)
    assert e.value.kind == K.InvalidArgument
    t1 = pipe.submit(views[1], bpp.VerifyAction.VerifyOnly, 256)
    assert pipe.collect(t1) == (None, None) and pipe.collect(t0) == (None, None)
    err = ctypes.create_string_buffer(64)
    assert eng2.lib.bpp_verify_collect(eng2.ctx, 987654, None, None, err, 64) == -3  # unknown ticket
    assert eng2.lib.bpp_ctx_pipeline_depth(eng2.ctx, 2) != 0  # already running
    t2 = pipe.submit(views[3], bpp.VerifyAction.VerifyOnly, 256)  # still in flight when the context goes: destroy waits
    assert t2
    p2.close()
    eng2.close()
    params.close()


def test_verifier_secrets_are_wiped(bpp, packed, engine):
    """seed nonces and recovered masks: present on the device while the batch lives, zero in the buffers the context
    keeps for the next upload once it is destroyed -- after a successful and after a failing verification"""
    eng = bpp.Engine(0)
    params, d = _inputs(bpp, packed, eng, 1, 1, 64, 7300)

    def secret_bytes(handle):
        n = ctypes.c_uint64()
        assert eng.lib.bpp_batch_secret_bytes(eng.ctx, handle, ctypes.byref(n)) == 0
        return n.value

    for tamper in (False, True):
        proofs = d["proofs"].copy()
        if tamper:
            proofs[3, 1 + 32 + 96] ^= 1
        rb = packed.ResidentBatch(params, proofs, d["commitments"], d["min_values"], d["min_present"], d["seeds"], LABEL)
        assert secret_bytes(rb.handle) > 64 * 24  # the nonces arrived
        if tamper:
            with pytest.raises(bpp.ProofError):
                rb.verify(bpp.VerifyAction.RecoverAndVerify, 0)
        else:
            masks = rb.verify(bpp.VerifyAction.RecoverAndVerify, 0)
            assert masks[0].blindings() == [bytes(d["blindings"][0, 0, 0])]
        assert secret_bytes(rb.handle) > 2 * 64 * 24  # nonces + masks
        rb.close()
        assert secret_bytes(0) == 0  # what the next upload adopts holds nothing
    # the one-call form leaves nothing behind either
    inp = packed.PackedInput(d["proofs"], d["commitments"], d["min_values"], d["min_present"], d["seeds"], LABEL)
    packed.verify_batch(params, inp, bpp.VerifyAction.RecoverAndVerify, 256)
    assert secret_bytes(0) == 0
    params.close()
    eng.close()


def test_sharded_entry_over_rccl_one_rank(bpp, packed, engine):
    """bpp_verify_sharded / bpp_verify_sharded_wave with a real RCCL communicator of ONE rank (what a 1-GPU box can run; the
    multi-rank rule is covered on CPU by tests/test_dist_gloo.py): both all_gathers run on device buffers, the verdicts,
    tiers and intermediates equal those of the single-call form on the same proofs"""
    dmod = importlib.import_module("bulletproofs-plus_amd.dist")
    params, d = _inputs(bpp, packed, engine, 1, 1, 600, 7400)
    K = bpp.ProofErrorKind
    comm = dmod.ShardComm(engine, 0, 1, dmod.ShardComm.unique_id())
    rb = _resident(packed, params, d, "packed", sl=slice(0, 300))
    assert comm.verify(rb, [300]) is True
    w_sharded, s_sharded, acc = rb.trace(3), rb.trace(4), rb.trace(6)
    rb.verify_only(0)  # the single-call form over the same 300 proofs: same weights, same scalars, same (identity) result
    assert (w_sharded, s_sharded, acc) == (rb.trace(3), rb.trace(4), rb.trace(6)) and acc == bytes(32)
    with pytest.raises(bpp.EngineError):
        comm.verify(rb, [299])  # counts[rank] must be the shard's size
    rb.close()

    def outcome(mut):
        pr = d["proofs"][:300].copy()
        mut(pr)
        r = packed.ResidentBatch(params, pr, d["commitments"][:300], d["min_values"][:300], d["min_present"][:300], None, LABEL)
        try:
            comm.verify(r, [300])
            return None
        except bpp.ProofError as e:
            return (int(e.kind), e.tier, e.rank, e.index)
        finally:
            r.close()

    def r1_bit(pr):
        pr[17, 1 + 32 + 96] ^= 1

    def bad_point(pr):
        pr[5, 1 + 32:1 + 64] = np.frombuffer(b"\x01" + bytes(31), dtype=np.uint8)

    def identity(pr):
        pr[30, 1 + 32:1 + 64] = 0

    def degree(pr):
        pr[12, 0] = 3
        pr[12, 1:1 + 32 * 8] = 0

    assert outcome(lambda pr: None) is None
    assert outcome(r1_bit) == (K.VerificationFailed, 7, -1, 0)
    assert outcome(bad_point) == (K.InvalidArgument, 6, 0, 5)
    assert outcome(identity) == (K.VerificationFailed, 5, 0, 30)
    assert outcome(lambda pr: (bad_point(pr), identity(pr))) == (K.VerificationFailed, 5, 0, 30)
    assert outcome(degree) == (K.InvalidArgument, 2, 0, 12)
    # a wave: three batches on three contexts, the middle one tampered; ONE all_gather per exchange for all three
    engs = [bpp.Engine(0) for _ in range(3)]
    pars = [params.share(e) for e in engs]
    rbs = []
    for i in range(3):
        pr = d["proofs"][200 * i:200 * (i + 1)].copy()
        if i == 1:
            pr[100, 1 + 32 + 96] ^= 1
        sl = slice(200 * i, 200 * (i + 1))
        rbs.append(packed.ResidentBatch(pars[i], pr, d["commitments"][sl], d["min_values"][sl], d["min_present"][sl], None, LABEL))
    for _ in range(2):
        res = comm.verify_wave(rbs, [200])
        assert [r["code"] for r in res] == [0, int(K.VerificationFailed), 0] and res[1]["tier"] == 7
    with pytest.raises(bpp.EngineError):
        comm.verify_wave([rbs[0], rbs[0]], [200])  # one context twice
    for x in rbs:
        x.close()
    for p in pars:
        p.close()
    for e in engs:
        e.close()
    comm.close()
    params.close()


@pytest.mark.parametrize("m,t,count,chunk", [(1, 1, 1, 0), (1, 1, 300, 0), (1, 1, 300, 64), (8, 1, 20, 0), (2, 3, 33, 16), (1, 1, 1100, 0)])
def test_half_scalar_plan_equals_the_full_one(bpp, packed, engine, opt, m, t, count, chunk):
    """small calls run the final MSM as a half-scalar plan: s = s_lo + 2^126 s_hi over (P, 2^126 P), 127-bit windows, half the
    doublings of the final Horner step (msm.h: k_split_shift_quad).  Forced on and off on the same resident batch: same accept /
    reject, and where the check fails the same (non-identity) group element comes out of both plans"""
    params, d = _inputs(bpp, packed, engine, m, t, count, 7500 + m + count)
    K = bpp.ProofErrorKind
    for tamper in (False, True):
        proofs = d["proofs"].copy()
        if tamper:
            proofs[count // 2, 1 + 32 * t + 96] ^= 1  # r1
        results = []
        for split in (0, 1):
            opt("msm_split", split)
            rb = packed.ResidentBatch(params, proofs, d["commitments"], d["min_values"], d["min_present"], None, LABEL)
            try:
                rb.verify_only(chunk)
                verdict = None
            except bpp.ProofError as e:
                verdict = int(e.kind)
            results.append((verdict, rb.trace(6), rb.trace(4), rb.trace(5)))
            rb.close()
        assert results[0] == results[1]
        assert results[0][0] == (K.VerificationFailed if tamper else None)
        groups = len(results[0][1]) // 32
        if not tamper:
            assert results[0][1] == bytes(32) * groups
        else:
            assert results[0][1] != bytes(32) * groups  # some group's sum is a real point, the same from both plans
    opt("msm_split", -1)
    params.close()


def test_sharded_entry_with_three_ranks_in_process(bpp, packed, engine):
    """bpp_verify_sharded with THREE ranks on one GPU: the ranks are threads of this process, each with its own context and a
    communicator of the in-process transport (bpp_comm_create_local: the all_gather is a rendezvous + device copies; everything
    else is the code the RCCL form runs).  Ragged shards (100 + 37 + 163 proofs): every rank must return the same outcome, the
    one the single-call form gives on the union -- verdict, tier, and for per-proof findings the rank and the index in the
    whole batch -- and each rank's weights must be its slice of the one weight chain over all 300 proofs."""
    import threading
    dmod = importlib.import_module("bulletproofs-plus_amd.dist")
    params, d = _inputs(bpp, packed, engine, 1, 1, 300, 7600)
    K = bpp.ProofErrorKind
    counts, world = [100, 37, 163], 3
    first = [0, 100, 137]
    engs = [bpp.Engine(0) for _ in range(world)]
    pars = [params.share(e) for e in engs]
    comms = [dmod.ShardComm(engs[r], r, world, local_group=4242) for r in range(world)]

    def run(proofs, commitments=None):
        comm_arr = d["commitments"] if commitments is None else commitments
        out, weights = [None] * world, [None] * world

        def rank_main(r):
            sl = slice(first[r], first[r] + counts[r])
            rb = packed.ResidentBatch(pars[r], proofs[sl], comm_arr[sl], d["min_values"][sl], d["min_present"][sl], None, LABEL)
            try:
                comms[r].verify(rb, counts)
                out[r] = None
                weights[r] = rb.trace(3)
            except bpp.ProofError as e:
                out[r] = (int(e.kind), e.tier, e.rank, e.index)
            except BaseException as e:  # noqa: BLE001
                out[r] = ("exception", repr(e))
            finally:
                rb.close()
        ths = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
        for t in ths:
            t.start()
        for t in ths:
            t.join(timeout=120)
        assert not any(t.is_alive() for t in ths), "a rank is stuck in a collective"
        assert out[0] == out[1] == out[2], out
        # the single-call form over the union
        rb = packed.ResidentBatch(params, proofs, comm_arr, d["min_values"], d["min_present"], None, LABEL)
        try:
            rb.verify_only(0)
            want = None
        except bpp.ProofError as e:
            want = int(e.kind)
        w_all = rb.trace(3)
        rb.close()
        assert (out[0][0] if out[0] else None) == want, (out[0], want)
        if out[0] is None:
            assert b"".join(weights) == w_all  # every rank used its slice of the one chain
        return out[0]

    def mut(*edits):
        pr = d["proofs"].copy()
        for i, what in edits:
            if what == "r1":
                pr[i, 1 + 32 + 96] ^= 1
            elif what == "badpoint":
                pr[i, 1 + 32:1 + 64] = np.frombuffer(b"\x01" + bytes(31), dtype=np.uint8)
            elif what == "identity":
                pr[i, 1 + 32:1 + 64] = 0
        return pr

    assert run(d["proofs"]) is None
    assert run(mut((120, "r1"))) == (K.VerificationFailed, 7, -1, 0)                      # only the sum notices
    assert run(mut((120, "badpoint"))) == (K.InvalidArgument, 6, 1, 120)                   # rank 1, proof 120 of the batch
    assert run(mut((250, "identity"))) == (K.VerificationFailed, 5, 2, 250)                # PASS 1 on rank 2
    assert run(mut((5, "badpoint"), (250, "identity"))) == (K.VerificationFailed, 5, 2, 250)   # PASS 1 of any proof first
    assert run(mut((136, "badpoint"), (20, "badpoint"))) == (K.InvalidArgument, 6, 0, 20)   # same tier: the earlier proof
    bad_comm = d["commitments"].copy()
    bad_comm[299, 0] = np.frombuffer(b"\x01" + bytes(31), dtype=np.uint8)
    assert run(mut((3, "identity")), bad_comm) == (K.InvalidArgument, 4, 2, 299)            # a statement point before PASS 1
    assert run(d["proofs"]) is None                                                          # still in step afterwards
    for c in comms:
        c.close()
    for p in pars:
        p.close()
    for e in engs:
        e.close()
    params.close()


def test_sharded_groups_with_three_ranks_in_process(bpp, packed, engine):
    """bpp_verify_sharded_groups: SEVEN reference batches of 60 proofs sharded raggedly (20 + 12 + 28) over three in-process
    ranks, each rank's seven shards resident as ONE batch on ONE context (every kernel launched once for all seven, the
    exchanges carrying all seven; the seven weight chains shared out 3 + 2 + 2 over the ranks and their weights gathered).  Per group every rank must report what bpp_verify_sharded reports for that batch alone
    (= what the single-call form says about the 60 proofs): verdict, tier, rank and index of the finding; untouched groups
    stay accepted next to failing ones; each rank's weights are its slices of the seven chains."""
    import threading
    dmod = importlib.import_module("bulletproofs-plus_amd.dist")
    params, d = _inputs(bpp, packed, engine, 1, 1, 420, 7700)
    K = bpp.ProofErrorKind
    counts, world, G, n = [20, 12, 28], 3, 7, 60
    first = [0, 20, 32]
    engs = [bpp.Engine(0) for _ in range(world)]
    pars = [params.share(e) for e in engs]
    comms = [dmod.ShardComm(engs[r], r, world, local_group=4343) for r in range(world)]

    def run(proofs):
        out, weights = [None] * world, [None] * world

        def rank_main(r):
            idx = np.concatenate([np.arange(n * g + first[r], n * g + first[r] + counts[r]) for g in range(G)])
            rb = packed.ResidentBatch(pars[r], proofs[idx], d["commitments"][idx], d["min_values"][idx], d["min_present"][idx], None, LABEL)
            try:
                res = comms[r].verify_groups(rb, G, counts)
                out[r] = [(x["code"], x["tier"], x["rank"], x["index"]) for x in res]
                weights[r] = rb.trace(3)
            except BaseException as e:  # noqa: BLE001
                out[r] = ("exception", repr(e))
            finally:
                rb.close()
        ths = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
        for t in ths:
            t.start()
        for t in ths:
            t.join(timeout=120)
        assert not any(t.is_alive() for t in ths), "a rank is stuck in a collective"
        assert out[0] == out[1] == out[2], out
        # every group against the single-call form on its 60 proofs
        for g in range(G):
            sl = slice(n * g, n * (g + 1))
            rb = packed.ResidentBatch(params, proofs[sl], d["commitments"][sl], d["min_values"][sl], d["min_present"][sl], None, LABEL)
            try:
                rb.verify_only(0)
                want = 0
            except bpp.ProofError as e:
                want = int(e.kind)
            w_all = rb.trace(3)
            rb.close()
            assert out[0][g][0] == want, (g, out[0][g], want)
            if want == 0:
                c0 = 0
                for r in range(world):  # rank r's weights of group g = its slice of the chain over the group's 60 proofs
                    got = weights[r][32 * g * counts[r]:32 * (g + 1) * counts[r]]
                    assert got == w_all[32 * c0:32 * (c0 + counts[r])], (g, r)
                    c0 += counts[r]
        return out[0]

    ok = (0, 0, -1, 0)
    assert run(d["proofs"]) == [ok] * G
    pr = d["proofs"].copy()
    pr[n * 1 + 25, 1 + 32 + 96] ^= 1                                                    # group 1: r1 of a proof on rank 1
    pr[n * 2 + 25, 1 + 32:1 + 64] = np.frombuffer(b"\x01" + bytes(31), dtype=np.uint8)   # group 2: non-canonical A on rank 1
    pr[n * 3 + 50, 1 + 32:1 + 64] = 0                                                    # group 3: identity A on rank 2
    pr[n * 3 + 2, 1 + 32:1 + 64] = np.frombuffer(b"\x01" + bytes(31), dtype=np.uint8)    # ... and a later-tier finding on rank 0
    pr[n * 6 + 59, 1 + 32 + 96] ^= 1                                                    # group 6 (replayed by rank 0): the last proof
    assert run(pr) == [ok, (int(K.VerificationFailed), 7, -1, 0), (int(K.InvalidArgument), 6, 1, 25), (int(K.VerificationFailed), 5, 2, 50),
                       ok, ok, (int(K.VerificationFailed), 7, -1, 0)]
    # verify()'s own consistency loop (a proof of another extension degree) precedes everything else of its group, whichever
    # rank holds it; the kernels still run on every item of the resident batch, the other groups keep their verdicts
    pr = d["proofs"].copy()
    pr[n * 4 + 40, 0] = 3
    pr[n * 4 + 40, 1:1 + 32 * 8] = 0                 # group 4, rank 2
    pr[n * 4 + 5, 1 + 32:1 + 64] = 0                 # ... and an identity A on rank 0 (a later tier)
    pr[n * 5 + 31, 1 + 32 + 96] ^= 1                 # group 5 fails in its sum
    assert run(pr) == [ok, ok, ok, ok, (int(K.InvalidArgument), 2, 2, 40), (int(K.VerificationFailed), 7, -1, 0), ok]
    assert run(d["proofs"]) == [ok] * G
    want_bad = [ok, ok, ok, ok, (int(K.InvalidArgument), 2, 2, 40), (int(K.VerificationFailed), 7, -1, 0), ok]
    # the pipelined form: every rank runs the tampered and the clean input as two slots of ONE call (two contexts, one host
    # thread, one communicator): the same per-group outcomes, in slot order
    engs_b = [bpp.Engine(0) for _ in range(world)]
    pars_b = [params.share(e) for e in engs_b]
    out2 = [None] * world

    def rank_wave(r):
        idx = np.concatenate([np.arange(n * g + first[r], n * g + first[r] + counts[r]) for g in range(G)])
        rbs = [packed.ResidentBatch(pp, arr[idx], d["commitments"][idx], d["min_values"][idx], d["min_present"][idx], None, LABEL)
               for pp, arr in ((pars[r], pr), (pars_b[r], d["proofs"]))]
        try:
            res = comms[r].verify_groups_wave(rbs, G, counts)
            out2[r] = [[(x["code"], x["tier"], x["rank"], x["index"]) for x in part] for part in res]
        except BaseException as e:  # noqa: BLE001
            out2[r] = ("exception", repr(e))
        finally:
            for rb in rbs:
                rb.close()
    ths = [threading.Thread(target=rank_wave, args=(r,)) for r in range(world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in ths), "a rank is stuck in a collective"
    assert out2[0] == out2[1] == out2[2] == [want_bad, [ok] * G], out2
    for p in pars_b:
        p.close()
    for e in engs_b:
        e.close()
    # an engine fault on ONE rank (its resident batch does not hold n_groups x counts[rank] proofs) strands nobody: the rank
    # still reaches every collective with empty payloads, and every rank reports the fault (tier 255, rank 1) for every group
    out3 = [None] * world

    def rank_fault(r):
        groups_here = G - 1 if r == 1 else G
        idx = np.concatenate([np.arange(n * g + first[r], n * g + first[r] + counts[r]) for g in range(groups_here)])
        rb = packed.ResidentBatch(pars[r], d["proofs"][idx], d["commitments"][idx], d["min_values"][idx], d["min_present"][idx], None, LABEL)
        try:
            out3[r] = [(x["code"] < 0, x["tier"], x["rank"]) for x in comms[r].verify_groups(rb, G, counts)]
        except BaseException as e:  # noqa: BLE001
            out3[r] = ("exception", repr(e))
        finally:
            rb.close()
    ths = [threading.Thread(target=rank_fault, args=(r,)) for r in range(world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in ths), "a rank is stuck in a collective"
    assert out3[0] == out3[1] == out3[2] == [(True, 255, 1)] * G, out3
    assert run(d["proofs"]) == [ok] * G  # and the communicators are still in step
    # one rank, one group: the grouped entry is bpp_verify_sharded
    c1 = dmod.ShardComm(engine, 0, 1, local_group=4344)
    sl = slice(0, 200)
    rb = packed.ResidentBatch(params, d["proofs"][sl], d["commitments"][sl], d["min_values"][sl], d["min_present"][sl], None, LABEL)
    assert [x["code"] for x in c1.verify_groups(rb, 1, [200])] == [0]
    assert [x["code"] for x in c1.verify_groups(rb, 4, [50])] == [0] * 4
    bad = c1.verify_groups(rb, 3, [50])  # 3 x 50 is not the batch: an engine fault of this rank, reported for every group
    assert all(x["code"] < 0 and x["tier"] == 255 for x in bad)
    assert [x["code"] for x in c1.verify_groups(rb, 2, [100])] == [0] * 2  # and the communicator is still in step
    rb.close()
    c1.close()
    for c in comms:
        c.close()
    for p in pars:
        p.close()
    for e in engs:
        e.close()
    params.close()


def test_sharded_groups_equal_the_chunked_form_at_baseline_size(bpp, packed, engine):
    """sixteen reference batches of 1024 proofs (BASELINE configs[1]'s batch) as ONE resident batch: bpp_verify_sharded_groups
    over a one-rank RCCL communicator and the single-process chunked form (verify_only(chunk = 1024)) must agree on every
    intermediate of every group -- weights, static and dynamic scalars, the groups' MSM results -- and on which groups fail
    when two of them are tampered with; the same through the pipelined entry with the batch cut in two slots"""
    dmod = importlib.import_module("bulletproofs-plus_amd.dist")
    G, n = 16, 1024
    params, d = _inputs(bpp, packed, engine, 1, 1, G * n, 7800)
    K = bpp.ProofErrorKind
    comm = dmod.ShardComm(engine, 0, 1, dmod.ShardComm.unique_id())
    pr = d["proofs"].copy()
    pr[5 * n + 1000, 1 + 32 + 96] ^= 1    # r1 of a proof of group 5
    pr[11 * n + 3, 1 + 32 + 128] ^= 2     # s1 of a proof of group 11
    for proofs, bad in ((d["proofs"], set()), (pr, {5, 11})):
        rb = packed.ResidentBatch(params, proofs, d["commitments"], d["min_values"], d["min_present"], None, LABEL)
        res = comm.verify_groups(rb, G, [n])
        assert {g for g, r in enumerate(res) if r["code"] != 0} == bad
        assert all(r["code"] == int(K.VerificationFailed) and r["tier"] == 7 for g, r in enumerate(res) if g in bad)
        sharded = [rb.trace(w) for w in (3, 4, 5, 6)]
        try:
            rb.verify_only(n)
            assert not bad
        except bpp.ProofError as e:
            assert bad and e.kind == K.VerificationFailed
        assert sharded == [rb.trace(w) for w in (3, 4, 5, 6)]
        acc = sharded[3]
        assert {g for g in range(G) if acc[32 * g:32 * g + 32] != bytes(32)} == bad
        rb.close()
        # two slots of eight groups on two contexts, one pipelined call
        eng_b = bpp.Engine(0)
        par_b = params.share(eng_b)
        h = G // 2 * n
        rbs = [packed.ResidentBatch(pp, proofs[sl], d["commitments"][sl], d["min_values"][sl], d["min_present"][sl], None, LABEL)
               for pp, sl in ((params, slice(0, h)), (par_b, slice(h, 2 * h)))]
        res2 = [r for part in comm.verify_groups_wave(rbs, G // 2, [n]) for r in part]
        assert [r["code"] for r in res2] == [r["code"] for r in res]
        assert rbs[0].trace(3) + rbs[1].trace(3) == sharded[0]
        for x in rbs:
            x.close()
        par_b.close()
        eng_b.close()
    comm.close()
    params.close()


@pytest.mark.parametrize("chunk,n", [(128, 16384), (256, 16384), (256, 16640)])  # the last: 65 groups x 29 windows, an odd number of windows
def test_many_small_groups_through_the_throughput_kernels(bpp, packed, engine, opt, chunk, n):
    """16 384 proofs cut into reference batches of 128 / 256 (the reference's own batch size): a throughput call (more than
    100 000 buckets) whose groups are small -- 8- and 9-bit windows, the four-wavefront prelude, two windows per wavefront in
    the bucket reduction (k_msm_window_rc2).  Against the one-window reduction (msm_rc2 = 0) and against the latency kernels
    (msm_quad = 1) on the same resident batch: the same scalars, the same per-group results -- identity everywhere, and the
    same non-identity element in the groups that were tampered with -- and the oracle's verdict on one clean and one bad group."""
    from oracle import cport
    params, d = _inputs(bpp, packed, engine, 1, 1, n, 7900)
    K = bpp.ProofErrorKind
    pr = d["proofs"].copy()
    bad_groups = {3, n // chunk - 1}
    for g in bad_groups:
        pr[g * chunk + 17, 1 + 32 + 96] ^= 1
    results = {}
    for name, options in (("rc2", {}), ("rc", {"msm_rc2": 0}), ("quad", {"msm_quad": 1})):
        for k, v in options.items():
            opt(k, v)
        rb = packed.ResidentBatch(params, pr, d["commitments"], d["min_values"], d["min_present"], None, LABEL)
        with pytest.raises(bpp.ProofError) as e:
            rb.verify_only(chunk)
        assert e.value.kind == K.VerificationFailed
        results[name] = (rb.trace(4), rb.trace(5), rb.trace(6))
        rb.close()
        for k in options:
            opt(k, -1)
    assert results["rc2"] == results["rc"] == results["quad"]
    acc = results["rc2"][2]
    assert {g for g in range(n // chunk) if acc[32 * g:32 * g + 32] != bytes(32)} == bad_groups
    cp = cport.Params(64, 1, 1)
    for g, want in ((0, 0), (3, int(K.VerificationFailed))):
        sl = range(g * chunk, (g + 1) * chunk)
        items = [{"proof": bytes(pr[i]), "commitments": [bytes(d["commitments"][i, 0])],
                  "min_values": [int(d["min_values"][i, 0]) if d["min_present"][i, 0] else None], "seed_nonce": None, "label": LABEL} for i in sl]
        rc, _, _ = cp.verify(items, action=0)
        assert rc == want
    cp.close()
    rb = packed.ResidentBatch(params, d["proofs"], d["commitments"], d["min_values"], d["min_present"], None, LABEL)
    rb.verify_only(chunk)
    assert rb.trace(6) == bytes(32) * (n // chunk)
    rb.close()
    params.close()
