"""GPU tests: reference batches of different sizes as the groups of one call (bpp_verify_resident_groups) and the pooling of
many callers' small calls (bpp_batcher) -- every caller must get the outcome of a call of its own."""
import importlib
import random
import threading

import numpy as np
import pytest

from tests.helpers import LABEL

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def packed():
    return importlib.import_module("bulletproofs-plus_amd.packed")


def _inputs(bpp, packed, engine, count, seed):
    import bench
    params = bpp.RangeParameters.init(64, 1, bpp.create_pedersen_gens_with_extension_degree(1), engine=engine)
    return params, bench.make_inputs(np, packed, params, count, seed=seed)


def _direct(bpp, packed, params, d, proofs, sl):
    """what a call of its own says about proofs[sl]: 0 or the ProofError kind"""
    inp = packed.PackedInput(proofs[sl], d["commitments"][sl], d["min_values"][sl], d["min_present"][sl], None, LABEL)
    try:
        packed.verify_batch(params, inp, bpp.VerifyAction.VerifyOnly, 0)
        return 0
    except bpp.ProofError as e:
        return int(e.kind)


def test_ragged_groups_equal_calls_of_their_own(bpp, packed, engine):
    """groups of 1, 7, 64, 200, 3 and 325 proofs in one resident batch: per group the verdict, tier and in-group index of the
    finding a call of its own gives; tampered groups next to clean ones"""
    params, d = _inputs(bpp, packed, engine, 600, 8100)
    K = bpp.ProofErrorKind
    bounds = [0, 1, 8, 72, 272, 275, 600]
    pr = d["proofs"].copy()
    pr[0, 1 + 32 + 96] ^= 1                                                        # the one-proof group fails in its sum
    pr[100, 1 + 32:1 + 64] = np.frombuffer(b"\x01" + bytes(31), dtype=np.uint8)       # group 3 (72..271): non-canonical A at in-group index 28
    pr[273, 1 + 32:1 + 64] = 0                                                      # group 4 (272..274): identity A at in-group index 1
    pr[274, 1 + 32:1 + 64] = np.frombuffer(b"\x01" + bytes(31), dtype=np.uint8)       # ... and a later-tier finding after it
    for proofs in (d["proofs"], pr):
        rb = packed.ResidentBatch(params, proofs, d["commitments"], d["min_values"], d["min_present"], None, LABEL)
        res = packed.verify_groups(rb, bounds)
        rb.close()
        want = [_direct(bpp, packed, params, d, proofs, slice(bounds[g], bounds[g + 1])) for g in range(len(bounds) - 1)]
        assert [r["code"] for r in res] == want
        if proofs is pr:
            assert want == [int(K.VerificationFailed), 0, 0, int(K.InvalidArgument), int(K.VerificationFailed), 0]
            assert (res[0]["tier"], res[3]["tier"], res[3]["index"], res[4]["tier"], res[4]["index"]) == (7, 6, 28, 5, 1)
            assert "canonical" in res[3]["msg"]
    rb = packed.ResidentBatch(params, d["proofs"], d["commitments"], d["min_values"], d["min_present"], None, LABEL)
    for bad_bounds in ([0, 10, 10, 600], [0, 10, 599]):  # an empty group; boundaries that do not cover the batch
        with pytest.raises(bpp.ProofError) as e:
            packed.verify_groups(rb, bad_bounds)
        assert e.value.kind == K.InvalidArgument
    assert [r["code"] for r in packed.verify_groups(rb, [0, 600])] == [0]
    rb.verify_only(64)  # and the equal-chunk form still lays the same batch out its own way afterwards
    rb.close()
    params.close()


def test_batcher_gives_every_caller_its_own_outcome(bpp, packed, engine):
    """sixteen host threads call Batcher.verify with batches of 1...200 proofs, a third of them tampered in different ways
    (a failing sum, a non-canonical point, an identity point, a non-canonical scalar = a construction error that makes the
    pooled upload fall back to single calls, a proof of another length = not poolable): every call returns what
    bpp_verify_batch_packed returns for that input alone; most calls went through pooled engine calls"""
    params, d = _inputs(bpp, packed, engine, 1200, 8200)
    sizes = [1, 2, 5, 16, 40, 64, 100, 200]
    cases = []
    rng = random.Random(99)
    for i in range(48):
        n = sizes[i % len(sizes)]
        lo = rng.randrange(0, 1200 - n)
        pr = d["proofs"][lo:lo + n].copy()
        kind = i % 9
        j = rng.randrange(n)
        if kind == 1:
            pr[j, 1 + 32 + 96] ^= 1
        elif kind == 2:
            pr[j, 1 + 32:1 + 64] = np.frombuffer(b"\x01" + bytes(31), dtype=np.uint8)
        elif kind == 3:
            pr[j, 1 + 32:1 + 64] = 0
        elif kind == 4:
            pr[j, 1 + 32 + 96:1 + 32 + 128] = 0xff  # r1 >= l: from_bytes would have refused this proof
        elif kind == 5:
            pr = np.concatenate([pr, np.zeros((n, 64), dtype=np.uint8)], axis=1)  # one (L, R) pair too many: another proof length
        sl = slice(lo, lo + n)
        inp = packed.PackedInput(pr, d["commitments"][sl], d["min_values"][sl], d["min_present"][sl], None, LABEL)
        try:
            packed.verify_batch(params, inp, bpp.VerifyAction.VerifyOnly, 0)
            want = 0
        except bpp.ProofError as e:
            want = int(e.kind)
        cases.append((inp, want))
    assert len({w for _, w in cases}) >= 3  # accept, VerificationFailed, InvalidArgument / InvalidLength ... are all there
    shape = packed.PackedInput(d["proofs"][:1], d["commitments"][:1], d["min_values"][:1], d["min_present"][:1], None, LABEL)
    bat = packed.Batcher(params, shape, lanes=2)
    problems = []

    def worker(k):
        r = random.Random(k)
        try:
            for _ in range(40):
                inp, want = cases[r.randrange(len(cases))]
                try:
                    bat.verify(inp)
                    got = 0
                except bpp.ProofError as e:
                    got = int(e.kind)
                if got != want:
                    problems.append((k, want, got))
        except BaseException as e:  # noqa: BLE001
            problems.append((k, "exception", repr(e)))
    ths = [threading.Thread(target=worker, args=(k,)) for k in range(16)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in ths), "a caller is stuck in the batcher"
    assert not problems, problems[:5]
    st = bat.stats()
    assert st["pooled_calls"] > 100 and st["engine_calls"] < 16 * 40
    bat.close()
    params.close()
