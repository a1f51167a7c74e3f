"""GPU parity at the edges of the parameter space: largest aggregation (mn = 2048, 11 rounds), extension degree 6,
smallest bit lengths, mixed aggregation with spare generator capacity.  Proofs come from the C oracle (fast)."""
import pytest

from oracle import cport
from oracle.pyref import protocol as O
from tests.helpers import LABEL, Prng, sb

pytestmark = pytest.mark.gpu


def _make(n, m, t, count, seed, m_max=None):
    """count proofs of aggregation m made by the C oracle -> list of item dicts (loader format)"""
    rng = Prng(seed)
    cp = cport.Params(n, m_max or m, t)
    rounds = (n * m).bit_length() - 1
    items, raw = [], []
    for _ in range(count):
        vals = [rng.next_u64() % (1 << max(n - 1, 1)) for _ in range(m)]
        blinds = [[sb(O.random_not_zero(rng)) for _ in range(t)] for _ in range(m)]
        mins = [(v // 3 if (i % 2 == 0) else None) for i, v in enumerate(vals)]
        seed_nonce = sb(O.random_not_zero(rng)) if m == 1 else None
        ext = rng.fill_bytes(32 * (rounds + 3))
        proof, comm = cp.prove(LABEL, vals, blinds, mins, seed_nonce, ext)
        items.append(dict(proof=proof, commitments=comm, min_values=mins, seed_nonce=seed_nonce, label=LABEL))
        raw.append(dict(vals=vals, blinds=blinds, mins=mins, seed=seed_nonce, ext=ext))
    return cp, items, raw


def _product(bpp, engine, n, m_max, t, items, private=True):
    params = bpp.RangeParameters.init(n, m_max, bpp.create_pedersen_gens_with_extension_degree(t), engine=engine)
    sts = [bpp.RangeStatement.init(params, it["commitments"], it["min_values"], it["seed_nonce"] if private else None) for it in items]
    proofs = [bpp.RangeProof.from_bytes(it["proof"]) for it in items]
    trs = [bpp.Transcript.new(LABEL) for _ in items]
    return params, sts, proofs, trs


@pytest.mark.parametrize("n,m,t,count", [(64, 32, 1, 2), (64, 16, 6, 2), (2, 1, 1, 3), (1, 2, 2, 2), (16, 8, 3, 3), (64, 1, 6, 4)])
def test_verifier_edge_shapes(bpp, engine, n, m, t, count, opt):
    cp, items, raw = _make(n, m, t, count, b"edge-%d-%d-%d" % (n, m, t))
    params, sts, proofs, trs = _product(bpp, engine, n, m, t, items)
    rc, want_masks, tr = cp.verify(items, action=1, want_trace=True)
    assert rc == 0
    # small inputs take the latency forms of PASS 1 and of the scalar-stage tables by default; the second pass forces the
    # one-lane-per-proof kernels of the throughput path (LDS-resident sponge, lane-built tables) onto the same shapes
    for force_lane_kernels in (False, True):
        if force_lane_kernels:
            opt("transcripts_wave", 0)
            opt("tables_wave", 0)
        rb = bpp.ResidentBatch(trs, sts, proofs)
        masks = rb.verify(bpp.VerifyAction.RecoverAndVerify, chunk=0)
        assert [mk.blindings() if mk else None for mk in masks] == want_masks
        assert rb.trace(1) == tr["challenges"] and rb.trace(2) == tr["rng_out"] and rb.trace(3) == tr["weights"]
        assert rb.trace(4) == tr["static_scalars"] and rb.trace(5) == tr["dynamic_scalars"]
        assert rb.trace(6) == tr["msm_result"] == bytes(32)
        rb.close()
    opt("transcripts_wave", -1)
    opt("tables_wave", -1)
    cp.close()
    # the engine's prover reproduces the same bytes at this shape
    wits = [bpp.RangeWitness.init([bpp.CommitmentOpening.new(r["vals"][j], r["blinds"][j]) for j in range(m)]) for r in raw]
    got = bpp.RangeProof.prove_batch(trs, sts, wits, [r["ext"] for r in raw])
    assert [g.to_bytes() for g in got] == [it["proof"] for it in items]


def test_mixed_aggregation_with_spare_capacity(bpp, engine):
    """batch [1, 8, 2, 4] against generators with capacity 8: zero padding of the static table, different round counts"""
    n, t, m_max = 8, 2, 8
    items = []
    cps = []
    for m in (1, 8, 2, 4):
        cp, it, _ = _make(n, m, t, 1, b"mix-%d" % m, m_max=m_max)
        items += it
        cps.append(cp)
    params, sts, proofs, trs = _product(bpp, engine, n, m_max, t, items)
    got = bpp.RangeProof.verify_batch(trs, sts, proofs, bpp.VerifyAction.RecoverAndVerify, chunk=0)
    rc, want, tr = cps[0].verify(items, action=1, want_trace=True)
    assert rc == 0 and [mk.blindings() if mk else None for mk in got] == want
    rb = bpp.ResidentBatch(trs, sts, proofs)
    rb.verify(bpp.VerifyAction.VerifyOnly, chunk=0)
    assert rb.trace(4) == tr["static_scalars"] and rb.trace(5) == tr["dynamic_scalars"] and rb.trace(1) == tr["challenges"]
    rb.close()
    for cp in cps:
        cp.close()
