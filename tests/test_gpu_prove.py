"""GPU parity tests for the batch prover (bpp_prove_batch): identical proof BYTES as the oracle for identical
(witness, statement, transcript, external RNG bytes), then the reference's prove_and_verify properties
(tests/ristretto.rs:152-373) with proofs made by the engine itself."""
import hashlib

import pytest

from oracle import cport
from oracle.pyref import curve as C
from oracle.pyref import merlin as M
from oracle.pyref import protocol as O
from tests.helpers import LABEL, Prng, make_batch, sb

pytestmark = pytest.mark.gpu


def _kind(bpp, fn):
    with pytest.raises(bpp.ProofError) as e:
        fn()
    return e.value.kind


def _inputs(bpp, params, n, m, t, count, seed, strategy="third", with_seed=None):
    """count (statement, witness, rng bytes) triples with the bench recipe; returns product objects + raw data"""
    rng = Prng(seed)
    rounds = (n * m).bit_length() - 1
    vals, blinds, mins, seeds, exts = [], [], [], [], []
    for _ in range(count):
        v, b, mn = [], [], []
        for _j in range(m):
            x = rng.next_u64() % (1 << (n - 1))
            v.append(x)
            mn.append({"none": None, "third": x // 3, "eq": x}[strategy])
            b.append([sb(O.random_not_zero(rng)) for _k in range(t)])
        vals.append(v)
        blinds.append(b)
        mins.append(mn)
        use_seed = (m == 1) if with_seed is None else with_seed
        seeds.append(sb(O.random_not_zero(rng)) if use_seed else None)
        exts.append(rng.fill_bytes(32 * (rounds + 3)))
    comms = params.commit_many([x for v in vals for x in v], [x for b in blinds for x in b])
    comms = [comms[i * m:(i + 1) * m] for i in range(count)]
    sts = [bpp.RangeStatement.init(params, comms[i], mins[i], seeds[i]) for i in range(count)]
    wits = [bpp.RangeWitness.init([bpp.CommitmentOpening.new(vals[i][j], blinds[i][j]) for j in range(m)]) for i in range(count)]
    return sts, wits, exts, dict(vals=vals, blinds=blinds, mins=mins, seeds=seeds, comms=comms)


@pytest.mark.parametrize("n,m,t,count,strategy", [(8, 1, 1, 3, "third"), (4, 4, 2, 2, "none"), (64, 1, 1, 5, "third"),
                                                  (64, 2, 3, 2, "eq"), (32, 4, 2, 2, "third"), (64, 4, 3, 3, "third"),
                                                  (2, 1, 1, 1, "none")])
def test_prover_bytes_equal_oracle(bpp, engine, n, m, t, count, strategy):
    params = bpp.RangeParameters.init(n, m, bpp.create_pedersen_gens_with_extension_degree(t), engine=engine)
    sts, wits, exts, raw = _inputs(bpp, params, n, m, t, count, b"prove-%d-%d-%d" % (n, m, t), strategy)
    got = bpp.RangeProof.prove_batch([bpp.Transcript.new(LABEL)] * count, sts, wits, exts)
    cp = cport.Params(n, m, t)
    for i in range(count):
        want, comm = cp.prove(LABEL, raw["vals"][i], raw["blinds"][i], raw["mins"][i], raw["seeds"][i], exts[i])
        assert comm == raw["comms"][i]
        assert got[i].to_bytes() == want, "proof %d differs from the oracle" % i
    cp.close()
    # and they verify, masks included
    masks = bpp.RangeProof.verify_batch([bpp.Transcript.new(LABEL)] * count, sts, got, bpp.VerifyAction.RecoverAndVerify)
    for i in range(count):
        if raw["seeds"][i] is not None:
            assert masks[i].blindings() == raw["blinds"][i][0]
        else:
            assert masks[i] is None


def test_prover_without_seed_nonce_and_with_transcript_state(bpp, engine):
    """m = 1 without a seed nonce (all nonces from the transcript RNG) and a caller transcript that is not fresh"""
    n, m, t = 16, 1, 2
    params = bpp.RangeParameters.init(n, m, bpp.create_pedersen_gens_with_extension_degree(t), engine=engine)
    sts, wits, exts, raw = _inputs(bpp, params, n, m, t, 2, b"noseed", with_seed=False)
    t0 = M.Transcript(b"outer protocol")
    t0.append_message(b"ctx", b"some application data")
    tr = bpp.Transcript.from_state(t0.strobe.to_bytes())
    got = bpp.RangeProof.prove_batch([tr, tr], sts, wits, exts)
    op = O.RangeParameters(n, m, O.PedersenGens(t))
    for i in range(2):
        ost = O.RangeStatement(op, [C.decompress(c) for c in raw["comms"][i]], raw["mins"][i], None)
        ow = O.RangeWitness([O.CommitmentOpening(raw["vals"][i][j], [int.from_bytes(x, "little") for x in raw["blinds"][i][j]])
                             for j in range(m)])
        want = O.prove_with_rng(t0.clone(), ost, ow, M.ByteStreamRng(exts[i]))
        assert got[i].to_bytes() == want.to_bytes()
    assert bpp.RangeProof.verify_batch([tr, tr], sts, got, bpp.VerifyAction.VerifyOnly) == [None, None]


def test_prover_error_paths(bpp, engine):
    """src/range_proof.rs:1672-1756 (inconsistent witness) and tests/ristretto.rs:230-241 (vmin > v)"""
    K = bpp.ProofErrorKind
    n, m, t = 8, 2, 1
    params = bpp.RangeParameters.init(n, m, bpp.create_pedersen_gens_with_extension_degree(t), engine=engine)
    sts, wits, exts, raw = _inputs(bpp, params, n, m, t, 1, b"errs")
    tr = [bpp.Transcript.new(LABEL)]
    P = bpp.RangeProof.prove_batch
    # wrong number of openings
    w1 = bpp.RangeWitness.init(wits[0].openings[:1])
    assert _kind(bpp, lambda: P(tr, sts, [w1], exts)) == K.InvalidLength
    # wrong extension degree
    w2 = bpp.RangeWitness.init([bpp.CommitmentOpening.new(o.v, o.r * 2) for o in wits[0].openings])
    assert _kind(bpp, lambda: P(tr, sts, [w2], exts)) == K.InvalidLength
    # value does not fit the bit length
    w3 = bpp.RangeWitness.init([bpp.CommitmentOpening.new(1 << n, wits[0].openings[0].r), wits[0].openings[1]])
    assert _kind(bpp, lambda: P(tr, sts, [w3], exts)) == K.InvalidLength
    # opening that does not match the commitment
    w4 = bpp.RangeWitness.init([bpp.CommitmentOpening.new(wits[0].openings[0].v ^ 1, wits[0].openings[0].r), wits[0].openings[1]])
    assert _kind(bpp, lambda: P(tr, sts, [w4], exts)) == K.InvalidArgument
    w5 = bpp.RangeWitness.init([bpp.CommitmentOpening.new(wits[0].openings[0].v, [sb(12345)]), wits[0].openings[1]])
    assert _kind(bpp, lambda: P(tr, sts, [w5], exts)) == K.InvalidArgument
    # minimum value promise above the value
    st6 = bpp.RangeStatement.init(params, raw["comms"][0], [raw["vals"][0][0] + 1, None], None)
    assert _kind(bpp, lambda: P(tr, [st6], wits, exts)) == K.InvalidArgument
    # not enough external randomness
    assert _kind(bpp, lambda: P(tr, sts, wits, [exts[0][:-32]])) == K.InvalidLength
    assert bpp.CommitmentOpening.new(0, []).r == [] and _kind(bpp, lambda: bpp.CommitmentOpening.new(0, []).r_len()) == K.InvalidLength
    assert _kind(bpp, lambda: bpp.RangeWitness.init([])) == K.InvalidLength


def test_prove_and_verify_round_trip_large(bpp, engine):
    """BASELINE configs[4] shape, smaller count: aggregated m=4, extension degree 3, 64-bit; engine-made proofs verify
    in one batch, tampering is caught, and a sample equals the C oracle byte for byte"""
    n, m, t, count = 64, 4, 3, 24
    params = bpp.RangeParameters.init(n, m, bpp.create_pedersen_gens_with_extension_degree(t), engine=engine)
    sts, wits, exts, raw = _inputs(bpp, params, n, m, t, count, b"cfg5")
    trs = [bpp.Transcript.new(LABEL)] * count
    proofs = bpp.RangeProof.prove_batch(trs, sts, wits, exts)
    assert all(len(p.to_bytes()) == 769 for p in proofs)
    assert bpp.RangeProof.verify_batch(trs, sts, proofs, bpp.VerifyAction.VerifyOnly) == [None] * count
    cp = cport.Params(n, m, t)
    for i in (0, count - 1):
        want, _ = cp.prove(LABEL, raw["vals"][i], raw["blinds"][i], raw["mins"][i], None, exts[i])
        assert proofs[i].to_bytes() == want
    cp.close()
    swapped = [proofs[1], proofs[0]] + proofs[2:]
    assert _kind(bpp, lambda: bpp.RangeProof.verify_batch(trs, sts, swapped, bpp.VerifyAction.VerifyOnly)) == \
        bpp.ProofErrorKind.VerificationFailed


@pytest.mark.parametrize("wbits", [8, 9, 10, 11])
def test_every_fixed_base_window_width(bpp, wbits, monkeypatch):
    """the prover's fixed-base tables pick their window width from the size of the parameter set (8..11 bits); force each
    one on a small set: proof bytes and Pedersen commitments must not depend on it"""
    monkeypatch.setenv("BPP_FB_WBITS", str(wbits))
    eng = bpp.Engine(0)
    try:
        n, m, t = 16, 2, 2
        params = bpp.RangeParameters.init(n, m, bpp.create_pedersen_gens_with_extension_degree(t), engine=eng)
        sts, wits, exts, raw = _inputs(bpp, params, n, m, t, 2, b"wbits", "third")
        got = bpp.RangeProof.prove_batch([bpp.Transcript.new(LABEL)] * 2, sts, wits, exts)
        cp = cport.Params(n, m, t)
        for i in range(2):
            want, comm = cp.prove(LABEL, raw["vals"][i], raw["blinds"][i], raw["mins"][i], raw["seeds"][i], exts[i])
            assert comm == raw["comms"][i]  # commit_many went through the resident Pedersen table of the same width
            assert got[i].to_bytes() == want
        cp.close()
    finally:
        eng.close()


def test_prover_baseline_cfg5_full_size(bpp, engine):
    """BASELINE configs[4]: 1024 aggregation-4 proofs, extension degree 3, 64 bits, in one bpp_prove_batch call.  Full-size
    properties: every proof verifies (round trip through the engine's verifier, one reference batch of 1024), a sample is
    byte-identical to the oracle's prover, and a proof bound to another statement is rejected"""
    n, m, t, count = 64, 4, 3, 1024
    params = bpp.RangeParameters.init(n, m, bpp.create_pedersen_gens_with_extension_degree(t), engine=engine)
    rng = Prng(b"cfg5-full")
    rounds = (n * m).bit_length() - 1
    blind = [sb(O.random_not_zero(rng)) for _ in range(t)]
    vals = [[rng.next_u64() % (1 << 63) for _ in range(m)] for _ in range(count)]
    mins = [[v // 3 for v in vs] for vs in vals]
    exts = [rng.fill_bytes(32 * (rounds + 3)) for _ in range(count)]
    comms = params.commit_many([x for v in vals for x in v], [blind] * (count * m))
    comms = [comms[i * m:(i + 1) * m] for i in range(count)]
    sts = [bpp.RangeStatement.init(params, comms[i], mins[i], None) for i in range(count)]
    wits = [bpp.RangeWitness.init([bpp.CommitmentOpening.new(vals[i][j], blind) for j in range(m)]) for i in range(count)]
    trs = [bpp.Transcript.new(LABEL)] * count
    proofs = bpp.RangeProof.prove_batch(trs, sts, wits, exts)
    assert len(proofs) == count and all(len(p.to_bytes()) == 1 + 32 * (t + 5 + 2 * rounds) for p in proofs)
    assert bpp.RangeProof.verify_batch(trs, sts, proofs, bpp.VerifyAction.VerifyOnly) == [None] * count
    cp = cport.Params(n, m, t)
    for i in (0, 1, 511, 1023):
        want, comm = cp.prove(LABEL, vals[i], [blind] * m, mins[i], None, exts[i])
        assert comm == comms[i] and proofs[i].to_bytes() == want
    cp.close()
    swapped = list(proofs)
    swapped[3], swapped[4] = swapped[4], swapped[3]
    assert _kind(bpp, lambda: bpp.RangeProof.verify_batch(trs, sts, swapped, bpp.VerifyAction.VerifyOnly)) == \
        bpp.ProofErrorKind.VerificationFailed
