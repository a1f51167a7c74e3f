"""GPU parity tests (run with -m gpu on an MI355X): the HIP path through the C ABI vs the oracle, bit-exact.

Mirrors the reference's integration tests (tests/ristretto.rs:24-142 -> prove_and_verify :152-373): same shapes, same
assertions, plus differential comparison of every intermediate the reference computes."""
import hashlib

import pytest

from oracle.pyref import curve as C
from oracle.pyref import protocol as O
from tests.helpers import make_batch, oracle_verify_trace, sb, trace_challenge_bytes

pytestmark = pytest.mark.gpu


def _h(tag, i, n=32):
    return hashlib.shake_256(b"%s-%d" % (tag, i)).digest(n)


def test_generators_match_oracle(bpp, engine):
    """BulletproofGens::new / masking base points (bulletproof_gens.rs:83-112, ristretto.rs:88-112) on the device"""
    params = bpp.RangeParameters.init(64, 2, bpp.create_pedersen_gens_with_extension_degree(6), engine=engine)
    og = O.BulletproofGens(64, 2)
    assert params.gi_base_compressed() == [p.compress() for p in og.g_iter(64, 2)]
    assert params.hi_base_compressed() == [p.compress() for p in og.h_iter(64, 2)]
    assert params.h_base_compressed() == C.BASEPOINT.compress()
    assert params.g_bases_compressed() == [p.compress() for p in O.ristretto_masking_basepoints()]


@pytest.mark.parametrize("n", [1, 2, 3, 17, 64, 200, 1000])
def test_msm_vartime_matches_oracle(bpp, engine, n):
    pts = [C.from_uniform_bytes(_h(b"p", i, 64)) for i in range(min(n, 40))]
    pts = [pts[i % len(pts)] for i in range(n)]  # repeated points on purpose
    scalars = [int.from_bytes(_h(b"s", i), "little") % C.L for i in range(n)]
    for i, v in zip(range(n), [0, 1, C.L - 1, 2**252, 2**128]):  # adversarial scalars
        scalars[i] = v
    got = engine.msm_vartime([sb(s) for s in scalars], [p.compress() for p in pts])
    assert got == C.multiscalar_mul(scalars, pts).compress()


def test_msm_edge_cases(bpp, engine):
    ident = bytes(32)
    b = C.BASEPOINT.compress()
    assert engine.msm_vartime([], []) == ident
    assert engine.msm_vartime([sb(0)], [b]) == ident
    assert engine.msm_vartime([sb(5), sb(C.L - 5)], [b, b]) == ident
    assert engine.msm_vartime([sb(7)], [ident]) == ident
    with pytest.raises(bpp.ProofError) as e:
        engine.msm_vartime([sb(1)], [b"\x01" + bytes(31)])  # negative s: not a valid encoding
    assert e.value.kind == bpp.ProofErrorKind.InvalidArgument
    with pytest.raises(bpp.ProofError):
        engine.msm_vartime([C.L.to_bytes(32, "little")], [b])  # non-canonical scalar


def test_msm_mixed_and_batched(bpp, engine):
    pts = [C.from_uniform_bytes(_h(b"q", i, 64)) for i in range(24)]
    sc = [int.from_bytes(_h(b"t", i), "little") % C.L for i in range(24)]
    pre = engine.precomputation([p.compress() for p in pts[:16]])
    got = pre.vartime_mixed_multiscalar_mul([sb(s) for s in sc[:10]], [sb(s) for s in sc[16:]],
                                            [p.compress() for p in pts[16:]])
    assert got == C.multiscalar_mul(sc[:10] + sc[16:], pts[:10] + pts[16:]).compress()
    pre.close()
    off = [0, 3, 3, 10, 24]
    got = engine.msm_vartime_batched([sb(s) for s in sc], [p.compress() for p in pts], off)
    want = [C.multiscalar_mul(sc[a:b], pts[a:b]).compress() for a, b in zip(off, off[1:])]
    assert got == want


def test_pedersen_commit(bpp, engine):
    params = bpp.RangeParameters.init(8, 1, bpp.create_pedersen_gens_with_extension_degree(3), engine=engine)
    og = O.PedersenGens(3)
    vals = [0, 1, 2**64 - 1, 123456789]
    blinds = [[int.from_bytes(_h(b"b", 3 * i + k), "little") % C.L for k in range(3)] for i in range(4)]
    got = params.commit_many(vals, [[sb(x) for x in b] for b in blinds])
    assert got == [og.commit(v, b).compress() for v, b in zip(vals, blinds)]
    assert params.commit(5, [sb(9)]) == og.commit(5, [9]).compress()  # 1..=t blindings accepted (ristretto.rs:152-176)
    with pytest.raises(bpp.ProofError) as e:
        params.commit(5, [])
    assert e.value.kind == bpp.ProofErrorKind.InvalidLength


SHAPES = [
    # (bit_length, aggregation batch, extension degree, min-value strategy)  -- tests/ristretto.rs:24-142
    (8, [1], 1, "none"),
    (64, [1], 2, "third"),
    (4, [4], 1, "none"),
    (32, [4], 2, "third"),
    (64, [1, 1], 3, "eq"),
    (64, [1, 2], 1, "third"),
]


@pytest.mark.parametrize("n,batch,t,strategy", SHAPES)
def test_prove_and_verify_shapes(bpp, engine, n, batch, t, strategy):
    case = make_batch(bpp, engine, n, batch, t, seed=b"shape-%d-%d" % (n, t), strategy=strategy)
    A = bpp.VerifyAction
    want_priv, _ = oracle_verify_trace(case, action=1)
    # 5. verify as the commitment owner: RecoverOnly / RecoverAndVerify / VerifyOnly  (tests/ristretto.rs:254-289)
    for action in (A.RecoverOnly, A.RecoverAndVerify):
        got = bpp.RangeProof.verify_batch(case.transcripts(), case.statements_private, case.proofs, action)
        assert [m.blindings() if m else None for m in got] == want_priv
        assert [m.blindings() if m else None for m in got] == \
            [[sb(x) for x in m] if m else None for m in case.expected_masks]
    got = bpp.RangeProof.verify_batch(case.transcripts(), case.statements_private, case.proofs, A.VerifyOnly)
    assert got == [None] * len(batch)
    # 6. public entity
    got = bpp.RangeProof.verify_batch(case.transcripts(), case.statements_public, case.proofs, A.VerifyOnly)
    assert got == [None] * len(batch)
    # 7. wrong seed nonce: still Ok, different masks (:291-318)
    if any(m == 1 for m in batch):
        changed = [bpp.RangeStatement.init(case.params, s.commitments_compressed, s.minimum_value_promises,
                                           sb((int.from_bytes(s.seed_nonce, "little") + 1) % C.L) if s.seed_nonce else None)
                   for s in case.statements_private]
        got = bpp.RangeProof.verify_batch(case.transcripts(), changed, case.proofs, A.RecoverAndVerify)
        assert [m.blindings() if m else None for m in got] != want_priv
    # 8. bumped minimum-value promise -> VerificationFailed (:320-356)
    bumped = [bpp.RangeStatement.init(case.params, s.commitments_compressed,
                                      [min(v + 1, 2**64 - 1) if v is not None else 1 for v in s.minimum_value_promises],
                                      None) for s in case.statements_public]
    with pytest.raises(bpp.ProofError) as e:
        bpp.RangeProof.verify_batch(case.transcripts(), bumped, case.proofs, A.VerifyOnly)
    assert e.value.kind == bpp.ProofErrorKind.VerificationFailed


@pytest.mark.parametrize("n,batch,t", [(8, [1, 2, 1, 4], 2), (64, [1, 1, 1], 1), (64, [2, 1], 1)])
def test_intermediates_bit_exact(bpp, engine, n, batch, t):
    """challenges, RNG outputs, weights, accumulated generator scalars, dynamic scalars, MSM result vs the oracle"""
    case = make_batch(bpp, engine, n, batch, t, seed=b"trace-%d" % n)
    _, tr = oracle_verify_trace(case, action=0)
    rb = bpp.ResidentBatch(case.transcripts(), case.statements_public, case.proofs)
    rb.verify(bpp.VerifyAction.VerifyOnly, chunk=0)
    shp = rb.shape()
    assert shp["max_mn"] == tr["max_mn"] and shp["groups"] == 1
    assert rb.trace(1) == trace_challenge_bytes(tr, shp["max_rounds"])
    assert rb.trace(2) == b"".join(tr["rng_outputs"])
    assert rb.trace(3) == b"".join(sb(w) for w in tr["weights"])
    static = b"".join(sb(g) + sb(h) for g, h in zip(tr["gi"], tr["hi"])) + b"".join(sb(x) for x in tr["g"]) + sb(tr["h"])
    assert rb.trace(4) == static
    assert rb.trace(5) == b"".join(sb(x) for x in tr["dynamic_scalars"])
    assert rb.trace(6) == tr["msm_result"] == bytes(32)
    rb.close()


@pytest.mark.parametrize("wave", ["0", "1"])
def test_both_pass1_kernels(bpp, engine, opt, wave):
    """PASS 1 has a one-lane-per-proof kernel (large inputs) and a one-wavefront-per-proof kernel on the cooperative
    sponge (small inputs); force each: challenges and transcript-RNG bytes must equal the oracle's, identity members must
    be reported by both"""
    opt("transcripts_wave", int(wave))
    case = make_batch(bpp, engine, 16, [1, 2, 1, 4], 2, seed=b"pass1-kernels")
    _, tr = oracle_verify_trace(case, action=0)
    rb = bpp.ResidentBatch(case.transcripts(), case.statements_public, case.proofs)
    rb.verify(bpp.VerifyAction.VerifyOnly, chunk=0)
    assert rb.trace(1) == trace_challenge_bytes(tr, rb.shape()["max_rounds"])
    assert rb.trace(2) == b"".join(tr["rng_outputs"])
    rb.close()
    raw = bytearray(case.proofs[2].to_bytes())
    raw[1 + 32 * 2 + 32:1 + 32 * 2 + 64] = bytes(32)  # A1 = identity encoding
    bad = list(case.proofs)
    bad[2] = bpp.RangeProof.from_bytes(bytes(raw))
    with pytest.raises(bpp.ProofError) as e:
        bpp.RangeProof.verify_batch(case.transcripts(), case.statements_public, bad, bpp.VerifyAction.VerifyOnly)
    assert e.value.kind == bpp.ProofErrorKind.VerificationFailed


@pytest.mark.parametrize("wave", ["0", "1"])
@pytest.mark.parametrize("n,ms", [(16, [1, 2, 1, 4]), (2, [1, 4, 2]), (64, [8, 1])])
def test_both_table_kernels(bpp, engine, opt, wave, n, ms):
    """the tables of the generator-row kernel are built by one lane per proof (large inputs) or one wavefront per proof (small
    inputs); force each: static and dynamic MSM scalars must equal the oracle's for mixed aggregation and small bit lengths"""
    opt("tables_wave", int(wave))
    case = make_batch(bpp, engine, n, ms, 2, seed=b"table-kernels-%d" % n)
    _, tr = oracle_verify_trace(case, action=0)
    rb = bpp.ResidentBatch(case.transcripts(), case.statements_public, case.proofs)
    assert rb.verify(bpp.VerifyAction.VerifyOnly, chunk=0) == [None] * len(ms)
    static = b"".join(sb(g) + sb(h) for g, h in zip(tr["gi"], tr["hi"])) + b"".join(sb(x) for x in tr["g"]) + sb(tr["h"])
    assert rb.trace(4) == static
    assert rb.trace(5) == b"".join(sb(x) for x in tr["dynamic_scalars"])
    assert rb.trace(6) == tr["msm_result"] == bytes(32)
    rb.close()


@pytest.mark.parametrize("bias", ["0", "1", "3", "7"])
def test_small_call_window_widths(bpp, engine, opt, bias):
    """small calls take wider MSM windows than the throughput rule gives (BPP_MSM_C_BIAS, default 3): every width from 4 to 11
    bits goes through the quad bucket kernels here, on multiscalar products of 1..300 terms and on small proof batches"""
    opt("msm_c_bias", int(bias))
    for n in (1, 2, 5, 40, 300):
        pts = [C.from_uniform_bytes(_h(b"cw-p", i, 64)) for i in range(min(n, 24))]
        pts = [pts[i % len(pts)] for i in range(n)]
        scalars = [int.from_bytes(_h(b"cw-s%s" % bias.encode(), i), "little") % C.L for i in range(n)]
        got = engine.msm_vartime([sb(s_) for s_ in scalars], [p_.compress() for p_ in pts])
        assert got == C.multiscalar_mul(scalars, pts).compress(), (bias, n)
    case = make_batch(bpp, engine, 16, [1, 2, 1], 1, seed=b"window-widths")
    rb = bpp.ResidentBatch(case.transcripts(), case.statements_public, case.proofs)
    assert rb.verify(bpp.VerifyAction.VerifyOnly, chunk=0) == [None] * 3
    assert rb.trace(6) == bytes(32)
    rb.close()
    bumped = [bpp.RangeStatement.init(case.params, s_.commitments_compressed, [(v or 0) + 1 for v in s_.minimum_value_promises], None)
              for s_ in case.statements_public]
    with pytest.raises(bpp.ProofError) as e:
        bpp.RangeProof.verify_batch(case.transcripts(), bumped, case.proofs, bpp.VerifyAction.VerifyOnly)
    assert e.value.kind == bpp.ProofErrorKind.VerificationFailed


@pytest.mark.parametrize("side", ["0", "1"])
def test_decompression_beside_pass1(bpp, engine, opt, side):
    """small inputs decompress on a second stream while PASS 1 runs; force each form: same dynamic points, same verdicts, and an
    undecodable point plus a transcript failure in one batch still surface in the reference's order"""
    opt("side_decompress", int(side))
    case = make_batch(bpp, engine, 16, [1, 2, 4, 1, 1], 1, seed=b"side-decompress")
    _, tr = oracle_verify_trace(case, action=0)
    for _ in range(3):  # the streams are reused call after call
        rb = bpp.ResidentBatch(case.transcripts(), case.statements_public, case.proofs)
        assert rb.verify(bpp.VerifyAction.VerifyOnly, chunk=0) == [None] * 5
        assert rb.trace(5) == b"".join(sb(x) for x in tr["dynamic_scalars"])
        assert rb.trace(6) == tr["msm_result"] == bytes(32)
        rb.close()
    K = bpp.ProofErrorKind
    V = lambda proofs: bpp.RangeProof.verify_batch(case.transcripts(), case.statements_public, proofs, bpp.VerifyAction.VerifyOnly)

    def patched(proofs, i, off, data):
        r = bytearray(proofs[i].to_bytes())
        r[off:off + len(data)] = data
        out = list(proofs)
        out[i] = bpp.RangeProof.from_bytes(bytes(r))
        return out
    offA = 1 + 32
    bad_point = patched(case.proofs, 0, offA, b"\x01" + bytes(31))  # negative field element: never a valid encoding
    with pytest.raises(bpp.ProofError) as e:
        V(bad_point)
    assert e.value.kind == K.InvalidArgument  # found by the decompression kernel
    with pytest.raises(bpp.ProofError) as e:
        V(patched(bad_point, 3, offA + 32, bytes(32)))  # identity A1 in a later proof: found by PASS 1, which precedes
    assert e.value.kind == K.VerificationFailed


@pytest.mark.parametrize("quad", ["0", "1"])
def test_both_bucket_kernel_forms(bpp, engine, opt, quad):
    """bucket accumulation / row-column reduction exist in a one-lane-per-bucket form (many buckets) and a quad form (few
    buckets, latency); force each on the same input: MSM result, accept / reject and the B1 multiscalar API must agree"""
    opt("msm_quad", int(quad))
    case = make_batch(bpp, engine, 32, [1, 2, 1, 1, 4, 1], 1, seed=b"bucket-forms")
    rb = bpp.ResidentBatch(case.transcripts(), case.statements_public, case.proofs)
    assert rb.verify(bpp.VerifyAction.VerifyOnly, chunk=0) == [None] * 6
    assert rb.trace(6) == bytes(32)
    rb.close()
    raw = bytearray(case.proofs[4].to_bytes())
    raw[1 + 32 + 96 + 5] ^= 2  # r1
    bad = list(case.proofs)
    bad[4] = bpp.RangeProof.from_bytes(bytes(raw))
    with pytest.raises(bpp.ProofError) as e:
        bpp.RangeProof.verify_batch(case.transcripts(), case.statements_public, bad, bpp.VerifyAction.VerifyOnly)
    assert e.value.kind == bpp.ProofErrorKind.VerificationFailed
    n = 300
    pts = [C.from_uniform_bytes(_h(b"bf-p", i, 64)) for i in range(40)]
    pts = [pts[i % 40] for i in range(n)]
    scalars = [int.from_bytes(_h(b"bf-s", i), "little") % C.L for i in range(n)]
    got = engine.msm_vartime([sb(s_) for s_ in scalars], [p_.compress() for p_ in pts])
    assert got == C.multiscalar_mul(scalars, pts).compress()
