"""GPU tests of round 6: the batch-weight chains as a device kernel (option "chain" = 1, csrc/chain_dev.h + csrc/wkeccak.h).

Reference: the weight transcript of RangeProof::verify (src/range_proof.rs:811 Transcript::new, :849 append_message per proof, :853
build_rng().finalize(NullRng), :894 one Scalar::random_not_zero per proof; src/protocols/scalar_protocol.rs:23-30 redraws a zero,
src/utils/nullrng.rs:16-40).  Option "chain" = 2 keeps the sponges on host cores and moves only
Scalar::from_bytes_mod_order_wide and the look for a zero weight to the device (the default for calls of 4096 proofs and more).
Either form must leave byte for byte the weights of the CPU oracle (oracle/c through
oracle.cport) and of the host form (csrc/chain_host.h), for whole reference batches of any size and for ragged groups."""
import importlib

import numpy as np
import pytest

from oracle import cport
from tests.helpers import LABEL

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def packed():
    return importlib.import_module("bulletproofs-plus_amd.packed")


@pytest.fixture(scope="module")
def cfg2(bpp, packed, engine):
    """4096 + 452 non-aggregated 64-bit proofs with bench.py's recipe (benches/range_proof.rs:206-262)"""
    import bench
    params = bpp.RangeParameters.init(64, 1, bpp.create_pedersen_gens_with_extension_degree(1), engine=engine)
    d = bench.make_inputs(np, packed, params, 4096 + 452, seed=20261005)
    yield params, d
    params.close()


def _resident(packed, params, d, lo, hi):
    return packed.ResidentBatch(params, d["proofs"][lo:hi], d["commitments"][lo:hi], d["min_values"][lo:hi], d["min_present"][lo:hi],
                                None, LABEL)


def _oracle_weights(d, lo, hi):
    cp = cport.Params(64, 1, 1)
    items = [{"proof": bytes(d["proofs"][i]), "commitments": [bytes(d["commitments"][i, 0])], "min_values": [int(d["min_values"][i, 0])],
              "seed_nonce": None, "label": LABEL} for i in range(lo, hi)]
    rc, _, tr = cp.verify(items, action=0, want_trace=True)
    cp.close()
    assert rc == 0
    return tr["weights"]


@pytest.mark.parametrize("chain", [1, 2])
@pytest.mark.parametrize("n", [1, 3, 4, 5, 64, 1024, 4096])
def test_device_chain_weights_equal_the_oracle(bpp, packed, engine, opt, cfg2, n, chain):
    """one reference batch of n proofs, chains on the device: accepted, MSM result the identity, and the n weights are the CPU
    oracle's and the host chain's, byte for byte (n = 3, 4, 5: a record of 45 bytes crosses the sponge's 166-byte block at the
    fourth proof; 1024 and 4096: the sizes BASELINE's configs quote).  chain = 2: the sponge on a host core, only the reduction
    mod l and the look for a zero on the device."""
    params, d = cfg2
    rb = _resident(packed, params, d, 0, n)
    opt("chain", chain)
    before = engine.device_chain_stats()
    rb.verify_only(chunk=0)
    after = engine.device_chain_stats()
    assert after[0] == before[0] + 1 and after[1] == before[1]  # the chain kernel ran, nothing was redrawn
    w_dev = rb.trace(3)
    assert rb.trace(6) == bytes(32)
    opt("chain", 0)
    rb.verify_only(chunk=0)
    assert engine.device_chain_stats() == after
    w_host = rb.trace(3)
    rb.close()
    assert len(w_dev) == 32 * n and w_dev == w_host
    assert w_dev == _oracle_weights(d, 0, n)


@pytest.mark.parametrize("chain", [1, 2])
def test_device_chain_per_group_and_ragged(bpp, packed, engine, opt, cfg2, chain):
    """several reference batches in one call (chunk = 1024 over 4548 proofs: four full groups and one of 452): every group's
    chain is its own wavefront; the weights are those of the host chains, and a group's weights are those of the same proofs
    verified alone.  Then ragged groups through bpp_verify_resident_groups (the batcher's form)."""
    params, d = cfg2
    n = 4096 + 452
    rb = _resident(packed, params, d, 0, n)
    opt("chain", chain)
    rb.verify_only(chunk=1024)
    assert rb.shape()["groups"] == 5
    w_dev = rb.trace(3)
    assert rb.trace(6) == bytes(32) * 5
    opt("chain", 0)
    rb.verify_only(chunk=1024)
    assert rb.trace(3) == w_dev
    # ragged groups: 1, 700, 1024, 2823 proofs
    bounds = [0, 1, 701, 1725, n]
    opt("chain", chain)
    res = packed.verify_groups(rb, bounds)
    assert all(r["code"] == 0 for r in res)
    w_rag = rb.trace(3)
    opt("chain", 0)
    res = packed.verify_groups(rb, bounds)
    assert all(r["code"] == 0 for r in res)
    assert rb.trace(3) == w_rag
    rb.close()
    assert w_rag[32 * 1:32 * 701] == _oracle_weights(d, 1, 701)
    one = _resident(packed, params, d, 4096, n)
    opt("chain", chain)
    one.verify_only(chunk=0)
    assert one.trace(3) == w_dev[32 * 4096:]
    one.close()


def test_device_chain_rejects_what_the_host_chain_rejects(bpp, packed, engine, opt, cfg2):
    """a tampered proof in the third of four 1024-proof batches: the same error kind, the same non-identity point for that batch and
    the identity for the others, whichever side ran the chains"""
    params, d = cfg2
    bad = d["proofs"][:4096].copy()
    bad[2 * 1024 + 77, 1 + 32 + 96 + 5] ^= 2  # r1
    out = {}
    for chain in (1, 2, 0):
        opt("chain", chain)
        rb = packed.ResidentBatch(params, bad, d["commitments"][:4096], d["min_values"][:4096], d["min_present"][:4096], None, LABEL)
        with pytest.raises(bpp.ProofError) as e:
            rb.verify_only(chunk=1024)
        out[chain] = (e.value.kind, rb.trace(6), rb.trace(3))
        rb.close()
    assert out[0] == out[1] == out[2]
    assert out[1][0] == bpp.ProofErrorKind.VerificationFailed
    pts = out[1][1]
    assert [pts[32 * g:32 * g + 32] == bytes(32) for g in range(4)] == [True, True, False, True]


@pytest.mark.parametrize("chain", [1, 2])
def test_device_chain_zero_weight_goes_back_to_the_host_chain(bpp, packed, engine, opt, cfg2, chain):
    """Scalar::random_not_zero redraws a zero weight (src/protocols/scalar_protocol.rs:23-30).  The device chain cannot (the next
    draw would shift every later weight): it reports the zero and the call runs once more with the chains on the host.  The
    test hook makes k_chain_finish report proof 700's weight as zero."""
    params, d = cfg2
    rb = _resident(packed, params, d, 0, 2048)
    opt("chain", chain)
    opt("chain_test_zero", 701)
    before = engine.device_chain_stats()
    rb.verify_only(chunk=1024)
    after = engine.device_chain_stats()
    assert after == (before[0] + 1, before[1] + 1)
    w = rb.trace(3)
    res = packed.verify_groups(rb, [0, 1024, 2048])
    assert all(r["code"] == 0 for r in res) and engine.device_chain_stats() == (before[0] + 2, before[1] + 2)
    opt("chain_test_zero", 0)
    opt("chain", 0)
    rb.verify_only(chunk=1024)
    assert rb.trace(3) == w
    rb.close()
    engine.set_option("chain_test_zero", 0)  # (the fixture restores options to -1: this one's neutral value is 0)


def test_wide_reduction_in_the_lanes_kernel_or_on_its_own(bpp, packed, engine, opt, cfg2):
    """chain = 2: the reduction mod l of the host sponges' 64 bytes per proof runs in k_scalars_lanes' prologue (option "wide_in_lanes",
    the rule) or as k_chain_finish_bytes in front of it (0, and whenever the generator columns are a matrix product): the same weights as
    the oracle's either way, ragged last workgroup included, and a zero weight found in the prologue sends the call back to the host
    chains just the same"""
    params, d = cfg2
    n = 1024 + 452
    rb = _resident(packed, params, d, 0, n)
    opt("chain", 2)
    want = _oracle_weights(d, 0, n)
    for in_lanes in (1, 0):
        opt("wide_in_lanes", in_lanes)
        rb.verify_only(chunk=0)
        assert rb.trace(6) == bytes(32) and rb.trace(3) == want, in_lanes
    opt("wide_in_lanes", 1)
    opt("chain_test_zero", n)  # the last proof, in the partly filled last workgroup
    before = engine.device_chain_stats()
    rb.verify_only(chunk=0)
    assert engine.device_chain_stats() == (before[0] + 1, before[1] + 1)
    assert rb.trace(6) == bytes(32) and rb.trace(3) == want
    opt("chain_test_zero", 0)
    engine.set_option("chain_test_zero", 0)
    rb.close()


def test_device_chain_stage_profile(bpp, packed, engine, opt, cfg2):
    """with stage profiling on the chain runs in line on the call's stream and its interval is reported as chain_device_ms; the
    host chain's wall time stays zero"""
    params, d = cfg2
    rb = _resident(packed, params, d, 0, 4096)
    engine.profile(True)
    try:
        opt("chain", 1)
        rb.verify_only(chunk=1024)
        pf = engine.last_profile()
        assert pf["chain_device_ms"] > 0.1 and pf["msm_final_ms"] > 0
        opt("chain", 0)
        rb.verify_only(chunk=1024)
        pf = engine.last_profile()
        assert pf["chain_device_ms"] == 0 and pf["chain_host_ms"] > 0
    finally:
        engine.profile(False)
        rb.close()


def test_north_star_target_shape_4096_aggregated_proofs_as_one_batch(bpp, packed, engine, opt):
    """BASELINE.json's target sentence: "aggregated 64-bit range-proof verifications ... on a batch of 4096 on one MI355X, bit-exact vs
    reference".  4096 aggregation-8 proofs (bench.py's recipe, benches/range_proof.rs:206-262) as ONE reference batch (chunk = 0: one
    weight chain over 4096 proofs, 1024 generator columns, a 119 809-term MSM; src/range_proof.rs:756-1065): accepted, and the weights,
    the accumulated generator scalars, every dynamic scalar and the final point are the C oracle's for the same batch -- with the
    generator columns as the engine's rule takes them (the int8 matrix product) and as per-proof products; a tampered proof leaves the
    same non-identity point in both forms and the oracle's verdict"""
    import struct
    import bench
    params = bpp.RangeParameters.init(64, 8, bpp.create_pedersen_gens_with_extension_degree(1), engine=engine)
    n = 4096
    d = bench.make_inputs(np, packed, params, n, seed=20261006)
    cp = cport.Params(64, 8, 1)
    items = bench.cpu_items(d, range(n))
    rc, _, tr = cp.verify(items, action=0, want_trace=True)
    assert rc == 0 and tr["msm_result"] == bytes(32)
    rb = packed.ResidentBatch(params, d["proofs"], d["commitments"], d["min_values"], d["min_present"], None, LABEL)
    forms = {}
    for gemm in (-1, 0):
        opt("static_gemm", gemm)
        rb.verify_only(chunk=0)
        shp = rb.shape()
        assert shp["groups"] == 1 and shp["max_mn"] == 512 and shp["total_dyn"] == n * 29
        forms[gemm] = struct.unpack("<4I", rb.trace(7))[0] & 2
        assert rb.trace(3) == tr["weights"], gemm
        assert rb.trace(4) == tr["static_scalars"], gemm
        assert rb.trace(5) == tr["dynamic_scalars"], gemm
        assert rb.trace(6) == tr["msm_result"], gemm
    assert forms[-1] == 2 and forms[0] == 0  # the rule took the matrix product; the other run did not
    rb.close()
    bad = d["proofs"].copy()
    bad[2999, 1 + 32 + 96 + 3] ^= 0x20  # r1 of one proof: the batch fails in its sum only
    items[2999] = dict(items[2999], proof=bad[2999].tobytes())
    rc_bad, _, tr_bad = cp.verify(items, action=0, want_trace=True)
    cp.close()
    assert rc_bad == int(bpp.ProofErrorKind.VerificationFailed)
    points = []
    rb = packed.ResidentBatch(params, bad, d["commitments"], d["min_values"], d["min_present"], None, LABEL)
    for gemm in (-1, 0):
        opt("static_gemm", gemm)
        with pytest.raises(bpp.ProofError) as e:
            rb.verify_only(chunk=0)
        assert e.value.kind == bpp.ProofErrorKind.VerificationFailed
        points.append(rb.trace(6))
    rb.close()
    params.close()
    assert points[0] == points[1] == tr_bad["msm_result"] != bytes(32)  # (the oracle's sum as well: the same group element)


def test_wait_memory_is_per_workload(bpp, packed):
    """A context remembers how long its calls take and sleeps most of that in one piece (csrc/engine.hip: gpu_wait_event) -- for the SAME
    work only: one build slept through 1024-proof prover calls on the memory of the 8192-proof calls the context had made before
    (bench.py's prover leg at a third of its rate).  A small call after large ones must take what it takes on a fresh context."""
    import time
    import bench
    eng = bpp.Engine(0)
    params = bpp.RangeParameters.init(64, 1, bpp.create_pedersen_gens_with_extension_degree(1), engine=eng)
    big = bench.make_inputs(np, packed, params, 8192, seed=5)  # one 8192-proof prover call on this context
    small = {k: (v[:256] if v is not None else None) for k, v in big.items()}

    def prove(p, d):
        return packed.prove(p, d["values"], d["blindings"], d["commitments"], d["min_values"], d["min_present"], d["seeds"], LABEL, d["ext"])

    def timed_small(p):
        prove(p, small)
        t0 = time.perf_counter()
        for _ in range(4):
            out = prove(p, small)
        return (time.perf_counter() - t0) / 4, out
    for _ in range(3):
        prove(params, big)
    after_big, out1 = timed_small(params)
    eng2 = bpp.Engine(0)
    params2 = params.share(eng2)
    fresh, out2 = timed_small(params2)
    assert (out1 == out2).all() and (out1 == big["proofs"][:256]).all()
    assert after_big < 2.5 * fresh + 1e-3, (after_big, fresh)
    # the verifier: one resident 8192-proof batch, then a 64-proof one on the same context
    rb_big = packed.ResidentBatch(params, big["proofs"], big["commitments"], big["min_values"], big["min_present"], None, LABEL)
    rb_small = packed.ResidentBatch(params, small["proofs"][:64], small["commitments"][:64], small["min_values"][:64], small["min_present"][:64], None, LABEL)
    eng.set_option("wait", 1)  # naps (and the remembered sleep) for calls of every size
    try:
        for _ in range(4):
            rb_big.verify_only(chunk=1024)
        rb_small.verify_only(chunk=0)
        t0 = time.perf_counter()
        for _ in range(4):
            rb_small.verify_only(chunk=0)
        small_ms = 1e3 * (time.perf_counter() - t0) / 4
    finally:
        eng.set_option("wait", -1)
    assert small_ms < 3.0, small_ms  # (0.5 ms for such a call; the 8192-proof call before it takes ~2 ms)
    rb_big.close()
    rb_small.close()
    params2.close()
    eng2.close()
    params.close()
    eng.close()


@pytest.mark.parametrize("n,m,t", [(16, 2, 4), (8, 1, 5), (64, 1, 6), (8, 4, 6), (64, 2, 6)])
def test_highest_extension_degrees(bpp, packed, n, m, t):
    """src/generators/pedersen_gens.rs:42-55: extension degrees up to AddFiveBasePoints (t = 6); the reference's own tests stop at
    t = 3 (tests/ristretto.rs:25-150).  t distinct blinding factors per commitment; prover bytes == the C oracle's in every `ct`
    setting, the proofs verify with the weight chains in each of their three forms (weights == the oracle's), the masks come back
    (m = 1), a flipped bit is VerificationFailed"""
    count = 9
    eng = bpp.Engine(0)
    params = bpp.RangeParameters.init(n, m, bpp.create_pedersen_gens_with_extension_degree(t), engine=eng)
    cp = cport.Params(n, m, t)
    rng = np.random.default_rng(4200 + 100 * t + 10 * m + n)
    rounds = (n * m).bit_length() - 1
    values = rng.integers(0, 1 << min(63, n - 1), size=(count, m), dtype=np.uint64)
    values[0, 0] = (1 << n) - 1 if n < 64 else np.uint64(2**64 - 1)  # the largest value of the range
    bl = rng.integers(0, 256, size=(count, m, t, 32), dtype=np.uint8)
    bl[..., 31] &= 0x0f
    bl[..., 0] |= 1
    min_values = values // np.uint64(3)
    min_present = (rng.integers(0, 3, size=(count, m)) > 0).astype(np.uint8)
    min_values[min_present == 0] = 0
    seeds = None
    if m == 1:
        seeds = rng.integers(0, 256, size=(count, 32), dtype=np.uint8)
        seeds[:, 31] &= 0x0f
    ext = rng.integers(0, 256, size=(count, 32 * (rounds + 3)), dtype=np.uint8)
    comm = packed.commit(params, values.reshape(-1), bl.reshape(count * m, t, 32)).reshape(count, m, 32)
    out = {}
    for ct in (0, 1, 2):
        eng.set_option("ct", ct)
        out[ct] = packed.prove(params, values, bl, comm, min_values, min_present, seeds, LABEL, ext)
    eng.set_option("ct", -1)
    proofs = out[1]
    assert (out[0] == proofs).all() and (out[2] == proofs).all() and proofs.shape[1] == 1 + 32 * (t + 5 + 2 * rounds)
    mins = [[int(v) if p else None for v, p in zip(min_values[i], min_present[i])] for i in range(count)]
    for i in (0, 4, 8):
        want, c = cp.prove(LABEL, [int(x) for x in values[i]], [[bytes(bl[i, j, k]) for k in range(t)] for j in range(m)], mins[i],
                           bytes(seeds[i]) if seeds is not None else None, bytes(ext[i]))
        assert bytes(proofs[i]) == want and [bytes(x) for x in comm[i]] == c
    items = [{"proof": bytes(proofs[i]), "commitments": [bytes(x) for x in comm[i]], "min_values": mins[i],
              "seed_nonce": bytes(seeds[i]) if seeds is not None else None, "label": LABEL} for i in range(count)]
    rc, omasks, tr = cp.verify(items, action=int(bpp.VerifyAction.RecoverAndVerify) if seeds is not None else 0, want_trace=True)
    assert rc == 0
    for chain in (0, 1, 2):
        eng.set_option("chain", chain)
        rb = packed.ResidentBatch(params, proofs, comm, min_values, min_present, seeds, LABEL)
        if seeds is not None:
            masks, present = rb.verify_arrays(int(bpp.VerifyAction.RecoverAndVerify), 0)
            assert bool(present.all()) and (masks == bl[:, 0]).all()
            assert [[bytes(masks[i, k]) for k in range(t)] for i in range(count)] == omasks
        else:
            rb.verify_only(0)
        assert rb.trace(3) == tr["weights"] and rb.trace(6) == bytes(32)
        rb.close()
    eng.set_option("chain", -1)
    bad = proofs.copy()
    bad[5, 1 + 32 * t + 32 * 3 + 7] ^= 0x10  # r1 of proof 5
    rb = packed.ResidentBatch(params, bad, comm, min_values, min_present, seeds, LABEL)
    with pytest.raises(bpp.ProofError) as e:
        rb.verify_only(0)
    assert e.value.kind == bpp.ProofErrorKind.VerificationFailed
    rb.close()
    cp.close()
    params.close()
    eng.close()
