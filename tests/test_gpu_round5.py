"""GPU tests of round 5: the caller-supplied transport of the sharded verifier (bpp_comm_create_callbacks), the multi-rank code as
real PROCESSES on the one GPU of the box, an ADOPTED RCCL communicator under a missed deadline (never aborted behind its owner's
back), the prover's wipe of witness-derived device memory, per-group result words.

Reference: the two couplings of RangeProof::verify that cross ranks (src/range_proof.rs:811-853 one weight transcript over all
proofs, :1050-1062 one group equation); Zeroizing of witness data in the prover (:300-301, :438-464)."""
import ctypes
import importlib
import json
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

from tests.helpers import LABEL

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def packed():
    return importlib.import_module("bulletproofs-plus_amd.packed")


def _make(bpp, packed, engine, m, count, seed, t=1):
    import bench
    params = bpp.RangeParameters.init(64, m, bpp.create_pedersen_gens_with_extension_degree(t), engine=engine)
    d = bench.make_inputs(np, packed, params, count, seed=seed)
    params.close()
    return d


class _ThreadTransport:
    """an all_gather between the threads of this process: what a caller's MPI / TCP channel would be, reduced to a rendezvous"""

    def __init__(self, world):
        self.world, self.slots = world, [None] * world
        self.bar = threading.Barrier(world, timeout=60)
        self.calls = [0] * world
        self.fail_rank = None

    def gather(self, rank):
        def fn(send):
            self.calls[rank] += 1
            if self.fail_rank == rank:
                raise RuntimeError("transport down")
            self.slots[rank] = bytes(send)
            self.bar.wait()
            out = b"".join(self.slots)
            self.bar.wait()
            return out
        return fn


def test_callback_transport_equals_the_in_process_group(bpp, packed, engine):
    """bpp_comm_create_callbacks: three ranks (threads, a context each) exchange through a CALLER-SUPPLIED all_gather on host
    bytes; verdicts, error kind / tier / rank / index on a tampered batch and every rank's batch weights equal what the
    in-process device-to-device group gives -- and that form is held to the single-call form, which is held to the oracle
    (tests/test_gpu_round3.py).  Ragged shards; the grouped entry as well."""
    dmod = importlib.import_module("bulletproofs-plus_amd.dist")
    params = bpp.RangeParameters.init(64, 1, bpp.create_pedersen_gens_with_extension_degree(1), engine=engine)
    d = _make(bpp, packed, engine, 1, 120, 5151)
    counts, world, first = [50, 17, 53], 3, [0, 50, 67]
    engs = [bpp.Engine(0) for _ in range(world)]
    pars = [params.share(e) for e in engs]
    tr = _ThreadTransport(world)
    forms = {"callbacks": [dmod.ShardComm.from_callbacks(engs[r], r, world, tr.gather(r)) for r in range(world)],
             "local": [dmod.ShardComm(engs[r], r, world, local_group=5150) for r in range(world)]}

    def run(comms, proofs):
        out, weights = [None] * world, [None] * world

        def rank_main(r):
            sl = slice(first[r], first[r] + counts[r])
            rb = packed.ResidentBatch(pars[r], proofs[sl], d["commitments"][sl], d["min_values"][sl], d["min_present"][sl], None, LABEL)
            try:
                comms[r].verify(rb, counts)
                out[r] = "ok"
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
        assert not any(t.is_alive() for t in ths), "a rank is stuck in an exchange"
        assert out[0] == out[1] == out[2], out
        return out[0], weights
    good = d["proofs"]
    bad = good.copy()
    bad[60, 1 + 32 + 5] ^= 1  # rank 1's proof 10: A no longer the prover's -> the sum fails (tier 7), or A does not decode (tier 5)
    for proofs in (good, bad):
        a, wa = run(forms["callbacks"], proofs)
        b, wb = run(forms["local"], proofs)
        assert a == b, (a, b)
        if a == "ok":
            assert wa == wb and all(w is not None for w in wa)
    assert run(forms["callbacks"], good)[0] == "ok"
    assert min(tr.calls) >= 4 and len(set(tr.calls)) == 1  # two exchanges per call, the same number on every rank
    # a transport that fails on one rank: that rank gets BPP_ERR_COMM at once, the handle is dead (the others are left to their
    # own transport's deadline: here the barrier's, which the test does not wait for -- it only runs the failing rank)
    tr.fail_rank = 1
    sl = slice(first[1], first[1] + counts[1])
    rb = packed.ResidentBatch(pars[1], good[sl], d["commitments"][sl], d["min_values"][sl], d["min_present"][sl], None, LABEL)
    for _ in range(2):  # the second call fails without calling the transport again
        with pytest.raises(bpp.EngineError, match=r"\(-4\)|callback|dead|aborted"):
            forms["callbacks"][1].verify(rb, counts)
    rb.verify_only(0)  # the context is fine
    rb.close()
    for f in forms.values():
        for c in f:
            c.close()
    for p in pars:
        p.close()
    for e in engs:
        e.close()
    params.close()


def test_two_ranks_as_processes_on_one_gpu_through_bench():
    """The multi-rank code as PROCESSES (rounds 1-4 only ever ran it as threads of one process, or with one rank): bench.py
    --gpus 2 under torch.distributed.run, both ranks on device 0, the sharded leg's all_gathers through the caller-supplied
    transport over gloo (RCCL refuses two ranks on one device).  The headline AND extra.wide (BASELINE configs[3]: one 4096-proof
    reference batch over the ranks) must complete on both ranks, the line must say what it is, both processes must exit 0."""
    import socket
    env = dict(os.environ, BPP_BENCH_WAVE_BATCHES="8", BPP_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    with socket.socket() as sk:  # a port nobody holds right now
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--batches-per-step", "8",
           "--preheat-ms", "0", "--wide-steps", "4", "--no-cpu-baseline", "--no-traffic"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [json.loads(x) for x in r.stdout.splitlines() if x.startswith("{") and '"metric"' in x]
    assert len(lines) == 1, r.stdout[-1500:]
    out = lines[0]
    # ONE device did the work of both ranks: the line says so at the top level (round 6: it used to read as a 2-GPU weak-scaling point)
    assert out["n_gpus"] == 1 and out["ranks"] == 2 and out["rehearsal"] is True and out["scaling"] is None
    assert out["all_steps_verified"] is True and out["steps_completed"] == 5 and out["value"] > 0
    assert "device 0" in out["config"]["devices"]
    wide = out["extra"]["wide"]
    assert "error" not in wide, wide
    # the line names the transport that really carried the all_gathers and how many ranks spoke RCCL (none: one device)
    assert wide["ranks"] == 2 and wide["rccl_ranks"] == 0 and wide["proofs_per_s"] > 0 and "gloo" in wide["transport"]
    assert wide["transport_requested"] == "gloo" and wide["transport_note"] is None
    ranks = out["per_rank"]
    assert [x["rank"] for x in ranks] == [0, 1] and all(x["host_threads"] >= 1 and x["host_chain_cpu_ms_per_step"] > 0 for x in ranks)
    assert out["host_chain_cpu_ms_per_step"] > 0 and out["host_cores_busy"] > 0
    assert out["host_bound"] in (True, False) and out["host_cores_busy_all_ranks"] > 0
    assert all(x["host_cores_busy"] > 0 and x["weight_chains"] in ("host", "host-wide", "device") for x in ranks)


_ADOPT_CHILD = r'''
import ctypes, importlib, json, os, sys, time
import numpy as np
sys.path.insert(0, {root!r})
import bench
bpp = importlib.import_module("bulletproofs-plus_amd")
packed = importlib.import_module("bulletproofs-plus_amd.packed")
dmod = importlib.import_module("bulletproofs-plus_amd.dist")
stub = ctypes.CDLL({stub!r})
eng = bpp.Engine(0)
params = bpp.RangeParameters.init(64, 1, bpp.create_pedersen_gens_with_extension_degree(1), engine=eng)
d = bench.make_inputs(np, packed, params, 16, seed=77)
rb = packed.ResidentBatch(params, d["proofs"], d["commitments"], d["min_values"], d["min_present"], None, bench.LABEL)

def counts():
    a, b, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    stub.stub_counts(ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
    return a.value, b.value, c.value

def shard(handle):
    c = dmod.ShardComm.__new__(dmod.ShardComm)
    c.engine, c.rank, c.world, c.lib, c.handle = eng, 0, 2, eng.lib, handle
    return c

def attempt(c):
    t0 = time.perf_counter()
    try:
        c.verify(rb, [16, 16])
        return "ok", time.perf_counter() - t0
    except bpp.EngineError as e:
        return str(e), time.perf_counter() - t0

res = {{}}
# 1. an ADOPTED communicator (the caller's ncclComm_t: here a made-up handle the stub never dereferences)
h = ctypes.c_void_p()
assert eng.lib.bpp_comm_adopt(eng.ctx, ctypes.c_void_p(0x1000), 0, 2, ctypes.byref(h)) == 0
c = shard(h)
c.set_timeout(300)
res["adopted_first"] = attempt(c)
res["adopted_counts_after_timeout"] = counts()
res["adopted_second"] = attempt(c)
stub.stub_release(1)  # the owner aborts his communicator: the spinning collective exits
t0 = time.perf_counter()
c.close()
res["adopted_destroy_s"] = time.perf_counter() - t0
res["adopted_counts_after_destroy"] = counts()
stub.stub_release(0)
# 2. a communicator the library CREATED: aborted by the library when the deadline passes
h2 = ctypes.c_void_p()
assert eng.lib.bpp_comm_create(eng.ctx, (ctypes.c_uint8 * 128)(), 0, 2, ctypes.byref(h2)) == 0
c2 = shard(h2)
c2.set_timeout(300)
res["own_first"] = attempt(c2)
res["own_counts_after_timeout"] = counts()
c2.close()
res["own_counts_after_destroy"] = counts()
rb.verify_only(0)  # the context and the batch are fine afterwards
res["context_ok"] = True
print("RESULT " + json.dumps(res))
'''


def test_adopted_communicator_is_never_aborted_or_destroyed(tmp_path):
    """a collective on an ADOPTED ncclComm_t (bpp_comm_adopt: the caller's, say a framework's process-group communicator)
    that misses its deadline must leave that communicator alone -- no ncclCommAbort, no ncclCommDestroy: the owner would later
    use or free a handle that is gone.  A stand-in librccl (tests/cpp/rccl_stub.hip: an all_gather that spins until aborted,
    and counts what it is asked) is loaded through BPP_RCCL_LIB in a child process.  The handle is dead afterwards, later calls
    fail at once, destroying it returns promptly once the owner has aborted; a communicator the library CREATED is aborted by
    the library, as before."""
    stub = str(tmp_path / "librccl_stub.so")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O1", "-shared", "-fPIC", "-o", stub,
                    os.path.join(ROOT, "tests", "cpp", "rccl_stub.hip")], check=True, timeout=600)
    script = tmp_path / "adopt_child.py"
    script.write_text(_ADOPT_CHILD.format(root=ROOT, stub=stub))
    r = subprocess.run([sys.executable, str(script)], env=dict(os.environ, BPP_RCCL_LIB=stub), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    res = json.loads([x for x in r.stdout.splitlines() if x.startswith("RESULT ")][0][7:])
    msg, took = res["adopted_first"]
    assert "(-4)" in msg and "did not complete within 300 ms" in msg and "left alone" in msg, msg
    assert 0.25 < took < 5.0
    aborts, destroys, gathers = res["adopted_counts_after_timeout"]
    assert (aborts, destroys) == (0, 0) and gathers == 1, res
    msg2, took2 = res["adopted_second"]
    assert "(-4)" in msg2 and took2 < 0.2, res  # dead: no second wait, no second collective
    assert res["adopted_counts_after_destroy"][:2] == [0, 0] and res["adopted_destroy_s"] < 3.0, res
    msg3, _ = res["own_first"]
    assert "(-4)" in msg3 and "was aborted" in msg3, msg3
    assert res["own_counts_after_timeout"][0] == 1, res  # the library's own communicator: aborted by the library
    assert res["context_ok"] is True


def test_prover_secrets_are_wiped(bpp, packed):
    """SURVEY 5 secret hygiene, prover side (the reference keeps witness bits, nonces and blinding accumulators in Zeroizing<>:
    src/range_proof.rs:300-301,325,438-464,542-571): after bpp_prove_batch returns -- with proofs, with "Witness opening is
    invalid!" found on the DEVICE, with an argument error found on the host -- the context's prover arena on the device and its
    page-locked staging in both directions hold no non-zero byte (bpp_prove_secret_bytes reads them back)."""
    import bench
    eng = bpp.Engine(0)

    def secret():
        examined, nonzero = ctypes.c_uint64(), ctypes.c_uint64()
        assert eng.lib.bpp_prove_secret_bytes(eng.ctx, ctypes.byref(examined), ctypes.byref(nonzero)) == 0
        return examined.value, nonzero.value
    assert secret() == (0, 0)  # nothing allocated before the first call
    params = bpp.RangeParameters.init(64, 2, bpp.create_pedersen_gens_with_extension_degree(2), engine=eng)
    d = bench.make_inputs(np, packed, params, 48, seed=5252)  # (proves on this context)
    args = lambda **kw: [params, kw.get("values", d["values"]), d["blindings"], kw.get("commitments", d["commitments"]),
                         kw.get("min_values", d["min_values"]), d["min_present"], None, LABEL, d["ext"]]
    proofs = packed.prove(*args())
    assert (proofs == d["proofs"]).all()
    examined, nonzero = secret()
    # the arena alone holds five scalar vectors of mn + a term list of 2 (2 mn + t + 1) scalars per proof
    assert examined > 48 * (5 * 128 + 2 * 259) * 32 and nonzero == 0, (examined, nonzero)
    bad = d["commitments"].copy()
    bad[17, 1] = d["commitments"][18, 1]  # a valid point that is not commit(v, r): found by the device-side witness check
    with pytest.raises(bpp.ProofError, match="Witness opening is invalid"):
        packed.prove(*args(commitments=bad))
    assert secret()[1] == 0
    mv = d["min_values"].copy()
    mv[5, 0] = d["values"][5, 0] + np.uint64(1)  # found on the host, after the witness bytes have been staged
    with pytest.raises(bpp.ProofError, match="Minimum value"):
        packed.prove(*args(min_values=mv))
    assert secret()[1] == 0
    assert (packed.prove(*args()) == d["proofs"]).all()  # and the context proves on
    assert secret()[1] == 0
    params.close()
    eng.close()


@pytest.mark.parametrize("m,t,seeded", [(1, 1, True), (2, 3, False), (4, 6, False), (1, 2, False)])
def test_uniform_access_path_gives_the_oracles_bytes(bpp, packed, m, t, seeded):
    """The secret-only terms through ct.h's uniform-access forms (option "ct": 1 = default, 2 = A1 / B too) -- bpp_pedersen_commit
    (src/generators/pedersen_gens.rs:112-122), the prover's witness check (src/range_proof.rs:275-284) and A1 / B (:572-584):
    commitments and whole proofs are the ORACLE's bytes, and the same bytes as with the fixed-base tables ("ct" = 0); edge
    scalars (0, 1, l - 1, 2^64 - 1) through the commit; a wrong opening is still refused; one launch per round and three."""
    import bench
    from oracle import cport
    from oracle.pyref import curve as C
    eng = bpp.Engine(0)
    params = bpp.RangeParameters.init(64, m, bpp.create_pedersen_gens_with_extension_degree(t), engine=eng)
    cp = cport.Params(64, m, t)
    # commit: edge scalars and random ones, 1..t blinding factors
    rng = np.random.default_rng(900 + 10 * m + t)
    edge = [0, 1, C.L - 1, C.L - 2, 2**252, 8, 7, int("8" * 63, 16) % C.L]
    for nb in range(1, t + 1):
        k = 24
        values = np.array([0, 1, 2**64 - 1, 2**63] + [int(x) for x in rng.integers(0, 2**63, size=k - 4)], dtype=np.uint64)
        bl = rng.integers(0, 256, size=(k, nb, 32), dtype=np.uint8)
        bl[..., 31] &= 0x0f
        for i, e in enumerate(edge):
            bl[i, i % nb] = np.frombuffer(int(e).to_bytes(32, "little"), dtype=np.uint8)
        got = {}
        for ct in (1, 0):
            eng.set_option("ct", ct)
            got[ct] = packed.commit(params, values, bl)
        assert (got[1] == got[0]).all()
        for i in range(k):
            assert bytes(got[1][i]) == cp.commit(int(values[i]), [bytes(bl[i, j]) for j in range(nb)]), (nb, i)
    # whole proofs
    eng.set_option("ct", 0)
    d = bench.make_inputs(np, packed, params, 20, seed=7000 + 10 * m + t)
    if not seeded:
        d["seeds"] = None
    args = [params, d["values"], d["blindings"], d["commitments"], d["min_values"], d["min_present"], d["seeds"], LABEL, d["ext"]]
    out = {}
    for ct in (0, 1, 2):  # 1 (default): the witness check uniform; 2: A1 and B as well (the ladder on the folded generators)
        for fused in (1, 0):
            eng.set_option("ct", ct)
            eng.set_option("prove_fused", fused)
            out[(ct, fused)] = packed.prove(*args)
    ref = out[(0, 1)]
    assert all((v == ref).all() for v in out.values())
    for i in (0, 7, 19):
        want, _ = cp.prove(LABEL, [int(x) for x in d["values"][i]], [[bytes(d["blindings"][i, j, k]) for k in range(t)] for j in range(m)],
                           [int(x) for x in d["min_values"][i]], bytes(d["seeds"][i]) if d["seeds"] is not None else None, bytes(d["ext"][i]))
        assert bytes(ref[i]) == want
    eng.set_option("ct", 1)
    eng.set_option("prove_fused", -1)
    bad = d["commitments"].copy()
    bad[3, 0] = d["commitments"][4, 0]
    with pytest.raises(bpp.ProofError, match="Witness opening is invalid"):
        packed.prove(params, d["values"], d["blindings"], bad, d["min_values"], d["min_present"], d["seeds"], LABEL, d["ext"])
    cp.close()
    params.close()
    eng.close()


@pytest.mark.parametrize("m,t,count", [(1, 1, 70), (4, 3, 40), (8, 1, 24), (2, 6, 30)])
def test_prover_slice_form_equals_workgroup_form(bpp, packed, m, t, count):
    """The rounds' fixed-base MSMs as independent one-wavefront slices whose partial sums the next round kernel adds up
    (k_fb_part + fb_reduce_pair; option "prove_parts", default) against one workgroup per output with its own reduction tree
    (k_fb_msm, "prove_parts" = 0): the same proof BYTES for every number of slices, in the fused and the unfused round form and
    with "ct" = 2 (whose ladder reads the summed points) -- and those bytes are the oracle's (src/range_proof.rs:401-607)."""
    import bench
    from oracle import cport
    eng = bpp.Engine(0)
    params = bpp.RangeParameters.init(64, m, bpp.create_pedersen_gens_with_extension_degree(t), engine=eng)
    eng.set_option("prove_parts", 0)
    d = bench.make_inputs(np, packed, params, count, seed=8100 + 10 * m + t)
    args = [params, d["values"], d["blindings"], d["commitments"], d["min_values"], d["min_present"], d["seeds"], LABEL, d["ext"]]
    ref = d["proofs"]
    cp = cport.Params(64, m, t)
    for i in (0, count - 1):
        want, _ = cp.prove(LABEL, [int(x) for x in d["values"][i]], [[bytes(d["blindings"][i, j, k]) for k in range(t)] for j in range(m)],
                           [int(x) for x in d["min_values"][i]], bytes(d["seeds"][i]) if d["seeds"] is not None else None, bytes(d["ext"][i]))
        assert bytes(ref[i]) == want
    cp.close()
    for parts in (-1, 1, 2, 5, 8):
        for fused, ct in ((1, 1), (0, 1), (1, 2)):
            eng.set_option("prove_parts", parts)
            eng.set_option("prove_fused", fused)
            eng.set_option("ct", ct)
            assert (packed.prove(*args) == ref).all(), (parts, fused, ct)
    # the fused round kernel on 1, 2 and 4 wavefronts per proof (two and more: L / R on a wavefront each, the Fiat-Shamir step's
    # transcript half and generator half side by side, the vector step over the whole workgroup), with and without slices
    for waves in (1, 2, 4):
        for parts, ct in ((-1, 1), (0, 1), (3, 2)):
            eng.set_option("prove_waves", waves)
            eng.set_option("prove_parts", parts)
            eng.set_option("prove_fused", 1)
            eng.set_option("ct", ct)
            assert (packed.prove(*args) == ref).all(), (waves, parts, ct)
    params.close()
    eng.close()


def test_status_words_of_an_earlier_batch_never_leak(bpp, packed):
    """k_results_out writes a block's status words only when the block holds a finding, and the host clears the blocks it skipped
    (settle_status); the page-locked buffer, and what is known about it, is recycled from batch to batch.  A finding at position 40
    of a 64-proof call must not resurface in later, clean calls of 2, 16, 40 and 200 proofs on the same context (it did, for one
    commit of round 5: the block was cleared up to the CURRENT batch's size only), nor after a call that fails before any kernel
    runs; the finding itself is reported every time it is really there."""
    import bench
    eng = bpp.Engine(0)
    params = bpp.RangeParameters.init(64, 1, bpp.create_pedersen_gens_with_extension_degree(1), engine=eng)
    d = bench.make_inputs(np, packed, params, 300, seed=5353)

    def call(n, tamper=None, lo=0):
        pr = d["proofs"][lo:lo + n].copy()
        if tamper == "identity":
            pr[min(40, n - 1), 1 + 32:1 + 64] = 0           # A = the identity's encoding: PASS 1 refuses (a status bit)
        elif tamper == "scalar":
            pr[0, 1 + 32 + 96:1 + 32 + 128] = 0xff           # r1 >= l: refused while parsing, no kernel runs
        sl = slice(lo, lo + n)
        inp = packed.PackedInput(pr, d["commitments"][sl], d["min_values"][sl], d["min_present"][sl], None, LABEL)
        try:
            packed.verify_batch(params, inp, bpp.VerifyAction.VerifyOnly, 0)
            return 0
        except bpp.ProofError as e:
            return int(e.kind)
    K = bpp.ProofErrorKind
    for rep in range(2):
        assert call(64, "identity") == int(K.VerificationFailed)
        assert [call(n, None, lo) for n, lo in ((2, 7), (16, 20), (40, 50), (200, 90), (1, 5), (64, 3))] == [0] * 6
        assert call(100, "scalar") == int(K.InvalidArgument)
        assert call(48, None, 100) == 0
        assert call(300, "identity") == int(K.VerificationFailed)  # block 0 of two
        assert call(300) == 0 and call(5) == 0
    params.close()
    eng.close()


_ENV_CHILD = r'''
import ctypes, importlib, json, sys
sys.path.insert(0, {root!r})
bpp = importlib.import_module("bulletproofs-plus_amd")
packed = importlib.import_module("bulletproofs-plus_amd.packed")
dmod = importlib.import_module("bulletproofs-plus_amd.dist")
eng = bpp.Engine(0)
note1 = eng.lib.bpp_ctx_last_error(eng.ctx).decode()
info = packed.runtime_info(eng)
c = dmod.ShardComm(eng, 0, 1, local_group=77)
note2 = eng.lib.bpp_ctx_last_error(eng.ctx).decode()
c.close()
print("RESULT " + json.dumps({{"note1": note1, "limit": info["small_call_limit"], "note2": note2}}))
'''


def test_malformed_environment_values_keep_the_defaults_and_say_so(tmp_path):
    """BPP_SMALL_CALLS_IN_FLIGHT and BPP_COMM_TIMEOUT_MS were read with atoi -- a typo became 0, i.e. "no gate" and "no
    deadline".  Now a value that is not a whole number keeps the default (12 calls; 60 000 ms) and leaves a note where
    bpp_ctx_last_error finds it; well-formed values still apply (a second child)."""
    script = tmp_path / "env_child.py"
    script.write_text(_ENV_CHILD.format(root=ROOT))

    def run(**env):
        r = subprocess.run([sys.executable, str(script)], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        return json.loads([x for x in r.stdout.splitlines() if x.startswith("RESULT ")][0][7:])
    bad = run(BPP_SMALL_CALLS_IN_FLIGHT="1x", BPP_COMM_TIMEOUT_MS="soon")
    assert bad["limit"] == 12 and "BPP_SMALL_CALLS_IN_FLIGHT" in bad["note1"] and "default" in bad["note1"], bad
    assert "BPP_COMM_TIMEOUT_MS" in bad["note2"] and "60000" in bad["note2"], bad
    good = run(BPP_SMALL_CALLS_IN_FLIGHT="5", BPP_COMM_TIMEOUT_MS="2500")
    assert good["limit"] == 5 and "BPP_" not in good["note1"] and "BPP_" not in good["note2"], good
