"""bench.py's own failure paths: a step that raises must never turn into a reported number."""
import json
import os
import subprocess
import sys
import threading

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _leg(bench, slots, one_step):
    leg = bench.Leg.__new__(bench.Leg)
    leg.slots, leg.chunk, leg.calls, leg.ok_steps = [None] * slots, 0, 0, 0
    leg.one_step = one_step
    return leg


def test_timed_region_is_exact_whatever_the_preheat():
    """CPU: timed() runs the warm-up, stretches it to `preheat_ms` at the warm-up's own rate, then times EXACTLY `steps` steps
    between two synchronisations; the pre-heat never leaks into the count or the latencies of the region"""
    sys.path.insert(0, ROOT)
    import time
    import bench
    lock, stamps, syncs = threading.Lock(), [], []

    def step(slot):
        time.sleep(0.002)
        with lock:
            stamps.append(time.perf_counter())
        return 0.002, {}
    for slots, preheat_ms in ((1, 0.0), (4, 0.0), (1, 60.0), (4, 60.0)):
        del stamps[:], syncs[:]
        leg = _leg(bench, slots, step)
        t0 = time.perf_counter()
        el, lat, profs = bench.timed(leg, 12, 3, lambda: syncs.append(time.perf_counter()), preheat_ms=preheat_ms)
        assert len(lat) == len(profs) == 12 and len(syncs) == 2
        inside = [t for t in stamps if syncs[0] <= t <= syncs[1]]
        assert len(inside) == 12 and len(stamps) >= 15
        assert syncs[0] - t0 >= preheat_ms * 1e-3 * 0.9  # the region starts behind the pre-heat
        assert 0 < el <= syncs[1] - syncs[0] + 1e-3
        if preheat_ms:
            assert len(stamps) > 15  # more untimed steps than the warm-up alone


def test_run_steps_propagates_worker_exceptions_and_counts():
    """CPU: Leg.run_steps with stand-in steps -- every requested step is accounted for, the first exception of any worker
    thread is re-raised on the caller (a thread that dies silently used to leave the step counter handing out its steps)"""
    sys.path.insert(0, ROOT)
    import bench
    lock, seen = threading.Lock(), []

    def ok(slot):
        with lock:
            seen.append(slot)
        return 0.001, {"msm_final_ms": 1.0}
    for slots in (1, 4):
        leg = _leg(bench, slots, ok)
        lat, profs = leg.run_steps(37)
        assert len(lat) == len(profs) == 37 and leg.ok_steps == 37
    n = [0]

    def failing(slot):
        with lock:
            n[0] += 1
            k = n[0]
        if k == 9:
            raise RuntimeError("step 9 failed")
        return 0.001, {}
    for slots in (1, 4):
        n[0] = 0
        leg = _leg(bench, slots, failing)
        with pytest.raises(RuntimeError, match="step 9 failed"):
            leg.run_steps(30)
        assert leg.ok_steps == 0  # nothing is credited for a region that did not complete


@pytest.mark.gpu
def test_forced_failing_step_makes_bench_exit_nonzero():
    """GPU: BPP_BENCH_FAIL_STEP makes the n-th engine call of a leg raise; bench.py must exit non-zero without a JSON line"""
    env = dict(os.environ, BPP_BENCH_FAIL_STEP="10")  # engine call 10 of the leg: inside the timed region (2 warm-up steps come first)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-extra", "--no-cpu-baseline", "--no-traffic", "--steps", "16",
           "--warmup", "2", "--batches-per-step", "4", "--preheat-ms", "0"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0, r.stdout[-500:]
    assert not any(line.startswith("{") and "value" in line for line in r.stdout.splitlines())
    assert "forced failure" in r.stderr or "did not complete" in r.stderr
    ok = subprocess.run(cmd, env=dict(os.environ), capture_output=True, text=True, timeout=600)
    assert ok.returncode == 0, ok.stderr[-800:]
    line = json.loads(ok.stdout.strip().splitlines()[-1])
    assert line["steps_completed"] == 16 and line["all_steps_verified"] is True and line["value"] > 0


def _oracle_inputs(np, n_bits, m, t, count, seed_nonces):
    """a make_inputs()-shaped data set made by the oracle's prover (CPU): what the engine's prover hands the legs on the box"""
    from oracle import cport
    rng = np.random.default_rng(11)
    rounds = (n_bits * m).bit_length() - 1
    plen = 1 + 32 * (t + 5 + 2 * rounds)
    values = rng.integers(0, 1 << 63, size=(count, m), dtype=np.uint64)
    one = rng.integers(0, 256, size=(count, m, 32), dtype=np.uint8)
    one[..., 31] &= 0x0f
    one[..., 0] |= 1
    blindings = np.ascontiguousarray(np.repeat(one[:, :, None, :], t, axis=2))
    seeds = None
    if seed_nonces:
        seeds = rng.integers(0, 256, size=(count, 32), dtype=np.uint8)
        seeds[:, 31] &= 0x0f
    ext = rng.integers(0, 256, size=(count, 32 * (rounds + 3)), dtype=np.uint8)
    min_values, min_present = values // np.uint64(3), np.ones((count, m), dtype=np.uint8)
    cp = cport.Params(n_bits, m, t)
    rc, _, proofs = cp.prove_timed_mt(b"BatchedRangeProofTest", values, blindings, min_values, min_present, seeds, ext, 1, 2, proof_len=plen)
    assert rc == 0
    commitments = np.zeros((count, m, 32), dtype=np.uint8)
    for i in range(count):
        for j in range(m):
            commitments[i, j] = np.frombuffer(cp.commit(int(values[i, j]), [bytes(blindings[i, j, k]) for k in range(t)]), dtype=np.uint8)
    cp.close()
    return {"proofs": proofs, "commitments": commitments, "min_values": min_values, "min_present": min_present, "values": values,
            "blindings": blindings, "seeds": seeds, "ext": ext}


def test_cpu_baseline_objects_of_every_leg():
    """CPU: the cpu_baseline objects bench.py attaches to every leg (configs[0], [2], [4], the wide batch, RecoverOnly) -- shape of
    the object (value, unit, cores, kind "port", sample), the all-cores part, and the prover part's byte comparison, on inputs made
    by the oracle in make_inputs()' layout.  A tampered proof must fail the leg's object loudly, not produce a number."""
    import numpy as np
    sys.path.insert(0, ROOT)
    import bench
    d = _oracle_inputs(np, 64, 1, 1, 8, seed_nonces=True)
    for action in (0, 2):
        o = bench.cpu_verify_baseline((64, 1, 1), d, 4, action=action, seconds=0.05)
        assert o["kind"] == "port" and o["cores"] == 1 and o["unit"] == "proofs/s" and o["value"] > 0 and "sample" in o
        assert o["all_cores"]["cores"] == bench.usable_cpus() and o["all_cores"]["value"] > 0
    one = bench.cpu_verify_baseline((64, 1, 1), d, 1, seconds=0.02, all_cores=False)
    assert "all_cores" not in one and one["ms_per_batch"] > 0
    bad = dict(d, proofs=d["proofs"].copy())
    bad["proofs"][0, 40] ^= 1
    with pytest.raises(RuntimeError, match="rejected"):
        bench.cpu_verify_baseline((64, 1, 1), bad, 4, seconds=0.02)
    d4 = _oracle_inputs(np, 64, 2, 2, 4, seed_nonces=False)
    o = bench.cpu_prove_baseline((64, 2, 2), d4, seconds=0.05)
    assert o["kind"] == "port" and o["cores"] == 1 and o["value"] > 0 and o["bytes_equal_engine"] is True
    assert o["all_cores"]["bytes_equal_engine"] is True
    d4["proofs"][1, 100] ^= 1  # "the engine's" bytes differ: reported, not hidden
    assert bench.cpu_prove_baseline((64, 2, 2), d4, seconds=0.05)["bytes_equal_engine"] is False


def _transport_worker(rank, world, port, stub, out_dir):
    """one rank of test_wide_transport_falls_back_to_gloo (a process of its own: BPP_RCCL_LIB is read once per process)"""
    import importlib
    import json
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.environ["BPP_RCCL_LIB"] = stub if rank == 0 else os.path.join(out_dir, "no-such-librccl.so")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    dmod = importlib.import_module("bulletproofs-plus_amd.dist")
    res = {"picked": bench.pick_wide_transport("rccl", dist, dmod, world), "asked_gloo": bench.pick_wide_transport("gloo", dist, dmod, world)}
    with open(os.path.join(out_dir, "rank%d.json" % rank), "w") as f:
        json.dump(res, f)
    dist.destroy_process_group()


def test_wide_transport_falls_back_to_gloo(tmp_path):
    """bench.py --gpus N, the sharded leg: when RCCL cannot be used on SOME rank (here: rank 1's library does not exist, rank 0 has a
    stand-in librccl, tests/cpp/rccl_stub.hip) every rank must come to the same answer BEFORE anyone enters ncclCommInitRank -- the
    caller-supplied transport over gloo -- and the note must name the failing rank.  World size 2 over gloo, no GPU."""
    import json
    import shutil
    import socket
    import torch.multiprocessing as mp
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    stub = str(tmp_path / "librccl_stub.so")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O1", "-shared", "-fPIC", "-o", stub, os.path.join(ROOT, "tests", "cpp", "rccl_stub.hip")],
                   check=True, timeout=600)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_transport_worker, args=(2, port, stub, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (json.load(open(tmp_path / ("rank%d.json" % r))) for r in (0, 1))
    assert r0 == r1  # the same decision on every rank
    transport, note = r0["picked"]
    assert transport == "gloo" and "rank 1" in note and "rank 0" not in note and "RCCL" in note
    assert r0["asked_gloo"] == ["gloo", None]
