#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric on MI355X: 64-bit range proofs verified / second (batch).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--no-extra] [--no-cpu-baseline]

Headline (value): BASELINE.json configs[1].  A step = one engine call = one pass of the hot path (RangeProof::verify:
transcript replay, decompression, scalar block, weight chain, final MSM) over one resident input of --batches-per-step
(default 64) INDEPENDENT reference batches of 1024 non-aggregated 64-bit proofs each, extension degree 1, resident in HBM.
Every 1024-proof batch keeps its own weight transcript, final MSM and identity test, exactly as if it had been passed
to the reference's verify() alone.  --concurrency steps are in flight per GPU and EXACTLY --steps are timed;
value = steps x batches-per-step x 1024 x N / elapsed.  All 64 batches of a step are DISTINCT proofs, made at start-up
on the box by the engine's own batch prover with the recipe of benches/range_proof.rs:206-262 (the prover's bytes are
pinned against the oracle in tests/test_gpu_prove.py); every slot holds the same 64 batches in another order.

The same JSON line carries, under "extra", the other BASELINE configs as their own timed legs, each with the roofline of
its dominant kernel: configs[2] (256 x aggregation-8), one 4096-proof reference batch on one GPU (north_star's target
sentence), single-call latency at 1 and 256 proofs (configs[0]'s shape, benches/range_proof.rs:115-119), the batch
prover on configs[4]; "cpu_baseline" times the oracle/c port on one host core AND on all host cores.

N > 1: `python bench.py --gpus N` starts `python -m torch.distributed.run` as a CHILD process (before anything touches
the GPU) and relays its JSON line; under torchrun it runs as one rank per GPU.  value = weak scaling, every rank runs
the N=1 step on its own (differently seeded) batches, verdicts combined by one all_reduce; "extra.wide" = BASELINE
configs[3]: 4096 proofs as ONE reference batch sharded over the N ranks (all_gather of the transcript-RNG bytes,
replayed weight chain, all_gather of the 128-byte accumulators over RCCL).
"""
import os

# More than four calls in flight only overlap if the HIP runtime may use more than its default four hardware queues (read
# once, when the runtime starts: before torch is imported).  The headline (three steps in flight) does not depend on it,
# the 4096-proof leg does: 5.4 -> 8.5 / 10.8 M proofs/s with eight / twelve calls in flight (sixteen need 24 queues: 11.9 M).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
# v_mad_u64_u32 issue peak: half rate = 64 lanes / clock / CU x 256 CUs x 2.4 GHz.  Measured with an 8 ms kernel at 8 waves per
# SIMD (tools/microbench/valu_rates.hip): 57.3 lanes / clock / CU at the 2.37 GHz the chip holds = 34.7 T/s
VALU_PEAK_TMAD = 39.3
LABEL = b"BatchedRangeProofTest"  # benches/range_proof.rs:49
FAIL_STEP = int(os.environ.get("BPP_BENCH_FAIL_STEP", "0"))  # harness self-test: the n-th step of a leg raises
HOST_CORES_PER_RANK = 4.0  # what a rank with host-side weight chains keeps busy at the headline rate, rounded up (RESULTS.md)
WATCHDOG_EXIT = 4  # exit code of every rank when the sharded leg's watchdog fires (the line is printed first)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=32)
    # steps in flight per GPU: 0 = the harness' rule (four; six with the weight chains on the device), see main()
    ap.add_argument("--concurrency", type=int, default=int(os.environ.get("BPP_BENCH_CONCURRENCY", "0")),
                    help="steps in flight per GPU (one engine/stream + one host thread each)")
    ap.add_argument("--batches-per-step", "--batches-per-launch", dest="batches_per_step", type=int,
                    default=int(os.environ.get("BPP_BENCH_BATCHES_PER_LAUNCH", "64")),
                    help="independent 1024-proof reference batches verified by one step (one engine call)")
    ap.add_argument("--preheat-ms", type=float, default=float(os.environ.get("BPP_BENCH_PREHEAT_MS", "1000")),
                    help="untimed run of the headline leg before the --warmup steps: the clock governor needs ~50 ms of full load to "
                         "reach the clock it then sustains (tools/clock_ramp.py), and 0.3-0.4 s after a load starts from idle the chip "
                         "holds back for ~40 ms (tools/rate_ramp.py: one 100 ms window at 17 M proofs/s and 2.15 GHz among windows at "
                         "27.5 M and 2.29 GHz) -- the timed region starts behind both; a service under load sees neither")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra legs (cfg3, 4096-wide, latency, prover)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=5.0, help="CPU work per part (one thread / all cores) of the headline's cpu_baseline")
    ap.add_argument("--cpu-leg-seconds", type=float, default=1.5,
                    help="CPU work per part of every other leg's cpu_baseline (configs[0], [2], [4], the 4096-proof batch, RecoverOnly)")
    ap.add_argument("--wide-steps", type=int, default=40)
    ap.add_argument("--wide-timeout", type=int, default=150, help="seconds after which the sharded leg (N > 1 / torchrun) is abandoned")
    ap.add_argument("--no-traffic", action="store_true",
                    help="skip the rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE, SQ_*) that fill roofline.traffic")
    ap.add_argument("--only", choices=("headline", "cfg3", "prover"), default=None,
                    help="run just this leg, no roofline post-processing: what the rocprofv3 child passes of measure_traffic() run")
    ap.add_argument("--host-in-calls", type=int, default=48, help="calls per timed region of the host-buffers-in leg")
    ap.add_argument("--transport", choices=("rccl", "gloo"), default=os.environ.get("BPP_BENCH_TRANSPORT", "rccl"),
                    help="N > 1, the sharded leg's all_gathers: RCCL on device buffers, or the caller-supplied transport of the C ABI "
                         "(bpp_comm_create_callbacks) over torch.distributed's gloo")
    ap.add_argument("--chain", choices=("host", "host-wide", "device", "auto"), default=os.environ.get("BPP_BENCH_CHAIN", "auto"),
                    help="where the batch-weight chains run (engine option \"chain\"): host cores, sponge and reduction mod l "
                         "(csrc/chain_host.h); host-wide: the sponges on host cores, the reduction on the device (the engine's own rule for "
                         "calls of 4096 proofs and more); device: one wavefront per reference batch (csrc/chain_dev.h); auto: the harness' rule")
    ap.add_argument("--one-device", action="store_true", default=os.environ.get("BPP_BENCH_ONE_DEVICE", "0") == "1",
                    help="every local rank on device 0 (a rehearsal of the multi-rank code as PROCESSES on a one-GPU box; RCCL refuses "
                         "two ranks on one device, so this implies --transport gloo); the line says so")
    return ap.parse_args()


def usable_cpus():
    """host threads this process may really run at once: affinity mask, capped by a cgroup CPU quota when there is one
    (a 1-GPU box of the pool reports 256 logical CPUs but schedules 16)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, (q + p // 2) // p))
        except (OSError, ValueError):
            pass
    return n


def thread_cpu_seconds():
    """{tid: (comm, user + system CPU seconds)} of every thread of this process (/proc/self/task): who is burning host
    cores during a timed region -- weight-chain workers, callers waiting on a stream, the runtime's own threads"""
    out = {}
    try:
        tick = os.sysconf("SC_CLK_TCK")
        for tid in os.listdir("/proc/self/task"):
            try:
                raw = open("/proc/self/task/%s/stat" % tid).read()
            except OSError:
                continue
            comm = raw[raw.index("(") + 1:raw.rindex(")")]
            f = raw[raw.rindex(")") + 2:].split()
            out[int(tid)] = (comm, (int(f[11]) + int(f[12])) / tick)
    except (OSError, ValueError):
        pass
    return out


def thread_cpu_delta(before, after, wall_s):
    """cores kept busy per thread NAME over a region of `wall_s` seconds (threads of one name summed), busiest first"""
    acc = {}
    for tid, (comm, sec) in after.items():
        d = sec - before.get(tid, (comm, 0.0))[1]
        if d > 0:
            e = acc.setdefault(comm, [0, 0.0, 0.0])
            e[0] += 1
            e[1] += d
            e[2] = max(e[2], d)
    rows = sorted(acc.items(), key=lambda kv: -kv[1][1])
    return [{"thread": k, "threads": v[0], "cores_busy": round(v[1] / wall_s, 3), "busiest_one": round(v[2] / wall_s, 3)}
            for k, v in rows if wall_s > 0][:8]


def resolve_chain_mode(args, world):
    """--chain auto: the host chains (the lowest latency per call) unless the ranks of this node would have to share fewer
    than HOST_CORES_PER_RANK schedulable cores each -- then the device chains, which leave the host alone"""
    if args.chain != "auto":
        return args.chain
    return "device" if usable_cpus() / max(1, world) < HOST_CORES_PER_RANK else "host-wide"


def self_launch(args):
    """plain `python bench.py --gpus N`: one rank per GPU under torch.distributed.run, as a child process (an exec from
    a process that has touched the GPU is forbidden on the pool; this one has not even imported torch yet)"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run(cmd, env=env)
    raise SystemExit(r.returncode)


def make_inputs(np, packed, params, count, seed, chunk=8192):
    """`count` (statement, proof) pairs with the recipe of benches/range_proof.rs:206-262: value = next_u64 mod 2^63,
    minimum-value promise value / 3, ONE random blinding repeated t times, seed nonce iff m == 1, label
    "BatchedRangeProofTest".  Proofs come from the engine's batch prover (bpp_prove_batch)."""
    n_bits, m, t = params.bit_length(), params.max_aggregation_factor(), int(params.extension_degree())
    rounds = (n_bits * m).bit_length() - 1
    rng = np.random.default_rng(seed)
    values = rng.integers(0, 1 << min(63, max(n_bits - 1, 1)), size=(count, m), dtype=np.uint64)  # (2^63 for the 64-bit configs)
    one = rng.integers(0, 256, size=(count, m, 32), dtype=np.uint8)
    one[..., 31] &= 0x0f  # < 2^252 < l: canonical (zero has probability 2^-252)
    one[..., 0] |= 1
    blindings = np.ascontiguousarray(np.repeat(one[:, :, None, :], t, axis=2))
    min_values = values // np.uint64(3)
    min_present = np.ones((count, m), dtype=np.uint8)
    seeds = None
    if m == 1:
        seeds = rng.integers(0, 256, size=(count, 32), dtype=np.uint8)
        seeds[:, 31] &= 0x0f
    ext = rng.integers(0, 256, size=(count, 32 * (rounds + 3)), dtype=np.uint8)
    commitments = packed.commit(params, values.reshape(-1), blindings.reshape(count * m, t, 32)).reshape(count, m, 32)
    proofs = []
    for lo in range(0, count, chunk):
        hi = min(count, lo + chunk)
        proofs.append(packed.prove(params, values[lo:hi], blindings[lo:hi], commitments[lo:hi], min_values[lo:hi],
                                   min_present[lo:hi], None if seeds is None else seeds[lo:hi], LABEL, ext[lo:hi]))
    return {"proofs": np.concatenate(proofs), "commitments": commitments, "min_values": min_values, "min_present": min_present,
            "values": values, "blindings": blindings, "seeds": seeds, "ext": ext}


def _profiler_usable():
    import shutil
    if not shutil.which("rocprofv3"):
        return False
    # already running under a profiler (the driver's, or tools/profile_round.sh): no nested profiler runs
    return not (any(k.startswith(("ROCPROF", "ROCP_", "ROCPROFILER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""))


def _counter_pass(only, counters, extra_args=()):
    """one child run of this script's `--only <leg>` under rocprofv3 --kernel-trace --pmc <counters> (kernel trace only, no
    other trace domain; the program directly after `--`; a child, not an exec: this process has initialised the GPU).
    Returns {kernel: {counter: sum, "dispatches": n, "dur_ns": total}} or None."""
    import shutil
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_summary
    d = tempfile.mkdtemp(prefix="bpp_pmc_", dir="/tmp")
    try:
        cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + list(counters) + ["--output-format", "csv", "-d", d, "--", sys.executable,
               os.path.abspath(__file__), "--only", only, "--no-extra", "--no-cpu-baseline", "--no-traffic", "--concurrency", "1",
               "--steps", "4", "--warmup", "1"] + list(extra_args)
        r = subprocess.run(cmd, env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
        if r.returncode != 0:
            return None
        return pmc_summary.counters(d)
    finally:
        shutil.rmtree(d, ignore_errors=True)


def _kernel_row(acc, name):
    """the row of kernel `name` (exact short name; k_decompress must not pick k_decompress_plain), else a substring match"""
    for exact in (True, False):
        for k, e in (acc or {}).items():
            if (k == name if exact else name in k) and e.get("dispatches"):
                return e
    return None


def measure_traffic(kernel="k_msm_accumulate", only="headline"):
    """HBM-side bytes of one launch of `kernel`, measured NOW: two child runs of this script (`--only <leg>`) under rocprofv3
    (`--pmc FETCH_SIZE`, then `--pmc WRITE_SIZE`: the TCC block cannot hold both in one pass), one step in flight so that the
    kernel's dispatches do not share the chip.  Returns None (traffic stays null) if the profiler is missing or a pass fails."""
    try:
        if not _profiler_usable():
            return None
        raw = {}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            e = _kernel_row(_counter_pass(only, [counter]), kernel)
            if not e or not e.get(counter):
                return None
            raw[counter] = (e[counter] * 1024.0 / e["dispatches"], int(e["dispatches"]))
        # calibration of the two counters for this engine's access patterns (tools/microbench/fetch_calib.hip on the same
        # hardware, profiles/r02_v18_fetch_calib.json): FETCH_SIZE tallies a full 128-byte line request at 64 bytes (x0.500 for
        # gathers of aligned 128-byte table entries, as for the documented wide-stream case), WRITE_SIZE reads x1.19 for
        # scattered 160-byte stores
        f, w = raw["FETCH_SIZE"][0], raw["WRITE_SIZE"][0]
        return {"traffic": f / 0.5 + w / 1.19, "traffic_raw_counters": {"FETCH_SIZE_bytes": f, "WRITE_SIZE_bytes": w,
                                                                       "dispatches": [raw["FETCH_SIZE"][1], raw["WRITE_SIZE"][1]]},
                "traffic_method": "two rocprofv3 child passes of `bench.py --only %s` (--pmc FETCH_SIZE / --pmc WRITE_SIZE, kernel trace "
                                  "only, one step in flight), per launch; corrected FETCH / 0.500 + WRITE / 1.19 "
                                  "(profiles/r02_v18_fetch_calib.json); bytes leaving the XCD L2s, served by the 256 MB Infinity Cache" % only}
    except (Exception, SystemExit):  # noqa: BLE001 - the headline number must not depend on the profiler
        return None


def measure_sq(kernels, only="headline"):
    """issue-side counters of `kernels` from one more child pass (SQ_INSTS_VALU, SQ_WAVE_CYCLES, SQ_WAIT_INST_ANY,
    SQ_ACTIVE_INST_VALU), one step in flight: VALU instructions per launch, fraction of wave cycles spent waiting"""
    try:
        if not _profiler_usable():
            return {}
        acc = _counter_pass(only, ["SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU"])
        out = {}
        for k in kernels:
            e = _kernel_row(acc, k)
            if e and e.get("SQ_WAVE_CYCLES"):
                n = e["dispatches"]
                out[k] = {"valu_instr_per_launch": e.get("SQ_INSTS_VALU", 0.0) / n, "kernel_ms_alone": e["dur_ns"] / n / 1e6,
                          "wait_inst_frac": e.get("SQ_WAIT_INST_ANY", 0.0) / e["SQ_WAVE_CYCLES"],
                          "active_valu_frac": e.get("SQ_ACTIVE_INST_VALU", 0.0) / e["SQ_WAVE_CYCLES"], "dispatches": int(n)}
        return out
    except (Exception, SystemExit):  # noqa: BLE001
        return {}


def _cport():
    """oracle/c, the CPU port of the reference's algorithms: imported by the cpu_baseline objects ONLY -- there it is the thing
    timed (kind "port"), never part of a product path"""
    from oracle import cport
    return cport


def cpu_items(data, idx, with_seeds=False):
    """the oracle's view of proofs `idx` of a make_inputs() set"""
    m = data["commitments"].shape[1]
    return [{"proof": data["proofs"][i].tobytes(), "commitments": [data["commitments"][i, j].tobytes() for j in range(m)],
             "min_values": [int(v) for v in data["min_values"][i]], "label": LABEL,
             "seed_nonce": data["seeds"][i].tobytes() if with_seeds and data.get("seeds") is not None else None} for i in idx]


def cpu_verify_baseline(shape, data, chunk, action=0, seconds=1.5, all_cores=True):
    """cpu_baseline object of one leg: RangeProof::verify over reference batches of `chunk` proofs of THIS leg's inputs through the
    oracle/c port -- one host thread (the reference is single-threaded), then one batch per schedulable core side by side; each
    part is bounded by `seconds` of wall time (at least one pass)."""
    cport = _cport()
    ncpu = usable_cpus()
    slices = max(1, min(ncpu if all_cores else 1, 4, data["proofs"].shape[0] // chunk))
    items = cpu_items(data, range(chunk * slices), with_seeds=action != 0)
    cp = cport.Params(*shape)
    try:
        rc, s1 = cp.verify_action_timed_mt(items[:chunk], chunk, action, 1, 1)
        iters = int(seconds / max(s1, 1e-4))
        if iters > 1:
            rc, s1 = cp.verify_action_timed_mt(items[:chunk], chunk, action, iters, 1)
        iters = max(1, iters)
        if rc != 0:
            raise RuntimeError("the CPU port rejected the leg's batch (rc %d)" % rc)
        verb = ("verify", "recover-and-verify", "recover-only")[action]
        out = {"value": chunk * iters / s1, "unit": "proofs/s", "cores": 1, "kind": "port", "ms_per_batch": 1e3 * s1 / iters,
               "sample": "%d x %s of one %d-proof reference batch of this leg's inputs, one thread, oracle/c (dalek's algorithms)"
                         % (iters, verb, chunk), "nproc": os.cpu_count(), "usable_cpus": ncpu}
        if all_cores and ncpu > 1:
            rc, sm = cp.verify_action_timed_mt(items, chunk, action, 1, ncpu)
            it_mt = int(seconds / max(sm, 1e-4))
            if it_mt > 1:
                rc, sm = cp.verify_action_timed_mt(items, chunk, action, it_mt, ncpu)
            it_mt = max(1, it_mt)
            if rc != 0:
                raise RuntimeError("the CPU port rejected the leg's batch on %d threads (rc %d)" % (ncpu, rc))
            out["all_cores"] = {"value": chunk * it_mt * ncpu / sm, "unit": "proofs/s", "cores": ncpu,
                                "sample": "%d threads, each %d x %s of a %d-proof reference batch" % (ncpu, it_mt, verb, chunk)}
        return out
    finally:
        cp.close()


def cpu_prove_baseline(shape, data, seconds=1.5):
    """cpu_baseline object of the prover leg: RangeProof::prove_with_rng (src/range_proof.rs:232-608, generator folding as the
    reference does it) over this leg's witnesses through the oracle/c port, one thread and one thread per schedulable core; the
    port's proof bytes are held against the engine's on the proofs both made."""
    import numpy as np
    cport = _cport()
    ncpu = usable_cpus()
    plen = data["proofs"].shape[1]
    cp = cport.Params(*shape)
    try:
        def run(k, threads):
            sl = slice(0, k)
            rc, sec, got = cp.prove_timed_mt(LABEL, data["values"][sl], data["blindings"][sl], data["min_values"][sl], data["min_present"][sl],
                                             None if data.get("seeds") is None else data["seeds"][sl], data["ext"][sl], 1, threads, proof_len=plen)
            if rc != 0:
                raise RuntimeError("the CPU port's prover failed (rc %d)" % rc)
            return sec, bool((got == data["proofs"][sl]).all())
        s1, eq1 = run(2, 1)
        k1 = max(2, min(data["proofs"].shape[0], int(seconds / (s1 / 2))))
        if k1 > 2:
            s1, eq1 = run(k1, 1)
        out = {"value": k1 / s1, "unit": "proofs/s", "cores": 1, "kind": "port", "ms_per_proof": 1e3 * s1 / k1,
               "sample": "%d proofs of this leg's witnesses, one thread, oracle/c (prove_with_rng with generator folding, as the reference)" % k1,
               "bytes_equal_engine": eq1, "nproc": os.cpu_count(), "usable_cpus": ncpu}
        if ncpu > 1:
            km = max(ncpu, min(data["proofs"].shape[0], int(seconds / (s1 / k1)) * ncpu))
            sm, eqm = run(km, ncpu)
            out["all_cores"] = {"value": km / sm, "unit": "proofs/s", "cores": ncpu, "bytes_equal_engine": eqm,
                                "sample": "%d proofs shared out over %d threads" % (km, ncpu)}
        return out
    finally:
        cp.close()


class Leg:
    """S slots (engine + stream + resident copy of the input) verifying `chunk`-proof reference batches"""

    def __init__(self, bpp, packed, torch, device, params0, data, batch_proofs, batches, slots, chunk, profile=True, action=0, options=None):
        import numpy as np
        self.bpp, self.chunk, self.slots, self.action = bpp, chunk, [], int(action)
        self.data_idx = []
        self.calls = self.ok_steps = 0
        self._pool = None
        self.proofs_per_step = batch_proofs * batches
        nb = data["proofs"].shape[0] // batch_proofs
        self.upload_s = self.marshal_s = 0.0
        for i in range(slots):
            stream = None  # the engine's own non-blocking stream (its HIP events time the kernels on it)
            eng = bpp.Engine(device.index)
            eng.profile(profile)  # stage events: nothing next to a 2.5 ms step, 10 % of a 0.8 ms call
            for name, value in (options or {}).items():  # per-context knobs of this leg (bpp_ctx_set_option)
                eng.set_option(name, value)
            params = params0.share(eng)  # ONE generator table for every slot (src/traits.rs:42 `Send + Sync`)
            # the same `nb` distinct batches in another order for every slot
            order = [(i * 11 + k) % nb for k in range(batches)]
            idx = np.concatenate([np.arange(b * batch_proofs, (b + 1) * batch_proofs) for b in order])
            seeds = data["seeds"][idx] if self.action and data.get("seeds") is not None else None  # (mask recovery needs the nonces)
            rb = packed.ResidentBatch(params, data["proofs"][idx], data["commitments"][idx], data["min_values"][idx],
                                      data["min_present"][idx], seeds, LABEL)
            self.upload_s, self.marshal_s = rb.upload_seconds, rb.marshal_seconds
            self.data_idx.append(idx)
            rb.prepare(chunk)   # group layout, MSM plan, work buffers: not in any timed call
            self._call(rb)  # every slot has run once (events, lazily built state) before anything is timed
            self.slots.append((stream, eng, params, rb))

    def _call(self, rb):
        if self.action:
            return rb.verify_arrays(self.action, self.chunk)  # masks as arrays: no per-item Python in the timed call
        rb.verify_only(self.chunk)  # raises on an invalid batch
        return None

    def set_profile(self, on):
        for _, eng, _, _ in self.slots:
            eng.profile(on)

    def one_step(self, slot):
        _, eng, _, rb = self.slots[slot]
        self.calls += 1
        if self.calls == FAIL_STEP:  # BPP_BENCH_FAIL_STEP: the harness' own failure path (tests/test_bench_harness.py)
            raise RuntimeError("forced failure of step %d (BPP_BENCH_FAIL_STEP)" % FAIL_STEP)
        t0 = time.perf_counter()
        self._call(rb)
        return time.perf_counter() - t0, eng.last_profile()

    def run_steps(self, count):
        """`count` complete steps, at most len(slots) in flight; returns (latencies, stage profiles).  A step that raises
        (invalid batch, engine fault) stops the leg: the first exception of any worker is re-raised here, and the number of
        completed steps is checked against `count` -- a number is never reported for steps that did not run."""
        S = min(len(self.slots), max(count, 1))
        if S == 1:
            res = [self.one_step(0) for _ in range(count)]
        else:
            nxt, lock, out, errors = [0], threading.Lock(), [[] for _ in range(S)], []

            def worker(slot):
                try:
                    while True:
                        with lock:
                            if nxt[0] >= count or errors:
                                return
                            nxt[0] += 1
                        out[slot].append(self.one_step(slot))
                except BaseException as e:  # noqa: BLE001 - re-raised on the calling thread
                    with lock:
                        errors.append(e)
            # persistent worker threads (created by the warm-up's first use): no thread start-up inside a timed region
            if getattr(self, "_pool", None) is None:
                from concurrent.futures import ThreadPoolExecutor
                self._pool = ThreadPoolExecutor(max_workers=len(self.slots))
            for fut in [self._pool.submit(worker, s) for s in range(S)]:
                fut.result()
            if errors:
                raise errors[0]
            res = [r for part in out for r in part]
        if len(res) != count:
            raise RuntimeError("%d of %d steps completed" % (len(res), count))
        self.ok_steps += len(res)
        return [r[0] for r in res], [r[1] for r in res]

    def close(self):
        if getattr(self, "_pool", None) is not None:
            self._pool.shutdown(wait=True)
            self._pool = None
        for _, eng, params, rb in self.slots:
            rb.close()
            params.close()
            eng.close()


def kernel_roofline(profs, alone_ms=None):
    """roofline object of k_msm_accumulate from per-step stage profiles (hipEvents on the engine's stream)"""
    profs = [p for p in profs if p and p.get("msm_accumulate_ms", 0) > 0]
    if not profs:
        return None, {}
    avg = {k: sum(p[k] for p in profs) / len(profs) for k in profs[0]}
    terms, K, acc_ms = int(avg["msm_terms"]), int(avg["msm_windows"]), avg["msm_accumulate_ms"]
    msm_bytes = 64 * terms  # 32 B scalar + 32 B compressed point per MSM term
    achieved = msm_bytes / (acc_ms * 1e-3) / 1e9
    # v_mad_u64_u32 per launch: a bucket's first term costs one field multiplication (100 mads), every further term a mixed
    # addition of seven (700); buckets = groups x windows x 2^(c-1), all of them occupied at these sizes
    buckets = min(int(avg["msm_groups"]) * K * (1 << (int(avg["msm_window_bits"]) - 1)), terms * K)
    mads = (terms * K - buckets) * 700.0 + buckets * 100.0
    out = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
           "traffic": None,  # filled in by measure_traffic() (two rocprofv3 child passes of this leg's own command) or left null
           "kernel": "k_msm_accumulate (Pippenger bucket accumulation of the final MSM)", "kernel_ms": acc_ms,
           "algorithmic_bytes": msm_bytes, "msm_terms_per_launch": terms, "msm_window_bits": int(avg["msm_window_bits"]),
           "msm_windows": K, "msm_groups": int(avg["msm_groups"]),
           "note": "integer-VALU bound, not HBM bound (SURVEY 8d): see valu",
           "valu": {"achieved_Tmad_per_s": mads / (acc_ms * 1e-3) / 1e12, "peak_Tmad_per_s": VALU_PEAK_TMAD,
                    "frac": mads / (acc_ms * 1e-3) / (VALU_PEAK_TMAD * 1e12)}}
    if alone_ms:  # same kernel, same launch, no other step in flight (3 launches after the timed region)
        out["alone"] = {"kernel_ms": alone_ms, "achieved": msm_bytes / (alone_ms * 1e-3) / 1e9,
                        "frac": msm_bytes / (alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                        "valu_frac": mads / (alone_ms * 1e-3) / (VALU_PEAK_TMAD * 1e12)}
    return out, {n: round(v, 4) for n, v in avg.items() if n.endswith("_ms")}


SIDE_PREHEAT_MS = float(os.environ.get("BPP_BENCH_SIDE_PREHEAT_MS", "600"))  # the extra legs' untimed run (see --preheat-ms)


def timed(leg, steps, warmup, sync, clock=None, preheat_ms=0.0):
    """`clock` (optional): callable that samples the shader clock; it runs on its own thread DURING the timed steps (one
    napping wavefront on its own context: nothing is added to the timed work) and its result is returned as 4th value.
    `preheat_ms`: the untimed run before the region lasts at least this long (more warm-up steps at the rate of the first)."""
    t_w = time.perf_counter()
    leg.run_steps(warmup)
    done = warmup
    while preheat_ms > 0 and done > 0:  # (a loop: the first estimate is off when the warm-up did not fill every slot)
        spent = time.perf_counter() - t_w
        more = int((preheat_ms * 1e-3 - spent) / max(spent / done, 1e-5))
        if more <= 0:
            break
        leg.run_steps(max(more, len(leg.slots)))
        done += max(more, len(leg.slots))
    sync()
    ghz = []
    th = threading.Thread(target=lambda: ghz.append(clock())) if clock else None
    if th:
        th.start()  # (before t0: the sampling wavefront naps through the region either way, its thread's start-up is not timed work)
    t0 = time.perf_counter()
    try:
        lat, profs = leg.run_steps(steps)
        sync()
        el = time.perf_counter() - t0
    finally:
        if th:
            th.join()  # (also when a step raised: the sampler's engine is closed by the caller right after)
    if th:
        return el, lat, profs, (ghz[0] if ghz else None)
    return el, lat, profs


def decompress_roofline(stages, n_points, sq, clock_ghz):
    """roofline object of k_decompress, the verifier's other dominant kernel (ristretto255 decoding of the proofs' points,
    src/range_proof.rs:1067-1109): 32 B in per point; one inverse square root = 254 squarings (55 multiply-adds each) and
    25 multiplications (100) per point"""
    ms = stages.get("decompress_ms")
    if not ms:
        return None
    mads = n_points * (254 * 55 + 25 * 100.0)
    out = {"bound": "hbm", "kernel": "k_decompress (ristretto255 -> affine-Niels, one lane per point)", "kernel_ms": ms,
           "points_per_launch": n_points, "algorithmic_bytes": 32 * n_points, "achieved": 32 * n_points / (ms * 1e-3) / 1e9,
           "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": 32 * n_points / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "traffic": None,
           "note": "integer-VALU bound (254 squarings + 25 multiplications per 32 bytes): see valu; kernel_ms is the stage interval of "
                   "the timed region (several steps share the chip) and includes the 32 B/proof copy of the transcript-RNG bytes",
           "valu": {"achieved_Tmad_per_s": mads / (ms * 1e-3) / 1e12, "peak_Tmad_per_s": VALU_PEAK_TMAD,
                    "frac": mads / (ms * 1e-3) / (VALU_PEAK_TMAD * 1e12)}}
    if clock_ghz:
        out["valu"]["frac_at_clock"] = out["valu"]["achieved_Tmad_per_s"] / (64 * 256 * clock_ghz * 1e9 / 1e12)
    e = (sq or {}).get("k_decompress")
    if e:  # one step in flight, from the SQ counter pass.  The kernel's dispatches there: one per verification (n_points proof
        # points) and ONE at upload over the statements' commitments (n_points / 15): per-launch averages are scaled to a
        # verification launch by points
        nd = e["dispatches"]
        scale = nd * n_points / ((nd - 1) * n_points + n_points / 15.0) if nd > 1 else 1.0
        e = dict(e, kernel_ms_alone=e["kernel_ms_alone"] * scale, valu_instr_per_launch=e["valu_instr_per_launch"] * scale)
        out["alone"] = {"kernel_ms": e["kernel_ms_alone"], "valu_instr_per_point": 64.0 * e["valu_instr_per_launch"] / n_points,
                        "wait_inst_frac": e["wait_inst_frac"], "active_valu_frac": e["active_valu_frac"],
                        "valu_frac": mads / (e["kernel_ms_alone"] * 1e-3) / (VALU_PEAK_TMAD * 1e12),
                        "note": "kernel_ms under the counter pass (slower than a plain launch); instructions counted per wavefront x 64 lanes"}
    return out


def host_in_leg(bpp, packed, np, device_index, params0, data, R, calls, sync):
    """host buffers in -> verdict out, what RangeProof::verify_batch(&[RangeStatement], &[RangeProof]) delivers
    (src/range_proof.rs:712-717): every call hands the engine R x 1024 proofs and statements in pageable host memory
    (bpp_packed_batch), the engine validates, packs, uploads, verifies (chunk = 1024) and releases.  Never `value`."""
    n = 1024 * R
    nb = data["proofs"].shape[0] // 1024
    inputs = []
    for k in range(4):  # four distinct call inputs (the same 64 batches in another order), rotated
        order = [(k * 17 + j) % nb for j in range(R)]
        idx = np.concatenate([np.arange(b * 1024, (b + 1) * 1024) for b in order])
        inputs.append(packed.PackedInput(data["proofs"][idx], data["commitments"][idx], data["min_values"][idx],
                                         data["min_present"][idx], None, LABEL))
    bytes_per_call = inputs[0].proofs.nbytes + inputs[0].commitments.nbytes + inputs[0].min_values.nbytes + inputs[0].min_present.nbytes
    out = {"workload": "BASELINE configs[1] with the inputs in HOST memory on every call: %d x 1024 proofs + statements as one "
                       "bpp_packed_batch (%.1f MB) -> validate, pack, DMA, verify (chunk = 1024), release; C-ABI calls, "
                       "ctypes marshalling of ONE struct per call" % (R, bytes_per_call / 1e6),
           "proofs_per_call": n, "host_bytes_per_call": bytes_per_call}

    def run(contexts, depth, calls):
        engs = [bpp.Engine(device_index) for _ in range(contexts)]
        pars = [params0.share(e) for e in engs]
        pipes = [packed.Pipeline(p, depth=depth) for p in pars]
        errors = []

        def worker(c, count):
            try:
                q = []
                for i in range(count):
                    q.append(pipes[c].submit(inputs[(i + c) % len(inputs)], bpp.VerifyAction.VerifyOnly, 1024))
                    if len(q) >= depth:
                        pipes[c].collect(q.pop(0))
                while q:
                    pipes[c].collect(q.pop(0))
            except BaseException as e:  # noqa: BLE001
                errors.append(e)

        def region(count):
            ths = [threading.Thread(target=worker, args=(c, count)) for c in range(contexts)]
            t0 = time.perf_counter()
            for th in ths:
                th.start()
            for th in ths:
                th.join()
            if errors:
                raise errors[0]
            return time.perf_counter() - t0
        region(max(2 * depth, 6))  # lanes, staging and work buffers exist; clocks up
        sync()
        el = region(calls)
        sync()
        for p in pars:
            p.close()
        for e in engs:
            e.close()
        return {"contexts": contexts, "pipeline_depth": depth, "calls": contexts * calls, "proofs_per_s": contexts * calls * n / el,
                "ms_per_call": 1e3 * el / calls / contexts, "host_to_device_GBps": contexts * calls * bytes_per_call / el / 1e9}
    out["one_context"] = run(1, 4, calls)
    out["four_contexts"] = run(4, 2, max(8, calls // 4))
    out["one_context_blocking"] = run(1, 1, max(8, calls // 4))  # bpp_verify_batch_packed's schedule: nothing overlaps
    return out


def small_calls_leg(bpp, packed, np, local_rank, params2, data2, callers=32, seconds=2.0):
    """The reference's own use: separate callers, each with ONE batch of 256 proofs per verify_batch call (its
    MAX_RANGE_PROOF_BATCH_SIZE), host buffers in.  (a) every caller on a context of its own (bpp_verify_batch_packed): a small
    call is a chain of latency-bound kernels and the chip runs about six of them side by side; the library's admission gate
    (bpp_small_call_limit, default 12) keeps the other callers waiting on the host instead of on the hardware queues -- the same
    with the gate switched off is measured next to it; (b) the same callers through ONE bpp_batcher, which pools the calls that
    are waiting into grouped engine calls (bpp_verify_resident_groups_actions); (c) the batcher with RecoverOnly calls (a wallet
    scanning outputs: seed nonces in, masks out, no final check)."""
    sl = slice(0, 256)
    inp = packed.PackedInput(data2["proofs"][sl], data2["commitments"][sl], data2["min_values"][sl], data2["min_present"][sl], None, LABEL)
    inp_seed = packed.PackedInput(data2["proofs"][sl], data2["commitments"][sl], data2["min_values"][sl], data2["min_present"][sl],
                                  data2["seeds"][sl], LABEL)
    want_masks = data2["blindings"][sl, 0, 0, :]
    eng0 = params2.engine
    limit0 = eng0.lib.bpp_small_call_limit(eng0.ctx, -1)
    out = {"workload": "%d host threads, each verifying one 256-proof reference batch per call from host buffers, %.1f s per form"
                       % (callers, seconds), "small_call_limit": limit0}
    for form in ("separate_contexts", "separate_contexts_gate_off", "batcher", "batcher_recover_only"):
        separate = form.startswith("separate")
        engs = [bpp.Engine(local_rank) for _ in range(callers if separate else 0)]
        pars = [params2.share(e) for e in engs]
        bat = None if separate else packed.Batcher(params2, inp, lanes=0)
        eng0.lib.bpp_small_call_limit(eng0.ctx, 0 if form.endswith("gate_off") else limit0)
        errors, cnt = [], [0] * callers
        info0 = packed.runtime_info(eng0)

        def call(k):
            if form == "batcher_recover_only":
                masks, present = bat.verify_action(inp_seed, bpp.VerifyAction.RecoverOnly)
                if not (present.all() and (masks[:, 0, :] == want_masks).all()):
                    raise RuntimeError("recovered masks differ from the prover's blindings")
            elif bat is not None:
                bat.verify(inp)
            else:
                packed.verify_batch(pars[k], inp, bpp.VerifyAction.VerifyOnly, 0)
        for k in range(callers if bat is None else 4):
            call(k)
        stop = time.time() + seconds

        def worker(k):
            try:
                while time.time() < stop:
                    call(k)
                    cnt[k] += 1
            except BaseException as e:  # noqa: BLE001
                errors.append(e)
        ths = [threading.Thread(target=worker, args=(k,)) for k in range(callers)]
        t0 = time.perf_counter()
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        el = time.perf_counter() - t0
        eng0.lib.bpp_small_call_limit(eng0.ctx, limit0)
        if errors:
            raise errors[0]
        info = packed.runtime_info(eng0)
        out[form] = {"calls_per_s": sum(cnt) / el, "proofs_per_s": 256 * sum(cnt) / el, "ms_per_call": 1e3 * el * callers / max(1, sum(cnt)),
                     "contexts": info["contexts"], "hw_queues": info["hw_queues"],
                     "calls_that_queued_at_the_gate": info["small_calls_queued"] - info0["small_calls_queued"]}
        if bat is not None:
            out[form]["engine_calls"] = bat.stats()["engine_calls"]
            out[form]["largest_pool"] = dict(zip(("calls", "proofs"), bat.largest_pool()))
            bat.close()
        for p in pars:
            p.close()
        for e in engs:
            e.close()
    return out


def pick_wide_transport(requested, dist, dmod, world):
    """COLLECTIVE.  The transport of the sharded leg's all_gathers: RCCL (`requested` = "rccl") only if EVERY rank can load it and
    make an id (bpp_comm_unique_id: dlopen + ncclGetUniqueId, no communication) -- a rank that cannot would leave the others waiting
    inside ncclCommInitRank; otherwise the caller-supplied transport over torch.distributed's gloo (bpp_comm_create_callbacks), and a
    note that says which ranks failed and how.  Returns (transport, note or None)."""
    if requested != "rccl":
        return "gloo", None
    ok, why = True, ""
    try:
        dmod.ShardComm.unique_id()
    except Exception as e:  # noqa: BLE001 - reported in the line
        ok, why = False, "%s: %s" % (type(e).__name__, e)
    every = [None] * world
    dist.all_gather_object(every, (ok, why))
    bad = ["rank %d (%s)" % (r, w) for r, (o, w) in enumerate(every) if not o]
    if bad:
        return "gloo", "RCCL is not usable on " + ", ".join(bad) + ": the all_gathers go through the caller-supplied transport over gloo instead"
    return "rccl", None


def make_wide_comm(transport, engine, dist, dmod, world):
    """COLLECTIVE.  One communicator of the sharded leg: over RCCL when `transport` says so and ncclCommInitRank succeeds on EVERY rank
    (a rank where it failed tells the others over the control plane; those that succeeded close theirs), else over gloo.
    Returns (communicator, transport actually used, note or None)."""
    if transport == "rccl":
        comm, why = None, ""
        try:
            comm = dmod.ShardComm.from_process_group(engine)
        except Exception as e:  # noqa: BLE001
            why = "%s: %s" % (type(e).__name__, e)
        every = [None] * world
        dist.all_gather_object(every, (comm is not None, why))
        bad = ["rank %d (%s)" % (r, w) for r, (o, w) in enumerate(every) if not o]
        if not bad:
            return comm, "rccl", None
        if comm is not None:
            comm.close()
        note = "bpp_comm_create failed on " + ", ".join(bad) + ": the all_gathers go through the caller-supplied transport over gloo instead"
        return dmod.ShardComm.from_process_group_gloo(engine), "gloo", note
    return dmod.ShardComm.from_process_group_gloo(engine), "gloo", None


def wide_leg(bpp, packed, np, torch, dist, device, local_rank, rank, world, params2, data2, args, sync):
    """BASELINE configs[3]: 4096 proofs as ONE reference batch sharded over the ranks, through the C ABI
    (bpp_verify_sharded_groups_wave: RCCL all_gathers on device buffers).  A rank's shards of G such batches are ONE resident
    batch on one context: every verifier kernel is launched once for all G, each all_gather carries all G, the weight chains
    (each over all 4096 proofs; shared out over the ranks and their weights gathered when there are several) run side by side on
    the host pool.  One call runs S such batches as a software pipeline of one host thread; W calls are in flight per rank
    (own communicator, own host thread each)."""
    dmod = importlib.import_module("bulletproofs-plus_amd.dist")
    # ONE call per rank, a pipeline of two times 64 batches, at every N.  (Several calls from several host threads, each with its
    # own communicator, do as well on one rank -- three times 32: the same rate -- but two communicators progressing from two
    # threads can reach their collectives in different orders on different ranks; that is legal for RCCL only while every
    # kernel involved can be resident at once, and this leg cannot be rehearsed on a multi-GPU node here: it takes the form
    # whose collectives are issued in one order on every rank by construction.)
    G = int(os.environ.get("BPP_BENCH_WAVE_BATCHES", "64"))
    W = int(os.environ.get("BPP_BENCH_WAVES", "1"))
    S = int(os.environ.get("BPP_BENCH_WAVE_SLOTS", "2"))
    n_local = 4096 // world
    counts = [n_local] * world
    nb = data2["proofs"].shape[0] // n_local
    calls = []
    transport, note = pick_wide_transport(args.transport, dist, dmod, world)
    for w in range(W):  # communicators are built collectively, in the same order on every rank
        engs = [bpp.Engine(local_rank) for _ in range(S)]
        pars = [params2.share(e) for e in engs]
        rbs = []
        for sl in range(S):
            # ranks were seeded differently: any n_local of this rank's proofs are a shard
            first = (w * S + sl) * G
            idx = np.concatenate([np.arange(((first + i) % nb) * n_local, ((first + i) % nb + 1) * n_local) for i in range(G)])
            rbs.append(packed.ResidentBatch(pars[sl], data2["proofs"][idx], data2["commitments"][idx], data2["min_values"][idx],
                                            data2["min_present"][idx], None, LABEL))
            rbs[-1].prepare(n_local if G > 1 else 0)
        comm, transport, note2 = make_wide_comm(transport, engs[0], dist, dmod, world)
        note = note or note2
        calls.append((engs, pars, rbs, comm))
    errors = []

    def worker(w, rounds):
        try:
            _, _, rbs, comm = calls[w]
            for _ in range(rounds):
                res = [r for part in comm.verify_groups_wave(rbs, G, counts) for r in part]
                if any(r["code"] != 0 for r in res):
                    raise RuntimeError("sharded batch failed: %r" % ([r for r in res if r["code"] != 0][:2],))
        except BaseException as e:  # noqa: BLE001 - NOTE: the other ranks' collectives of this call are stranded; the run is lost
            errors.append(e)

    def region(rounds):
        ths = [threading.Thread(target=worker, args=(w, rounds)) for w in range(W)]
        t0 = time.perf_counter()
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        if errors:
            raise errors[0]
        return time.perf_counter() - t0
    rounds = max(1, args.wide_steps)
    region(max(3, rounds // 4))
    sync()
    wel = region(rounds)
    sync()
    tt = torch.tensor([wel], dtype=torch.float64)  # (CPU tensor: the control plane is gloo)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    wel = float(tt.item())
    batches = W * S * G * rounds
    wave_ms = calls[0][3].last_timing()
    for engs, pars, rbs, comm in calls:
        comm.close()
        for rb in rbs:
            rb.close()
        for par in pars:
            par.close()
        for eng in engs:
            eng.close()
    return {"workload": "BASELINE configs[3]: 4096 non-aggregated 64-bit proofs as ONE reference batch, %d per rank, through "
                        "bpp_verify_sharded_groups_wave: all_gather (%s) of 32 B/proof transcript-RNG bytes, weight "
                        "chains replayed (shared out over the ranks and gathered when there are several), all_gather of the "
                        "128-byte accumulators, sum + identity test on the device; %d calls in flight per rank, each a pipeline of %d "
                        "times %d batches resident as one" % (n_local, "RCCL, device buffers" if transport == "rccl" else "gloo, host buffers", W, S, G),
            "rccl_ranks": world if transport == "rccl" else 0, "ranks": world,
            "transport": "rccl" if transport == "rccl" else "gloo through bpp_comm_create_callbacks", "transport_requested": args.transport,
            "transport_note": note,
            "proofs_per_s": 4096 * batches / wel, "ms_per_batch": 1e3 * wel / batches,
            "batches": batches, "in_flight": W * S * G, "waves": W, "slots_per_call": S, "batches_per_wave": G,
            "last_wave_host_ms": {k: round(v, 3) for k, v in wave_ms.items()}}


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        self_launch(args)

    # Several ranks on one host share its cores: each process's pool of weight-chain workers (default: up to 32) is sized to
    # its share before the library is loaded, so that eight ranks do not put 256 runnable threads on the node.
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    # ranks as processes share device memory through dmabuf handles only on this pool (RCCL's P2P setup otherwise fails with
    # hipIpcGetMemHandle: invalid argument); the GPU boxes export it already -- kept when the launcher's environment does not
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world_env > 1 and "BPP_HOST_THREADS" not in os.environ:
        os.environ["BPP_HOST_THREADS"] = str(max(4, min(32, usable_cpus() // world_env)))
    # where the weight chains run: read by the library once per context, when it is created (BPP_CHAIN)
    chain_mode = resolve_chain_mode(args, world_env)
    if args.chain != "auto" or chain_mode == "device":  # (auto with cores to spare: the engine's own rule, call by call)
        os.environ["BPP_CHAIN"] = {"host": "0", "device": "1", "host-wide": "2"}[chain_mode]
    import numpy as np
    import torch
    import torch.distributed as dist
    bpp = importlib.import_module("bulletproofs-plus_amd")
    packed = importlib.import_module("bulletproofs-plus_amd.packed")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.one_device else int(os.environ.get("LOCAL_RANK", "0"))
    if args.one_device:
        args.transport = "gloo"
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE is %d" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    use_dist = "RANK" in os.environ  # launched by torch.distributed.run (also with one rank: exercises the RCCL path)
    if use_dist:
        # the harness' own control plane (barriers, the MAX over ranks of the elapsed time, the verdict AND, rank 0's
        # ncclUniqueId) runs over gloo on the CPU: the ONLY RCCL in this process is the one the engine loads for its data path
        # (the librccl next to its own HIP runtime, csrc/engine_shard.h) -- torch's bundled copy is never initialised
        dist.init_process_group("gloo")

    def sync():
        torch.cuda.synchronize(device)
        if use_dist:
            dist.barrier()

    G = bpp.create_pedersen_gens_with_extension_degree
    eng0 = bpp.Engine(local_rank)
    eng0.profile(True)
    t_setup = time.perf_counter()

    def cpu_side(obj, fn):
        """north_star: "the reference's own CPU path timed on the GPU box's host cores ... in the same run" -- every leg's object
        gets a cpu_baseline from the same inputs (a failure is reported inside it, the leg's own numbers stand)"""
        if args.no_cpu_baseline or args.only:
            return obj
        try:
            obj["cpu_baseline"] = fn()
        except Exception as e:  # noqa: BLE001
            obj["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
        return obj

    # ------------------------------------------------------------------ legs other than the headline, as functions
    def cfg3_leg(steps=48, warmup=8, only=False):
        p3 = bpp.RangeParameters.init(64, 8, G(1), engine=eng0)
        R3 = int(os.environ.get("BPP_BENCH_CFG3_BATCHES", "64"))
        S3 = 1 if only else int(os.environ.get("BPP_BENCH_CFG3_INFLIGHT", "4"))
        d3 = make_inputs(np, packed, p3, 256 * R3, seed=8675309 + 3)
        leg3 = Leg(bpp, packed, torch, device, p3, d3, 256, R3, S3, 256)
        el3, lat3, pr3 = timed(leg3, steps, warmup, sync, preheat_ms=0.0 if only else SIDE_PREHEAT_MS)
        sync()
        al3 = [leg3.one_step(0)[1].get("msm_accumulate_ms", 0.0) for _ in range(3)]
        roof3, st3 = kernel_roofline(pr3, sum(al3) / 3 if min(al3) > 0 else None)
        leg3.close()
        res = {"workload": "BASELINE configs[2]: reference batches of 256 x aggregation-8 64-bit proofs, extension degree 1; "
                           "one step = %d such batches, %d steps in flight" % (R3, S3),
               "proofs_per_s": 256 * R3 * steps / el3, "ms_per_step": 1e3 * el3 / steps, "steps": steps, "roofline": roof3,
               "stages_ms": st3}
        if not only and d3["proofs"].shape[0] >= 4096:
            # north_star's target sentence, literally: AGGREGATED proofs "on a batch of 4096 on one MI355X" -- 4096 of this leg's
            # aggregation-8 proofs as ONE reference batch per call (chunk = 0: one weight chain over 4096 proofs, 1024 generator
            # columns, one 119 809-term MSM; held to the oracle at this size by tests/test_gpu_round6.py), several calls in flight
            try:
                Sa = int(os.environ.get("BPP_BENCH_AGG4096_INFLIGHT", "8"))
                lega = Leg(bpp, packed, torch, device, p3, d3, 4096, 1, Sa, 0, profile=False)
                na = 10 * Sa
                ela, lata, _ = timed(lega, na, 3 * Sa, sync, preheat_ms=SIDE_PREHEAT_MS)
                sync()
                ala = [lega.one_step(0)[0] for _ in range(5)]
                lega.close()
                agg = {"workload": "4096 x aggregation-8 64-bit proofs as ONE reference batch per call (chunk = 0), %d calls in flight, "
                                   "each on another 4096 proofs" % Sa, "proofs_per_s": 4096 * na / ela,
                       "values_per_s": 8 * 4096 * na / ela, "ms_per_batch_in_flight": 1e3 * sum(lata) / len(lata),
                       "ms_per_batch_alone": 1e3 * sum(ala) / len(ala), "calls": na}
                res["agg4096"] = cpu_side(agg, lambda: cpu_verify_baseline((64, 8, 1), d3, 4096, seconds=args.cpu_leg_seconds / 2,
                                                                          all_cores=False))
            except Exception as e:  # noqa: BLE001 - reported inside the object, the leg's own numbers stand
                res["agg4096"] = {"error": "%s: %s" % (type(e).__name__, e)}
        p3.close()
        # benches/range_proof.rs:206-262 on the host: the same 256 x aggregation-8 batches through the CPU port
        return cpu_side(res, lambda: cpu_verify_baseline((64, 8, 1), d3, 256, seconds=args.cpu_leg_seconds))

    def recover_only_leg(data, params, steps=32, warmup=6):
        """SURVEY 8(f)2 / src/range_proof.rs:941-969,1040-1043: a wallet scanning outputs.  RecoverOnly over the headline's 65 536
        resident proofs (seed nonces resident too) in 1024-proof reference batches, as many steps in flight as the headline: PASS 1, decompression,
        k_masks; no weight chains, no PASS 2, no MSM.  Masks come back as arrays and are checked against the prover's blindings."""
        Rr, Sr = max(1, args.batches_per_step), max(1, args.concurrency)
        legr = Leg(bpp, packed, torch, device, params, data, 1024, Rr, Sr, 1024, action=int(bpp.VerifyAction.RecoverOnly))
        elr, latr, prr = timed(legr, steps, warmup, sync, preheat_ms=SIDE_PREHEAT_MS)
        sync()
        masks, present = legr._call(legr.slots[0][3])
        ok = bool(present.all()) and bool((masks[:, 0, :] == data["blindings"][legr.data_idx[0], 0, 0, :]).all())
        avg = {k: sum(p[k] for p in prr) / len(prr) for k in prr[0]}
        n_pts = 1024 * Rr * 15
        dec_ms, msk_ms, tr_ms = avg["decompress_ms"], avg["masks_ms"], avg["transcripts_ms"]
        legr.close()
        return cpu_side({"workload": "VerifyAction::RecoverOnly over %d resident 64-bit proofs with seed nonces, as %d reference batches of 1024 per step, "
                            "%d steps in flight: PASS 1 + decompression + mask recovery (src/range_proof.rs:941-969), no weight chain, no MSM "
                            "(:1040-1043); masks returned as arrays" % (1024 * Rr, Rr, Sr),
                "proofs_per_s": 1024 * Rr * steps / elr, "ms_per_step": 1e3 * elr / steps, "steps": steps, "masks_equal_blindings": ok,
                "stages_ms": {"transcripts_ms": round(tr_ms, 4), "decompress_ms": round(dec_ms, 4), "masks_ms": round(msk_ms, 4)},
                "roofline": {"bound": "hbm", "kernel": "k_decompress (15 proof points per proof; the dominant kernel without an MSM)",
                             "kernel_ms": dec_ms, "algorithmic_bytes": 32 * n_pts, "achieved": 32 * n_pts / (dec_ms * 1e-3) / 1e9,
                             "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": 32 * n_pts / (dec_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "traffic": None,
                             "note": "integer-VALU bound (254 squarings + 25 multiplications per point): see valu",
                             "valu": {"achieved_Tmad_per_s": n_pts * (254 * 55 + 25 * 100) / (dec_ms * 1e-3) / 1e12,
                                      "peak_Tmad_per_s": VALU_PEAK_TMAD,
                                      "frac": n_pts * (254 * 55 + 25 * 100) / (dec_ms * 1e-3) / (VALU_PEAK_TMAD * 1e12)}}},
                        lambda: cpu_verify_baseline((64, 1, 1), data, 256, action=2, seconds=args.cpu_leg_seconds))

    def prover_leg(iters5=8):
        p5 = bpp.RangeParameters.init(64, 4, G(3), engine=eng0)
        d5 = make_inputs(np, packed, p5, 1024, seed=8675309 + 5)  # also builds the fixed-base tables
        for _ in range(2):  # untimed, like the headline's warm-up steps (tools/bench_prover_leg.py does the same)
            packed.prove(p5, d5["values"], d5["blindings"], d5["commitments"], d5["min_values"], d5["min_present"], None, LABEL, d5["ext"])
        t0 = time.perf_counter()
        for _ in range(iters5):
            packed.prove(p5, d5["values"], d5["blindings"], d5["commitments"], d5["min_values"], d5["min_present"], None, LABEL,
                         d5["ext"])
        el5 = time.perf_counter() - t0
        pp = eng0.last_prove_profile()
        fb_bytes = 64 * pp["fb_terms"]
        # the same call from four contexts at once (one fixed-base table, Arc semantics): one call's Fiat-Shamir steps (a
        # handful of wavefronts) overlap the others' fixed-base MSMs, as the verifier's legs do with four steps in flight
        conc = {}
        if iters5 >= 4:
            engs = [bpp.Engine(local_rank) for _ in range(4)]
            pars = [p5.share(e) for e in engs]
            errs = []

            def prove_worker(k, n):
                try:
                    for _ in range(n):
                        packed.prove(pars[k], d5["values"], d5["blindings"], d5["commitments"], d5["min_values"], d5["min_present"], None,
                                     LABEL, d5["ext"])
                except BaseException as e:  # noqa: BLE001
                    errs.append(e)
            for n in (1, iters5):
                ths = [threading.Thread(target=prove_worker, args=(k, n)) for k in range(4)]
                t1 = time.perf_counter()
                for th in ths:
                    th.start()
                for th in ths:
                    th.join()
                elc = time.perf_counter() - t1
                if errs:
                    raise errs[0]
            conc = {"calls_in_flight": 4, "proofs_per_s": 4 * 1024 * iters5 / elc, "calls": 4 * iters5}
            for q in pars:
                q.close()
            for e in engs:
                e.close()
        # the same call with A1 and B in the uniform-access form as well ("ct" = 2: the reference's constant-time `&P * Scalar`,
        # src/range_proof.rs:572-584): what the property costs, in the same run
        ct2 = {}
        if iters5 >= 4:
            eng0.set_option("ct", 2)
            try:
                for _ in range(2):
                    out2 = packed.prove(p5, d5["values"], d5["blindings"], d5["commitments"], d5["min_values"], d5["min_present"], None, LABEL, d5["ext"])
                t2 = time.perf_counter()
                for _ in range(iters5):
                    out2 = packed.prove(p5, d5["values"], d5["blindings"], d5["commitments"], d5["min_values"], d5["min_present"], None, LABEL,
                                        d5["ext"])
                el2 = time.perf_counter() - t2
                ct2 = {"proofs_per_s": 1024 * iters5 / el2, "ms_per_call": 1e3 * el2 / iters5,
                       "cost_against_default": 1.0 - (1024 * iters5 / el2) / (1024 * iters5 / el5),
                       "bytes_equal_default": bool((out2 == d5["proofs"]).all()),
                       "note": "option ct = 2: no secret scalar of A1 / B addresses a table; default ct = 1 (commit and the witness check only)"}
            finally:
                eng0.set_option("ct", -1)
        p5.close()
        return cpu_side({"workload": "BASELINE configs[4]: bpp_prove_batch over 1024 x aggregation-4 64-bit proofs, extension "
                            "degree 3; host witness buffers in, proof bytes out (PCIe-inclusive)",
                "proofs_per_s": 1024 * iters5 / el5, "ms_per_call": 1e3 * el5 / iters5, "calls": iters5, "four_calls_in_flight": conc,
                "uniform_access_A1_B": ct2,
                "roofline": {"bound": "hbm",
                             "kernel": "k_fb_part (fixed-base MSM of every L and R -- and of A1's four public points, or of A1 and B with "
                                       "ct < 2 -- as one-wavefront slices; the secret-only terms run in the uniform-access forms of ct.h)",
                             "kernel_ms": pp["fb_msm_ms"], "launches": pp["fb_launches"], "algorithmic_bytes": fb_bytes,
                             "achieved": fb_bytes / (pp["fb_msm_ms"] * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                             "frac": fb_bytes / (pp["fb_msm_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, "traffic": None,
                             "fb_window_bits": pp["fb_window_bits"], "fb_terms": pp["fb_terms"],
                             "additions_per_s_kernel_events": pp["fb_terms"] * pp["fb_windows"] / (pp["fb_msm_ms"] * 1e-3),
                             "additions_per_s_call_wall": pp["fb_terms"] * pp["fb_windows"] / (pp["total_ms"] * 1e-3),
                             "engine_call_ms": pp["total_ms"],
                             "note": "achieved / algorithmic_bytes / kernel_ms are sums over the %d launches of ONE call (both "
                                     "sub-batch streams); 64 B per term.  The launches of the two sub-batch streams OVERLAP, so the "
                                     "summed event time exceeds the wall time they cover: additions_per_s_call_wall divides by the "
                                     "whole call instead (start-up, Fiat-Shamir steps and the final step included)" % pp["fb_launches"]}},
                        # benches/range_proof.rs:43-107 on the host: the same witnesses through the CPU port's prover
                        lambda: cpu_prove_baseline((64, 4, 3), d5, seconds=args.cpu_leg_seconds))

    if args.only in ("cfg3", "prover"):  # a rocprofv3 child pass of measure_traffic(): just the kernels, a short line
        res = cfg3_leg(args.steps, args.warmup, only=True) if args.only == "cfg3" else prover_leg(2)
        print(json.dumps({"only": args.only, "proofs_per_s": res["proofs_per_s"]}))
        eng0.close()
        return

    # ------------------------------------------------------------------ headline: BASELINE configs[1]
    # steps in flight: four (round 6: callers nap instead of spinning, a fourth costs nothing and sustains 1-3 % more over long
    # regions); the device chains put ~2.5 ms of lone-wavefront latency on every step's critical path: six
    if args.concurrency <= 0:
        args.concurrency = 6 if chain_mode == "device" else 4
    R, S = max(1, args.batches_per_step), max(1, args.concurrency)
    params2 = bpp.RangeParameters.init(64, 1, G(1), engine=eng0)
    data2 = make_inputs(np, packed, params2, 1024 * R, seed=8675309 + 1000 * rank)
    # the timed region carries the roofline kernel's two events per step and no others (profile level 2): an event at every stage
    # boundary -- thirteen per step -- read 0-4 % lower on one box and no different on another (profiles/r06_stage_events_ab.txt:
    # inside the +-5 % that identical runs differ by, profiles/r06_run_to_run.txt); the stage table comes from a short pass with
    # all events on AFTER the timed region, the same steps in flight (BPP_BENCH_STAGE_EVENTS=1: all events in the timed region,
    # as until round 5; =0: none)
    ev_level = {"1": 1, "0": 0}.get(os.environ.get("BPP_BENCH_STAGE_EVENTS", ""), 2)
    leg = Leg(bpp, packed, torch, device, params2, data2, 1024, R, S, 1024, profile=ev_level)
    gen_s = time.perf_counter() - t_setup
    # the shader clock held during the timed region (a light kernel sees 2.4 GHz, this load 2.0-2.2): sampled over the
    # middle of it by one napping wavefront on a context of its own
    clk_eng = bpp.Engine(local_rank)
    est_ms = 2.7 * args.steps
    # Untimed pre-heat: right after a load increase the chip runs at 1.9-2.0 GHz and takes ~50 ms of full load to reach the
    # 2.3+ GHz it then sustains (tools/clock_ramp.py); 5 warm-up steps are 15 ms.  Without this a 20-step timed region (55 ms)
    # runs mostly at the ramp's clock and reads 10-15 % low; `shader_clock_ghz` in the line is what the timed region held.
    # Round 6: the rate per 100 ms window of a leg that starts from idle (tools/rate_ramp.py, profiles/r06_rate_ramp.jsonl) is flat
    # at 27.5 M proofs/s from the second window on EXCEPT one window 0.3-0.4 s after the start (17 M at 2.15 GHz: the power
    # controller's answer to the step in load); with 150 ms of pre-heat that window fell inside a 256-step region (0.66 s) and
    # cost it 6-8 % (profiles/r06_preheat_ab.txt: 25.0-25.4 M against 27.3-27.8 M, whatever the wait policy or the chain form; the
    # same step as a LATER leg of the same process read 27 M all along, profiles/r06_position_ab.txt).  1000 ms puts the region behind it.
    if float(os.environ.get("BPP_BENCH_COOL_MS", "0")) > 0:  # (an experiment knob: idle time between input generation and the pre-heat)
        time.sleep(float(os.environ["BPP_BENCH_COOL_MS"]) / 1e3)
    sync()  # under torch.distributed the first barrier builds the communicator (100s of ms): not between warm-up and timing
    step_error, elapsed, lat, profs, clock_ghz = None, 0.0, [], [], None
    pool_cpu0 = proc_cpu0 = 0
    thr0 = {}
    try:
        if args.preheat_ms > 0 and not args.only:
            leg.run_steps(max(1, int(args.preheat_ms / 2.7)))
        leg.run_steps(args.warmup)  # (the warm-up, outside the host-CPU accounting below; timed() then starts at once)
        pool_cpu0, proc_cpu0 = bpp.host_pool_cpu_ns(), time.process_time()
        thr0 = thread_cpu_seconds()
        elapsed, lat, profs, clock_ghz = timed(leg, args.steps, 0, sync,
                                               clock=lambda: bpp.shader_clock_ghz(clk_eng, int(1e3 * max(5.0, min(200.0, 0.6 * est_ms)))))
    except Exception as e:  # noqa: BLE001 - a failed step: no number; the other ranks still get their collective
        step_error = e
    pool_cpu_ms = (bpp.host_pool_cpu_ns() - pool_cpu0) / 1e6 / max(1, args.steps)
    proc_cpu_ms = (time.process_time() - proc_cpu0) * 1e3 / max(1, args.steps)
    thr_busy = thread_cpu_delta(thr0, thread_cpu_seconds(), elapsed) if step_error is None else []
    local_ms = 1e3 * elapsed / max(1, args.steps)
    clk_eng.close()
    # every timed step ran and raised nothing (each step verifies all its batches or raises)
    ok_all = 1 if (step_error is None and len(lat) == args.steps) else 0
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64)  # (CPU tensors: the control plane is gloo)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        flag = torch.tensor([ok_all], dtype=torch.int32)  # verdicts of the independent shards
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok_all = int(flag.item())
    if ok_all != 1:
        sys.stderr.write("bench.py: the timed region did not complete on every rank: %r\n" % (step_error,))
        if use_dist:
            dist.destroy_process_group()
        raise SystemExit(3)
    if args.only == "headline":
        print(json.dumps({"only": "headline", "proofs_per_s": 1024 * R * args.steps / elapsed}))
        leg.close()
        params2.close()
        eng0.close()
        return
    # calibration after the timed region: the roofline kernel with nothing co-running (in the timed region several steps
    # share the chip, so each launch of it is stretched by its neighbours)
    sync()
    alone = [leg.one_step(0)[1].get("msm_accumulate_ms", 0.0) for _ in range(3)]
    alone_ms = sum(alone) / len(alone) if min(alone) > 0 else None
    sync()
    stage_profs = profs
    if ev_level == 2:  # the stage intervals: a few steps with every event on, the same steps in flight, outside the timed region
        leg.set_profile(1)
        leg.run_steps(len(leg.slots))
        _, stage_profs = leg.run_steps(max(8, 2 * len(leg.slots)))
        leg.set_profile(2)
        sync()
    roof, _ = kernel_roofline(profs, alone_ms)
    _, stages = kernel_roofline(stage_profs)
    if roof and clock_ghz:
        # the multiplier's peak at the clock the chip really held: 64 lanes / clock / CU x 256 CUs x clock
        at_clock = 64 * 256 * clock_ghz * 1e9 / 1e12
        roof["valu"]["shader_clock_ghz"] = clock_ghz
        roof["valu"]["peak_at_clock_Tmad_per_s"] = at_clock
        roof["valu"]["frac_at_clock"] = roof["valu"]["achieved_Tmad_per_s"] / at_clock
    total = 1024 * R * world * args.steps
    out = {
        "metric": "64-bit range proofs verified/sec (batch)", "value": total / elapsed, "unit": "proofs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: reference batches of 1024 x aggregation-1 64-bit proofs, extension degree 1, "
                               "VerifyOnly, resident in HBM; one step = %d such batches (all distinct proofs) in one engine call" % R,
                   "proofs_per_reference_batch": 1024, "batches_per_step": R, "proofs_per_step_per_gpu": 1024 * R,
                   "mode": "each 1024-proof batch is one reference batch: own weight transcript, own final MSM, own identity test",
                   "steps_in_flight_per_gpu": len(leg.slots), "parallelism": "proof-sharded x%d" % world,
                   "inputs": "%d distinct proofs per rank, proved on the box by bpp_prove_batch (recipe of "
                             "benches/range_proof.rs:206-262), %.1f s" % (1024 * R, gen_s)},
        "step_latency_ms": 1e3 * sum(lat) / len(lat), "steps_completed": len(lat), "all_steps_verified": bool(ok_all),
        "host_threads": bpp.host_threads(), "nproc": os.cpu_count(), "usable_cpus": usable_cpus(),
        "weight_chains": chain_mode, "shader_clock_ghz": clock_ghz,
        # the one sequential part of a verification stays on the host (the batch weight chains, src/range_proof.rs:849-853,894):
        # CPU time of the library's host pool per timed step, and of the whole process (time.process_time: every thread of it, the
        # callers' naps, the shader-clock sampler and the runtime's own threads included); cores kept busy = cpu ms / step ms.  With
        # N ranks on one node, N x host_cores_busy against the node's cores tells a host-bound scaling curve from a GPU-bound one.
        "host_chain_cpu_ms_per_step": pool_cpu_ms, "host_process_cpu_ms_per_step": proc_cpu_ms,
        "host_cores_busy": proc_cpu_ms / local_ms if local_ms > 0 else None,
        # the same by thread name (/proc/self/task, 10 ms ticks: meaningful for regions of 0.5 s and more)
        "host_cores_busy_by_thread": thr_busy,
    }
    # N ranks of this host want N x host_cores_busy schedulable cores: more than there are = the scaling curve is the host's
    out["host_bound"] = bool(out["host_cores_busy"] is not None and world * out["host_cores_busy"] > usable_cpus())
    if args.one_device:  # every rank on device 0: ONE GPU did all of it -- a rehearsal of the multi-rank code, never a scaling point
        out.update(n_gpus=1, ranks=world, rehearsal=True, scaling=None)
        out["config"]["devices"] = "ALL %d ranks on device 0 (--one-device: a rehearsal of the multi-rank code, not a scaling point)" % world
    if use_dist:
        mine = {"rank": rank, "device": local_rank, "host_threads": bpp.host_threads(), "usable_cpus": usable_cpus(),
                "ms_per_step_local": local_ms, "host_chain_cpu_ms_per_step": pool_cpu_ms, "host_process_cpu_ms_per_step": proc_cpu_ms}
        mine["host_cores_busy"] = proc_cpu_ms / local_ms if local_ms > 0 else None
        mine["weight_chains"] = chain_mode
        every = [None] * world
        dist.all_gather_object(every, mine)
        out["per_rank"] = every
        # ranks of ONE node share its cores (the driver's runs are single-node): all of them together against what is schedulable
        want = sum(e["host_cores_busy"] or 0.0 for e in every)
        out["host_cores_busy_all_ranks"] = want
        out["host_bound"] = bool(want > min(e["usable_cpus"] for e in every))
    profiler_legs = rank == 0 and world == 1 and not use_dist and not args.no_traffic
    if roof:
        sq = {}
        if profiler_legs:
            tr = measure_traffic("k_msm_accumulate", "headline")
            if tr:
                roof.update(tr)
            sq = measure_sq(["k_decompress", "k_msm_accumulate"], "headline")
            if sq.get("k_msm_accumulate"):
                roof.setdefault("alone", {}).update({"wait_inst_frac": sq["k_msm_accumulate"]["wait_inst_frac"],
                                                     "valu_instr_per_launch": sq["k_msm_accumulate"]["valu_instr_per_launch"]})
        out["roofline"] = roof
        out["stages_ms"] = stages
        out["stage_events"] = {2: "timed region: the roofline kernel's two events per step only (roofline.kernel_ms); stages_ms: a pass with every "
                                  "stage event on after the timed region, the same steps in flight",
                               1: "every stage event in the timed region", 0: "none"}[ev_level]
        # the verifier's other dominant kernel: 15 proof points per non-aggregated 64-bit proof (A, A1, B, 6 L, 6 R)
        out["roofline_decompress"] = decompress_roofline(stages, 15 * 1024 * R, sq, clock_ghz)
    # bpp_batch_upload_packed alone: host proof/statement buffers -> parsed, packed and resident (R batches of 1024); never `value`
    out["pcie_inclusive_upload_ms"] = 1e3 * leg.upload_s
    leg.close()

    extra = {}
    # ------------------------------------------------------------------ host buffers in -> verdict out (what verify_batch receives)
    if rank == 0 and world == 1 and not args.no_extra:
        try:
            extra["host_in"] = host_in_leg(bpp, packed, np, local_rank, params2, data2, R, args.host_in_calls, sync)
            out["pcie_inclusive_value"] = extra["host_in"]["one_context"]["proofs_per_s"]
        except Exception as e:  # noqa: BLE001 - a side leg: reported, never allowed to take the line down
            extra["host_in"] = {"error": "%s: %s" % (type(e).__name__, e)}
    # ------------------------------------------------------------------ N > 1: BASELINE configs[3], one batch over all ranks
    if use_dist:
        # The headline above is complete; this leg must not be able to take it down.  An exception is reported inside the
        # object; a collective that never returns (this leg cannot be rehearsed on a multi-GPU node by the builder) is cut
        # short by a watchdog: after --wide-timeout seconds every rank gives up at once, rank 0 prints the line it has
        # (extra.wide = the timeout) and EVERY rank ends with a non-zero code without waiting for the stuck communicator: a
        # kernel or collective left hanging behind must never read as success to whoever started the run.
        def give_up():
            if rank == 0:
                out["extra"] = dict(extra, wide={"error": "the sharded leg did not finish within %d s; headline unaffected" % args.wide_timeout,
                                                 "rccl_ranks": world})
                print(json.dumps(out), flush=True)
            sys.stderr.write("bench.py: rank %d: the sharded leg hung; giving up with exit code %d\n" % (rank, WATCHDOG_EXIT))
            sys.stderr.flush()
            os._exit(WATCHDOG_EXIT)
        dog = threading.Timer(args.wide_timeout, give_up)
        dog.daemon = True
        dog.start()
        try:
            extra["wide"] = wide_leg(bpp, packed, np, torch, dist, device, local_rank, rank, world, params2, data2, args, sync)
        except Exception as e:  # noqa: BLE001
            extra["wide"] = {"error": "%s: %s" % (type(e).__name__, e), "rccl_ranks": world}
        dog.cancel()

    legs_wanted = [x for x in os.environ.get("BPP_BENCH_EXTRA_LEGS", "").split(",") if x]  # (A/B runs: only these extra legs)

    def side_leg(name, fn):
        """an extra leg: its failure is reported inside its own object, the headline line is printed either way"""
        if legs_wanted and name not in legs_wanted:
            extra[name] = {"skipped": "BPP_BENCH_EXTRA_LEGS"}
            return False
        try:
            extra[name] = fn()
        except Exception as e:  # noqa: BLE001
            extra[name] = {"error": "%s: %s" % (type(e).__name__, e)}
        return "error" not in extra[name]

    if rank == 0 and world == 1 and not args.no_extra:
        # -------------------------------------------------------------- the headline's step with the weight chains on the OTHER side
        def other_chain_leg():
            """the A/B the line owes: the same resident step with the weight chains on the device (one wavefront per
            reference batch, six steps in flight) when the headline kept the sponges on host cores -- or the other way round --
            with what each costs the host"""
            other = "host-wide" if chain_mode == "device" else "device"
            if os.environ.get("BPP_BENCH_OTHER_CHAIN") == "same":  # (a position check: the headline's own step once more, later in the run)
                other = chain_mode
            So = 6 if other == "device" else 4
            lego = Leg(bpp, packed, torch, device, params2, data2, 1024, R, So, 1024, profile=False,
                       options={"chain": {"host": 0, "device": 1, "host-wide": 2}[other]})
            steps = max(40, min(args.steps, 120))
            clk = bpp.Engine(local_rank)
            timed(lego, 1, 2 * So, sync, preheat_ms=SIDE_PREHEAT_MS)
            pool0, cpu0 = bpp.host_pool_cpu_ns(), time.process_time()
            try:
                el, _, _, ghz = timed(lego, steps, 0, sync, clock=lambda: bpp.shader_clock_ghz(clk, int(1e3 * max(5.0, min(200.0, 1.5 * steps)))))
            finally:
                clk.close()
            cpu_ms, pool_ms = (time.process_time() - cpu0) * 1e3 / steps, (bpp.host_pool_cpu_ns() - pool0) / 1e6 / steps
            lego.close()
            return {"workload": "the headline's step (64 reference batches of 1024 proofs, resident) with weight chains = %s, %d steps in flight, "
                                "%d timed steps" % (other, So, steps), "weight_chains": other, "proofs_per_s": 1024 * R * steps / el,
                    "ms_per_step": 1e3 * el / steps, "steps": steps, "shader_clock_ghz": ghz, "host_cores_busy": cpu_ms / (1e3 * el / steps),
                    "host_chain_cpu_ms_per_step": pool_ms, "headline_weight_chains": chain_mode}
        side_leg("other_chain", other_chain_leg)
        # -------------------------------------------------------------- configs[2]: 256 x aggregation-8
        side_leg("cfg3", cfg3_leg)
        if profiler_legs and extra["cfg3"].get("roofline"):
            tr = measure_traffic("k_msm_accumulate", "cfg3")
            if tr:
                extra["cfg3"]["roofline"].update(tr)
        # -------------------------------------------------------------- one 4096-proof reference batch (north_star's sentence)
        def wide4096_leg():
            Sw = int(os.environ.get("BPP_BENCH_WIDE_INFLIGHT", "12"))
            legw = Leg(bpp, packed, torch, device, params2, data2, 4096, 1, Sw, 0, profile=False)
            nw = 40 * Sw
            elw, latw, _ = timed(legw, nw, 12 * Sw, sync, preheat_ms=SIDE_PREHEAT_MS)  # the rate: no stage events; untimed run as the headline's pre-heat
            legw.set_profile(True)
            _, _, prw = timed(legw, 4 * Sw, Sw, sync)  # stage times with the same calls in flight
            sync()
            alw = [legw.one_step(0) for _ in range(5)]
            roofw, stw = kernel_roofline(prw, sum(a[1].get("msm_accumulate_ms", 0.0) for a in alw) / 5)
            res = {"workload": "4096 non-aggregated 64-bit proofs as ONE reference batch (chunk = 0: one weight chain over "
                                             "4096, one 65 667-term MSM) on one GPU; %d calls in flight, each on another 4096 proofs" % Sw,
                                 "proofs_per_s": 4096 * nw / elw, "ms_per_batch_in_flight": 1e3 * sum(latw) / len(latw),
                                 "ms_per_batch_alone": 1e3 * sum(a[0] for a in alw) / 5, "steps": nw, "roofline": roofw, "stages_ms": stw}
            legw.close()
            return cpu_side(res, lambda: cpu_verify_baseline((64, 1, 1), data2, 4096, seconds=args.cpu_leg_seconds))
        side_leg("wide4096", wide4096_leg)
        # -------------------------------------------------------------- single-call latency (configs[0]'s shape)
        def latency_leg():
            lat_out = {}
            for nb in (1, 64, 256):
                legl = Leg(bpp, packed, torch, device, params2, data2, nb, 1, 1, 0, profile=False)
                legl.run_steps(10)
                ls, _ = legl.run_steps(50)  # the latency: no stage events
                ls.sort()
                legl.set_profile(True)
                legl.run_steps(3)
                lsp, lp = legl.run_steps(20)  # stage times (the events add ~0.1 ms to the call)
                lsp.sort()
                rl, sl = kernel_roofline(lp)
                lat_out["batch_%d" % nb] = {"ms_per_call_median": 1e3 * ls[len(ls) // 2], "ms_per_call_min": 1e3 * ls[0],
                                            "ms_per_call_median_with_stage_events": 1e3 * lsp[len(lsp) // 2],
                                            "proofs_per_s": nb / ls[len(ls) // 2], "roofline": rl, "stages_ms": sl}
                legl.close()
                # configs[0] is the reference's own single-call CPU path (benches/range_proof.rs:115-119,199-203): the same call
                # through the CPU port on one core, beside the engine's
                cpu_side(lat_out["batch_%d" % nb], lambda: cpu_verify_baseline((64, 1, 1), data2, nb, seconds=args.cpu_leg_seconds / 2,
                                                                              all_cores=False))
            cb = {k: {"ms_per_call": v["cpu_baseline"]["ms_per_batch"], "proofs_per_s": v["cpu_baseline"]["value"]}
                  for k, v in lat_out.items() if "value" in v.get("cpu_baseline", {})}
            if cb:
                lat_out["cpu_baseline"] = dict(cb, cores=1, kind="port", unit="ms per call / proofs/s",
                                               sample="one verify call of 1, 64 and 256 proofs at a time through oracle/c, one thread")
            return dict(lat_out, workload="BASELINE configs[0]'s shape through the engine: ONE call at a time, 1, 64 and 256 "
                                          "non-aggregated 64-bit proofs (benches/range_proof.rs:115-119,199-203), resident input; calls of "
                                          "up to ~1200 proofs run the final MSM as a half-scalar plan (s = s_lo + 2^126 s_hi: half the "
                                          "Horner doublings)")
        side_leg("latency", latency_leg)
        # -------------------------------------------------------------- many callers, one 256-proof verify_batch call each
        side_leg("small_calls", lambda: small_calls_leg(bpp, packed, np, local_rank, params2, data2))
        side_leg("recover_only", lambda: recover_only_leg(data2, params2))
        # -------------------------------------------------------------- configs[4]: batch prover
        if side_leg("prover", prover_leg) and profiler_legs:
            tr = measure_traffic("k_fb_part", "prover")
            if tr:  # the counters are per dispatch: scaled to the launches of one call, like achieved / algorithmic_bytes
                rp = extra["prover"]["roofline"]
                tr["traffic_per_launch_avg"] = tr["traffic"]
                tr["traffic"] = tr["traffic"] * rp["launches"]
                rp.update(tr)
    if extra and rank == 0:
        out["extra"] = extra

    if rank == 0 and world == 1 and not args.no_cpu_baseline:  # (the contract: on rank 0 at N = 1 only)
        from oracle import cport  # cpu_baseline leg only: the oracle is the thing timed here, never the product path
        cp = cport.Params(64, 1, 1)
        ncpu = usable_cpus()
        k = 256 * max(1, min(ncpu, 4))
        pr, cm, mv = data2["proofs"], data2["commitments"], data2["min_values"]
        sample = [{"proof": pr[i].tobytes(), "commitments": [cm[i, 0].tobytes()], "min_values": [int(mv[i, 0])],
                   "seed_nonce": None, "label": LABEL} for i in range(k)]
        rc, sec1 = cp.verify_timed(sample[:256], 256, 1)
        iters = max(1, int(args.cpu_seconds / max(sec1, 1e-3)))
        rc, sec = cp.verify_timed(sample[:256], 256, iters)
        assert rc == 0
        rcm, secm1 = cp.verify_timed_mt(sample, 256, 1, ncpu)  # calibration: logical CPUs may outnumber the schedulable ones
        iters_mt = max(1, min(iters, int(args.cpu_seconds / max(secm1, 1e-3))))
        rcm, secm = cp.verify_timed_mt(sample, 256, iters_mt, ncpu)
        assert rcm == 0
        out["cpu_baseline"] = {"value": 256 * iters / sec, "unit": "proofs/s", "cores": 1, "kind": "port",
                               "sample": "%d x verify of one 256-proof reference batch (MAX_RANGE_PROOF_BATCH_SIZE) of the workload, "
                                         "single thread, oracle/c port with dalek's algorithms (the reference is single-threaded)" % iters,
                               "all_cores": {"value": 256 * iters_mt * ncpu / secm, "unit": "proofs/s", "cores": ncpu,
                                             "sample": "%d threads, each %d x verify of a 256-proof reference batch" % (ncpu, iters_mt)},
                               "nproc": os.cpu_count(), "usable_cpus": ncpu}
        cp.close()
    if rank == 0:
        print(json.dumps(out))
    params2.close()
    eng0.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
