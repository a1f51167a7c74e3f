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
