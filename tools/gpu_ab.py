#!/usr/bin/env python3
"""ONE alternating A/B runner for the GPU box (replaces the family of one-off tools/gpu_*_ab*.sh scripts of rounds 2-6).

  tools/gpu_ab.py --leg LEG [--reps N] [--out FILE] [--args "common arguments"] ARM [ARM ...]

Every ARM is one setting of the comparison, written as a shell-like string of environment assignments and/or arguments of its
own, e.g. "BPP_CT=2 BPP_CT_BACK=1", "--concurrency 4 --steps 20", "BPP_LIB_PATH=gpurun_in/other_build.so" (another build of the
library: bulletproofs-plus_amd/_lib.py loads it INSTEAD of the product's .so, which is never swapped in place), "" (the defaults).
The arms run one after the other, the whole sequence N times (alternating on one box is what makes two arms comparable: boxes
differ by several per cent, and so does one box over minutes).  One summary line per run goes to FILE and to stdout.

Legs (what is run and what the summary shows):
  headline          bench.py --no-extra --no-cpu-baseline --no-traffic         proofs/s, ms per step, clock, host cores, stage times
  headline-again    the headline, then the SAME step as a later leg of the process  both rates and clocks (does the position matter?)
  prover            tools/bench_prover_leg.py (configs[4], one call at a time)  proofs/s, ms per call, MSM event time
  prover-inflight   tools/bench_prove_concurrent.py (1, 2, 4 calls in flight)   proofs/s per number of calls in flight
  latency           tools/bench_latency.py --no-cpu                             median ms per call and size
  cmd               the command given with --cmd                                its last line of output

Examples (the A/Bs on record in HISTORY.md, as they would be run today):
  tools/gpu_ab.py --leg headline --reps 3 "--chain host --concurrency 3" "--chain host-wide --concurrency 4" "--chain device --concurrency 5"
  tools/gpu_ab.py --leg headline --args "--steps 20 --warmup 5" "BPP_WAIT=0" "BPP_WAIT=1"
  tools/gpu_ab.py --leg prover "BPP_CT=1" "BPP_CT=2 BPP_CT_BACK=1" "BPP_CT=2 BPP_CT_BACK=2"
  tools/gpu_ab.py --leg headline "BPP_STATIC_GEMM=0" "BPP_STATIC_GEMM=1"
  tools/gpu_ab.py --leg prover "BPP_PROVE_SUBS=1" "BPP_PROVE_SUBS=2" "BPP_PROVE_SUBS=3"
  tools/gpu_ab.py --leg headline-again "--preheat-ms 150" "--preheat-ms 1000" "--preheat-ms 4000"
"""
import argparse
import json
import os
import shlex
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def split_arm(arm):
    """"A=1 B=2 --flag x" -> ({"A": "1", "B": "2"}, ["--flag", "x"])"""
    env, args = {}, []
    for tok in shlex.split(arm):
        if not args and "=" in tok and not tok.startswith("-") and tok.split("=", 1)[0].replace("_", "").isalnum():
            k, v = tok.split("=", 1)
            env[k] = v
        else:
            args.append(tok)
    return env, args


def last_json(text):
    for line in reversed(text.strip().splitlines()):
        line = line.strip()
        if line.startswith("{"):
            try:
                return json.loads(line)
            except ValueError:
                continue
    return None


def summarise(leg, out):
    if leg == "headline":
        d = last_json(out)
        if not d:
            return "no line"
        st = d.get("stages_ms") or {}
        return ("%.2f M proofs/s  %.3f ms/step  latency %.2f ms  clock %.2f GHz  host cores %.2f  chains %s | %s"
                % (d["value"] / 1e6, d["ms_per_step"], d.get("step_latency_ms", 0), d.get("shader_clock_ghz") or 0, d.get("host_cores_busy") or 0,
                   d.get("weight_chains"), " ".join("%s:%.2f(%d threads, busiest %.2f)" % (e["thread"], e["cores_busy"], e["threads"], e.get("busiest_one", 0)) for e in d.get("host_cores_busy_by_thread") or [])
                   + " | " + " ".join("%s %.3f" % (k[:-3], v) for k, v in st.items() if v)))
    if leg == "headline-again":
        d = last_json(out)
        if not d:
            return "no line"
        o = (d.get("extra") or {}).get("other_chain", {})
        return "headline %.2f M proofs/s at %.3f GHz (%s, %d steps) | as a later leg: %.2f M at %.3f GHz %s" % (
            d["value"] / 1e6, d.get("shader_clock_ghz") or 0, d.get("weight_chains"), d["steps"], o.get("proofs_per_s", 0) / 1e6,
            o.get("shader_clock_ghz") or 0, o.get("error", ""))
    if leg == "prover":
        d = last_json(out)
        if not d:
            return "no line"
        return "%.1f k proofs/s  %.3f ms per call  engine %.3f ms  MSM events %.3f ms" % (d["proofs_per_s"] / 1e3, d["ms_per_call"],
                                                                                         d["engine_total_ms"], d["fb_msm_ms"])
    if leg == "prover-inflight":
        rows = [json.loads(x) for x in out.splitlines() if x.startswith("{")]
        return "  ".join("%d in flight %.1f k" % (r["contexts"], r["proofs_per_s"] / 1e3) for r in rows) or "no line"
    if leg == "latency":
        rows = [json.loads(x) for x in out.splitlines() if x.startswith("{")]
        return "  ".join("%d: %.3f ms" % (r["batch"], r["gpu_ms_median"]) for r in rows) or "no line"
    lines = out.strip().splitlines()
    return lines[-1] if lines else "no output"


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--leg", choices=("headline", "headline-again", "prover", "prover-inflight", "latency", "cmd"), default="headline")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--out", default=os.path.join("gpurun_out", "ab.txt"))
    ap.add_argument("--args", default="", help="arguments every arm gets (after the leg's own)")
    ap.add_argument("--cmd", default="", help="--leg cmd: the command line to run")
    ap.add_argument("--timeout", type=int, default=300, help="seconds per run")
    ap.add_argument("arms", nargs="+")
    a = ap.parse_args()
    base = {"headline": [sys.executable, "bench.py", "--no-extra", "--no-cpu-baseline", "--no-traffic"],
            "headline-again": [sys.executable, "bench.py", "--no-cpu-baseline", "--no-traffic"],
            "prover": [sys.executable, "tools/bench_prover_leg.py"], "prover-inflight": [sys.executable, "tools/bench_prove_concurrent.py"],
            "latency": [sys.executable, "tools/bench_latency.py", "--no-cpu"], "cmd": shlex.split(a.cmd)}[a.leg]
    if not base:
        raise SystemExit("--leg cmd needs --cmd")
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, "w") as f:
        for rep in range(1, a.reps + 1):
            for arm in a.arms:
                env, args = split_arm(arm)
                if a.leg == "headline-again":
                    env = dict({"BPP_BENCH_EXTRA_LEGS": "other_chain", "BPP_BENCH_OTHER_CHAIN": "same"}, **env)
                cmd = base + shlex.split(a.args) + args
                try:
                    r = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, **env), capture_output=True, text=True, timeout=a.timeout)
                    line = summarise(a.leg, r.stdout) if r.returncode == 0 else "FAILED rc %d: %s" % (r.returncode, r.stderr.strip()[-200:])
                except subprocess.TimeoutExpired:  # a GPU step that ran into its limit: nothing further is started
                    line = "TIMEOUT after %d s" % a.timeout
                    print("rep=%d [%s] %s" % (rep, arm, line), flush=True)
                    f.write("rep=%d [%s] %s\n" % (rep, arm, line))
                    raise SystemExit(1)
                text = "rep=%d [%s] %s" % (rep, arm, line)
                print(text, flush=True)
                f.write(text + "\n")
                f.flush()


if __name__ == "__main__":
    main()
