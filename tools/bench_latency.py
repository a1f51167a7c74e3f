#!/usr/bin/env python3
"""Latency view (the reference's own Criterion benches time ONE verify_batch call for batch sizes 1..256,
benches/range_proof.rs:206-262): one resident batch of B non-aggregated 64-bit proofs, one call at a time, median wall time
and the per-stage HIP-event times; next to it the oracle/c port on one core for the same batch.  Not the headline metric."""
import argparse
import importlib
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="1,4,16,64,256,1024")
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()
    bpp = importlib.import_module("bulletproofs-plus_amd")
    from tests.golden.loader import load_bench
    data = load_bench("bench_cfg2.bin")
    eng = bpp.Engine(0)
    eng.profile(True)
    params = bpp.RangeParameters.init(data["bit_length"], data["m"], bpp.create_pedersen_gens_with_extension_degree(data["t"]),
                                      engine=eng)
    cp = None
    if not args.no_cpu:
        from oracle import cport
        cp = cport.Params(data["bit_length"], data["m"], data["t"])
    for B in [int(x) for x in args.sizes.split(",")]:
        its = (data["items"] * ((B + len(data["items"]) - 1) // len(data["items"])))[:B]  # sizes above the fixture: repeated proofs
        sts = [bpp.RangeStatement.init(params, it["commitments"], it["min_values"], None) for it in its]
        proofs = [bpp.RangeProof.from_bytes(it["proof"]) for it in its]
        rb = bpp.ResidentBatch([bpp.Transcript.new(data["label"]) for _ in its], sts, proofs)
        for _ in range(20):
            rb.verify(bpp.VerifyAction.VerifyOnly, chunk=0)
        lat, prof = [], {}
        for _ in range(args.iters):
            t0 = time.perf_counter()
            rb.verify(bpp.VerifyAction.VerifyOnly, chunk=0)
            lat.append(time.perf_counter() - t0)
            for k, v in eng.last_profile().items():
                prof[k] = prof.get(k, 0.0) + v
        # the same without per-stage events (what a caller sees)
        eng.profile(False)
        for _ in range(20):
            rb.verify_only(chunk=0)
        plain = []
        for _ in range(args.iters):
            t0 = time.perf_counter()
            rb.verify_only(chunk=0)
            plain.append(time.perf_counter() - t0)
        eng.profile(True)
        out = {"batch": B, "gpu_ms_median": 1e3 * statistics.median(plain), "gpu_ms_min": 1e3 * min(plain),
               "gpu_ms_median_profiled": 1e3 * statistics.median(lat),
               "stages_ms": {k: round(v / args.iters, 4) for k, v in prof.items() if k.endswith("_ms")}}
        if cp is not None:
            rc, sec1 = cp.verify_timed(its, B, 1)
            iters = max(1, min(200, int(1.0 / max(sec1, 1e-4))))
            rc, sec = cp.verify_timed(its, B, iters)
            assert rc == 0
            out["cpu_port_ms_1core"] = 1e3 * sec / iters
        print(json.dumps(out))
        rb.close()
    if cp is not None:
        cp.close()
    eng.close()


if __name__ == "__main__":
    main()
