#!/usr/bin/env python3
"""The prover leg of bench.py alone (configs[4]: 1024 x aggregation-4, extension degree 3): proofs/s one call at a time over 8
calls, k_fb_msm's summed event time and rate per call.  One line."""
import importlib
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import bench
    bpp = importlib.import_module("bulletproofs-plus_amd")
    packed = importlib.import_module("bulletproofs-plus_amd.packed")
    eng = bpp.Engine(0)
    eng.profile(True)
    p5 = bpp.RangeParameters.init(64, 4, bpp.create_pedersen_gens_with_extension_degree(3), engine=eng)
    d5 = bench.make_inputs(np, packed, p5, 1024, seed=8675309 + 5)
    iters = int(os.environ.get("PROVER_ITERS", "8"))
    for _ in range(2):
        packed.prove(p5, d5["values"], d5["blindings"], d5["commitments"], d5["min_values"], d5["min_present"], None, bench.LABEL, d5["ext"])
    t0 = time.perf_counter()
    for _ in range(iters):
        out = packed.prove(p5, d5["values"], d5["blindings"], d5["commitments"], d5["min_values"], d5["min_present"], None, bench.LABEL, d5["ext"])
    el = time.perf_counter() - t0
    pp = eng.last_prove_profile()
    print(json.dumps({"proofs_per_s": round(1024 * iters / el), "ms_per_call": round(1e3 * el / iters, 3), "fb_msm_ms": round(pp["fb_msm_ms"], 3),
                      "fb_G_adds_per_s": round(pp["fb_terms"] * pp["fb_windows"] / (pp["fb_msm_ms"] * 1e-3) / 1e9, 2), "engine_total_ms": round(pp["total_ms"], 3)}))
    p5.close()
    eng.close()


if __name__ == "__main__":
    main()
