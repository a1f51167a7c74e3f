#!/usr/bin/env python3
"""Throughput of the resident verifier against the aggregation factor, with the generator columns by Montgomery products per
(proof, generator), by products summed per workgroup under one reduction, and as a matrix product over the proofs of a group on
the matrix cores: 64 reference batches of 256 aggregation-m proofs per step, three steps in flight.  One JSON line per (m, form).

    python tools/agg_probe.py "1,2,4,8" [steps]"""
import importlib
import json
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    import bench
    bpp = importlib.import_module("bulletproofs-plus_amd")
    packed = importlib.import_module("bulletproofs-plus_amd.packed")
    ms = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,2,4,8").split(",")]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 48
    device = torch.device("cuda", 0)
    eng0 = bpp.Engine(0)

    def sync():
        torch.cuda.synchronize(device)
    for m in ms:
        params = bpp.RangeParameters.init(64, m, bpp.create_pedersen_gens_with_extension_degree(1), engine=eng0)
        data = bench.make_inputs(np, packed, params, 256 * 64, seed=8675309 + m)
        for name, env in (("per-proof products", {"BPP_STATIC_GEMM": "0", "BPP_LAZY_COLUMNS": "0"}),
                          ("one reduction per workgroup", {"BPP_STATIC_GEMM": "0", "BPP_LAZY_COLUMNS": "1"}),
                          ("matrix product", {"BPP_STATIC_GEMM": "1", "BPP_LAZY_COLUMNS": "1"})):
            os.environ.update(env)  # read when the slots' contexts are created
            leg = bench.Leg(bpp, packed, torch, device, params, data, 256, 64, 3, 256)
            el, lat, profs = bench.timed(leg, steps, 8, sync)
            print(json.dumps({"m": m, "columns": name, "proofs_per_s": round(256 * 64 * steps / el), "ms_per_step": round(1e3 * el / steps, 4)}),
                  flush=True)
            leg.close()
        params.close()
    eng0.close()


if __name__ == "__main__":
    main()
