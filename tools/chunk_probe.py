#!/usr/bin/env python3
"""Throughput of the resident verifier against the size of the reference batch (the reference's own verify_batch cuts its
input into batches of at most 256 proofs, src/range_proof.rs:73-76; BASELINE configs[1] is quoted on 1024): 65 536 proofs
per step, four steps in flight, `chunk` proofs per reference batch.  One JSON line per chunk with the MSM plan it got.

    python tools/chunk_probe.py "256,512,1024,2048,4096" [steps]"""
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
    chunks = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "256,512,1024,2048,4096").split(",")]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    device = torch.device("cuda", 0)
    eng0 = bpp.Engine(0)
    params = bpp.RangeParameters.init(64, 1, bpp.create_pedersen_gens_with_extension_degree(1), engine=eng0)
    data = bench.make_inputs(np, packed, params, 65536, seed=8675309)

    def sync():
        torch.cuda.synchronize(device)
    for chunk in chunks:
        leg = bench.Leg(bpp, packed, torch, device, params, data, chunk, 65536 // chunk, 4, chunk)
        el, lat, profs = bench.timed(leg, steps, 16, sync)
        shape = leg.slots[0][3].shape()
        profs = [p for p in profs if p]
        stages = {k: round(sum(float(p[k]) for p in profs) / len(profs), 3) for k in (profs[0] if profs else {})
                  if isinstance(profs[0][k], (int, float))}
        print(json.dumps({"chunk": chunk, "groups": shape.get("groups"), "proofs_per_s": 65536 * steps / el, "ms_per_step": 1e3 * el / steps,
                          "profile_mean": stages}), flush=True)
        leg.close()
    params.close()
    eng0.close()


if __name__ == "__main__":
    main()
