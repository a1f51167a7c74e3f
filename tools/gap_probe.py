#!/usr/bin/env python3
"""What does the idle gap in front of a SHORT timed region cost?  The headline leg runs 120 untimed steps, the device is
synchronised, the host sleeps `gap` ms, then 20 steps are timed between synchronisations (the driver's form of bench.py).
Five repetitions per gap; also the same 20 steps cut out of a run that never paused (steady state)."""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    import bench
    bpp = importlib.import_module("bulletproofs-plus_amd")
    packed = importlib.import_module("bulletproofs-plus_amd.packed")
    device = torch.device("cuda", 0)
    eng0 = bpp.Engine(0)
    params = bpp.RangeParameters.init(64, 1, bpp.create_pedersen_gens_with_extension_degree(1), engine=eng0)
    data = bench.make_inputs(np, packed, params, 1024 * 64, seed=1)
    S = int(os.environ.get("GAP_INFLIGHT", "4"))
    K = int(os.environ.get("GAP_STEPS", "20"))
    leg = bench.Leg(bpp, packed, torch, device, params, data, 1024, 64, S, 1024, profile=False)
    leg.run_steps(400)
    # GAP_STAGGER_US: worker k of a timed region starts k x this late (do steps that start in phase stay in phase?)
    stagger = float(os.environ.get("GAP_STAGGER_US", "0")) * 1e-6
    one, first = leg.one_step, {}

    def staggered(slot):
        if stagger and first.get(slot):
            first[slot] = False
            time.sleep(slot * stagger)
        return one(slot)
    leg.one_step = staggered
    for gap_ms in [float(x) for x in os.environ.get("GAP_LIST_MS", "0,0.2,0.5,1,2,5,20,100").split(",")]:
        rates = []
        for _ in range(5):
            leg.run_steps(120)
            torch.cuda.synchronize(device)
            if gap_ms:
                time.sleep(gap_ms * 1e-3)
            first.update({k: True for k in range(S)})
            t0 = time.perf_counter()
            leg.run_steps(K)
            torch.cuda.synchronize(device)
            rates.append(K * 65536 / (time.perf_counter() - t0) / 1e6)
        print(json.dumps({"gap_ms": gap_ms, "steps": K, "in_flight": S, "M_proofs_per_s": [round(r, 2) for r in rates], "mean": round(sum(rates) / len(rates), 2)}),
              flush=True)
    t0 = time.perf_counter()
    leg.run_steps(800)
    torch.cuda.synchronize(device)
    print(json.dumps({"steady_state_800_steps_M_proofs_per_s": round(800 * 65536 / (time.perf_counter() - t0) / 1e6, 2)}))
    leg.close()


if __name__ == "__main__":
    main()
