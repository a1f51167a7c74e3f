#!/usr/bin/env python3
"""How does the verifier's RATE develop after the load starts?  The bench's headline leg (64 x 1024 proofs per step, four steps in
flight) runs for ~4 s without a pause after the GPU idled; every step's completion time is kept and the rate is printed per
100 ms window, with the shader clock of the same window.  Then the same after idle gaps of 5 / 50 / 500 ms.
  python tools/rate_ramp.py [seconds]"""
import importlib
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
    import numpy as np
    import torch
    import bench
    bpp = importlib.import_module("bulletproofs-plus_amd")
    packed = importlib.import_module("bulletproofs-plus_amd.packed")
    device = torch.device("cuda", 0)
    eng0 = bpp.Engine(0)
    eng0.profile(os.environ.get("RAMP_ENG0_PROFILE", "0") == "1")
    params = bpp.RangeParameters.init(64, 1, bpp.create_pedersen_gens_with_extension_degree(1), engine=eng0)
    data = bench.make_inputs(np, packed, params, 1024 * 64, seed=1)
    leg = bench.Leg(bpp, packed, torch, device, params, data, 1024, 64, 4, 1024, profile=int(os.environ.get("RAMP_PROFILE", "0")))
    clk = bpp.Engine(0)
    done = []
    one = leg.one_step

    def stamped(slot):
        r = one(slot)
        done.append(time.perf_counter())
        return r
    leg.one_step = stamped
    for idle_ms, secs in ((int(os.environ.get("RAMP_FIRST_IDLE_MS", "2000")), seconds), (500, 1.5), (50, 1.0), (5, 1.0)):
        torch.cuda.synchronize(device)
        time.sleep(idle_ms * 1e-3)
        del done[:]
        steps = int(secs / 2.5e-3)
        th = threading.Thread(target=lambda: leg.run_steps(steps))
        t0 = time.perf_counter()
        th.start()
        clocks = []
        while th.is_alive():
            t = time.perf_counter() - t0
            clocks.append((t, bpp.shader_clock_ghz(clk, 20000)))
        th.join()
        rows, w = [], 0.1
        k = 0
        while k * w < done[-1] - t0:
            n = sum(1 for d in done if k * w <= d - t0 < (k + 1) * w)
            ck = [c for t, c in clocks if k * w <= t < (k + 1) * w]
            rows.append((round(k * w, 1), round(n * 65536 / w / 1e6, 2), round(sum(ck) / len(ck), 3) if ck else None))
            k += 1
        print(json.dumps({"idle_before_ms": idle_ms, "steps": steps, "overall_M_per_s": round(steps * 65536 / (done[-1] - t0) / 1e6, 2),
                          "window_start_s, M proofs/s, GHz": rows}))
    leg.close()


if __name__ == "__main__":
    main()
