#!/usr/bin/env python3
"""How fast does the shader clock come up under the verifier's load?  Idles the GPU, starts the bench's headline leg and
samples bpp_shader_clock over consecutive 10 ms windows."""
import importlib
import json
import os
import sys
import threading
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
    leg = bench.Leg(bpp, packed, torch, device, params, data, 1024, 64, 4, 1024)
    clk = bpp.Engine(0)
    for idle_ms in (500, 50, 5):
        torch.cuda.synchronize(device)
        time.sleep(idle_ms * 1e-3)
        th = threading.Thread(target=lambda: leg.run_steps(150))
        t0 = time.perf_counter()
        th.start()
        samples = []
        while th.is_alive():
            t = time.perf_counter() - t0
            samples.append((round(1e3 * t, 1), round(bpp.shader_clock_ghz(clk, 10000), 3)))
        th.join()
        print(json.dumps({"idle_before_ms": idle_ms, "ghz_by_ms": samples[:40]}))
    leg.close()


if __name__ == "__main__":
    main()
