#!/usr/bin/env python3
"""Where the wall time of one bpp_prove_batch call goes on the host (configs[4]): Python marshalling, the C call as a whole, the
engine's own interval (first enqueue -> last stream drained)."""
import ctypes
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
    d = bench.make_inputs(np, packed, p5, 1024, seed=8675309 + 5)
    args = (p5, d["values"], d["blindings"], d["commitments"], d["min_values"], d["min_present"], None, bench.LABEL, d["ext"])
    real = eng.lib.bpp_prove_batch
    spans = []

    class Wrap:
        def __call__(self, *a):
            t0 = time.perf_counter()
            rc = real(*a)
            spans.append(time.perf_counter() - t0)
            return rc
    eng.lib.bpp_prove_batch = Wrap()
    for _ in range(3):
        packed.prove(*args)
    spans.clear()
    tot, eng_ms = [], []
    for _ in range(8):
        t0 = time.perf_counter()
        packed.prove(*args)
        tot.append(time.perf_counter() - t0)
        eng_ms.append(eng.last_prove_profile()["total_ms"])
    med = lambda v: sorted(v)[len(v) // 2]
    print(json.dumps({"python_call_ms": round(1e3 * med(tot), 3), "c_call_ms": round(1e3 * med(spans), 3), "engine_interval_ms": round(med(eng_ms), 3),
                      "python_marshalling_ms": round(1e3 * (med(tot) - med(spans)), 3), "c_host_outside_engine_interval_ms": round(1e3 * med(spans) - med(eng_ms), 3)}))


if __name__ == "__main__":
    main()
