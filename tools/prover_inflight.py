#!/usr/bin/env python3
"""configs[4]'s batch prover with C calls in flight (a context and a host thread each, ONE parameter handle): proofs/s for every
C in PROVER_INFLIGHT (default 1,2,4).  One JSON line per C.  BPP_CT / BPP_CT_BACK choose the form of A1 and B."""
import importlib
import json
import os
import sys
import threading
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import bench
    bpp = importlib.import_module("bulletproofs-plus_amd")
    packed = importlib.import_module("bulletproofs-plus_amd.packed")
    eng0 = bpp.Engine(0)
    p5 = bpp.RangeParameters.init(64, 4, bpp.create_pedersen_gens_with_extension_degree(3), engine=eng0)
    d5 = bench.make_inputs(np, packed, p5, 1024, seed=8675309 + 5)
    iters = int(os.environ.get("PROVER_ITERS", "8"))
    for C in [int(x) for x in os.environ.get("PROVER_INFLIGHT", "1,2,4").split(",")]:
        engs = [bpp.Engine(0) for _ in range(C)]
        pars = [p5.share(e) for e in engs]
        errs = []

        def worker(k, n):
            try:
                for _ in range(n):
                    packed.prove(pars[k], d5["values"], d5["blindings"], d5["commitments"], d5["min_values"], d5["min_present"], None,
                                 bench.LABEL, d5["ext"])
            except BaseException as e:  # noqa: BLE001
                errs.append(e)
        for n in (2, iters):
            ths = [threading.Thread(target=worker, args=(k, n)) for k in range(C)]
            t0 = time.perf_counter()
            for th in ths:
                th.start()
            for th in ths:
                th.join()
            el = time.perf_counter() - t0
            if errs:
                raise errs[0]
        print(json.dumps({"calls_in_flight": C, "proofs_per_s": round(C * 1024 * iters / el), "ms_per_call_and_context": round(1e3 * el / iters, 3),
                          "ct": os.environ.get("BPP_CT", "default"), "ct_back": os.environ.get("BPP_CT_BACK", "default")}), flush=True)
        for q in pars:
            q.close()
        for e in engs:
            e.close()
    p5.close()
    eng0.close()


if __name__ == "__main__":
    main()
