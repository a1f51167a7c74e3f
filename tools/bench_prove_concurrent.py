#!/usr/bin/env python3
"""The batch prover with several calls in flight (own context each, ONE shared parameter handle and fixed-base table):
BASELINE configs[4]'s shape, 1024 x aggregation-4, extension degree 3, per call.  One JSON line per thread count."""
import argparse
import importlib
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", default="1,2,4")
    ap.add_argument("--calls", type=int, default=8)
    ap.add_argument("--m", type=int, default=4)
    ap.add_argument("--t", type=int, default=3)
    args = ap.parse_args()
    import numpy as np
    bpp = importlib.import_module("bulletproofs-plus_amd")
    packed = importlib.import_module("bulletproofs-plus_amd.packed")
    import bench
    eng0 = bpp.Engine(0)
    p0 = bpp.RangeParameters.init(64, args.m, bpp.create_pedersen_gens_with_extension_degree(args.t), engine=eng0)
    d = bench.make_inputs(np, packed, p0, 1024, seed=99)  # also builds the fixed-base table
    for S in [int(x) for x in args.threads.split(",")]:
        engs = [bpp.Engine(0) for _ in range(S)]
        ps = [p0.share(e) for e in engs]

        def worker(k):
            for _ in range(args.calls):
                packed.prove(ps[k], d["values"], d["blindings"], d["commitments"], d["min_values"], d["min_present"], d["seeds"],
                             bench.LABEL, d["ext"])
        for k in range(S):
            worker.__call__  # noqa
        # warm every context (arena, streams)
        for k in range(S):
            packed.prove(ps[k], d["values"], d["blindings"], d["commitments"], d["min_values"], d["min_present"], d["seeds"], bench.LABEL,
                         d["ext"])
        th = [threading.Thread(target=worker, args=(k,)) for k in range(S)]
        t0 = time.perf_counter()
        for x in th:
            x.start()
        for x in th:
            x.join()
        el = time.perf_counter() - t0
        print(json.dumps({"metric": "range proofs created/sec (batch), calls in flight", "contexts": S, "aggregation": args.m,
                          "extension_degree": args.t, "proofs_per_call": 1024, "proofs_per_s": 1024 * args.calls * S / el,
                          "ms_per_call": 1e3 * el / args.calls}))
        for p in ps:
            p.close()
        for e in engs:
            e.close()
    p0.close()
    eng0.close()


if __name__ == "__main__":
    main()
