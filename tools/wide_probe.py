#!/usr/bin/env python3
"""Calls in flight from independent callers: S contexts, each verifying ONE reference batch of N proofs per call, as fast as
it can (S host threads).  WIDE_PROBE_N proofs per call (default 4096; 256 = the reference's own batch size), WIDE_PROBE_S list
of S, WIDE_PROBE_HOST=1: every call starts from host buffers (bpp_verify_batch_packed) instead of a resident batch,
WIDE_PROBE_HOST=2: the same calls through ONE bpp_batcher (the library pools the calls that are waiting into grouped engine
calls).  One JSON line per S."""
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
    n = int(os.environ.get("WIDE_PROBE_N", "4096"))
    mode = int(os.environ.get("WIDE_PROBE_HOST", "0"))
    host = mode >= 1
    eng0 = bpp.Engine(0)
    p0 = bpp.RangeParameters.init(64, 1, bpp.create_pedersen_gens_with_extension_degree(1), engine=eng0)
    d = bench.make_inputs(np, packed, p0, n, seed=5)
    inp = packed.PackedInput(d["proofs"], d["commitments"], d["min_values"], d["min_present"], None, bench.LABEL)
    for S in [int(x) for x in os.environ.get("WIDE_PROBE_S", "1,4,6,8").split(",")]:
        engs = [bpp.Engine(0) for _ in range(S if mode < 2 else 0)]
        ps = [p0.share(e) for e in engs]
        bat = packed.Batcher(p0, inp, lanes=int(os.environ.get("WIDE_PROBE_LANES", "0"))) if mode == 2 else None
        rbs = [] if host else [packed.ResidentBatch(ps[k], d["proofs"], d["commitments"], d["min_values"], d["min_present"], None, bench.LABEL)
                               for k in range(S)]

        def call(k):
            if mode == 2:
                bat.verify(inp)
            elif host:
                packed.verify_batch(ps[k], inp, bpp.VerifyAction.VerifyOnly, 0)
            else:
                rbs[k].verify_only(chunk=0)
        for k in range(S):
            for _ in range(5):
                call(k)
        cnt = [0] * S
        stop = time.time() + 2.0

        def w(k):
            while time.time() < stop:
                call(k)
                cnt[k] += 1
        th = [threading.Thread(target=w, args=(k,)) for k in range(S)]
        t0 = time.perf_counter()
        for x in th:
            x.start()
        for x in th:
            x.join()
        el = time.perf_counter() - t0
        print(json.dumps({"proofs_per_call": n, "form": ("resident", "packed", "batcher")[mode], "host_buffers_in": host, "in_flight": S, "calls_per_s": sum(cnt) / el, "proofs_per_s": n * sum(cnt) / el,
                          "ms_per_call_per_context": 1e3 * el * S / max(1, sum(cnt))}), flush=True)
        if bat is not None:
            bat.close()
        for rb in rbs:
            rb.close()
        for p in ps:
            p.close()
        for e in engs:
            e.close()
    p0.close()
    eng0.close()


if __name__ == "__main__":
    main()
