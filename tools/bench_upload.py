#!/usr/bin/env python3
"""Host-buffer-inclusive rate through the C ABI: every call gets proof / statement bytes in host memory, uploads them
(bpp_batch_upload: parse, pack, page-locked staging, DMA), verifies (bpp_verify_resident, chunk = 1024) and releases the
batch -- what a service that verifies fresh proofs call after call pays.  The ctypes item array is built once (marshalling
is the Python harness, not the product).  Not the headline metric (bench.py keeps its inputs resident)."""
import argparse
import ctypes
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
    ap.add_argument("--batches-per-call", type=int, default=64)
    ap.add_argument("--calls", type=int, default=24)
    ap.add_argument("--threads", default="1,2,4")
    args = ap.parse_args()
    bpp = importlib.import_module("bulletproofs-plus_amd")
    from tests.golden.loader import load_bench
    data = load_bench("bench_cfg2.bin")
    R = args.batches_per_call
    its = data["items"] * R
    n = len(its)
    for S in [int(x) for x in args.threads.split(",")]:
        lanes = []
        for _ in range(S):
            eng = bpp.Engine(0)
            params = bpp.RangeParameters.init(data["bit_length"], data["m"],
                                              bpp.create_pedersen_gens_with_extension_degree(data["t"]), engine=eng)
            sts = [bpp.RangeStatement.init(params, it["commitments"], it["min_values"], None) for it in its]
            proofs = [bpp.RangeProof.from_bytes(it["proof"]) for it in its]
            trs = [bpp.Transcript.new(data["label"]) for _ in its]
            items, keep = bpp.RangeProof._items(trs, sts, proofs)
            lanes.append((eng, params, items, keep))
        split = {"upload": 0.0, "verify": 0.0}

        def worker(slot, calls):
            eng, params, items, _ = lanes[slot]
            lib = eng.lib
            masks = (ctypes.c_uint8 * (n * 32))()
            present = (ctypes.c_uint8 * n)()
            err = ctypes.create_string_buffer(256)
            for _ in range(calls):
                h = ctypes.c_uint64()
                t0 = time.perf_counter()
                assert lib.bpp_batch_upload(eng.ctx, params.handle, items, n, ctypes.byref(h), err, 256) == 0, err.value
                t1 = time.perf_counter()
                assert lib.bpp_verify_resident(eng.ctx, h, 0, 1024, masks, present, err, 256) == 0, err.value
                t2 = time.perf_counter()
                lib.bpp_batch_destroy(eng.ctx, h)
                if slot == 0:
                    split["upload"] += t1 - t0
                    split["verify"] += t2 - t1

        for warm in (True, False):
            calls = 3 if warm else args.calls
            if not warm:
                split["upload"] = split["verify"] = 0.0
            th = [threading.Thread(target=worker, args=(k, calls)) for k in range(S)]
            t0 = time.perf_counter()
            for t in th:
                t.start()
            for t in th:
                t.join()
            dt = time.perf_counter() - t0
        print(json.dumps({"metric": "64-bit range proofs verified/sec, host buffers in", "contexts": S, "proofs_per_call": n,
                          "proofs_per_s": S * args.calls * n / dt, "upload_ms_per_call": 1e3 * split["upload"] / args.calls,
                          "verify_ms_per_call": 1e3 * split["verify"] / args.calls}))
        for eng, _, _, _ in lanes:
            eng.close()


if __name__ == "__main__":
    main()
