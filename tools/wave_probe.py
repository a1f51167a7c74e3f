#!/usr/bin/env python3
"""Sweep of the sharded wave form on ONE rank (a real RCCL communicator of size 1, no torchrun needed): waves x batches per
wave of 4096-proof reference batches through bpp_verify_sharded_wave; prints one JSON line per configuration with the
host-side split of the last wave (bpp_comm_last_timing).  Usage: tools/wave_probe.py "2x8,3x6,4x4" [rounds]"""
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
    dmod = importlib.import_module("bulletproofs-plus_amd.dist")
    configs = [tuple(int(x) for x in c.split("x")) for c in (sys.argv[1] if len(sys.argv) > 1 else "2x8,3x6,4x4,2x12").split(",")]
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    eng0 = bpp.Engine(0)
    params = bpp.RangeParameters.init(64, 1, bpp.create_pedersen_gens_with_extension_degree(1), engine=eng0)
    data = bench.make_inputs(np, packed, params, 4096 * 8, seed=8675309)
    for W, K in configs:
        waves = []
        for w in range(W):
            engs = [bpp.Engine(0) for _ in range(K)]
            pars = [params.share(e) for e in engs]
            rbs = []
            for i, p in enumerate(pars):
                sl = slice(((w * K + i) % 8) * 4096, ((w * K + i) % 8 + 1) * 4096)
                rbs.append(packed.ResidentBatch(p, data["proofs"][sl], data["commitments"][sl], data["min_values"][sl],
                                                data["min_present"][sl], None, bench.LABEL))
                rbs[-1].prepare(0)
            waves.append((engs, pars, rbs, dmod.ShardComm(engs[0], 0, 1, dmod.ShardComm.unique_id())))
        errors = []

        def worker(w, n):
            try:
                for _ in range(n):
                    res = waves[w][3].verify_wave(waves[w][2], [4096])
                    assert all(r["code"] == 0 for r in res), res
            except BaseException as e:  # noqa: BLE001
                errors.append(e)

        def region(n):
            ths = [threading.Thread(target=worker, args=(w, n)) for w in range(W)]
            t0 = time.perf_counter()
            for th in ths:
                th.start()
            for th in ths:
                th.join()
            if errors:
                raise errors[0]
            return time.perf_counter() - t0
        region(5)
        el = region(rounds)
        print(json.dumps({"waves": W, "batches_per_wave": K, "proofs_per_s": 4096 * W * K * rounds / el,
                          "ms_per_wave": 1e3 * el / rounds, "last_wave_host_ms": {k: round(v, 3) for k, v in waves[0][3].last_timing().items()}}),
              flush=True)
        for engs, pars, rbs, comm in waves:
            comm.close()
            for x in rbs:
                x.close()
            for p in pars:
                p.close()
            for e in engs:
                e.close()
    params.close()
    eng0.close()


if __name__ == "__main__":
    main()
