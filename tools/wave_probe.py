#!/usr/bin/env python3
"""Sweep of the sharded wave form on ONE rank (a real RCCL communicator of size 1, no torchrun needed): waves x batches per
wave of 4096-proof reference batches through bpp_verify_sharded_wave; prints one JSON line per configuration with the
host-side split of the last wave (bpp_comm_last_timing).  Usage: tools/wave_probe.py "2x8,3x6,4x4" [rounds] [shard]
`shard` (default 4096) is the number of proofs per batch on this rank: 512 is what one of eight ranks holds of a 4096-proof
batch (its kernels, not its weight chain, which runs over all 4096 proofs on every rank)."""
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
    # "WxK": W waves of K batches on K contexts (bpp_verify_sharded_wave); "WgG": W threads, each ONE resident batch of G groups
    # on one context (bpp_verify_sharded_groups)
    # "WpSgG": W threads, each ONE pipelined call over S grouped batches of G groups on S contexts (bpp_verify_sharded_groups_wave)
    def parse(c):
        if "p" in c:
            w, rest = c.split("p")
            sl, g = rest.split("g")
            return (int(w), int(g), "pipe%d" % int(sl))
        if "g" in c:
            return tuple(int(x) for x in c.split("g")) + ("groups",)
        return tuple(int(x) for x in c.split("x")) + ("wave",)
    configs = [parse(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "2x8,3x6,4x4,2x12").split(",")]
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    shard = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
    eng0 = bpp.Engine(0)
    params = bpp.RangeParameters.init(64, 1, bpp.create_pedersen_gens_with_extension_degree(1), engine=eng0)
    data = bench.make_inputs(np, packed, params, 4096 * 8, seed=8675309)
    for W, K, mode in configs:
        waves = []
        for w in range(W):
            S = int(mode[4:]) if mode.startswith("pipe") else 1
            engs = [bpp.Engine(0) for _ in range(K if mode == "wave" else S)]
            pars = [params.share(e) for e in engs]
            rbs = []
            if mode == "wave":
                for i, p in enumerate(pars):
                    sl = slice(((w * K + i) % 8) * 4096, ((w * K + i) % 8) * 4096 + shard)
                    rbs.append(packed.ResidentBatch(p, data["proofs"][sl], data["commitments"][sl], data["min_values"][sl],
                                                    data["min_present"][sl], None, bench.LABEL))
                    rbs[-1].prepare(0)
            else:  # K groups of `shard` proofs in one resident batch (S of them for a pipelined call)
                for sl_i in range(S):
                    idx = np.concatenate([np.arange((((w * S + sl_i) * K + i) % 8) * 4096, (((w * S + sl_i) * K + i) % 8) * 4096 + shard) for i in range(K)])
                    rbs.append(packed.ResidentBatch(pars[sl_i], data["proofs"][idx], data["commitments"][idx], data["min_values"][idx],
                                                    data["min_present"][idx], None, bench.LABEL))
                    rbs[-1].prepare(shard if K > 1 else 0)
            waves.append((engs, pars, rbs, dmod.ShardComm(engs[0], 0, 1, dmod.ShardComm.unique_id())))
        errors = []

        def worker(w, n):
            try:
                for _ in range(n):
                    if mode == "wave":
                        res = waves[w][3].verify_wave(waves[w][2], [shard])
                    elif mode == "groups":
                        res = waves[w][3].verify_groups(waves[w][2][0], K, [shard])
                    else:
                        res = [r for part in waves[w][3].verify_groups_wave(waves[w][2], K, [shard]) for r in part]
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
        print(json.dumps({"form": mode, "waves": W, "batches_per_wave": K, "proofs_per_batch": shard, "proofs_per_s": shard * W * K * (int(mode[4:]) if mode.startswith("pipe") else 1) * rounds / el,
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
