#!/usr/bin/env python3
"""Differential soak of the batch prover under concurrency (GPU box): several contexts share ONE parameter handle (one
fixed-base table) and keep proving random statements of random batch sizes; a sample of every call's proofs is compared
byte for byte with the CPU oracle's prover (same external-RNG bytes), and every proof is verified by the engine.  Test
infrastructure, not part of the product.

    python tools/soak_prove.py --seconds 60 --threads 3
prints one JSON line: calls, proofs, oracle comparisons, mismatches (must be 0)."""
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
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--threads", type=int, default=3)
    ap.add_argument("--m", type=int, default=2)
    ap.add_argument("--t", type=int, default=2)
    ap.add_argument("--bits", type=int, default=32)
    args = ap.parse_args()
    import numpy as np
    bpp = importlib.import_module("bulletproofs-plus_amd")
    packed = importlib.import_module("bulletproofs-plus_amd.packed")
    from oracle import cport
    label = b"soak-prove"
    n_bits, m, t = args.bits, args.m, args.t
    rounds = (n_bits * m).bit_length() - 1
    eng0 = bpp.Engine(0)
    p0 = bpp.RangeParameters.init(n_bits, m, bpp.create_pedersen_gens_with_extension_degree(t), engine=eng0)
    stop = time.time() + args.seconds
    stats = {"calls": 0, "proofs": 0, "compared": 0, "mismatch": 0}
    lock = threading.Lock()

    def worker(k):
        rng = np.random.default_rng(1000 + k)
        eng = bpp.Engine(0)
        params = p0.share(eng)
        cp = cport.Params(n_bits, m, t)
        while time.time() < stop:
            cnt = int(rng.choice([1, 3, 17, 64, 300]))
            values = rng.integers(0, 1 << (n_bits - 1), size=(cnt, m), dtype=np.uint64)
            bl = rng.integers(0, 256, size=(cnt, m, t, 32), dtype=np.uint8)
            bl[..., 31] &= 0x0f
            bl[..., 0] |= 1
            mins = values // np.uint64(2)
            present = rng.integers(0, 2, size=(cnt, m), dtype=np.uint8)
            mins = mins * present
            ext = rng.integers(0, 256, size=(cnt, 32 * (rounds + 3)), dtype=np.uint8)
            comm = packed.commit(params, values.reshape(-1), bl.reshape(cnt * m, t, 32)).reshape(cnt, m, 32)
            proofs = packed.prove(params, values, bl, comm, mins, present, None, label, ext)
            bad = 0
            for i in sorted(set([0, cnt - 1, int(rng.integers(0, cnt))])):
                want, wc = cp.prove(label, [int(v) for v in values[i]], [[bytes(bl[i, j, q]) for q in range(t)] for j in range(m)],
                                    [int(mins[i, j]) if present[i, j] else None for j in range(m)], None, bytes(ext[i]))
                if want != bytes(proofs[i]) or b"".join(wc) != bytes(comm[i].reshape(-1)):
                    bad += 1
            rb = packed.ResidentBatch(params, proofs, comm, mins, present, None, label)
            try:
                rb.verify_only(chunk=0)
            except bpp.ProofError:
                bad += 1
            rb.close()
            with lock:
                stats["calls"] += 1
                stats["proofs"] += cnt
                stats["compared"] += 3 if cnt > 2 else cnt
                stats["mismatch"] += bad
        cp.close()
        params.close()
        eng.close()

    th = [threading.Thread(target=worker, args=(k,)) for k in range(args.threads)]
    for x in th:
        x.start()
    while any(x.is_alive() for x in th):  # a progress line every 30 s (a silent GPU job is taken to be hung)
        time.sleep(1.0)
        if int(time.time()) % 30 == 0:
            with lock:
                print("progress", json.dumps(stats), file=sys.stderr, flush=True)
    for x in th:
        x.join()
    print(json.dumps(dict(stats, seconds=args.seconds, threads=args.threads, shape={"bits": n_bits, "m": m, "t": t})))
    return 1 if stats["mismatch"] else 0


if __name__ == "__main__":
    sys.exit(main())
