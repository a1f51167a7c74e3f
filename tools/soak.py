#!/usr/bin/env python3
"""Differential soak of the verifier under concurrency (GPU box): several contexts at once keep verifying random sub-batches
of the 1024-proof fixture, about half of them with one random bit of one proof (or commitment) flipped; every verdict is
compared with the CPU oracle's (oracle/c: accept / reject and the error kind) for the same input.  Exercises the kernels in
the mixed, overlapping conditions of the throughput path with inputs no unit test enumerates.  Test infrastructure (uses the
oracle as the checker), not part of the product.

    python tools/soak.py --seconds 120 --threads 4
prints one JSON line: calls, rejected inputs, mismatches (must be 0)."""
import argparse
import importlib
import json
import os
import random
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--threads", type=int, default=4)
    ap.add_argument("--seed", type=int, default=20260704)
    args = ap.parse_args()
    bpp = importlib.import_module("bulletproofs-plus_amd")
    from oracle import cport
    from tests.golden.loader import load_bench
    data = load_bench("bench_cfg2.bin")
    items = data["items"]
    n_bits, m, t, label = data["bit_length"], data["m"], data["t"], data["label"]
    stop = time.time() + args.seconds
    stats = {"calls": 0, "rejected": 0, "mismatch": 0, "proofs": 0}
    lock = threading.Lock()
    problems = []

    def worker(k):
        rng = random.Random(args.seed + k)
        eng = bpp.Engine(0)
        params = bpp.RangeParameters.init(n_bits, m, bpp.create_pedersen_gens_with_extension_degree(t), engine=eng)
        cp = cport.Params(n_bits, m, t)
        while time.time() < stop:
            cnt = rng.choice([1, 2, 3, 7, 16, 64, 200, 256])
            chunk = rng.choice([0, 0, 8, 64])
            sub = [dict(items[i]) for i in rng.sample(range(len(items)), cnt)]
            mutated = rng.random() < 0.5
            if mutated:
                j = rng.randrange(cnt)
                if rng.random() < 0.85:
                    b = bytearray(sub[j]["proof"])
                    pos = rng.randrange(1, len(b))  # byte 0 is the extension degree: its own error path, covered by unit tests
                    b[pos] ^= 1 << rng.randrange(8)
                    sub[j]["proof"] = bytes(b)
                else:
                    c = bytearray(sub[j]["commitments"][0])
                    c[rng.randrange(32)] ^= 1 << rng.randrange(8)
                    sub[j]["commitments"] = [bytes(c)]
            # expectation: the oracle verifies each chunk like one reference call, first failing chunk wins
            want = 0
            step = chunk if chunk else cnt
            for lo in range(0, cnt, step):
                rc, _, _ = cp.verify(sub[lo:lo + step], action=0)
                if rc != 0:
                    want = rc
                    break
            got = 0
            try:
                sts = [bpp.RangeStatement.init(params, it["commitments"], it["min_values"], None) for it in sub]
                proofs = [bpp.RangeProof.from_bytes(it["proof"]) for it in sub]
                trs = [bpp.Transcript.new(label) for _ in sub]
                bpp.RangeProof.verify_batch(trs, sts, proofs, bpp.VerifyAction.VerifyOnly, chunk=chunk)
            except bpp.ProofError as e:
                got = int(e.kind)
            with lock:
                stats["calls"] += 1
                stats["proofs"] += cnt
                stats["rejected"] += 1 if want else 0
                if got != want:
                    stats["mismatch"] += 1
                    if len(problems) < 5:
                        problems.append({"thread": k, "count": cnt, "chunk": chunk, "mutated": mutated, "oracle": want, "engine": got})
        cp.close()

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
    print(json.dumps(dict(stats, seconds=args.seconds, threads=args.threads, problems=problems)))
    return 1 if stats["mismatch"] else 0


if __name__ == "__main__":
    sys.exit(main())
