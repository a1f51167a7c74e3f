#!/usr/bin/env python3
"""Secondary benchmark: the batch prover on BASELINE configs[4] (1024 x aggregation-4, extension degree 3, 64-bit) and on
the m=1 shape of configs[1].  Prints one JSON line per shape: GPU proofs/s (bpp_prove_batch, host buffers in, proof bytes
out) and the oracle/c port's single-thread rate on a bounded sample.  Not the headline metric (that is bench.py)."""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--count", type=int, default=1024)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--cpu-sample", type=int, default=16)
    args = ap.parse_args()
    bpp = importlib.import_module("bulletproofs-plus_amd")
    from oracle import cport
    from oracle.pyref import protocol as O
    from tests.helpers import LABEL, Prng, sb
    eng = bpp.Engine(0)
    for (n, m, t) in [(64, 4, 3), (64, 1, 1), (64, 8, 1)]:
        params = bpp.RangeParameters.init(n, m, bpp.create_pedersen_gens_with_extension_degree(t), engine=eng)
        rng = Prng(b"bench-prove-%d-%d" % (m, t))
        rounds = (n * m).bit_length() - 1
        vals, blinds, mins, seeds, exts = [], [], [], [], []
        for _ in range(args.count):
            v = [rng.next_u64() % (1 << 63) for _ in range(m)]
            b = [sb(O.random_not_zero(rng))] * t
            vals.append(v)
            blinds.append([b] * m)
            mins.append([x // 3 for x in v])
            seeds.append(sb(O.random_not_zero(rng)) if m == 1 else None)
            exts.append(rng.fill_bytes(32 * (rounds + 3)))
        comms = params.commit_many([x for v in vals for x in v], [x for b in blinds for x in b])
        comms = [comms[i * m:(i + 1) * m] for i in range(args.count)]
        sts = [bpp.RangeStatement.init(params, comms[i], mins[i], seeds[i]) for i in range(args.count)]
        wits = [bpp.RangeWitness.init([bpp.CommitmentOpening.new(vals[i][j], blinds[i][j]) for j in range(m)])
                for i in range(args.count)]
        trs = [bpp.Transcript.new(LABEL)] * args.count
        proofs = bpp.RangeProof.prove_batch(trs, sts, wits, exts)  # warm-up: builds the fixed-base tables
        t0 = time.perf_counter()
        for _ in range(args.iters):
            proofs = bpp.RangeProof.prove_batch(trs, sts, wits, exts)
        gpu = args.count * args.iters / (time.perf_counter() - t0)
        # the C-ABI call alone (what a compiled caller sees): marshalling and RangeProof parsing are Python's
        marshalled = bpp.RangeProof._prove_marshal(trs, sts, wits, exts)
        t0 = time.perf_counter()
        for _ in range(args.iters):
            bpp.RangeProof._prove_call(marshalled, parse=False)
        gpu_call = args.count * args.iters / (time.perf_counter() - t0)
        cp = cport.Params(n, m, t)
        k = min(args.cpu_sample, args.count)
        t0 = time.perf_counter()
        for i in range(k):
            want, _ = cp.prove(LABEL, vals[i], blinds[i], mins[i], seeds[i], exts[i])
            assert want == proofs[i].to_bytes()
        cpu = k / (time.perf_counter() - t0)
        cp.close()
        assert bpp.RangeProof.verify_batch(trs[:64], sts[:64], proofs[:64], bpp.VerifyAction.VerifyOnly) == [None] * 64
        print(json.dumps({"metric": "range proofs created/sec (batch)", "shape": {"bit_length": n, "aggregation": m,
                          "extension_degree": t, "batch": args.count}, "gpu_proofs_per_s": gpu_call, "gpu_proofs_per_s_incl_python": gpu,
                          "cpu_port_proofs_per_s_1core": cpu, "cpu_sample": k, "bytes_equal_oracle": True}))
    eng.close()


if __name__ == "__main__":
    main()
