#!/usr/bin/env python3
"""Differential soak of the verifier under concurrency (GPU box): several contexts at once keep verifying random sub-batches
of the 1024-proof fixture, about half of them with one random bit of one proof (or commitment) flipped; every verdict is
compared with the CPU oracle's (oracle/c: accept / reject and the error kind) for the same input.  Exercises the kernels in
the mixed, overlapping conditions of the throughput path with inputs no unit test enumerates.  Test infrastructure (uses the
oracle as the checker), not part of the product.

Every call goes through one of the engine's entry paths, picked at random (--paths): the item form (bpp_verify_batch), the
packed blocking form (bpp_verify_batch_packed), the pipelined form (bpp_verify_submit_packed / collect, several tickets
outstanding, collected in random order) and the sharded form over a one-rank RCCL communicator (bpp_verify_sharded: the whole
call is one reference batch) and the grouped sharded form (bpp_verify_sharded_groups: the call cut into equal groups, each its
own reference batch, every group's outcome compared with the oracle's; half of the even splits as two slots of one pipelined
call, bpp_verify_sharded_groups_wave), and through ONE bpp_batcher shared by all threads (the call is pooled with whatever the
other threads hand in at that moment; its verdict must be that of a call of its own).  Most calls are small (half-scalar MSM plan); about one in twenty-five has 1000 or 1500 proofs (full plan with
the latency kernels / the throughput kernels).

    python tools/soak.py --seconds 120 --threads 4
prints one JSON line: calls (per path), rejected inputs, mismatches (must be 0)."""
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
    ap.add_argument("--paths", default="items,packed,pipeline,sharded,groups,batcher")
    args = ap.parse_args()
    import numpy as np
    bpp = importlib.import_module("bulletproofs-plus_amd")
    packed = importlib.import_module("bulletproofs-plus_amd.packed")
    dmod = importlib.import_module("bulletproofs-plus_amd.dist")
    paths = args.paths.split(",")
    from oracle import cport
    from tests.golden.loader import load_bench
    data = load_bench("bench_cfg2.bin")
    items = data["items"]
    n_bits, m, t, label = data["bit_length"], data["m"], data["t"], data["label"]
    stop = time.time() + args.seconds
    stats = {"calls": 0, "rejected": 0, "mismatch": 0, "proofs": 0}
    stats.update({"calls_" + p: 0 for p in paths})
    lock = threading.Lock()
    problems = []

    shared = {}  # the one bpp_batcher all worker threads call into (path "batcher")

    def packed_input(sub, seeds=False):
        n = len(sub)
        proofs = np.frombuffer(b"".join(it["proof"] for it in sub), dtype=np.uint8).reshape(n, -1)
        comm = np.frombuffer(b"".join(it["commitments"][0] for it in sub), dtype=np.uint8).reshape(n, 1, 32)
        mins = np.array([[it["min_values"][0] or 0] for it in sub], dtype=np.uint64)
        pres = np.array([[0 if it["min_values"][0] is None else 1] for it in sub], dtype=np.uint8)
        sn = sp = None
        if seeds:  # mask recovery: the statements' seed nonces travel with the call
            sn = np.frombuffer(b"".join(it["seed_nonce"] or bytes(32) for it in sub), dtype=np.uint8).reshape(n, 32)
            sp = np.array([1 if it["seed_nonce"] else 0 for it in sub], dtype=np.uint8)
        return packed.PackedInput(proofs, comm, mins, pres, sn, label, seed_present=sp)

    def worker(k):
        rng = random.Random(args.seed + k)
        eng = bpp.Engine(0)
        params = bpp.RangeParameters.init(n_bits, m, bpp.create_pedersen_gens_with_extension_degree(t), engine=eng)
        pipe = packed.Pipeline(params, depth=3) if "pipeline" in paths else None
        comm = dmod.ShardComm(eng, 0, 1, dmod.ShardComm.unique_id()) if ("sharded" in paths or "groups" in paths) else None
        eng_b = bpp.Engine(0) if "groups" in paths else None  # second slot of the pipelined grouped form
        params_b = params.share(eng_b) if eng_b else None
        cp = cport.Params(n_bits, m, t)
        pending = []  # pipeline tickets: (ticket, expected code, record)

        def record(path, cnt, chunk, mutated, want, got):
            with lock:
                stats["calls"] += 1
                stats["calls_" + path] += 1
                stats["proofs"] += cnt
                stats["rejected"] += 1 if want else 0
                if got != want:
                    stats["mismatch"] += 1
                    if len(problems) < 5:
                        problems.append({"thread": k, "path": path, "count": cnt, "chunk": chunk, "mutated": mutated, "oracle": want, "engine": got})

        def collect(entry):
            ticket, want, rec = entry
            got = 0
            try:
                pipe.collect(ticket)
            except bpp.ProofError as e:
                got = int(e.kind)
            record(*rec, want, got)

        while time.time() < stop:
            path = rng.choice(paths)
            cnt = rng.choice([1, 2, 3, 7, 16, 64, 200, 256])
            if rng.random() < 0.04:  # now and then a call beyond the half-scalar plan (1000 proofs) and beyond the small-call plan (1500)
                cnt = rng.choice([1000, 1500])
            chunk = 0 if path in ("sharded", "batcher") else rng.choice([0, 0, 8, 64])
            if path == "groups":  # bpp_verify_sharded_groups: `cnt` proofs as equal groups, every group its own reference batch
                groups = rng.choice([g for g in (1, 2, 4, 8, 16) if cnt % g == 0 and cnt // g >= 1] if cnt < 1000 else [1, 2, 4])
                chunk = cnt // groups
            sub = [dict(items[i]) for i in (rng.sample(range(len(items)), cnt) if cnt <= len(items) else rng.choices(range(len(items)), k=cnt))]
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
            # round 4: the packed and the pooled path take every VerifyAction; the masks are compared as well
            action = rng.choice([0, 0, 1, 2]) if path in ("packed", "batcher") else 0
            if action == 0:
                for it in sub:
                    it["seed_nonce"] = None
            # expectation: the oracle verifies each chunk like one reference call, first failing chunk wins
            want, want_masks = 0, []
            step = chunk if chunk else cnt
            for lo in range(0, cnt, step):
                rc, mk, _ = cp.verify(sub[lo:lo + step], action=action)
                if rc != 0:
                    want = rc
                    break
                want_masks += mk
            got = 0
            try:
                if path == "items":
                    sts = [bpp.RangeStatement.init(params, it["commitments"], it["min_values"], None) for it in sub]
                    proofs = [bpp.RangeProof.from_bytes(it["proof"]) for it in sub]
                    trs = [bpp.Transcript.new(label) for _ in sub]
                    bpp.RangeProof.verify_batch(trs, sts, proofs, bpp.VerifyAction.VerifyOnly, chunk=chunk)
                elif path == "packed":
                    mk, pr = packed.verify_batch(params, packed_input(sub, seeds=action != 0), action, chunk)
                    if action and [[bytes(mk[i, 0])] if pr[i] else None for i in range(cnt)] != want_masks:
                        got = -2000  # masks differ from the oracle's
                elif path == "pipeline":
                    ticket = pipe.submit(packed_input(sub), bpp.VerifyAction.VerifyOnly, chunk)  # construction errors raise here
                    pending.append((ticket, want, ("pipeline", cnt, chunk, mutated)))
                    if len(pending) >= 3:
                        collect(pending.pop(rng.randrange(len(pending))))
                    continue
                elif path == "batcher":  # pooled with whatever the other threads are calling at the moment
                    mk, pr = shared["bat"].verify_action(packed_input(sub, seeds=action != 0), action)
                    if action and [[bytes(mk[i, 0])] if pr[i] else None for i in range(cnt)] != want_masks:
                        got = -2000
                    if not action and pr.any():
                        got = -2001  # masks out of a VerifyOnly call
                elif path == "groups":
                    n_groups = cnt // chunk
                    if n_groups % 2 == 0 and rng.random() < 0.5:  # two slots of one pipelined call (bpp_verify_sharded_groups_wave)
                        half = n_groups // 2 * chunk
                        rbs = [packed.ResidentBatch(pp, *[getattr(packed_input(part), a) for a in ("proofs", "commitments", "min_values", "min_present")],
                                                    None, label) for pp, part in ((params, sub[:half]), (params_b, sub[half:]))]
                        try:
                            res = [r for part in comm.verify_groups_wave(rbs, n_groups // 2, [chunk]) for r in part]
                        finally:
                            for rb in rbs:
                                rb.close()
                    else:
                        rb = packed.ResidentBatch(params, *[getattr(packed_input(sub), a) for a in ("proofs", "commitments", "min_values", "min_present")],
                                                  None, label)
                        try:
                            res = comm.verify_groups(rb, n_groups, [chunk])
                        finally:
                            rb.close()
                    # per group the oracle's verdict on that group; the call's outcome for the record: the first failing group's
                    for gi, r in enumerate(res):
                        rc_g, _, _ = cp.verify(sub[gi * chunk:(gi + 1) * chunk], action=0)
                        if r["code"] != rc_g:
                            got = -1000 - gi  # a per-group mismatch shows up as a mismatch of the call
                            break
                    else:
                        got = next((r["code"] for r in res if r["code"] != 0), 0)
                else:
                    rb = packed.ResidentBatch(params, *[getattr(packed_input(sub), a) for a in ("proofs", "commitments", "min_values", "min_present")],
                                              None, label)
                    try:
                        comm.verify(rb, [cnt])
                    finally:
                        rb.close()
            except bpp.ProofError as e:
                got = int(e.kind)
            record(path, cnt, chunk, mutated, want, got)
        while pending:
            collect(pending.pop())
        if comm is not None:
            comm.close()
        if params_b is not None:
            params_b.close()
            eng_b.close()
        params.close()
        eng.close()
        cp.close()

    eng_b = params_b = None
    if "batcher" in paths:
        eng_b = bpp.Engine(0)
        params_b = bpp.RangeParameters.init(n_bits, m, bpp.create_pedersen_gens_with_extension_degree(t), engine=eng_b)
        shared["bat"] = packed.Batcher(params_b, packed_input(items[:1]), lanes=2, max_wait_us=5000)  # (the threads spend most of their time in the oracle: a leader waits for company)
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
    if eng_b is not None:
        stats["batcher"] = shared["bat"].stats()
        shared["bat"].close()
        params_b.close()
        eng_b.close()
    print(json.dumps(dict(stats, seconds=args.seconds, threads=args.threads, problems=problems)))
    return 1 if stats["mismatch"] else 0


if __name__ == "__main__":
    sys.exit(main())
