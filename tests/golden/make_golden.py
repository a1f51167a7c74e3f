#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/ (run in the build container only).

* kat_libsodium.json  -- ristretto255 vectors computed by libsodium 1.0.18 (/opt/conda/lib/libsodium.so): an
                         implementation independent of everything in this repo.  from_hash, scalarmult, add.
* protocol_small.json -- proofs + every verifier intermediate from oracle/pyref for small shapes (regression pin).
* bench_cfg2.bin / bench_cfg3.bin -- BASELINE.json configs[1] / configs[2] inputs (1024 x m=1, 256 x m=8 64-bit proofs)
                         produced by the C oracle's prover with the bench recipe of benches/range_proof.rs:206-262.
Format of *.bin: see tests/golden/loader.py.
"""
import ctypes
import hashlib
import json
import os
import struct
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import cport  # noqa: E402
from oracle.pyref import curve as C  # noqa: E402
from oracle.pyref import protocol as O  # noqa: E402
from tests.helpers import Prng, make_oracle_batch, oracle_verify_trace, sb  # noqa: E402


def libsodium_kats():
    so = ctypes.CDLL("/opt/conda/lib/libsodium.so")
    so.sodium_init()
    out = {"from_hash": [], "scalarmult": [], "add": []}
    prev = None
    for i in range(24):
        h = hashlib.sha512(b"bpp-kat-%d" % i).digest()
        p = ctypes.create_string_buffer(32)
        so.crypto_core_ristretto255_from_hash(p, h)
        out["from_hash"].append([h.hex(), p.raw.hex()])
        k = (int.from_bytes(hashlib.sha256(b"k%d" % i).digest(), "little") % C.L).to_bytes(32, "little")
        q = ctypes.create_string_buffer(32)
        assert so.crypto_scalarmult_ristretto255(q, k, p.raw) == 0
        out["scalarmult"].append([k.hex(), p.raw.hex(), q.raw.hex()])
        if prev is not None:
            s = ctypes.create_string_buffer(32)
            so.crypto_core_ristretto255_add(s, prev, p.raw)
            out["add"].append([prev.hex(), p.raw.hex(), s.raw.hex()])
        prev = p.raw
    return out


def protocol_small():
    cases = []
    for (n, batch, t, strat) in [(8, [1], 1, "none"), (8, [2, 1], 2, "third"), (4, [1, 2, 4], 3, "eq"), (64, [1, 2], 1, "third")]:
        c = make_oracle_batch(n, batch, t, seed=b"golden-%d-%d" % (n, t), strategy=strat)
        masks, tr = oracle_verify_trace(c, action=1)
        cases.append({
            "bit_length": n, "aggregation": batch, "extension_degree": t, "label": c.label.decode(),
            "proofs": [p.to_bytes().hex() for p in c.o_proofs],
            "commitments": [[x.hex() for x in s.commitments_compressed] for s in c.o_statements_private],
            "min_values": [s.minimum_value_promises for s in c.o_statements_private],
            "seed_nonces": [sb(s.seed_nonce).hex() if s.seed_nonce is not None else None for s in c.o_statements_private],
            "masks": [[x.hex() for x in m] if m else None for m in masks],
            "challenges": [[sb(y).hex(), sb(z).hex(), [sb(e).hex() for e in r], sb(ef).hex()] for (y, z, r, ef) in tr["challenges"]],
            "rng_outputs": [x.hex() for x in tr["rng_outputs"]],
            "weights": [sb(x).hex() for x in tr["weights"]],
            "gi": [sb(x).hex() for x in tr["gi"]], "hi": [sb(x).hex() for x in tr["hi"]],
            "g": [sb(x).hex() for x in tr["g"]], "h": sb(tr["h"]).hex(),
            "dynamic_scalars": [sb(x).hex() for x in tr["dynamic_scalars"]],
        })
    og = O.BulletproofGens(64, 2)
    anchors = {
        "masking_basepoints": [p.compress().hex() for p in O.ristretto_masking_basepoints()],
        "G[0][0]": og.g_vec[0][0].compress().hex(), "H[0][0]": og.h_vec[0][0].compress().hex(),
        "G[0][63]": og.g_vec[0][63].compress().hex(), "H[0][63]": og.h_vec[0][63].compress().hex(),
        "G[1][0]": og.g_vec[1][0].compress().hex(), "H[1][0]": og.h_vec[1][0].compress().hex(),
        "nonce(1,alpha,None,0)": sb(O.nonce(1, "alpha", None, 0)).hex(),
        "nonce(1,dL,3,2)": sb(O.nonce(1, "dL", 3, 2)).hex(),
        "nonce(1,eta,None,None)": sb(O.nonce(1, "eta", None, None)).hex(),
    }
    return {"cases": cases, "anchors": anchors}


def bench_file(path, n_proofs, m, t, seed):
    """benches/range_proof.rs:206-262 recipe: v = next_u64 % 2^63, min = v/3, one blinding repeated t times,
    seed nonce iff m == 1, label BatchedRangeProofTest."""
    label = b"BatchedRangeProofTest"
    rng = Prng(seed)
    cp = cport.Params(64, m, t)
    rounds = (64 * m).bit_length() - 1
    with open(path, "wb") as f:
        f.write(struct.pack("<4sIIIII", b"BPPB", 1, n_proofs, 64, m, t))
        f.write(struct.pack("<I", len(label)) + label)
        for _ in range(n_proofs):
            vals, blinds, mins = [], [], []
            for _j in range(m):
                v = rng.next_u64() % (1 << 63)
                vals.append(v)
                mins.append(v // 3)
                blinds.append([sb(O.random_not_zero(rng))] * t)
            sn = sb(O.random_not_zero(rng)) if m == 1 else None
            proof, comm = cp.prove(label, vals, blinds, mins, sn, rng.fill_bytes(32 * (rounds + 3)))
            f.write(struct.pack("<I", len(proof)) + proof + b"".join(comm))
            f.write(b"".join(struct.pack("<Q", x) for x in mins))
            f.write(b"\x01" + sn if sn else b"\x00" + bytes(32))
    cp.close()


if __name__ == "__main__":
    json.dump(libsodium_kats(), open(os.path.join(HERE, "kat_libsodium.json"), "w"), indent=0)
    json.dump(protocol_small(), open(os.path.join(HERE, "protocol_small.json"), "w"), indent=0)
    bench_file(os.path.join(HERE, "bench_cfg2.bin"), 1024, 1, 1, b"8675309-cfg2")
    bench_file(os.path.join(HERE, "bench_cfg3.bin"), 256, 8, 1, b"8675309-cfg3")
    print("fixtures written")
