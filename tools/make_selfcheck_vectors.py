#!/usr/bin/env python3
"""Exercises tests/test_ref_golden.py without the reference: writes a file in the schema of rust/ref-dump's output, but
made by the ORACLE (so it pins nothing -- it only proves that the consumer test runs end to end).  Never commit its output
as tests/golden/ref_vectors.json; use:  BPP_REF_VECTORS=/tmp/selfcheck.json python -m pytest tests/test_ref_golden.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.pyref import curve as C  # noqa: E402
from oracle.pyref import merlin as M  # noqa: E402
from oracle.pyref import protocol as O  # noqa: E402
from tests.helpers import Prng  # noqa: E402


class Rec:
    def __init__(self, inner):
        self.inner, self.log = inner, b""

    def fill_bytes(self, n):
        b = self.inner.fill_bytes(n)
        self.log += b
        return b


def res(fn):
    try:
        return {"ok": [None if m is None else [C.scalar_bytes(x).hex() for x in m] for m in fn()]}
    except O.ProofError as e:
        return {"err": int(e.kind), "msg": e.msg}


def case(name, n, batch, t, strategy):
    rng, label = Prng(name.encode()), b"BatchedRangeProofTest"
    items, priv, pub, proofs = [], [], [], []
    for m in batch:
        params = O.RangeParameters(n, m, O.PedersenGens(t))
        vals = [rng.next_u64() % (1 << (n - 1)) for _ in range(m)]
        mins = [{"none": None, "third": v // 3, "eq": v}[strategy] for v in vals]
        blinds = [[O.random_not_zero(rng)] * t for _ in range(m)]
        comms = [params.pc_gens.commit(v, b) for v, b in zip(vals, blinds)]
        seed = O.random_not_zero(rng) if m == 1 else None
        sp, su = O.RangeStatement(params, comms, mins, seed), O.RangeStatement(params, comms, mins, None)
        rec = Rec(rng)
        proof = O.prove_with_rng(M.Transcript(label), sp, O.RangeWitness([O.CommitmentOpening(v, b) for v, b in zip(vals, blinds)]), rec)
        items.append({"m": m, "values": vals, "blindings": [[C.scalar_bytes(x).hex() for x in b] for b in blinds], "min_values": mins,
                      "seed_nonce": C.scalar_bytes(seed).hex() if seed is not None else None,
                      "commitments": [c.compress().hex() for c in comms], "rng_bytes": rec.log.hex(), "proof": proof.to_bytes().hex()})
        priv.append(sp)
        pub.append(su)
        proofs.append(proof)
    T = lambda: [M.Transcript(label) for _ in proofs]
    wrong = [O.RangeStatement(s.generators, s.commitments, s.minimum_value_promises, (s.seed_nonce + 1) % C.L if s.seed_nonce is not None else None) for s in priv]
    bumped = [O.RangeStatement(s.generators, s.commitments, [(v + 1 if v is not None else 1) for v in s.minimum_value_promises], None) for s in pub]
    return {"name": name, "bit_length": n, "aggregation": batch, "extension_degree": t, "label": label.decode(), "items": items,
            "verify": {"private_recover_only": res(lambda: O.verify_batch(T(), priv, proofs, 2)),
                       "private_recover_and_verify": res(lambda: O.verify_batch(T(), priv, proofs, 1)),
                       "private_verify_only": res(lambda: O.verify_batch(T(), priv, proofs, 0)),
                       "public_verify_only": res(lambda: O.verify_batch(T(), pub, proofs, 0)),
                       "wrong_seed_recover_and_verify": res(lambda: O.verify_batch(T(), wrong, proofs, 1)),
                       "bumped_promise_verify_only": res(lambda: O.verify_batch(T(), bumped, proofs, 0))}}


def build_doc():
    cases = [case("single_n8_t1", 8, [1], 1, "none"), case("aggregated4_n4_t2", 4, [4], 2, "third"), case("mixed_1_2_n8_t3", 8, [1, 2], 3, "eq")]
    p = O.RangeParameters(64, 2, O.PedersenGens(6))
    anchors = {"h_base": p.pc_gens.h_base_compressed.hex(), "g_bases": [g.hex() for g in p.pc_gens.g_base_compressed_vec],
               "gi": [g.compress().hex() for g in p.gi_base()], "hi": [g.compress().hex() for g in p.hi_base()]}
    return {"source": "ORACLE self-check (not the reference)", "cases": cases, "anchors_n64_m2_t6": anchors}


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/selfcheck_ref_vectors.json"
    json.dump(build_doc(), open(out, "w"))
    print("wrote", out)
