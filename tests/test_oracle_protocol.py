"""CPU suite: the oracle re-states every property the reference's own tests assert, against the fixtures."""
import json
import os

import pytest

from oracle.pyref import curve as C
from oracle.pyref import merlin as M
from oracle.pyref import protocol as O
from tests.helpers import make_oracle_batch, oracle_verify_trace, sb

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _case_from_fixture(cs):
    params = O.RangeParameters(cs["bit_length"], max(cs["aggregation"]), O.PedersenGens(cs["extension_degree"]))
    sts, proofs = [], []
    for comm, mins, sn, pr in zip(cs["commitments"], cs["min_values"], cs["seed_nonces"], cs["proofs"]):
        pts = [C.decompress(bytes.fromhex(x)) for x in comm]
        sts.append(O.RangeStatement(params, pts, mins, int.from_bytes(bytes.fromhex(sn), "little") if sn else None))
        proofs.append(O.RangeProof.from_bytes(bytes.fromhex(pr)))
    return sts, proofs, cs["label"].encode()


def test_fixture_proofs_verify_and_intermediates_are_stable():
    for cs in json.load(open(os.path.join(GOLD, "protocol_small.json")))["cases"]:
        sts, proofs, label = _case_from_fixture(cs)
        tr = {}
        masks = O.verify([M.Transcript(label) for _ in proofs], sts, proofs, O.RECOVER_AND_VERIFY, trace=tr)
        assert [[sb(x).hex() for x in m] if m else None for m in masks] == cs["masks"]
        assert [sb(x).hex() for x in tr["weights"]] == cs["weights"]
        assert [sb(x).hex() for x in tr["gi"]] == cs["gi"] and [sb(x).hex() for x in tr["hi"]] == cs["hi"]
        assert [sb(x).hex() for x in tr["dynamic_scalars"]] == cs["dynamic_scalars"]
        assert [x.hex() for x in tr["rng_outputs"]] == cs["rng_outputs"]


@pytest.mark.parametrize("n,batch,t,strategy", [(8, [1], 1, "none"), (4, [4], 2, "third"), (8, [1, 2], 3, "eq")])
def test_prove_and_verify_properties(n, batch, t, strategy):
    """tests/ristretto.rs:152-373 against the oracle itself"""
    c = make_oracle_batch(n, batch, t, seed=b"orc-%d" % n, strategy=strategy)
    tr = lambda: [M.Transcript(c.label) for _ in c.o_proofs]
    want = [m for m in c.expected_masks]
    assert O.verify_batch(tr(), c.o_statements_private, c.o_proofs, O.RECOVER_ONLY) == want
    assert O.verify_batch(tr(), c.o_statements_private, c.o_proofs, O.RECOVER_AND_VERIFY) == want
    assert O.verify_batch(tr(), c.o_statements_public, c.o_proofs, O.VERIFY_ONLY) == [None] * len(batch)
    bumped = [O.RangeStatement(s.generators, s.commitments, [(v + 1 if v is not None else 1) for v in s.minimum_value_promises], None)
              for s in c.o_statements_public]
    with pytest.raises(O.ProofError) as e:
        O.verify_batch(tr(), bumped, c.o_proofs, O.VERIFY_ONLY)
    assert e.value.kind == O.VERIFICATION_FAILED
    for p in c.o_proofs:
        assert O.RangeProof.from_bytes(p.to_bytes()) == p


def test_prover_rejects_min_value_above_value():
    params = O.RangeParameters(8, 1, O.PedersenGens(1))
    st = O.RangeStatement(params, [params.pc_gens.commit(5, [7])], [6], None)
    with pytest.raises(O.ProofError) as e:
        O.prove_with_rng(M.Transcript(b"x"), st, O.RangeWitness([O.CommitmentOpening(5, [7])]), M.NullRng())
    assert e.value.kind == O.INVALID_ARGUMENT


def test_from_bytes_errors():
    """src/range_proof.rs:1339-1435"""
    c = make_oracle_batch(4, [1], 1, seed=b"ser")
    raw = c.o_proofs[0].to_bytes()
    for bad, kind in [(b"", O.INVALID_LENGTH), (raw[:-1], O.INVALID_LENGTH), (raw + b"\0", O.INVALID_LENGTH),
                      (raw + bytes(32), O.INVALID_LENGTH), (b"\x07" + raw[1:], O.INVALID_ARGUMENT),
                      (b"\x00" + raw[1:], O.INVALID_ARGUMENT), (raw[:1 + 32 * 6], O.INVALID_LENGTH),
                      (raw[:1] + b"\xff" * 32 + raw[33:], O.INVALID_ARGUMENT)]:
        with pytest.raises(O.ProofError) as e:
            O.RangeProof.from_bytes(bad)
        assert e.value.kind == kind


def test_nonce_domain_separation():
    """src/utils/generic.rs:107-199"""
    s = 12345
    vals = {O.nonce(s, "test", None, None), O.nonce(s, "test", 0, None), O.nonce(s, "test", None, 0),
            O.nonce(s, "test", 0, 0), O.nonce(s, "test", 1, 0), O.nonce(s, "test", 0, 1), O.nonce(s + 1, "test", 0, 1),
            O.nonce(s, "tesu", 0, 1)}
    assert len(vals) == 8
    O.nonce(s, "a" * 16, None, None)
    with pytest.raises(O.ProofError):
        O.nonce(s, "a" * 17, None, None)
    assert O.compute_generator_padding(64, 1, 4) == 384 and O.compute_generator_padding(64, 4, 4) == 0
