"""GPU parity fuzz: mutated proofs through the whole product path (from_bytes -> C ABI -> HIP kernels) must end exactly
like the oracle's verify on the same bytes: same Ok(masks) or the same ProofError kind, whichever check trips first.
Mirrors the intent of the reference's fuzz target (fuzz/fuzz_targets/proofs.rs) one level up: not only the parser but the
verifier's verdict and its error precedence."""
import pytest

from oracle.pyref import merlin as M
from oracle.pyref import protocol as O
from tests.helpers import Prng, make_batch, sb

pytestmark = pytest.mark.gpu


def _mutations(raw, t, rng, count):
    """deterministic byte-level mutations of one proof: bit flips, field surgery, truncation / extension, header edits"""
    n_fields = (len(raw) - 1) // 32
    out = []
    for k in range(count):
        r = bytearray(raw)
        kind = k % 8
        f = 1 + 32 * (rng.next_u64() % n_fields)
        if kind == 0:  # single bit anywhere
            bit = rng.next_u64() % (8 * len(r))
            r[bit // 8] ^= 1 << (bit % 8)
        elif kind == 1:  # identity / zero field
            r[f:f + 32] = bytes(32)
        elif kind == 2:  # non-canonical field element or scalar
            r[f:f + 32] = b"\xff" * 32
        elif kind == 3:  # a valid member in the wrong place
            g = 1 + 32 * (rng.next_u64() % n_fields)
            r[f:f + 32] = raw[g:g + 32]
        elif kind == 4:  # drop or add whole 32-byte members
            d = 32 * (1 + rng.next_u64() % 3)
            r = r[:-d] if k % 16 == 4 else r + bytes(rng.fill_bytes(d))
        elif kind == 5:  # ragged length
            r = r[:len(r) - 1 - rng.next_u64() % 31]
        elif kind == 6:  # extension-degree byte
            r[0] = rng.next_u64() % 8
        else:  # low bit of a field (sign / parity of an encoding, +-1 on a scalar)
            r[f] ^= 1
        out.append(bytes(r))
    return out


def _product(bpp, c, proofs_bytes, sts, action):
    try:
        proofs = [bpp.RangeProof.from_bytes(b) for b in proofs_bytes]
        masks = bpp.RangeProof.verify_batch(c.transcripts(), sts, proofs, action)
        return ("ok", [m.blindings() if m is not None else None for m in masks])
    except bpp.ProofError as e:
        return ("err", int(e.kind))


def _oracle(c, proofs_bytes, sts, action):
    try:
        proofs = [O.RangeProof.from_bytes(b) for b in proofs_bytes]
        masks = O.verify([M.Transcript(c.label) for _ in proofs], sts, proofs, action)
        return ("ok", [[sb(x) for x in m] if m is not None else None for m in masks])
    except O.ProofError as e:
        return ("err", int(e.kind))


@pytest.mark.parametrize("n,agg,t,seed", [(8, [1, 1, 1], 1, b"fuzz-a"), (4, [2, 1], 2, b"fuzz-b")])
def test_mutated_proofs_end_like_the_oracle(bpp, engine, n, agg, t, seed):
    c = make_batch(bpp, engine, n, agg, t, seed=seed)
    raw = [p.to_bytes() for p in c.o_proofs]
    rng = Prng(seed + b"-mut")
    A = bpp.VerifyAction
    outcomes = set()
    for which in range(len(raw)):
        for k, mutated in enumerate(_mutations(raw[which], t, rng, 24)):
            blobs = list(raw)
            blobs[which] = mutated
            # alternate public / private statements: mask recovery runs on garbage too and must agree
            private = (k % 3 == 0) and all(m == 1 for m in agg)
            action = A.RecoverAndVerify if private else A.VerifyOnly
            got = _product(bpp, c, blobs, c.statements_private if private else c.statements_public, action)
            want = _oracle(c, blobs, c.o_statements_private if private else c.o_statements_public, int(action))
            assert got == want, "proof %d mutation %d: product %r, oracle %r" % (which, k, got, want)
            outcomes.add(want[0] if want[0] == "ok" else want[1])
    # the corpus must actually exercise several verdicts, not only the parser
    assert len(outcomes) >= 3, outcomes
