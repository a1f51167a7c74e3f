"""CPU suite: the plain-C port (oracle/c, the cpu_baseline) against pyref and the fixtures."""
import ctypes
import hashlib

import pytest

from oracle import cport
from oracle.pyref import curve as C
from oracle.pyref import merlin as M
from oracle.pyref import protocol as O
from tests.golden.loader import load_bench
from tests.helpers import make_oracle_batch, oracle_verify_trace, sb, trace_challenge_bytes


def _items(c):
    return [dict(proof=p.to_bytes(), commitments=s.commitments_compressed, min_values=s.minimum_value_promises,
                 seed_nonce=sb(s.seed_nonce) if s.seed_nonce is not None else None, label=c.label)
            for p, s in zip(c.o_proofs, c.o_statements_private)]


def test_c_primitives():
    L = cport.lib()
    for i in range(16):
        u = hashlib.shake_256(b"u%d" % i).digest(64)
        o = (ctypes.c_uint8 * 32)()
        L.oracle_from_uniform(u, o)
        assert bytes(o) == C.from_uniform_bytes(u).compress()
        w = hashlib.shake_256(b"w%d" % i).digest(64)
        L.oracle_sc_wide(w, o)
        assert int.from_bytes(bytes(o), "little") == int.from_bytes(w, "little") % C.L
    for n in [1, 7, 189, 190, 520, 801]:  # Straus below 190 terms, Pippenger w=6/7/8 above
        pts = [C.from_uniform_bytes(hashlib.shake_256(b"p%d" % (i % 12)).digest(64)) for i in range(n)]
        scs = [int.from_bytes(hashlib.shake_256(b"s%d" % i).digest(32), "little") % C.L for i in range(n)]
        o = (ctypes.c_uint8 * 32)()
        assert L.oracle_msm(b"".join(sb(s) for s in scs), b"".join(p.compress() for p in pts), n, o) == 1
        assert bytes(o) == C.multiscalar_mul(scs, pts).compress()
    seed = sb(7)
    o = (ctypes.c_uint8 * 32)()
    L.oracle_nonce(seed, b"dR", 3, 1, o)
    assert bytes(o) == sb(O.nonce(7, "dR", 3, 1))


@pytest.mark.parametrize("n,batch,t", [(8, [1, 2, 1], 2), (64, [1, 2], 1), (4, [4], 3)])
def test_c_verifier_and_prover_match_pyref(n, batch, t):
    c = make_oracle_batch(n, batch, t, seed=b"cport%d" % n)
    cp = cport.Params(n, max(batch), t)
    items = _items(c)
    rc, masks, tr = cp.verify(items, action=1, want_trace=True)
    want_masks, otr = oracle_verify_trace(c, action=1)
    assert rc == 0 and masks == want_masks
    assert tr["challenges"] == trace_challenge_bytes(otr, max(len(p.li) for p in c.o_proofs))
    assert tr["rng_out"] == b"".join(otr["rng_outputs"])
    assert tr["weights"] == b"".join(sb(w) for w in otr["weights"])
    static = b"".join(sb(a) + sb(b) for a, b in zip(otr["gi"], otr["hi"])) + b"".join(sb(x) for x in otr["g"]) + sb(otr["h"])
    assert tr["static_scalars"] == static
    assert tr["dynamic_scalars"] == b"".join(sb(x) for x in otr["dynamic_scalars"])
    assert tr["msm_result"] == bytes(32)
    items[0]["min_values"] = [(v + 1 if v is not None else 1) for v in items[0]["min_values"]]
    assert cp.verify(items, action=0)[0] == O.VERIFICATION_FAILED
    for st, w, pr in zip(c.o_statements_private, c.o_witnesses, c.o_proofs):  # identical proof bytes
        ext = hashlib.shake_256(b"ext").digest(32 * (len(pr.li) + 3))
        want = O.prove_with_rng(M.Transcript(c.label), st, w, M.ByteStreamRng(ext)).to_bytes()
        got, comm = cp.prove(c.label, [o.v for o in w.openings], [[sb(x) for x in o.r] for o in w.openings],
                             st.minimum_value_promises, sb(st.seed_nonce) if st.seed_nonce is not None else None, ext)
        assert got == want and comm == st.commitments_compressed
    cp.close()


def test_bench_fixtures_verify_with_c_port():
    """the committed BASELINE configs[1]/[2] inputs are valid proofs (sample of each, full-chunk for cfg2)"""
    b2 = load_bench("bench_cfg2.bin")
    assert len(b2["items"]) == 1024 and b2["m"] == 1 and all(len(i["proof"]) == 577 for i in b2["items"])
    cp = cport.Params(64, 1, 1)
    assert cp.verify(b2["items"][:256], action=0)[0] == 0
    rc, masks, _ = cp.verify(b2["items"][256:264], action=2)
    assert rc == 0 and all(m is not None for m in masks)
    cp.close()
    b3 = load_bench("bench_cfg3.bin")
    assert len(b3["items"]) == 256 and b3["m"] == 8 and all(len(i["proof"]) == 769 for i in b3["items"])
    cp = cport.Params(64, 8, 1)
    assert cp.verify(b3["items"][:16], action=0)[0] == 0
    cp.close()
