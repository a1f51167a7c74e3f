"""CPU suite: the reference's fuzz target (fuzz/fuzz_targets/proofs.rs:10-15: from_bytes ok => to_bytes round-trips) as a
hypothesis property, plus agreement of the product's host-side parser with the oracle's on arbitrary byte strings."""
import importlib

from hypothesis import given, settings
from hypothesis import strategies as st

from oracle.pyref import protocol as O
from tests.helpers import make_oracle_batch

_BASE = make_oracle_batch(4, [1], 2, seed=b"fuzz").o_proofs[0].to_bytes()


def _engine_parser():
    import ctypes
    pkg = importlib.import_module("bulletproofs-plus_amd")
    lib = ctypes.CDLL(pkg._build.build_hosttest())
    lib.ht_parse_proof.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]
    return lib


_HT = _engine_parser()


def _outcome(parse, data):
    try:
        return ("ok", parse(data).to_bytes())
    except Exception as e:  # ProofError of either implementation
        return ("err", int(e.kind))


@settings(max_examples=400, deadline=None)
@given(st.one_of(
    st.binary(max_size=700),
    st.tuples(st.integers(0, len(_BASE) - 1), st.integers(0, 255)).map(lambda t: _BASE[:t[0]] + bytes([t[1]]) + _BASE[t[0] + 1:]),
    st.integers(0, len(_BASE) + 70).map(lambda n: (_BASE + bytes(70))[:n]),
    st.tuples(st.integers(1, 6), st.integers(0, 24)).map(lambda t: bytes([t[0]]) + bytes(32 * t[1])),
))
def test_from_bytes_round_trip_and_parity(data):
    bpp = importlib.import_module("bulletproofs-plus_amd")
    got = _outcome(bpp.RangeProof.from_bytes, data)
    want = _outcome(O.RangeProof.from_bytes, data)
    assert got == want
    if got[0] == "ok":
        assert got[1] == bytes(data)  # canonical: deserialise-then-serialise is the identity
    # the engine's own parser (csrc/upload_host.h, applied to raw bytes at bpp_batch_upload) gives the same verdict
    import ctypes
    t, r = ctypes.c_uint32(), ctypes.c_uint32()
    rc = _HT.ht_parse_proof(bytes(data), len(data), ctypes.byref(t), ctypes.byref(r))
    assert (("ok", None) if rc == 0 else ("err", rc))[0] == want[0] and (rc == 0 or rc == want[1])
    if rc == 0:
        assert len(data) == 1 + 32 * (t.value + 5 + 2 * r.value)
