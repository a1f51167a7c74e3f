"""GPU tests of the array entry points (bulletproofs-plus_amd/packed.py) and the small ABI additions of round 2:
the packed forms build the same bpp_verify_item / bpp_prove_item arrays as the object API, so every result must be
byte-identical to the object API's (which tests/test_gpu_prove.py / test_gpu_verify.py hold to the oracle)."""
import importlib

import numpy as np
import pytest

from tests.helpers import LABEL
from tests.test_gpu_prove import _inputs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,m,t,count", [(64, 1, 1, 5), (16, 4, 2, 3), (64, 2, 3, 2)])
def test_packed_equals_object_api(bpp, engine, n, m, t, count):
    packed = importlib.import_module("bulletproofs-plus_amd.packed")
    params = bpp.RangeParameters.init(n, m, bpp.create_pedersen_gens_with_extension_degree(t), engine=engine)
    sts, wits, exts, raw = _inputs(bpp, params, n, m, t, count, b"packed-%d-%d-%d" % (n, m, t), "third")
    want = [p.to_bytes() for p in bpp.RangeProof.prove_batch([bpp.Transcript.new(LABEL)] * count, sts, wits, exts)]
    values = np.array(raw["vals"], dtype=np.uint64)
    blindings = np.frombuffer(b"".join(b"".join(b"".join(o) for o in item) for item in raw["blinds"]), dtype=np.uint8).reshape(count, m, t, 32)
    comm = packed.commit(params, values.reshape(-1), blindings.reshape(count * m, t, 32)).reshape(count, m, 32)
    assert [[bytes(comm[i, j]) for j in range(m)] for i in range(count)] == raw["comms"]
    mins = np.array([[(v if v is not None else 0) for v in it] for it in raw["mins"]], dtype=np.uint64)
    pres = np.array([[(1 if v is not None else 0) for v in it] for it in raw["mins"]], dtype=np.uint8)
    seeds = None
    if m == 1:
        seeds = np.frombuffer(b"".join(raw["seeds"]), dtype=np.uint8).reshape(count, 32)
    ext = np.frombuffer(b"".join(exts), dtype=np.uint8).reshape(count, -1)
    engine.profile(True)
    got = packed.prove(params, values, blindings, comm, mins, pres, seeds, LABEL, ext)
    assert [bytes(got[i]) for i in range(count)] == want
    pp = engine.last_prove_profile()
    rounds = (n * m).bit_length() - 1
    # terms through the fixed-base tables: the rounds' L and R, A1 + B (the witness check's m (1 + t) terms per proof take the
    # uniform-access form since round 5: option "ct", default 1)
    assert pp["fb_terms"] == count * (rounds * 2 * (n * m + t + 1) + 2 * n * m + 2 * t + 2)
    assert pp["fb_msm_ms"] > 0 and pp["fb_launches"] >= rounds + 1 and pp["total_ms"] >= pp["fb_msm_ms"] / max(pp["sub_batches"], 1)
    engine.profile(False)
    # resident batch from arrays: same verdicts and the same intermediates as the object form
    rb = packed.ResidentBatch(params, got, comm, mins, pres, None, LABEL)
    rb.prepare(0)
    rb.prepare(0)  # idempotent
    rb.verify_only(0)
    pub = [bpp.RangeStatement.init(params, s.commitments_compressed, s.minimum_value_promises, None) for s in sts]
    ro = bpp.ResidentBatch([bpp.Transcript.new(LABEL)] * count, pub, [bpp.RangeProof.from_bytes(w) for w in want])
    ro.verify(bpp.VerifyAction.VerifyOnly, chunk=0)
    for what in (1, 2, 3, 4, 5, 6):
        assert rb.trace(what) == ro.trace(what)
    assert rb.trace(6) == bytes(32)
    rb.close()
    ro.close()
    bad = got.copy()
    bad[count - 1, 1 + 32 * t + 96] ^= 1  # r1 of the last proof
    rb = packed.ResidentBatch(params, bad, comm, mins, pres, None, LABEL)
    with pytest.raises(bpp.ProofError) as e:
        rb.verify_only(0)
    assert e.value.kind == bpp.ProofErrorKind.VerificationFailed
    rb.close()
    with pytest.raises(bpp.ProofError):  # wrong array shape is refused before the engine is called
        packed.ResidentBatch(params, got, comm[:, :, :16], mins, pres, None, LABEL)
    params.close()


def test_host_threads_and_prepare_errors(bpp, engine):
    assert 1 <= bpp.host_threads() <= 256
    assert engine.lib.bpp_batch_prepare(engine.ctx, 123456789, 0) == -3  # unknown batch handle
