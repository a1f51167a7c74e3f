"""GPU tests added in round 2: error precedence of verify()'s two consistency loops (src/range_proof.rs:637-682) alone
and against earlier chunks' verdicts, shared RangeParameters across contexts (`Precomputation: Send + Sync`,
src/traits.rs:42, src/generators/bulletproof_gens.rs:52,103), handle lifetime."""
import threading

import pytest

from oracle import cport
from tests.helpers import LABEL, make_batch, sb

pytestmark = pytest.mark.gpu


def _kind(bpp, fn):
    with pytest.raises(bpp.ProofError) as e:
        fn()
    return e.value.kind


def test_two_loop_precedence(bpp, engine):
    """the extension degree of ALL items is checked (:637-659) before ANY minimum-value promise (:674-682): item 0 with an
    oversized promise + item 1 with another degree -> InvalidArgument, not item 0's InvalidLength"""
    K, A = bpp.ProofErrorKind, bpp.VerifyAction
    c = make_batch(bpp, engine, 4, [1, 1, 1], 1, seed=b"two-loop")
    c2 = make_batch(bpp, engine, 4, [1], 2, seed=b"two-loop-2")  # proofs with extension degree 2
    big = bpp.RangeStatement.init(c.params, c.statements_public[0].commitments_compressed, [1 << 4], None)
    V = lambda sts, proofs, chunk=0: bpp.RangeProof.verify_batch(c.transcripts()[:len(proofs)], sts, proofs, A.VerifyOnly,
                                                                 chunk=chunk)
    sts = [big, c.statements_public[1], c.statements_public[2]]
    assert _kind(bpp, lambda: V(sts, c.proofs)) == K.InvalidLength
    mixed = [c.proofs[0], c2.proofs[0], c.proofs[2]]
    assert _kind(bpp, lambda: V(sts, mixed)) == K.InvalidArgument
    assert _kind(bpp, lambda: V(c.statements_public, mixed)) == K.InvalidArgument
    # with the reference's own chunking (one verify() per chunk, here chunk = 1) an EARLIER chunk's MSM verdict comes
    # first, a LATER chunk's does not
    raw = bytearray(c.proofs[0].to_bytes())
    raw[1 + 32 + 96] ^= 1  # r1: only the final MSM notices
    flipped = bpp.RangeProof.from_bytes(bytes(raw))
    assert _kind(bpp, lambda: V(c.statements_public, [flipped, c2.proofs[0], c.proofs[2]], chunk=1)) == K.VerificationFailed
    assert _kind(bpp, lambda: V(c.statements_public, [c.proofs[0], c2.proofs[0], flipped], chunk=1)) == K.InvalidArgument
    assert _kind(bpp, lambda: V([c.statements_public[0], big, c.statements_public[2]],
                                [flipped, c.proofs[1], c.proofs[2]], chunk=1)) == K.VerificationFailed
    assert _kind(bpp, lambda: V([c.statements_public[0], big, c.statements_public[2]], c.proofs, chunk=1)) == K.InvalidLength
    # a construction error (from_bytes: non-canonical scalar) is an upload error and beats everything
    raw2 = bytearray(c.proofs[2].to_bytes())
    raw2[1 + 32 + 96 + 31] = 0xff
    assert _kind(bpp, lambda: V(c.statements_public, [flipped, c2.proofs[0], bytes(raw2)], chunk=1)) == K.InvalidArgument
    # the untouched batch still verifies, also when the degree-2 item sits in the resident form's phase 1
    assert V(c.statements_public, c.proofs) == [None] * 3
    rb = bpp.ResidentBatch(c.transcripts(), c.statements_public, mixed)
    assert _kind(bpp, rb.phase1) == K.InvalidArgument
    rb.close()


def test_params_shared_by_four_contexts(bpp):
    """ONE RangeParameters handle (one generator table, one fixed-base table) used by four contexts at once: two verify,
    two prove; device memory grows by one table, not four; the handle outlives the context that created it"""
    import ctypes
    from tests.test_gpu_prove import _inputs

    def free_bytes():  # hipMemGetInfo of the HIP runtime the engine itself runs on (no second runtime in the process)
        hip = ctypes.CDLL("libamdhip64.so")
        free, total = ctypes.c_size_t(), ctypes.c_size_t()
        assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
        return free.value
    K, A = bpp.ProofErrorKind, bpp.VerifyAction
    n, m, t = 16, 2, 1
    owner = bpp.Engine(0)
    free0 = free_bytes()
    params = bpp.RangeParameters.init(n, m, bpp.create_pedersen_gens_with_extension_degree(t), engine=owner)
    sts, wits, exts, raw = _inputs(bpp, params, n, m, t, 4, b"shared-params", "third")
    first = bpp.RangeProof.prove_batch([bpp.Transcript.new(LABEL)] * 4, sts, wits, exts)  # builds the fixed-base table
    free1 = free_bytes()
    table_cost = free0 - free1
    assert table_cost > 50 << 20  # 65 generators x 24 windows x 1024 entries x 128 B = 204 MB
    cp = cport.Params(n, m, t)
    want = [cp.prove(LABEL, raw["vals"][i], raw["blinds"][i], raw["mins"][i], raw["seeds"][i], exts[i])[0] for i in range(4)]
    cp.close()
    assert [p.to_bytes() for p in first] == want
    engines = [bpp.Engine(0) for _ in range(4)]
    shared = [params.share(e) for e in engines]
    results, errors = {}, []

    def rebind(objs, p):
        out = []
        for s in objs:
            st = bpp.RangeStatement.init(p, s.commitments_compressed, s.minimum_value_promises, s.seed_nonce)
            out.append(st)
        return out

    def prover(idx):
        try:
            mine = rebind(sts, shared[idx])
            out = []
            for _ in range(3):
                got = bpp.RangeProof.prove_batch([bpp.Transcript.new(LABEL)] * 4, mine, wits, exts)
                out.append([g.to_bytes() for g in got] == want)
            results[idx] = out
        except Exception as e:  # noqa: BLE001
            errors.append((idx, repr(e)))

    def verifier(idx, bad):
        try:
            mine = rebind(sts, shared[idx])
            proofs = [bpp.RangeProof.from_bytes(w) for w in want]
            if bad:
                r = bytearray(want[2])
                r[1 + 32 * t + 96] ^= 1
                proofs[2] = bpp.RangeProof.from_bytes(bytes(r))
            out = []
            for _ in range(4):
                try:
                    bpp.RangeProof.verify_batch([bpp.Transcript.new(LABEL)] * 4, mine, proofs, A.VerifyOnly)
                    out.append("ok")
                except bpp.ProofError as e:
                    out.append(int(e.kind))
            results[idx] = out
        except Exception as e:  # noqa: BLE001
            errors.append((idx, repr(e)))

    threads = [threading.Thread(target=prover, args=(0,)), threading.Thread(target=verifier, args=(1, False)),
               threading.Thread(target=prover, args=(2,)), threading.Thread(target=verifier, args=(3, True))]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    assert results[0] == [True] * 3 and results[2] == [True] * 3
    assert results[1] == ["ok"] * 4 and results[3] == [int(K.VerificationFailed)] * 4
    free2 = free_bytes()
    # four more users: their own work buffers only (arena, staging), no second copy of the 200 MB table
    assert free1 - free2 < table_cost // 2, (table_cost, free1 - free2)
    # the creating context goes away; the holders keep using the tables
    owner.close()
    got = bpp.RangeProof.prove_batch([bpp.Transcript.new(LABEL)] * 4, rebind(sts, shared[0]), wits, exts)
    assert [g.to_bytes() for g in got] == want
    # a context that never retained the handle may use it while it lives, but cannot drop it
    stranger = bpp.Engine(0)
    assert stranger.lib.bpp_params_destroy(stranger.ctx, params.handle) == -3
    for sp in shared:
        sp.close()
    # every reference dropped: the handle is dead for everyone, the memory is back
    with pytest.raises(bpp.EngineError):
        bpp.RangeProof.prove_batch([bpp.Transcript.new(LABEL)] * 4, rebind(sts, shared[0]), wits, exts)
    for e in engines:
        e.close()
    stranger.close()
    free3 = free_bytes()
    assert free3 - free2 > table_cost // 2, (table_cost, free3 - free2)


def test_params_outlive_destroy_while_batch_resident(bpp):
    """bpp_params_destroy while a resident batch still references the parameters: the batch keeps them alive"""
    eng = bpp.Engine(0)
    c = make_batch(bpp, eng, 8, [1, 1], 1, seed=b"lifetime")
    rb = bpp.ResidentBatch(c.transcripts(), c.statements_public, c.proofs)
    c.params.close()
    assert rb.verify(bpp.VerifyAction.VerifyOnly, chunk=0) == [None, None]
    rb.close()
    with pytest.raises(bpp.EngineError):
        bpp.ResidentBatch(c.transcripts(), c.statements_public, c.proofs)
    eng.close()


def test_oversized_proof_does_not_inflate_the_batch(bpp, engine):
    """a proof claiming far more rounds than any statement allows (mn <= 2048 -> at most 11) is refused with the reference's
    InvalidLength (:886-888; SizeOverflow from 32 rounds on, :882-885) -- next to valid proofs and in its own chunk -- and the
    batch's per-proof buffers stay
    sized for the statements: challenge slots are capped (shape()['max_rounds'] <= 11), so are the scalar-stage tables"""
    K, A = bpp.ProofErrorKind, bpp.VerifyAction
    c = make_batch(bpp, engine, 8, [1, 1, 1, 1], 1, seed=b"oversized")
    raw = c.proofs[1].to_bytes()
    lr = raw[-64:]
    for extra, want in ((1, K.InvalidLength), (9, K.InvalidLength), (40, K.SizeOverflow)):  # 4, 12, 43 rounds; the statement has 3
        big = bpp.RangeProof.from_bytes(raw + lr * extra)
        proofs = [c.proofs[0], big, c.proofs[2], c.proofs[3]]
        V = lambda chunk: bpp.RangeProof.verify_batch(c.transcripts(), c.statements_public, proofs, A.VerifyOnly, chunk=chunk)
        assert _kind(bpp, lambda: V(0)) == want
        assert _kind(bpp, lambda: V(1)) == want  # chunk 0 verifies, chunk 1 (the oversized proof alone) fails
        assert _kind(bpp, lambda: V(2)) == want
        rb = bpp.ResidentBatch(c.transcripts(), c.statements_public, proofs)
        assert rb.shape()["max_rounds"] <= 11
        assert _kind(bpp, lambda: rb.verify(A.VerifyOnly, chunk=1)) == want
        assert len(rb.trace(1)) == 4 * (rb.shape()["max_rounds"] + 3) * 32
        rb.close()
    # the engine's own cap (layout.h BPP_MAX_WIRE_ROUNDS = 64 pairs): at the cap the proof is still replayed and reported with
    # PASS-2 precedence (an identity member of a LATER proof, a PASS-1 finding, comes first); one pair more is refused at
    # upload, same error kind, and quickly: no kernel ever walks a proof longer than the cap
    import time
    at_cap = bpp.RangeProof.from_bytes(raw + lr * 61)
    r3 = bytearray(c.proofs[3].to_bytes())
    r3[1 + 32:1 + 64] = bytes(32)
    ident = bpp.RangeProof.from_bytes(bytes(r3))
    V = lambda proofs: bpp.RangeProof.verify_batch(c.transcripts(), c.statements_public, proofs, A.VerifyOnly, chunk=0)
    assert _kind(bpp, lambda: V([c.proofs[0], at_cap, c.proofs[2], c.proofs[3]])) == K.SizeOverflow
    assert _kind(bpp, lambda: V([c.proofs[0], at_cap, c.proofs[2], ident])) == K.VerificationFailed
    t0 = time.perf_counter()
    assert _kind(bpp, lambda: V([c.proofs[0], raw + lr * 62, c.proofs[2], ident])) == K.SizeOverflow
    assert _kind(bpp, lambda: V([c.proofs[0], raw + lr * 100000, c.proofs[2], ident])) == K.SizeOverflow  # a 6.4 MB "proof"
    assert time.perf_counter() - t0 < 2.0
    # the same proofs without the oversized one still verify on the same engine
    assert bpp.RangeProof.verify_batch(c.transcripts(), c.statements_public, c.proofs, A.VerifyOnly, chunk=2) == [None] * 4


def test_shader_clock_probe(bpp, engine):
    """bpp_shader_clock: one napping wavefront reads the shader-clock counter against the constant 100 MHz one"""
    g = bpp.shader_clock_ghz(engine, window_us=2000)
    assert 0.3 < g < 3.5, g
