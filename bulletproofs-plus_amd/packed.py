"""Array forms of the batch entry points for large inputs (tens of thousands of proofs per call).

api.py mirrors the reference's object interface (one RangeStatement / RangeProof / RangeWitness object per proof) and
marshals item by item with ctypes, ~20 us per proof; a caller that already holds its proofs, commitments and openings
as contiguous arrays uses these instead: the bpp_verify_item / bpp_prove_item arrays of include/bpp.h are built with
numpy pointer arithmetic (no per-item Python), the calls are the same C ABI calls.

Shapes (n items, aggregation m, extension degree t, rounds = log2(m * bit_length)):
    proofs        uint8 [n, 1 + 32 (t + 5 + 2 rounds)]     RangeProof::to_bytes()          (src/range_proof.rs:1120-1150)
    commitments   uint8 [n, m, 32]                         statement.commitments_compressed (src/range_statement.rs:27)
    min_values    uint64 [n, m], min_present uint8 [n, m]  Option<u64> promises             (src/range_statement.rs:29)
    seed_nonces   uint8 [n, 32] or None                                                      (src/range_statement.rs:31)
    values        uint64 [n, m], blindings uint8 [n, m, t, 32]   RangeWitness openings      (src/commitment_opening.rs:14-37)
    rng_bytes     uint8 [n, 32 (rounds + 3)]               what the external RNG hands out  (src/range_proof.rs:232-608)
"""
import ctypes
import time
from ctypes import POINTER, byref, c_size_t, c_uint64

import numpy as np

from . import _lib, api

_VERIFY_ITEM = np.dtype([("proof", "<u8"), ("proof_len", "<u8"), ("commitments32", "<u8"), ("m", "<u4"), ("_pad", "<u4"),
                         ("min_values", "<u8"), ("min_present", "<u8"), ("seed_nonce32", "<u8"), ("transcript_state", "<u8"),
                         ("transcript_label", "<u8"), ("label_len", "<u8")])
_PROVE_ITEM = np.dtype([("values", "<u8"), ("blindings32", "<u8"), ("commitments32", "<u8"), ("m", "<u4"), ("_pad", "<u4"),
                        ("min_values", "<u8"), ("min_present", "<u8"), ("seed_nonce32", "<u8"), ("transcript_state", "<u8"),
                        ("transcript_label", "<u8"), ("label_len", "<u8"), ("rng_bytes", "<u8"), ("rng_len", "<u8")])
assert _VERIFY_ITEM.itemsize == ctypes.sizeof(_lib.VerifyItem) and _PROVE_ITEM.itemsize == ctypes.sizeof(_lib.ProveItem)


def _c(a, dtype, shape):
    a = np.ascontiguousarray(a, dtype=dtype)
    if a.shape != tuple(shape):
        raise api.ProofError(api.ProofErrorKind.InvalidLength, "array of shape %s expected, got %s" % (tuple(shape), a.shape))
    return a


def _rows(a):
    """addresses of the rows of a C-contiguous array"""
    return a.ctypes.data + np.arange(a.shape[0], dtype=np.uint64) * np.uint64(a.strides[0])


def commit(params, values, blindings):
    """PedersenGens::commit for k openings (src/generators/pedersen_gens.rs:112-122): uint64 [k], uint8 [k, nb, 32] -> [k, 32]"""
    values = np.ascontiguousarray(values, dtype=np.uint64)
    k = values.shape[0]
    blindings = np.ascontiguousarray(blindings, dtype=np.uint8)
    nb = blindings.shape[1]
    blindings = _c(blindings, np.uint8, (k, nb, 32))
    out = np.zeros((k, 32), dtype=np.uint8)
    eng = params.engine
    rc = eng.lib.bpp_pedersen_commit(eng.ctx, params.handle, values.ctypes.data, blindings.ctypes.data, nb, k, out.ctypes.data)
    api._check(rc, eng.ctx)
    return out


def prove(params, values, blindings, commitments, min_values, min_present, seed_nonces, label, rng_bytes):
    """n x RangeProof::prove_with_rng in one bpp_prove_batch call -> uint8 [n, proof_len]"""
    n_bits, t = params.bit_length(), int(params.extension_degree())
    values = np.ascontiguousarray(values, dtype=np.uint64)
    n, m = values.shape
    rounds = max((n_bits * m).bit_length() - 1, 0)
    blindings = _c(blindings, np.uint8, (n, m, t, 32))
    commitments = _c(commitments, np.uint8, (n, m, 32))
    min_values = _c(min_values, np.uint64, (n, m))
    min_present = _c(min_present, np.uint8, (n, m))
    rng_bytes = _c(rng_bytes, np.uint8, (n, 32 * (rounds + 3)))
    lbl = np.frombuffer(bytes(label), dtype=np.uint8).copy() if len(label) else np.zeros(1, dtype=np.uint8)
    items = np.zeros(n, dtype=_PROVE_ITEM)
    items["values"] = _rows(values)
    items["blindings32"] = _rows(blindings)
    items["commitments32"] = _rows(commitments)
    items["m"] = m
    items["min_values"] = _rows(min_values)
    items["min_present"] = _rows(min_present)
    if seed_nonces is not None:
        seed_nonces = _c(seed_nonces, np.uint8, (n, 32))
        items["seed_nonce32"] = _rows(seed_nonces)
    items["transcript_label"] = lbl.ctypes.data
    items["label_len"] = len(label)
    items["rng_bytes"] = _rows(rng_bytes)
    items["rng_len"] = rng_bytes.shape[1]
    plen = 1 + 32 * (t + 5 + 2 * rounds)
    out = np.empty((n, plen), dtype=np.uint8)  # (every byte is written by the call, or the call raises)
    got = c_size_t()
    err = ctypes.create_string_buffer(256)
    eng = params.engine
    rc = eng.lib.bpp_prove_batch(eng.ctx, params.handle, items.ctypes.data_as(POINTER(_lib.ProveItem)), n, out.ctypes.data, plen,
                                 byref(got), err, 256)
    api._check(rc, eng.ctx, err)
    assert got.value == plen
    return out


class PackedInput:
    """A homogeneous batch as contiguous arrays + the bpp_packed_batch that describes it (include/bpp.h).  Holds the
    arrays alive; `.struct` can be passed to any *_packed entry point any number of times."""

    def __init__(self, proofs, commitments, min_values, min_present, seed_nonces, label, seed_present=None, state=None):
        self.proofs = np.ascontiguousarray(proofs, dtype=np.uint8)
        n, plen = self.proofs.shape
        commitments = np.ascontiguousarray(commitments, dtype=np.uint8)
        m = commitments.shape[1]
        self.commitments = _c(commitments, np.uint8, (n, m, 32))
        self.min_values = None if min_values is None else _c(min_values, np.uint64, (n, m))
        self.min_present = None if min_present is None else _c(min_present, np.uint8, (n, m))
        self.seed_nonces = None if seed_nonces is None else _c(seed_nonces, np.uint8, (n, 32))
        self.seed_present = None if seed_present is None else _c(seed_present, np.uint8, (n,))
        self.label = np.frombuffer(bytes(label), dtype=np.uint8).copy() if len(label) else np.zeros(1, dtype=np.uint8)
        self.state = None if state is None else np.frombuffer(bytes(state), dtype=np.uint8).copy()
        self.n, self.m = n, m
        ptr = lambda a: None if a is None else a.ctypes.data  # noqa: E731
        self.struct = _lib.PackedBatch(n, ptr(self.proofs), plen, self.proofs.strides[0], ptr(self.commitments), m,
                                       ptr(self.min_values), ptr(self.min_present), ptr(self.seed_nonces),
                                       ptr(self.seed_present), ptr(self.state), ptr(self.label), len(label))


def verify_batch(params, inp, action=api.VerifyAction.VerifyOnly, chunk=api.MAX_RANGE_PROOF_BATCH_SIZE):
    """RangeProof::verify_batch over a PackedInput in ONE C call (bpp_verify_batch_packed: upload, verify, release).
    Returns (masks uint8 [n, t, 32], present uint8 [n])."""
    eng, t = params.engine, int(params.extension_degree())
    masks = np.zeros((inp.n, t, 32), dtype=np.uint8)
    present = np.zeros(inp.n, dtype=np.uint8)
    err = ctypes.create_string_buffer(256)
    rc = eng.lib.bpp_verify_batch_packed(eng.ctx, params.handle, byref(inp.struct), int(action), chunk, masks.ctypes.data,
                                         present.ctypes.data, err, 256)
    api._check(rc, eng.ctx, err)
    return masks, present


class Batcher:
    """bpp_batcher: many host threads, each calling verify(inp) with ONE reference batch; the calls that are waiting are pooled
    into grouped engine calls (bpp_verify_resident_groups), every caller gets the outcome of a call of its own.  `shape`: a
    PackedInput whose proof length, aggregation factor and label say what can be pooled."""

    def __init__(self, params, shape, lanes=0, max_wait_us=0, max_calls=64):
        self.params, self.engine = params, params.engine
        self.handle = ctypes.c_void_p()
        api._check(self.engine.lib.bpp_batcher_create(self.engine.ctx, params.handle, byref(shape.struct), lanes, max_wait_us, max_calls,
                                                      byref(self.handle)), self.engine.ctx)

    def verify(self, inp):
        """blocks; raises ProofError exactly as verify_batch(params, inp, VerifyOnly, chunk=0) would"""
        err = ctypes.create_string_buffer(256)
        api._check(self.engine.lib.bpp_batcher_verify(self.handle, byref(inp.struct), err, 256), None, err)

    def verify_action(self, inp, action):
        """any VerifyAction through the pool: returns (masks uint8 [n, t, 32], present uint8 [n]) exactly as
        verify_batch(params, inp, action, chunk=0) would, or raises its ProofError"""
        t = int(self.params.extension_degree())
        masks = np.zeros((inp.n, t, 32), dtype=np.uint8)
        present = np.zeros(inp.n, dtype=np.uint8)
        err = ctypes.create_string_buffer(256)
        api._check(self.engine.lib.bpp_batcher_verify_action(self.handle, byref(inp.struct), int(action), masks.ctypes.data,
                                                             present.ctypes.data, err, 256), None, err)
        return masks, present

    def largest_pool(self):
        c, p = ctypes.c_uint32(), ctypes.c_uint32()
        self.engine.lib.bpp_batcher_largest_pool(self.handle, byref(c), byref(p))
        return c.value, p.value

    def set_limits(self, max_calls=0, max_proofs=0):
        api._check(self.engine.lib.bpp_batcher_set_limits(self.handle, max_calls, max_proofs), None)

    def stats(self):
        v = [c_uint64() for _ in range(3)]
        self.engine.lib.bpp_batcher_stats(self.handle, *[byref(x) for x in v])
        return dict(zip(("pooled_calls", "engine_calls", "solo_calls"), [x.value for x in v]))

    def close(self):
        if self.handle:
            self.engine.lib.bpp_batcher_destroy(self.handle)
            self.handle = ctypes.c_void_p()


def verify_groups_actions(rb, bounds, actions):
    """bpp_verify_resident_groups_actions: one VerifyAction per group -> (result dicts, masks [n, t, 32], present [n])"""
    G = len(bounds) - 1
    arr = (ctypes.c_uint32 * (G + 1))(*bounds)
    act = (ctypes.c_int * G)(*[int(a) for a in actions])
    out = (_lib.ShardResult * G)()
    masks = np.zeros((rb.n, rb.t, 32), dtype=np.uint8)
    present = np.zeros(rb.n, dtype=np.uint8)
    api._check(rb.engine.lib.bpp_verify_resident_groups_actions(rb.engine.ctx, rb.handle, arr, G, act, out, masks.ctypes.data,
                                                                present.ctypes.data), rb.engine.ctx)
    return [{"code": r.code, "tier": r.tier, "index": r.index, "msg": r.msg.decode(errors="replace")} for r in out], masks, present


def runtime_info(engine):
    """bpp_runtime_info_get as a dict"""
    info = _lib.RuntimeInfo()
    api._check(engine.lib.bpp_runtime_info_get(engine.ctx, byref(info)), engine.ctx)
    return {n: getattr(info, n) for n, _ in _lib.RuntimeInfo._fields_}


def verify_groups(rb, bounds):
    """bpp_verify_resident_groups on a resident batch: group g = proofs [bounds[g], bounds[g+1]) -> list of result dicts"""
    G = len(bounds) - 1
    arr = (ctypes.c_uint32 * (G + 1))(*bounds)
    out = (_lib.ShardResult * G)()
    api._check(rb.engine.lib.bpp_verify_resident_groups(rb.engine.ctx, rb.handle, arr, G, out), rb.engine.ctx)
    return [{"code": r.code, "tier": r.tier, "index": r.index, "msg": r.msg.decode(errors="replace")} for r in out]


class Pipeline:
    """bpp_verify_submit_packed / bpp_verify_collect on one engine: upload k+1 overlaps verify k inside ONE context"""

    def __init__(self, params, depth=None):
        self.params, self.engine, self.t = params, params.engine, int(params.extension_degree())
        if depth is not None:
            api._check(self.engine.lib.bpp_ctx_pipeline_depth(self.engine.ctx, int(depth)), self.engine.ctx)
        self._n = {}

    def submit(self, inp, action=api.VerifyAction.VerifyOnly, chunk=api.MAX_RANGE_PROOF_BATCH_SIZE):
        ticket = c_uint64()
        err = ctypes.create_string_buffer(256)
        rc = self.engine.lib.bpp_verify_submit_packed(self.engine.ctx, self.params.handle, byref(inp.struct), int(action), chunk,
                                                      byref(ticket), err, 256)
        api._check(rc, None, err)
        self._n[ticket.value] = (inp.n, int(action))
        return ticket.value

    def collect(self, ticket):
        n, action = self._n.pop(ticket)
        err = ctypes.create_string_buffer(256)
        if action == int(api.VerifyAction.VerifyOnly):
            api._check(self.engine.lib.bpp_verify_collect(self.engine.ctx, ticket, None, None, err, 256), None, err)
            return None, None
        masks = np.zeros((n, self.t, 32), dtype=np.uint8)
        present = np.zeros(n, dtype=np.uint8)
        api._check(self.engine.lib.bpp_verify_collect(self.engine.ctx, ticket, masks.ctypes.data, present.ctypes.data, err, 256),
                   None, err)
        return masks, present


class ResidentBatch(api.ResidentBatch):
    """api.ResidentBatch (bpp_batch_upload / bpp_verify_resident / traces) uploaded from arrays.  form="packed": through
    bpp_batch_upload_packed (one bpp_packed_batch, no per-item structs); form="items": a bpp_verify_item array built with
    numpy pointer arithmetic.  Both end in the same resident batch (tests/test_gpu_packed.py)."""

    def __init__(self, params, proofs, commitments, min_values, min_present, seed_nonces, label, form="packed"):
        self.params, self.engine, self.t = params, params.engine, int(params.extension_degree())
        self.handle = c_uint64()
        err = ctypes.create_string_buffer(256)
        if form == "packed":
            t0 = time.perf_counter()
            inp = PackedInput(proofs, commitments, min_values, min_present, seed_nonces, label)
            self.n = inp.n
            t1 = time.perf_counter()
            rc = self.engine.lib.bpp_batch_upload_packed(self.engine.ctx, params.handle, byref(inp.struct), byref(self.handle), err, 256)
            api._check(rc, self.engine.ctx, err)
            self.marshal_seconds, self.upload_seconds = t1 - t0, time.perf_counter() - t1
            return
        t0 = time.perf_counter()
        proofs = np.ascontiguousarray(proofs, dtype=np.uint8)
        n, plen = proofs.shape
        commitments = np.ascontiguousarray(commitments, dtype=np.uint8)
        m = commitments.shape[1]
        commitments = _c(commitments, np.uint8, (n, m, 32))
        min_values = _c(min_values, np.uint64, (n, m))
        min_present = _c(min_present, np.uint8, (n, m))
        lbl = np.frombuffer(bytes(label), dtype=np.uint8).copy() if len(label) else np.zeros(1, dtype=np.uint8)
        items = np.zeros(n, dtype=_VERIFY_ITEM)
        items["proof"] = _rows(proofs)
        items["proof_len"] = plen
        items["commitments32"] = _rows(commitments)
        items["m"] = m
        items["min_values"] = _rows(min_values)
        items["min_present"] = _rows(min_present)
        if seed_nonces is not None:
            seed_nonces = _c(seed_nonces, np.uint8, (n, 32))
            items["seed_nonce32"] = _rows(seed_nonces)
        items["transcript_label"] = lbl.ctypes.data
        items["label_len"] = len(label)
        t1 = time.perf_counter()
        self.n = n
        rc = self.engine.lib.bpp_batch_upload(self.engine.ctx, params.handle, items.ctypes.data_as(POINTER(_lib.VerifyItem)), n,
                                              byref(self.handle), err, 256)
        api._check(rc, self.engine.ctx, err)
        self.marshal_seconds, self.upload_seconds = t1 - t0, time.perf_counter() - t1

    def verify_arrays(self, action, chunk=0):
        """bpp_verify_resident with the masks as arrays (no per-item Python): (masks uint8 [n, t, 32], present uint8 [n])"""
        if not hasattr(self, "_masks"):
            self._masks = np.zeros((self.n, self.t, 32), dtype=np.uint8)
            self._present = np.zeros(self.n, dtype=np.uint8)
        err = ctypes.create_string_buffer(256)
        rc = self.engine.lib.bpp_verify_resident(self.engine.ctx, self.handle, int(action), chunk, self._masks.ctypes.data,
                                                 self._present.ctypes.data, err, 256)
        api._check(rc, self.engine.ctx, err)
        return self._masks, self._present
