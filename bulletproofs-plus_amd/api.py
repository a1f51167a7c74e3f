"""Host-side mirror of the reference's public interface for the hot path, on top of the C ABI (include/bpp.h).

Same names, argument meaning and error behaviour as tari_bulletproofs_plus 0.4.1, so the parity tests read like
the reference's own tests (tests/ristretto.rs):

    reference (Rust)                                   here
    -------------------------------------------------  ------------------------------------------
    ristretto::create_pedersen_gens_with_extension_..  create_pedersen_gens_with_extension_degree
    RangeParameters::init(n, m, pc_gens)               RangeParameters.init(n, m, pc_gens)
    RangeStatement::init(gens, commitments, mins, sn)  RangeStatement.init(...)
    RangeProof::from_bytes / to_bytes                  RangeProof.from_bytes / to_bytes
    RangeProof::verify_batch(ts, sts, proofs, action)  RangeProof.verify_batch(...)
    VerifyAction / ProofError / ExtendedMask           same names
    merlin::Transcript::new(label)                     Transcript.new(label)

Points are 32-byte ristretto255 encodings, scalars 32-byte canonical little-endian `bytes`.
Everything heavy runs in libbpp_hip.so on the GPU; nothing here touches oracle/.
"""
import ctypes
import enum
from ctypes import byref, c_int, c_size_t, c_uint32, c_uint64, c_void_p

from . import _lib

L_ORDER = 2**252 + 27742317777372353535851937790883648493


class ProofErrorKind(enum.IntEnum):
    """src/errors.rs:11-28"""
    VerificationFailed = 1
    InvalidArgument = 2
    InvalidLength = 3
    InvalidBlake2b = 4
    SizeOverflow = 5


class ProofError(Exception):
    def __init__(self, kind, msg=""):
        self.kind = ProofErrorKind(kind)
        self.msg = msg
        super().__init__("%s: %s" % (self.kind.name, msg))


class EngineError(RuntimeError):
    """negative return codes of the C ABI: HIP failures, no gfx950 device, bad handles"""


class VerifyAction(enum.IntEnum):
    """src/range_proof.rs:46-54"""
    VerifyOnly = 0
    RecoverAndVerify = 1
    RecoverOnly = 2


class ExtensionDegree(enum.IntEnum):
    """src/generators/pedersen_gens.rs:40-55"""
    DefaultPedersen = 1
    AddOneBasePoint = 2
    AddTwoBasePoints = 3
    AddThreeBasePoints = 4
    AddFourBasePoints = 5
    AddFiveBasePoints = 6

    @staticmethod
    def try_from(v):
        try:
            return ExtensionDegree(int(v))
        except ValueError:
            raise ProofError(ProofErrorKind.InvalidArgument, "Extension degree not valid")


MAX_RANGE_PROOF_BATCH_SIZE = 256  # src/range_proof.rs:76


def _buf(data):
    return (ctypes.c_uint8 * len(data)).from_buffer_copy(data) if len(data) else (ctypes.c_uint8 * 1)()


def _check(rc, ctx=None, errbuf=None):
    if rc == 0:
        return
    msg = ""
    if errbuf is not None and errbuf.value:
        msg = errbuf.value.decode(errors="replace")
    elif ctx is not None:
        m = _lib.load().bpp_ctx_last_error(ctx)
        msg = m.decode(errors="replace") if m else ""
    if rc > 0:
        raise ProofError(rc, msg)
    raise EngineError("bpp engine error %d: %s" % (rc, msg))


class Engine:
    """One bpp_ctx: a device + stream.  `stream` may be a raw hipStream_t (e.g. torch.cuda.current_stream().cuda_stream)."""

    _default = None

    def __init__(self, device=0, stream=None):
        self.lib = _lib.load()
        self.ctx = c_void_p()
        if stream:
            rc = self.lib.bpp_ctx_create_on_stream(byref(self.ctx), int(device), c_void_p(int(stream)))
        else:
            rc = self.lib.bpp_ctx_create(byref(self.ctx), int(device))
        if rc != 0:
            raise EngineError("bpp_ctx_create failed (%d): a gfx950 device is required, there is no CPU fallback" % rc)
        self.device = device

    @classmethod
    def default(cls):
        if cls._default is None:
            cls._default = Engine(0)
        return cls._default

    def close(self):
        if self.ctx:
            self.lib.bpp_ctx_destroy(self.ctx)
            self.ctx = c_void_p()

    # ---- B1 ----
    def msm_vartime(self, scalars, points):
        """P::vartime_multiscalar_mul(scalars, points) -> compressed point"""
        assert len(scalars) == len(points)
        out = (ctypes.c_uint8 * 32)()
        rc = self.lib.bpp_msm_vartime(self.ctx, _buf(b"".join(scalars)), _buf(b"".join(points)), len(scalars), out)
        _check(rc, self.ctx)
        return bytes(out)

    def msm_vartime_batched(self, scalars, points, group_off):
        g = len(group_off) - 1
        out = (ctypes.c_uint8 * (32 * g))()
        go = (c_uint32 * len(group_off))(*group_off)
        rc = self.lib.bpp_msm_vartime_batched(self.ctx, _buf(b"".join(scalars)), _buf(b"".join(points)), go, g, out)
        _check(rc, self.ctx)
        raw = bytes(out)
        return [raw[32 * i:32 * i + 32] for i in range(g)]

    def precomputation(self, static_points):
        return Precomputation(self, static_points)

    def profile(self, on=True):
        """bpp_profile_enable: False / 0 off, True / 1 an event at every stage boundary, 2 the roofline kernel's two events only"""
        self.lib.bpp_profile_enable(self.ctx, int(on))

    def set_option(self, name, value):
        """bpp_ctx_set_option: per-context knob (tests, A/B timing); -1 restores the engine's own rule"""
        _check(self.lib.bpp_ctx_set_option(self.ctx, name.encode(), int(value)), self.ctx)

    def device_chain_stats(self):
        """bpp_device_chain_stats: (verifications whose weight chains ran on the device, those sent back to the host chains)"""
        calls, redraws = c_uint64(), c_uint64()
        _check(self.lib.bpp_device_chain_stats(self.ctx, byref(calls), byref(redraws)), self.ctx)
        return int(calls.value), int(redraws.value)

    def last_profile(self):
        p = _lib.Profile()
        self.lib.bpp_profile_get(self.ctx, byref(p))
        return {n: getattr(p, n) for n, _ in _lib.Profile._fields_}

    def last_prove_profile(self):
        p = _lib.ProveProfile()
        self.lib.bpp_prove_profile_get(self.ctx, byref(p))
        return {n: getattr(p, n) for n, _ in _lib.ProveProfile._fields_}


class Precomputation:
    """VartimeRistrettoPrecomputation (src/ristretto.rs:62-64, built at src/generators/bulletproof_gens.rs:103)."""

    def __init__(self, engine, static_points):
        self.engine = engine
        self.count = len(static_points)
        self.handle = c_uint64()
        rc = engine.lib.bpp_precomp_create(engine.ctx, _buf(b"".join(static_points)), self.count, byref(self.handle))
        _check(rc, engine.ctx)

    def vartime_mixed_multiscalar_mul(self, static_scalars, dynamic_scalars, dynamic_points):
        assert len(dynamic_scalars) == len(dynamic_points)
        out = (ctypes.c_uint8 * 32)()
        rc = self.engine.lib.bpp_msm_mixed(self.engine.ctx, self.handle, _buf(b"".join(static_scalars)),
                                           len(static_scalars), _buf(b"".join(dynamic_scalars)),
                                           _buf(b"".join(dynamic_points)), len(dynamic_scalars), out)
        _check(rc, self.engine.ctx)
        return bytes(out)

    def close(self):
        self.engine.lib.bpp_precomp_destroy(self.engine.ctx, self.handle)


class Transcript:
    """merlin::Transcript as far as the boundary needs it: a label (fresh transcript) or a 203-byte STROBE state."""

    def __init__(self, label=None, state=None):
        self.label = label
        self.state = state

    @staticmethod
    def new(label):
        return Transcript(label=bytes(label))

    @staticmethod
    def from_state(state203):
        assert len(state203) == 203
        return Transcript(state=bytes(state203))

    def clone(self):
        return Transcript(self.label, self.state)

    def strobe_state(self):
        if self.state is not None:
            return self.state
        out = (ctypes.c_uint8 * 203)()
        _lib.load().bpp_transcript_new(_buf(self.label), len(self.label), out)
        return bytes(out)


class PedersenGens:
    """src/generators/pedersen_gens.rs:25-36; base points are fixed by the extension degree (src/ristretto.rs:67-76)."""

    def __init__(self, extension_degree, h_base=None, g_base_vec=None):
        self.extension_degree = ExtensionDegree.try_from(extension_degree)
        self.h_base_compressed = h_base
        self.g_base_compressed_vec = g_base_vec


def create_pedersen_gens_with_extension_degree(extension_degree):
    return PedersenGens(extension_degree)


class RangeParameters:
    """src/range_parameters.rs:32-113 -- owns the device generator tables."""

    def __init__(self):
        raise TypeError("use RangeParameters.init")

    @classmethod
    def init(cls, bit_length, max_aggregation_factor, pc_gens, engine=None):
        self = object.__new__(cls)
        self.engine = engine or Engine.default()
        self.handle = c_uint64()
        h = _buf(pc_gens.h_base_compressed) if pc_gens.h_base_compressed else None
        g = _buf(b"".join(pc_gens.g_base_compressed_vec)) if pc_gens.g_base_compressed_vec else None
        rc = self.engine.lib.bpp_params_create(self.engine.ctx, bit_length, max_aggregation_factor,
                                               int(pc_gens.extension_degree), h, g, byref(self.handle))
        _check(rc, self.engine.ctx)
        self._n, self._m, self._t = bit_length, max_aggregation_factor, int(pc_gens.extension_degree)
        nm = bit_length * max_aggregation_factor
        gi = (ctypes.c_uint8 * (32 * nm))()
        hi = (ctypes.c_uint8 * (32 * nm))()
        hb = (ctypes.c_uint8 * 32)()
        gb = (ctypes.c_uint8 * (32 * self._t))()
        _check(self.engine.lib.bpp_params_export(self.engine.ctx, self.handle, gi, hi, hb, gb), self.engine.ctx)
        self._gi = [bytes(gi[32 * i:32 * i + 32]) for i in range(nm)]
        self._hi = [bytes(hi[32 * i:32 * i + 32]) for i in range(nm)]
        self._h = bytes(hb)
        self._g = [bytes(gb[32 * i:32 * i + 32]) for i in range(self._t)]
        self.pc_gens = PedersenGens(self._t, self._h, self._g)
        return self

    def share(self, engine):
        """Arc::clone for another context of the same device (bpp_params_retain): the SAME device tables, usable from
        `engine` concurrently with every other holder (src/traits.rs:42 `Precomputation: Send + Sync`)."""
        _check(engine.lib.bpp_params_retain(engine.ctx, self.handle), engine.ctx)
        other = object.__new__(RangeParameters)
        other.__dict__.update(self.__dict__)
        other.engine = engine
        return other

    def close(self):
        """drop this holder's reference (bpp_params_destroy)"""
        if self.handle.value:
            self.engine.lib.bpp_params_destroy(self.engine.ctx, self.handle)
            self.handle = c_uint64()

    def bit_length(self):
        return self._n

    def max_aggregation_factor(self):
        return self._m

    def extension_degree(self):
        return ExtensionDegree(self._t)

    def h_base_compressed(self):
        return self._h

    def g_bases_compressed(self):
        return list(self._g)

    def gi_base_compressed(self):
        return list(self._gi)

    def hi_base_compressed(self):
        return list(self._hi)

    def commit(self, value, blindings):
        """PedersenGens::commit (src/generators/pedersen_gens.rs:112-122) -> compressed commitment"""
        return self.commit_many([value], [blindings])[0]

    def commit_many(self, values, blindings_list):
        n = len(values)
        nb = len(blindings_list[0]) if n else 0
        if any(len(b) != nb for b in blindings_list):
            raise ProofError(ProofErrorKind.InvalidLength, "blinding vector")
        vals = (c_uint64 * max(n, 1))(*values)
        out = (ctypes.c_uint8 * (32 * max(n, 1)))()
        rc = self.engine.lib.bpp_pedersen_commit(self.engine.ctx, self.handle, vals,
                                                 _buf(b"".join(b"".join(b) for b in blindings_list)), nb, n, out)
        _check(rc, self.engine.ctx)
        raw = bytes(out)
        return [raw[32 * i:32 * i + 32] for i in range(n)]


class RangeStatement:
    """src/range_statement.rs:21-73"""

    def __init__(self):
        raise TypeError("use RangeStatement.init")

    @classmethod
    def init(cls, generators, commitments, minimum_value_promises, seed_nonce):
        n = len(commitments)
        if n == 0 or n & (n - 1):
            raise ProofError(ProofErrorKind.InvalidArgument, "Number of commitments must be a power of two")
        if len(minimum_value_promises) != n:
            raise ProofError(ProofErrorKind.InvalidArgument, "Incorrect number of minimum value promises")
        if generators.max_aggregation_factor() < n:
            raise ProofError(ProofErrorKind.InvalidArgument, "Not enough generators for this statement")
        if seed_nonce is not None and n > 1:
            raise ProofError(ProofErrorKind.InvalidArgument,
                             "Mask recovery is not supported with an aggregated statement")
        self = object.__new__(cls)
        self.generators = generators
        self.commitments_compressed = [bytes(c) for c in commitments]
        self.minimum_value_promises = list(minimum_value_promises)
        self.seed_nonce = seed_nonce
        return self


class CommitmentOpening:
    """src/commitment_opening.rs:14-37: value + extended blinding factors (32-byte canonical scalars)"""

    def __init__(self, v, r):
        self.v = int(v)
        self.r = [bytes(x) for x in r]

    @staticmethod
    def new(v, r):
        return CommitmentOpening(v, r)

    def r_len(self):
        if not self.r:
            raise ProofError(ProofErrorKind.InvalidLength, "Extended blinding factors cannot be empty")
        return len(self.r)


class RangeWitness:
    """src/range_witness.rs:15-41"""

    def __init__(self):
        raise TypeError("use RangeWitness.init")

    @classmethod
    def init(cls, openings):
        if not openings:
            raise ProofError(ProofErrorKind.InvalidLength, "Vector openings cannot be empty")
        t = openings[0].r_len()
        for o in openings[1:]:
            if o.r_len() != t:
                raise ProofError(ProofErrorKind.InvalidLength, "Extended blinding factors must have consistent length")
        self = object.__new__(cls)
        self.openings = list(openings)
        self.extension_degree = ExtensionDegree.try_from(t)
        return self


class ExtendedMask:
    """src/extended_mask.rs:14-41"""

    def __init__(self, blindings):
        self._blindings = list(blindings)

    @staticmethod
    def assign(extension_degree, blindings):
        if len(blindings) == 0 or len(blindings) != int(extension_degree):
            raise ProofError(ProofErrorKind.InvalidLength,
                             "Extended mask length must correspond to the extension degree")
        return ExtendedMask(blindings)

    def blindings(self):
        if not self._blindings:
            raise ProofError(ProofErrorKind.InvalidLength, "Extended mask values not assigned yet")
        return list(self._blindings)

    def __eq__(self, o):
        return isinstance(o, ExtendedMask) and self._blindings == o._blindings

    def __repr__(self):
        return "ExtendedMask(%s)" % [b.hex() for b in self._blindings]


class RangeProof:
    """src/range_proof.rs:58-68 -- held as its canonical wire bytes plus the parsed view."""

    def __init__(self, raw, t, rounds):
        self._raw = raw
        self._t = t
        self._rounds = rounds

    def extension_degree(self):
        return ExtensionDegree(self._t)

    def to_bytes(self):
        return self._raw

    def __eq__(self, o):
        return isinstance(o, RangeProof) and self._raw == o._raw

    @staticmethod
    def from_bytes(data):
        """src/range_proof.rs:1155-1257 (host-only structural parse; points are not validated, like the reference)."""
        data = bytes(data)
        if len(data) < 1:
            raise ProofError(ProofErrorKind.InvalidLength, "Serialized proof is too short")
        t = int(ExtensionDegree.try_from(data[0]))
        body = len(data) - 1
        nchunks, rem = divmod(body, 32)

        def chunk(i):
            if i >= nchunks:
                raise ProofError(ProofErrorKind.InvalidLength, "Serialized proof is too short")
            return data[1 + 32 * i:33 + 32 * i]

        def scalar(i):
            if int.from_bytes(chunk(i), "little") >= L_ORDER:
                raise ProofError(ProofErrorKind.InvalidArgument, "Invalid parsing")

        for k in range(t):
            scalar(k)
        chunk(t), chunk(t + 1), chunk(t + 2)
        scalar(t + 3), scalar(t + 4)
        rest = nchunks - (t + 5)
        if rest // 2 == 0:
            raise ProofError(ProofErrorKind.InvalidLength, "Serialized proof is too short")
        if rest % 2 or rem:
            raise ProofError(ProofErrorKind.InvalidLength, "Unused data after deserialization")
        return RangeProof(data, t, rest // 2)

    @staticmethod
    def extension_degree_from_proof_bytes(data):
        if len(data) < 1:
            raise ProofError(ProofErrorKind.InvalidLength, "Serialized proof is too short")
        return ExtensionDegree.try_from(data[0])

    # ---- proving ----
    @staticmethod
    def rounds_for(statement):
        mn = statement.generators.bit_length() * len(statement.commitments_compressed)
        return max(mn.bit_length() - 1, 0)

    @staticmethod
    def prove_batch(transcripts, statements, witnesses, rng_bytes):
        """n x RangeProof::prove_with_rng (src/range_proof.rs:232-608) in one engine call.

        rng_bytes[i]: the bytes the external RNG hands out for proof i, 32 per draw, (rounds + 3) draws."""
        return RangeProof._prove_call(RangeProof._prove_marshal(transcripts, statements, witnesses, rng_bytes))

    @staticmethod
    def _prove_marshal(transcripts, statements, witnesses, rng_bytes):
        """host checks of :248-260 + the bpp_prove_item array (ctypes marshalling, kept apart so that it can be timed apart)"""
        if not statements or len(statements) != len(witnesses) or len(transcripts) != len(statements) or \
                len(rng_bytes) != len(statements):
            raise ProofError(ProofErrorKind.InvalidArgument, "Range statements, witnesses, transcripts length mismatch")
        params = statements[0].generators
        n = len(statements)
        items = (_lib.ProveItem * n)()
        keep = []
        for i, (tr, st, w, rb) in enumerate(zip(transcripts, statements, witnesses, rng_bytes)):
            if st.generators is not params:
                raise ProofError(ProofErrorKind.InvalidArgument, "one RangeParameters object per prove batch")
            m = len(st.commitments_compressed)
            # :248-260
            if len(w.openings) != m:
                raise ProofError(ProofErrorKind.InvalidLength, "Witness openings and statement commitments do not match!")
            if int(w.extension_degree) != int(params.extension_degree()):
                raise ProofError(ProofErrorKind.InvalidLength, "Witness and statement extension degrees do not match!")
            vals = (c_uint64 * m)(*[o.v for o in w.openings])
            bl = _buf(b"".join(b"".join(o.r) for o in w.openings))
            cb = _buf(b"".join(st.commitments_compressed))
            mv = (c_uint64 * m)(*[(v if v is not None else 0) for v in st.minimum_value_promises])
            mp = (ctypes.c_uint8 * m)(*[(1 if v is not None else 0) for v in st.minimum_value_promises])
            rbuf = _buf(rb)
            keep += [vals, bl, cb, mv, mp, rbuf]
            it = items[i]
            it.values = ctypes.cast(vals, c_void_p)
            it.blindings32 = ctypes.cast(bl, c_void_p)
            it.commitments32 = ctypes.cast(cb, c_void_p)
            it.m = m
            it.min_values = ctypes.cast(mv, c_void_p)
            it.min_present = ctypes.cast(mp, c_void_p)
            it.rng_bytes = ctypes.cast(rbuf, c_void_p)
            it.rng_len = len(rb)
            if st.seed_nonce is not None:
                sb = _buf(st.seed_nonce)
                keep.append(sb)
                it.seed_nonce32 = ctypes.cast(sb, c_void_p)
            if tr.state is not None:
                tb = _buf(tr.state)
                keep.append(tb)
                it.transcript_state = ctypes.cast(tb, c_void_p)
            else:
                lb = _buf(tr.label)
                keep.append(lb)
                it.transcript_label = ctypes.cast(lb, c_void_p)
                it.label_len = len(tr.label)
        return params, items, n, keep

    @staticmethod
    def _prove_call(marshalled, parse=True):
        params, items, n, _keep = marshalled
        eng = params.engine
        stride = 1 + 32 * (6 + 5 + 2 * 12)
        out = (ctypes.c_uint8 * (stride * n))()
        plen = c_size_t()
        err = ctypes.create_string_buffer(256)
        rc = eng.lib.bpp_prove_batch(eng.ctx, params.handle, items, n, out, stride, byref(plen), err, 256)
        _check(rc, eng.ctx, err)
        raw = bytes(out)
        if not parse:
            return [raw[i * stride:i * stride + plen.value] for i in range(n)]
        return [RangeProof.from_bytes(raw[i * stride:i * stride + plen.value]) for i in range(n)]

    @staticmethod
    def prove_with_rng(transcript, statement, witness, rng):
        """RangeProof::prove_with_rng; `rng` = object with fill_bytes(n) (the external RNG) or the bytes themselves"""
        need = 32 * (RangeProof.rounds_for(statement) + 3)
        rb = rng if isinstance(rng, (bytes, bytearray)) else b"".join(rng.fill_bytes(32) for _ in range(need // 32))
        return RangeProof.prove_batch([transcript], [statement], [witness], [bytes(rb)])[0]

    @staticmethod
    def prove(transcript, statement, witness):
        """RangeProof::prove (OsRng, src/range_proof.rs:222-228)"""
        import os as _os
        return RangeProof.prove_with_rng(transcript, statement, witness, _os.urandom(32 * (RangeProof.rounds_for(statement) + 3)))

    # ---- verification ----
    @staticmethod
    def _items(transcripts, statements, proofs):
        keep = []
        items = (_lib.VerifyItem * len(proofs))()
        for i, (tr, st, pr) in enumerate(zip(transcripts, statements, proofs)):
            raw = pr.to_bytes() if isinstance(pr, RangeProof) else bytes(pr)
            m = len(st.commitments_compressed)
            pb = _buf(raw)
            cb = _buf(b"".join(st.commitments_compressed))
            mv = (c_uint64 * m)(*[(v if v is not None else 0) for v in st.minimum_value_promises])
            mp = (ctypes.c_uint8 * m)(*[(1 if v is not None else 0) for v in st.minimum_value_promises])
            keep += [pb, cb, mv, mp]
            it = items[i]
            it.proof = ctypes.cast(pb, c_void_p)
            it.proof_len = len(raw)
            it.commitments32 = ctypes.cast(cb, c_void_p)
            it.m = m
            it.min_values = ctypes.cast(mv, c_void_p)
            it.min_present = ctypes.cast(mp, c_void_p)
            if st.seed_nonce is not None:
                sb = _buf(st.seed_nonce)
                keep.append(sb)
                it.seed_nonce32 = ctypes.cast(sb, c_void_p)
            if tr.state is not None:
                tb = _buf(tr.state)
                keep.append(tb)
                it.transcript_state = ctypes.cast(tb, c_void_p)
            else:
                lb = _buf(tr.label)
                keep.append(lb)
                it.transcript_label = ctypes.cast(lb, c_void_p)
                it.label_len = len(tr.label)
        return items, keep

    @staticmethod
    def _check_batch_args(transcripts, statements, proofs):
        # src/range_proof.rs:719-734
        if not statements or not proofs or not transcripts:
            raise ProofError(ProofErrorKind.InvalidArgument, "Range statements or proofs length empty")
        if len(statements) != len(proofs):
            raise ProofError(ProofErrorKind.InvalidArgument, "Range statements and proofs length mismatch")
        if len(transcripts) != len(statements):
            raise ProofError(ProofErrorKind.InvalidArgument, "Range statements and transcripts length mismatch")
        # verify_statements_and_generators_consistency (:610-709): one parameter object per batch
        g0 = statements[0].generators
        for st in statements[1:]:
            g = st.generators
            if g is g0:
                continue
            if g.g_bases_compressed() != g0.g_bases_compressed():
                raise ProofError(ProofErrorKind.InvalidArgument, "Inconsistent G generator point in batch statement")
            if g.h_base_compressed() != g0.h_base_compressed():
                raise ProofError(ProofErrorKind.InvalidArgument, "Inconsistent H generator point in batch statement")
            if g.bit_length() != g0.bit_length():
                raise ProofError(ProofErrorKind.InvalidArgument, "Inconsistent bit length in batch statement")
            if g.extension_degree() != g0.extension_degree():
                raise ProofError(ProofErrorKind.InvalidArgument, "Inconsistent extension degree")

    @staticmethod
    def _largest_params(statements):
        # the statement with the largest aggregation capacity carries the precomputation (:666-673, :778)
        return max((st.generators for st in statements), key=lambda g: g.max_aggregation_factor())

    @staticmethod
    def verify_batch(transcripts, statements, proofs, action, chunk=MAX_RANGE_PROOF_BATCH_SIZE):
        """RangeProof::verify_batch (src/range_proof.rs:712-752) -> list[ExtendedMask | None].

        `chunk` proofs form one reference batch (own weight transcript, own final MSM); unlike the reference's
        wrapper every chunk is verified.  chunk=0: the whole input is one batch (the private `verify`)."""
        RangeProof._check_batch_args(transcripts, statements, proofs)
        params = RangeProof._largest_params(statements)
        eng = params.engine
        items, keep = RangeProof._items(transcripts, statements, proofs)
        n, t = len(proofs), int(params.extension_degree())
        masks = (ctypes.c_uint8 * (n * t * 32))()
        present = (ctypes.c_uint8 * n)()
        err = ctypes.create_string_buffer(256)
        rc = eng.lib.bpp_verify_batch(eng.ctx, params.handle, items, n, int(action), chunk, masks, present, err, 256)
        _check(rc, eng.ctx, err)
        raw = bytes(masks)
        out = []
        for i in range(n):
            if present[i]:
                out.append(ExtendedMask.assign(t, [raw[(i * t + k) * 32:(i * t + k) * 32 + 32] for k in range(t)]))
            else:
                out.append(None)
        return out


def verify_batch_with_challenges(statements, proofs, challenges, rng_outputs, action, chunk=MAX_RANGE_PROOF_BATCH_SIZE):
    """bpp_verify_batch_with_challenges: PASS 1 done by the caller.  challenges[i] = [y, z, e_0.., e_final] (32-byte
    scalars), rng_outputs[i] = the 32 transcript-RNG bytes of proof i."""
    trs = [Transcript.new(b"")] * len(proofs)
    RangeProof._check_batch_args(trs, statements, proofs)
    params = RangeProof._largest_params(statements)
    eng = params.engine
    items, keep = RangeProof._items(trs, statements, proofs)
    n, t = len(proofs), int(params.extension_degree())
    bufs = [_buf(b"".join(c)) for c in challenges]
    ptrs = (c_void_p * n)(*[ctypes.cast(b, c_void_p) for b in bufs])
    masks = (ctypes.c_uint8 * (n * t * 32))()
    present = (ctypes.c_uint8 * n)()
    err = ctypes.create_string_buffer(256)
    rc = eng.lib.bpp_verify_batch_with_challenges(eng.ctx, params.handle, items, n, ptrs, _buf(b"".join(rng_outputs)),
                                                  int(action), chunk, masks, present, err, 256)
    _check(rc, eng.ctx, err)
    raw = bytes(masks)
    return [ExtendedMask.assign(t, [raw[(i * t + k) * 32:(i * t + k) * 32 + 32] for k in range(t)]) if present[i] else None
            for i in range(n)]


class ResidentBatch:
    """A batch kept in HBM (bpp_batch_upload / bpp_verify_resident), plus the parity trace accessors."""

    def __init__(self, transcripts, statements, proofs):
        RangeProof._check_batch_args(transcripts, statements, proofs)
        self.params = RangeProof._largest_params(statements)
        self.engine = self.params.engine
        self.n = len(proofs)
        self.t = int(self.params.extension_degree())
        import time as _time
        t0 = _time.perf_counter()
        items, keep = RangeProof._items(transcripts, statements, proofs)
        t1 = _time.perf_counter()
        self.handle = c_uint64()
        err = ctypes.create_string_buffer(256)
        rc = self.engine.lib.bpp_batch_upload(self.engine.ctx, self.params.handle, items, self.n, byref(self.handle),
                                              err, 256)
        _check(rc, self.engine.ctx, err)
        # ctypes marshalling (Python only) / the C-ABI call: host buffers -> parsed, packed, resident in HBM
        self.marshal_seconds, self.upload_seconds = t1 - t0, _time.perf_counter() - t1

    def verify(self, action=VerifyAction.VerifyOnly, chunk=0):
        masks = (ctypes.c_uint8 * (self.n * self.t * 32))()
        present = (ctypes.c_uint8 * self.n)()
        err = ctypes.create_string_buffer(256)
        rc = self.engine.lib.bpp_verify_resident(self.engine.ctx, self.handle, int(action), chunk, masks, present, err,
                                                 256)
        _check(rc, self.engine.ctx, err)
        raw = bytes(masks)
        t = self.t
        return [ExtendedMask.assign(t, [raw[(i * t + k) * 32:(i * t + k) * 32 + 32] for k in range(t)])
                if present[i] else None for i in range(self.n)]

    def verify_only(self, chunk=0):
        """VerifyAction::VerifyOnly without the mask vector (all None by definition): no per-item work on the host"""
        err = ctypes.create_string_buffer(256)
        rc = self.engine.lib.bpp_verify_resident(self.engine.ctx, self.handle, int(VerifyAction.VerifyOnly), chunk, None, None,
                                                 err, 256)
        _check(rc, self.engine.ctx, err)

    def prepare(self, chunk=0):
        """bpp_batch_prepare: the one-off planning / allocation of verify(chunk=...) done ahead of the first call"""
        _check(self.engine.lib.bpp_batch_prepare(self.engine.ctx, self.handle, chunk), self.engine.ctx)

    def phase1(self):
        out = (ctypes.c_uint8 * (self.n * 32))()
        err = ctypes.create_string_buffer(256)
        _check(self.engine.lib.bpp_verify_phase1(self.engine.ctx, self.handle, out, err, 256), self.engine.ctx, err)
        return bytes(out)

    def phase2(self, weights32):
        assert len(weights32) == 32 * self.n
        acc = (ctypes.c_uint8 * 128)()
        err = ctypes.create_string_buffer(256)
        _check(self.engine.lib.bpp_verify_phase2(self.engine.ctx, self.handle, _buf(weights32), acc, err, 256),
               self.engine.ctx, err)
        return bytes(acc)

    def shape(self):
        v = [c_uint32() for _ in range(5)]
        _check(self.engine.lib.bpp_batch_shape(self.engine.ctx, self.handle, *[byref(x) for x in v]), self.engine.ctx)
        return dict(zip(("n_items", "max_rounds", "max_mn", "total_dyn", "groups"), [x.value for x in v]))

    def trace(self, what):
        need = c_size_t()
        self.engine.lib.bpp_batch_trace(self.engine.ctx, self.handle, what, None, 0, byref(need))
        out = (ctypes.c_uint8 * max(need.value, 1))()
        _check(self.engine.lib.bpp_batch_trace(self.engine.ctx, self.handle, what, out, need.value, byref(need)),
               self.engine.ctx)
        return bytes(out)[:need.value]

    def close(self):
        if self.handle.value:
            self.engine.lib.bpp_batch_destroy(self.engine.ctx, self.handle)
            self.handle = c_uint64()


def shader_clock_ghz(engine, window_us=20000):
    """the shader clock the device holds while this call naps on `engine`'s stream (other engines keep it busy)"""
    g = ctypes.c_double()
    _check(engine.lib.bpp_shader_clock(engine.ctx, int(window_us), byref(g)), engine.ctx)
    return g.value


def host_threads():
    """size of the engine's host worker pool (weight chains, upload packer)"""
    return int(_lib.load().bpp_host_threads())


def host_pool_cpu_ns():
    """bpp_host_pool_cpu_ns: CPU time this process has spent in the library's host jobs (weight chains, batch parsing)"""
    return int(_lib.load().bpp_host_pool_cpu_ns())


def weights_from_chain(rng32_all):
    """The batch-weight transcript (src/range_proof.rs:811,849,853,894) over 32-byte transcript-RNG outputs."""
    n = len(rng32_all) // 32
    out = (ctypes.c_uint8 * (32 * max(n, 1)))()
    rc = _lib.load().bpp_weights_from_chain(_buf(rng32_all), n, out)
    _check(rc)
    return bytes(out)[:32 * n]


def weights_from_chains(rng32_all, n_groups):
    """n_groups independent weight transcripts over equal slices of rng32_all (vectorised lockstep on the host)"""
    n = len(rng32_all) // 32
    assert n % n_groups == 0
    out = (ctypes.c_uint8 * (32 * max(n, 1)))()
    _check(_lib.load().bpp_weights_from_chains(_buf(rng32_all), n_groups, n // n_groups, out))
    return bytes(out)[:32 * n]


def accumulators_sum_is_identity(engine, accumulators128):
    n = len(accumulators128) // 128
    flag = c_int()
    _check(engine.lib.bpp_accumulators_sum_is_identity(engine.ctx, _buf(accumulators128), n, byref(flag)), engine.ctx)
    return bool(flag.value)
