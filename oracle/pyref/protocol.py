"""ORACLE (test infrastructure only -- never imported by the product path).

Line-by-line big-int restatement of the reference's Bulletproofs+ protocol layer
(tari_bulletproofs_plus 0.4.1).  Every function cites the reference lines it follows.
Parity status: the reference holds NO golden byte vectors -> byte parity with
the Rust crate is "unpinned"; this restatement is pinned by RFC 9496 / merlin / hashlib
KATs and libsodium (build container only).
"""
import hashlib

from . import curve as C
from .curve import L, Point
from .merlin import NullRng, Transcript

# ---- errors: src/errors.rs:11-28; numeric codes are the C ABI's (include/bpp.h) ----
VERIFICATION_FAILED, INVALID_ARGUMENT, INVALID_LENGTH, INVALID_BLAKE2B, SIZE_OVERFLOW = 1, 2, 3, 4, 5
_KIND_NAMES = {1: "VerificationFailed", 2: "InvalidArgument", 3: "InvalidLength", 4: "InvalidBlake2b",
               5: "SizeOverflow"}


class ProofError(Exception):
    def __init__(self, kind, msg=""):
        super().__init__("%s: %s" % (_KIND_NAMES[kind], msg))
        self.kind = kind
        self.msg = msg


VERIFY_ONLY, RECOVER_AND_VERIFY, RECOVER_ONLY = 0, 1, 2  # src/range_proof.rs:46-54
MAX_RANGE_PROOF_BIT_LENGTH = 64  # :71
MAX_RANGE_PROOF_BATCH_SIZE = 256  # :76


# ---- generators ----

class GeneratorsChain:
    """src/generators/generators_chain.rs:23-49: SHAKE256("GeneratorsChain" || label), 64-byte blocks."""

    def __init__(self, label):
        self._shake = hashlib.shake_256(b"GeneratorsChain" + label)
        self._n = 0

    def take(self, count):
        stream = self._shake.digest(64 * (self._n + count))
        out = [C.from_uniform_bytes(stream[64 * i:64 * i + 64]) for i in range(self._n, self._n + count)]
        self._n += count
        return out


class BulletproofGens:
    """src/generators/bulletproof_gens.rs:83-134."""

    def __init__(self, gens_capacity, party_capacity):
        self.gens_capacity = gens_capacity
        self.party_capacity = party_capacity
        self.g_vec, self.h_vec = [], []
        for i in range(party_capacity):
            self.g_vec.append(GeneratorsChain(b"G" + i.to_bytes(4, "little")).take(gens_capacity))
            self.h_vec.append(GeneratorsChain(b"H" + i.to_bytes(4, "little")).take(gens_capacity))

    def g_iter(self, n, m):  # aggregated_gens_iter.rs:18-43, party-major
        return [self.g_vec[j][i] for j in range(m) for i in range(n)]

    def h_iter(self, n, m):
        return [self.h_vec[j][i] for j in range(m) for i in range(n)]

    def precomp_points(self):
        """Interleaved G0,H0,G1,H1,... over all parties (bulletproof_gens.rs:99-103)."""
        g = [p for v in self.g_vec for p in v]
        h = [p for v in self.h_vec for p in v]
        out = []
        for a, b in zip(g, h):
            out += [a, b]
        return out


def hash_from_bytes_sha3_512(data):
    """src/protocols/curve_point_protocol.rs:31-35."""
    return C.from_uniform_bytes(hashlib.sha3_512(data).digest())


_MASKING = {}


def ristretto_masking_basepoints():
    """src/ristretto.rs:88-112."""
    if not _MASKING:
        for i in range(1, 7):
            _MASKING[i] = hash_from_bytes_sha3_512(b"RISTRETTO_MASKING_BASEPOINT_" + str(i).encode())
    return [_MASKING[i] for i in range(1, 7)]


class PedersenGens:
    """src/generators/pedersen_gens.rs:25-122, src/ristretto.rs:67-76."""

    def __init__(self, extension_degree):
        if not 1 <= extension_degree <= 6:
            raise ProofError(INVALID_ARGUMENT, "Extension degree not valid")
        self.extension_degree = extension_degree
        self.h_base = C.BASEPOINT
        self.h_base_compressed = self.h_base.compress()
        self.g_base_vec = ristretto_masking_basepoints()[:extension_degree]
        self.g_base_compressed_vec = [g.compress() for g in self.g_base_vec]

    def commit(self, value, blindings):
        if len(blindings) == 0 or len(blindings) > self.extension_degree:
            raise ProofError(INVALID_LENGTH, "blinding vector")
        return C.multiscalar_mul([value] + list(blindings), [self.h_base] + self.g_base_vec[:len(blindings)])


class RangeParameters:
    """src/range_parameters.rs:32-113."""

    def __init__(self, bit_length, max_aggregation_factor, pc_gens):
        if max_aggregation_factor < 1 or max_aggregation_factor & (max_aggregation_factor - 1):
            raise ProofError(INVALID_ARGUMENT, "Aggregation factor size must be a power of two")
        if bit_length < 1 or bit_length & (bit_length - 1):
            raise ProofError(INVALID_ARGUMENT, "Bit length must be a power of two")
        if bit_length > MAX_RANGE_PROOF_BIT_LENGTH:
            raise ProofError(INVALID_ARGUMENT, "Bit length must be <= 64")
        self.bp_gens = BulletproofGens(bit_length, max_aggregation_factor)
        self.pc_gens = pc_gens

    def bit_length(self):
        return self.bp_gens.gens_capacity

    def max_aggregation_factor(self):
        return self.bp_gens.party_capacity

    def extension_degree(self):
        return self.pc_gens.extension_degree

    def h_base(self):
        return self.pc_gens.h_base

    def g_bases(self):
        return self.pc_gens.g_base_vec

    def gi_base(self):
        return self.bp_gens.g_iter(self.bit_length(), self.max_aggregation_factor())

    def hi_base(self):
        return self.bp_gens.h_iter(self.bit_length(), self.max_aggregation_factor())


class RangeStatement:
    """src/range_statement.rs:21-73. commitments: list[Point]; minimum_value_promises: list[int|None]."""

    def __init__(self, generators, commitments, minimum_value_promises, seed_nonce):
        n = len(commitments)
        if n == 0 or n & (n - 1):
            raise ProofError(INVALID_ARGUMENT, "Number of commitments must be a power of two")
        if len(minimum_value_promises) != n:
            raise ProofError(INVALID_ARGUMENT, "Incorrect number of minimum value promises")
        if generators.max_aggregation_factor() < n:
            raise ProofError(INVALID_ARGUMENT, "Not enough generators for this statement")
        if seed_nonce is not None and n > 1:
            raise ProofError(INVALID_ARGUMENT, "Mask recovery is not supported with an aggregated statement")
        self.generators = generators
        self.commitments = list(commitments)
        self.commitments_compressed = [c.compress() for c in commitments]
        self.minimum_value_promises = list(minimum_value_promises)
        self.seed_nonce = seed_nonce


class CommitmentOpening:
    """src/commitment_opening.rs:14-37."""

    def __init__(self, v, r):
        self.v = v
        self.r = list(r)


class RangeWitness:
    """src/range_witness.rs:24-41."""

    def __init__(self, openings):
        if not openings:
            raise ProofError(INVALID_LENGTH, "Vector openings cannot be empty")
        t = len(openings[0].r)
        for o in openings:
            if len(o.r) == 0:
                raise ProofError(INVALID_LENGTH, "Extended blinding factors cannot be empty")
            if len(o.r) != t:
                raise ProofError(INVALID_LENGTH, "Extended blinding factors must have consistent length")
        if not 1 <= t <= 6:
            raise ProofError(INVALID_ARGUMENT, "Extension degree not valid")
        self.openings = list(openings)
        self.extension_degree = t


# ---- utils ----

def nonce(seed_nonce, label, index_j, index_k):
    """src/utils/generic.rs:30-60 (keyed, personalised BLAKE2b-512, empty message) ->
    src/protocols/scalar_protocol.rs:32-36 (wide reduction)."""
    enc = label.encode()
    if len(enc) > 16:
        raise ProofError(INVALID_LENGTH, "Bad nonce label encoding")
    key = b"\x00" + C.scalar_bytes(seed_nonce)
    if index_j is not None:
        key += b"j" + int(index_j).to_bytes(4, "little")
    if index_k is not None:
        key += b"k" + int(index_k).to_bytes(4, "little")
    h = hashlib.blake2b(b"", digest_size=64, key=key, salt=b"", person=enc)
    return C.scalar_from_wide(h.digest())


def compute_generator_padding(bit_length, aggregation_factor, max_aggregation_factor):
    """src/utils/generic.rs:63-82."""
    pad = 2 * bit_length * max_aggregation_factor - 2 * bit_length * aggregation_factor
    if pad < 0:
        raise ProofError(SIZE_OVERFLOW)
    return pad


def random_not_zero(rng):
    """src/protocols/scalar_protocol.rs:23-30 with dalek Scalar::random = 64 bytes -> wide reduce."""
    v = 0
    while v == 0:
        v = C.scalar_from_wide(rng.fill_bytes(64))
    return v


# ---- transcript wrapper: src/transcripts.rs ----

def _validate_and_append_point(t, label, comp):
    # src/protocols/transcript_protocol.rs:48-61: identity encoding = 32 zero bytes
    if comp == bytes(32):
        raise ProofError(VERIFICATION_FAILED, "Identity element cannot be added to the transcript")
    t.append_message(label, comp)


def _challenge_scalar(t, label):
    # transcript_protocol.rs:67-78
    v = C.scalar_from_wide(t.challenge_bytes(label, 64))
    if v == 0:
        raise ProofError(VERIFICATION_FAILED, "Transcript challenge cannot be zero")
    return v


class RangeProofTranscript:
    def __init__(self, transcript, h_base_compressed, g_base_compressed, bit_length, extension_degree,
                 aggregation_factor, statement, witness, external_rng):
        # transcripts.rs:59-121
        t = transcript
        t.append_message(b"dom-sep", b"Bulletproofs+ Range Proof")
        _validate_and_append_point(t, b"H", h_base_compressed)
        for item in g_base_compressed:
            _validate_and_append_point(t, b"G", item)
        t.append_u64(b"N", bit_length)
        t.append_u64(b"T", extension_degree)
        t.append_u64(b"M", aggregation_factor)
        for item in statement.commitments_compressed:
            t.append_message(b"Ci", item)
        for item in statement.minimum_value_promises:
            t.append_u64(b"vi - minimum_value", item if item is not None else 0)
        self.bytes = None
        if witness is not None:
            wb = b""
            for o in witness.openings:
                wb += int(o.v).to_bytes(8, "little")
                for r in o.r:
                    wb += C.scalar_bytes(r)
            self.bytes = wb
        self.transcript = t
        self.external_rng = external_rng
        self.transcript_rng = self._build_rng()

    def _build_rng(self):
        # transcripts.rs:185-194
        b = self.transcript.build_rng()
        if self.bytes is not None:
            b = b.rekey_with_witness_bytes(b"witness", self.bytes)
        return b.finalize(self.external_rng)

    def challenges_y_z(self, a):
        _validate_and_append_point(self.transcript, b"A", a)
        self.transcript_rng = self._build_rng()
        return _challenge_scalar(self.transcript, b"y"), _challenge_scalar(self.transcript, b"z")

    def challenge_round_e(self, l, r):
        _validate_and_append_point(self.transcript, b"L", l)
        _validate_and_append_point(self.transcript, b"R", r)
        self.transcript_rng = self._build_rng()
        return _challenge_scalar(self.transcript, b"e")

    def challenge_final_e(self, a1, b):
        _validate_and_append_point(self.transcript, b"A1", a1)
        _validate_and_append_point(self.transcript, b"B", b)
        self.transcript_rng = self._build_rng()
        return _challenge_scalar(self.transcript, b"e")

    def to_verifier_rng(self, r1, s1, d1):
        self.transcript.append_message(b"r1", C.scalar_bytes(r1))
        self.transcript.append_message(b"s1", C.scalar_bytes(s1))
        for item in d1:
            self.transcript.append_message(b"d1", C.scalar_bytes(item))
        self.transcript_rng = self._build_rng()
        return self.transcript_rng


# ---- the proof ----

class RangeProof:
    """src/range_proof.rs:58-68. a, a1, b, li[], ri[] are 32-byte compressed encodings; scalars are ints."""

    def __init__(self, a, a1, b, r1, s1, d1, li, ri, extension_degree):
        self.a, self.a1, self.b = a, a1, b
        self.r1, self.s1, self.d1 = r1, s1, list(d1)
        self.li, self.ri = list(li), list(ri)
        self.extension_degree = extension_degree

    def __eq__(self, o):
        return self.to_bytes() == o.to_bytes()

    def to_bytes(self):
        """:1120-1150."""
        buf = bytes([self.extension_degree])
        for d in self.d1:
            buf += C.scalar_bytes(d)
        buf += self.a + self.a1 + self.b + C.scalar_bytes(self.r1) + C.scalar_bytes(self.s1)
        for l, r in zip(self.li, self.ri):
            buf += l + r
        return buf

    @staticmethod
    def from_bytes(data):
        """:1155-1257."""
        if len(data) < 1:
            raise ProofError(INVALID_LENGTH, "Serialized proof is too short")
        t = data[0]
        if not 1 <= t <= 6:
            raise ProofError(INVALID_ARGUMENT, "Extension degree not valid")
        body = data[1:]
        chunks = [body[i:i + 32] for i in range(0, len(body) - len(body) % 32, 32)]
        remainder = len(body) % 32
        pos = [0]

        def nxt():
            if pos[0] >= len(chunks):
                raise ProofError(INVALID_LENGTH, "Serialized proof is too short")
            c = chunks[pos[0]]
            pos[0] += 1
            return c

        def parse_scalar():
            v = C.scalar_from_canonical(nxt())
            if v is None:
                raise ProofError(INVALID_ARGUMENT, "Invalid parsing")
            return v

        d1 = [parse_scalar() for _ in range(t)]
        a, a1, b = nxt(), nxt(), nxt()
        r1, s1 = parse_scalar(), parse_scalar()
        rest = chunks[pos[0]:]
        li = [rest[2 * i] for i in range(len(rest) // 2)]
        ri = [rest[2 * i + 1] for i in range(len(rest) // 2)]
        if not li or not ri:
            raise ProofError(INVALID_LENGTH, "Serialized proof is too short")
        if len(rest) % 2 or remainder:
            raise ProofError(INVALID_LENGTH, "Unused data after deserialization")
        return RangeProof(a, a1, b, r1, s1, d1, li, ri, t)


def prove_with_rng(transcript, statement, witness, rng):
    """src/range_proof.rs:232-608. `transcript` is advanced in place; rng.fill_bytes(32) is drawn r+3 times."""
    gens = statement.generators
    bit_length = gens.bit_length()
    aggregation_factor = len(statement.commitments)
    extension_degree = gens.extension_degree()
    full_length = bit_length * aggregation_factor

    if len(witness.openings) != len(statement.commitments):  # :248
        raise ProofError(INVALID_LENGTH, "Witness openings and statement commitments do not match!")
    if witness.extension_degree != extension_degree:  # :256
        raise ProofError(INVALID_LENGTH, "Witness and statement extension degrees do not match!")
    for o in witness.openings:  # :264-271
        if bit_length < 64 and (o.v >> bit_length) > 0:
            raise ProofError(INVALID_LENGTH, "Value exceeds bit vector capacity!")
    for o, c in zip(witness.openings, statement.commitments):  # :275-284
        if gens.pc_gens.commit(o.v, o.r) != c:
            raise ProofError(INVALID_ARGUMENT, "Witness opening is invalid!")

    rpt = RangeProofTranscript(transcript, gens.pc_gens.h_base.compress(), gens.pc_gens.g_base_compressed_vec,
                               bit_length, extension_degree, aggregation_factor, statement, witness, rng)

    a_li, a_ri = [], []  # :300-322
    for vmin, o in zip(statement.minimum_value_promises, witness.openings):
        if vmin is not None:
            if o.v < vmin:
                raise ProofError(INVALID_ARGUMENT, "Minimum value is larger than value")
            off = o.v - vmin
        else:
            off = o.v
        for i in range(bit_length):
            bit = (off >> i) & 1
            a_li.append(bit)
            a_ri.append((bit - 1) % L)

    alpha = []  # :325-333
    for k in range(extension_degree):
        if statement.seed_nonce is not None:
            alpha.append(nonce(statement.seed_nonce, "alpha", None, k))
        else:
            alpha.append(random_not_zero(rpt.transcript_rng))
    compute_generator_padding(bit_length, aggregation_factor, gens.max_aggregation_factor())
    gi_all, hi_all = gens.gi_base(), gens.hi_base()
    # :339-345 -- static scalars interleaved (a_li[0], a_ri[0], a_li[1], ...) over G0,H0,G1,H1,...
    A = C.multiscalar_mul(alpha, gens.g_bases())
    for i in range(full_length):
        if a_li[i]:
            A = A + gi_all[i]
        else:
            A = A - hi_all[i]  # a_ri = -1
    a_comp = A.compress()

    y, z = rpt.challenges_y_z(a_comp)  # :348
    z_square = z * z % L
    y_powers = [1]  # :353-359
    for _ in range(full_length + 1):
        y_powers.append(y_powers[-1] * y % L)
    d = [z_square]  # :362-373
    for _ in range(1, bit_length):
        d.append(2 * d[-1] % L)
    for j in range(1, aggregation_factor):
        for i in range(bit_length):
            d.append(d[(j - 1) * bit_length + i] * z_square % L)

    a_li = [(x - z) % L for x in a_li]  # :376-381
    for i in range(full_length):
        a_ri[i] = (a_ri[i] + d[i] * y_powers[full_length - i] + z) % L
    z_even_powers = 1  # :382-392
    for o in witness.openings:
        z_even_powers = z_even_powers * z_square % L
        for k in range(len(o.r)):
            alpha[k] = (alpha[k] + z_even_powers * o.r[k] % L * y_powers[full_length + 1]) % L

    gi_base = gi_all[:full_length]  # :395-396
    hi_base = hi_all[:full_length]
    g_base = gens.g_bases()
    h_base = gens.h_base()

    li, ri = [], []
    n = full_length
    rnd = 0
    while n > 1:  # :409-538
        n //= 2
        a_lo, a_hi = a_li[:n], a_li[n:]
        b_lo, b_hi = a_ri[:n], a_ri[n:]
        gi_lo, gi_hi = gi_base[:n], gi_base[n:]
        hi_lo, hi_hi = hi_base[:n], hi_base[n:]
        if y_powers[n] == 0:
            raise ProofError(INVALID_ARGUMENT, "Cannot invert a zero valued Scalar")
        y_n_inverse = C.scalar_inv(y_powers[n])
        a_lo_offset = [s * y_n_inverse % L for s in a_lo]
        a_hi_offset = [s * y_powers[n] % L for s in a_hi]
        if statement.seed_nonce is not None:
            d_l = [nonce(statement.seed_nonce, "dL", rnd, k) for k in range(extension_degree)]
            d_r = [nonce(statement.seed_nonce, "dR", rnd, k) for k in range(extension_degree)]
        else:
            d_l = [random_not_zero(rpt.transcript_rng) for _ in range(extension_degree)]
            d_r = [random_not_zero(rpt.transcript_rng) for _ in range(extension_degree)]
        rnd += 1
        c_l = sum(a * yp % L * b for a, yp, b in zip(a_lo, y_powers[1:], b_hi)) % L
        c_r = sum(a * yp % L * b for a, yp, b in zip(a_hi, y_powers[n + 1:], b_lo)) % L
        Lp = C.multiscalar_mul([c_l] + d_l + a_lo_offset + b_hi, [h_base] + g_base + gi_hi + hi_lo)
        Rp = C.multiscalar_mul([c_r] + d_r + a_hi_offset + b_lo, [h_base] + g_base + gi_lo + hi_hi)
        li.append(Lp)
        ri.append(Rp)
        e = rpt.challenge_round_e(Lp.compress(), Rp.compress())
        e_square = e * e % L
        e_inverse = C.scalar_inv(e)
        e_inverse_square = e_inverse * e_inverse % L
        e_y_n_inverse = e * y_n_inverse % L
        gi_base = [lo * e_inverse + hi * e_y_n_inverse for lo, hi in zip(gi_lo, gi_hi)]
        hi_base = [lo * e + hi * e_inverse for lo, hi in zip(hi_lo, hi_hi)]
        a_li = [(lo * e + hi * e_inverse) % L for lo, hi in zip(a_lo, a_hi_offset)]
        a_ri = [(lo * e_inverse + hi * e) % L for lo, hi in zip(b_lo, b_hi)]
        for k in range(extension_degree):
            alpha[k] = (alpha[k] + d_l[k] * e_square + d_r[k] * e_inverse_square) % L

    r = random_not_zero(rpt.transcript_rng)  # :542-571
    s = random_not_zero(rpt.transcript_rng)
    if statement.seed_nonce is not None:
        dd = [nonce(statement.seed_nonce, "d", None, k) for k in range(extension_degree)]
        eta = [nonce(statement.seed_nonce, "eta", None, k) for k in range(extension_degree)]
    else:
        dd = [random_not_zero(rpt.transcript_rng) for _ in range(extension_degree)]
        eta = [random_not_zero(rpt.transcript_rng) for _ in range(extension_degree)]

    a1 = gi_base[0] * r + hi_base[0] * s + h_base * ((r * y_powers[1] % L * a_ri[0] + s * y_powers[1] % L * a_li[0]) % L)
    b = h_base * (r * y_powers[1] % L * s % L)
    for g, dk in zip(g_base, dd):
        a1 = a1 + g * dk
    for g, ek in zip(g_base, eta):
        b = b + g * ek
    e = rpt.challenge_final_e(a1.compress(), b.compress())  # :587
    e_square = e * e % L
    r1 = (r + a_li[0] * e) % L
    s1 = (s + a_ri[0] * e) % L
    d1 = [(eta[k] + dd[k] * e + alpha[k] * e_square) % L for k in range(extension_degree)]
    return RangeProof(a_comp, a1.compress(), b.compress(), r1, s1, d1,
                      [p.compress() for p in li], [p.compress() for p in ri], extension_degree)


def _consistency(statements, proofs):
    """verify_statements_and_generators_consistency, src/range_proof.rs:610-709."""
    if not statements:
        raise ProofError(INVALID_ARGUMENT, "Empty proof statements")
    if not proofs:
        raise ProofError(INVALID_ARGUMENT, "Empty proofs")
    if len(statements) != len(proofs):
        raise ProofError(INVALID_ARGUMENT, "Range statements and proofs length mismatch")
    first = statements[0]
    g_base_vec = first.generators.g_bases()
    h_base = first.generators.h_base()
    bit_length = first.generators.bit_length()
    max_mn = len(first.commitments) * bit_length
    max_index = 0
    t = first.generators.extension_degree()

    def ext_from_len(n):
        if not 1 <= n <= 6:
            raise ProofError(INVALID_ARGUMENT, "Extension degree not valid")
        return n

    if t != ext_from_len(len(proofs[0].d1)):
        raise ProofError(INVALID_ARGUMENT, "Inconsistent extension degree")
    for i in range(1, len(statements)):
        st, pr = statements[i], proofs[i]
        if len(g_base_vec) != len(st.generators.g_bases()) or any(a != b for a, b in zip(g_base_vec, st.generators.g_bases())):
            raise ProofError(INVALID_ARGUMENT, "Inconsistent G generator point in batch statement")
        if h_base != st.generators.h_base():
            raise ProofError(INVALID_ARGUMENT, "Inconsistent H generator point in batch statement")
        if bit_length != st.generators.bit_length():
            raise ProofError(INVALID_ARGUMENT, "Inconsistent bit length in batch statement")
        if t != st.generators.extension_degree() or t != ext_from_len(len(pr.d1)):
            raise ProofError(INVALID_ARGUMENT, "Inconsistent extension degree")
        fl = len(st.commitments) * st.generators.bit_length()
        if fl > max_mn:
            max_mn, max_index = fl, i
    max_st = statements[max_index]
    for i, st in enumerate(statements):
        for v in st.minimum_value_promises:
            if v is not None and bit_length < 64 and (v >> bit_length) > 0:
                raise ProofError(INVALID_LENGTH, "Minimum value promise exceeds bit vector capacity")
        if i == max_index:
            continue
        if any(a != b for a, b in zip(st.generators.gi_base(), max_st.generators.gi_base())):
            raise ProofError(INVALID_ARGUMENT, "Inconsistent Gi generator point vector in batch statement")
        if any(a != b for a, b in zip(st.generators.hi_base(), max_st.generators.hi_base())):
            raise ProofError(INVALID_ARGUMENT, "Inconsistent Hi generator point vector in batch statement")
    return max_mn, max_index


def verify_batch(transcripts, statements, proofs, action):
    """src/range_proof.rs:712-752, INCLUDING the first-chunk-only behaviour."""
    if not statements or not proofs or not transcripts:
        raise ProofError(INVALID_ARGUMENT, "Range statements or proofs length empty")
    if len(statements) != len(proofs):
        raise ProofError(INVALID_ARGUMENT, "Range statements and proofs length mismatch")
    if len(transcripts) != len(statements):
        raise ProofError(INVALID_ARGUMENT, "Range statements and transcripts length mismatch")
    n = MAX_RANGE_PROOF_BATCH_SIZE
    return verify(transcripts, statements[:n], proofs[:n], action)


def verify(transcripts, statements, proofs, action, trace=None, weights_override=None, check=True):
    """src/range_proof.rs:756-1065 (no batch-size limit). Returns list of masks (list[int] | None).
    `trace`, if a dict, receives every intermediate the GPU path is diffed against.
    Test-only knobs for the sharded (multi-GPU) form: `weights_override` replaces the weights this call would draw
    from its own chain (a shard of a larger batch gets its weights from the global chain); `check=False` skips the
    identity test so the caller can combine trace["accumulator"] across shards."""
    max_mn, max_index = _consistency(statements, proofs)
    first, max_st = statements[0], statements[max_index]
    g_base_vec = first.generators.g_bases()
    h_base = first.generators.h_base()
    bit_length = first.generators.bit_length()
    t = first.generators.extension_degree()
    g_bases_compressed = first.generators.pc_gens.g_base_compressed_vec
    h_base_compressed = first.generators.pc_gens.h_base_compressed
    two_n_minus_one = (pow(2, bit_length, L) - 1) % L

    g_base_scalars = [0] * t
    h_base_scalar = 0
    gi_base_scalars = [0] * max_mn
    hi_base_scalars = [0] * max_mn
    dynamic_scalars, dynamic_points = [], []
    masks = []

    weight_transcript = Transcript(b"Bulletproofs+ verifier weights")  # :811
    batch_challenges = []
    rng_outputs = []
    for proof, st, tr in zip(proofs, statements, transcripts):  # :816-850
        rpt = RangeProofTranscript(tr, h_base_compressed, g_bases_compressed, bit_length, t,
                                   len(st.commitments), st, None, NullRng())
        y, z = rpt.challenges_y_z(proof.a)
        round_e = [rpt.challenge_round_e(l, r) for l, r in zip(proof.li, proof.ri)]
        e = rpt.challenge_final_e(proof.a1, proof.b)
        batch_challenges.append((y, z, round_e, e))
        trng = rpt.to_verifier_rng(proof.r1, proof.s1, proof.d1)
        bts = trng.fill_bytes(32)
        rng_outputs.append(bts)
        weight_transcript.append_message(b"proof", bts)
    weight_rng = weight_transcript.build_rng().finalize(NullRng())  # :853
    weights = []

    for proof, st, (y, z, challenges, e) in zip(proofs, statements, batch_challenges):  # :856-1033
        def dec(b, what):
            p = C.decompress(b)
            if p is None:
                raise ProofError(INVALID_ARGUMENT, "Member '%s' was not the canonical encoding of a point" % what)
            return p

        a = dec(proof.a, "a")
        a1 = dec(proof.a1, "a1")
        b = dec(proof.b, "b")
        li = [dec(x, "L") for x in proof.li]
        ri = [dec(x, "L") for x in proof.ri]
        r1, s1, d1 = proof.r1, proof.s1, proof.d1
        aggregation_factor = len(st.commitments)
        full_length = aggregation_factor * bit_length
        rounds = len(li)
        if len(li) != len(ri):
            raise ProofError(INVALID_LENGTH, "Vector L length not equal to vector R length")
        if rounds >= 32:
            raise ProofError(SIZE_OVERFLOW)
        if (1 << rounds) != full_length:
            raise ProofError(INVALID_LENGTH, "Vector L/R length not adequate")

        weight = random_not_zero(weight_rng)  # :894
        if weights_override is not None:
            weight = weights_override[len(weights)]
        weights.append(weight)

        # :897-905 batch_invert([e_j..., y, y-1]); inverse(0)=0 convention
        inv_in = list(challenges) + [y, (y - 1) % L]
        inv_out = [C.scalar_inv(x) for x in inv_in]
        prod_inv = 1
        for x in inv_in:
            prod_inv = prod_inv * C.scalar_inv(x) % L
        challenges_inv_prod = prod_inv * y % L * ((y - 1) % L) % L
        y_1_inverse = inv_out.pop()
        y_inverse = inv_out.pop()
        challenges_inv = inv_out

        z_square = z * z % L
        e_square = e * e % L
        challenges_sq = [c * c % L for c in challenges]
        challenges_sq_inv = [c * c % L for c in challenges_inv]
        y_nm = pow(y, full_length, L)
        y_nm_1 = y_nm * y % L
        y_sum = y * ((y_nm - 1) % L) % L * y_1_inverse % L

        d = [z_square]  # :919-929
        for _ in range(1, bit_length):
            d.append(2 * d[-1] % L)
        for j in range(1, aggregation_factor):
            for i in range(bit_length):
                d.append(d[(j - 1) * bit_length + i] * z_square % L)

        d_sum = z_square  # :932-938
        d_sum_temp_z = z_square
        for _ in range(aggregation_factor.bit_length() - 1):
            d_sum = (d_sum + d_sum * d_sum_temp_z) % L
            d_sum_temp_z = d_sum_temp_z * d_sum_temp_z % L
        d_sum = d_sum * two_n_minus_one % L

        if action == VERIFY_ONLY:  # :941-969
            masks.append(None)
        else:
            if st.seed_nonce is not None:
                sn = st.seed_nonce
                temp = []
                for k in range(min(len(d1), t)):
                    m = (d1[k] - nonce(sn, "eta", None, k) - e * nonce(sn, "d", None, k)) % L * C.scalar_inv(e_square) % L
                    m = (m - nonce(sn, "alpha", None, k)) % L
                    for j, (csq, csqi) in enumerate(zip(challenges_sq, challenges_sq_inv)):
                        m = (m - csq * nonce(sn, "dL", j, k)) % L
                        m = (m - csqi * nonce(sn, "dR", j, k)) % L
                    m = m * C.scalar_inv(z_square * y_nm_1 % L) % L
                    temp.append(m)
                masks.append(temp)
            else:
                masks.append(None)
            if action == RECOVER_ONLY:
                continue

        y_inv_i = 1  # :972-1003
        y_nm_i = y_nm
        s = [challenges_inv_prod]
        for i in range(1, full_length):
            log_i = i.bit_length() - 1
            j = 1 << log_i
            s.append(s[i - j] * challenges_sq[rounds - log_i - 1] % L)
        r1_e = r1 * e % L
        s1_e = s1 * e % L
        e_square_z = e_square * z % L
        for i in range(min(full_length, max_mn)):
            g = r1_e * y_inv_i % L * s[i] % L
            h = s1_e * s[full_length - 1 - i] % L
            gi_base_scalars[i] = (gi_base_scalars[i] + weight * (g + e_square_z)) % L
            hi_base_scalars[i] = (hi_base_scalars[i] + weight * (h - e_square * (d[i] * y_nm_i + z))) % L
            y_inv_i = y_inv_i * y_inverse % L
            y_nm_i = y_nm_i * y_inverse % L

        z_even_powers = 1  # :1006-1015
        for vmin in st.minimum_value_promises:
            z_even_powers = z_even_powers * z_square % L
            weighted = weight * ((-e_square) % L * z_even_powers % L * y_nm_1 % L) % L
            dynamic_scalars.append(weighted)
            if vmin is not None:
                h_base_scalar = (h_base_scalar - weighted * vmin) % L
        dynamic_points.extend(st.commitments)

        h_base_scalar = (h_base_scalar + weight * (r1 * y % L * s1 + e_square * (y_nm_1 * z % L * d_sum + (z_square - z) * y_sum))) % L  # :1017
        for k in range(min(t, len(d1))):
            g_base_scalars[k] = (g_base_scalars[k] + weight * d1[k]) % L

        dynamic_scalars.append(weight * (-e) % L)  # :1022-1032
        dynamic_points.append(a1)
        dynamic_scalars.append((-weight) % L)
        dynamic_points.append(b)
        dynamic_scalars.append(weight * (-e_square) % L)
        dynamic_points.append(a)
        dynamic_scalars.extend(weight * (-e_square) % L * c % L for c in challenges_sq)
        dynamic_points.extend(li)
        dynamic_scalars.extend(weight * (-e_square) % L * c % L for c in challenges_sq_inv)
        dynamic_points.extend(ri)

    if trace is not None:
        trace.update(challenges=batch_challenges, rng_outputs=rng_outputs, weights=weights,
                     gi=list(gi_base_scalars), hi=list(hi_base_scalars), g=list(g_base_scalars),
                     h=h_base_scalar, dynamic_scalars=list(dynamic_scalars), max_mn=max_mn)
    if action == RECOVER_ONLY:
        return masks

    dynamic_scalars.extend(g_base_scalars)  # :1039-1042
    dynamic_points.extend(g_base_vec)
    dynamic_scalars.append(h_base_scalar)
    dynamic_points.append(h_base)

    compute_generator_padding(max_st.generators.bit_length(), len(max_st.commitments),
                              max_st.generators.max_aggregation_factor())
    gi_pts, hi_pts = max_st.generators.gi_base(), max_st.generators.hi_base()
    acc = C.multiscalar_mul(gi_base_scalars, gi_pts[:max_mn])
    acc = acc + C.multiscalar_mul(hi_base_scalars, hi_pts[:max_mn])
    acc = acc + C.multiscalar_mul(dynamic_scalars, dynamic_points)
    if trace is not None:
        trace["msm_result"] = acc.compress()
        trace["accumulator"] = acc
    if check and acc != Point.identity():
        raise ProofError(VERIFICATION_FAILED, "Range proof batch not valid")
    return masks
