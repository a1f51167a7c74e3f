"""ORACLE (test infrastructure only).

Restatement of merlin 3.0.0 (Transcript, TranscriptRng) on STROBE-128/Keccak-f[1600].
merlin is not under /root/reference (pinned supply-chain/config.toml:132-133); the
published algorithm (merlin.cool, STROBE v1.0.2 spec) is restated and pinned by merlin's
own `equivalence_simple` known-answer test (tests/test_oracle_kats.py).

Reference call sites: src/transcripts.rs:59-200, src/protocols/transcript_protocol.rs:39-79,
src/range_proof.rs:811,849,853.
"""

_MASK = (1 << 64) - 1
_RC = [
    0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000,
    0x000000000000808B, 0x0000000080000001, 0x8000000080008081, 0x8000000000008009,
    0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
    0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003,
    0x8000000000008002, 0x8000000000000080, 0x000000000000800A, 0x800000008000000A,
    0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008,
]
_ROT = [
    [0, 36, 3, 41, 18],
    [1, 44, 10, 45, 2],
    [62, 6, 43, 15, 61],
    [28, 55, 25, 21, 56],
    [27, 20, 39, 8, 14],
]

PERMUTATION_COUNT = 0  # instrumentation only


def _rol(v, n):
    n %= 64
    return ((v << n) | (v >> (64 - n))) & _MASK if n else v


def keccak_f1600(lanes):
    """lanes: list of 25 u64, index x + 5*y. In place."""
    global PERMUTATION_COUNT
    PERMUTATION_COUNT += 1
    A = lanes
    for rnd in range(24):
        C = [A[x] ^ A[x + 5] ^ A[x + 10] ^ A[x + 15] ^ A[x + 20] for x in range(5)]
        Dv = [C[(x - 1) % 5] ^ _rol(C[(x + 1) % 5], 1) for x in range(5)]
        for i in range(25):
            A[i] ^= Dv[i % 5]
        B = [0] * 25
        for x in range(5):
            for y in range(5):
                B[y + 5 * ((2 * x + 3 * y) % 5)] = _rol(A[x + 5 * y], _ROT[x][y])
        for y in range(5):
            for x in range(5):
                A[x + 5 * y] = B[x + 5 * y] ^ ((~B[(x + 1) % 5 + 5 * y]) & B[(x + 2) % 5 + 5 * y] & _MASK)
        A[0] ^= _RC[rnd]
    return A


def _permute_bytes(state):
    lanes = [int.from_bytes(state[8 * i:8 * i + 8], "little") for i in range(25)]
    keccak_f1600(lanes)
    for i in range(25):
        state[8 * i:8 * i + 8] = lanes[i].to_bytes(8, "little")


STROBE_R = 166
FLAG_I, FLAG_A, FLAG_C, FLAG_T, FLAG_M, FLAG_K = 1, 2, 4, 8, 16, 32


class Strobe128:
    def __init__(self, protocol_label=None):
        self.state = bytearray(200)
        self.pos = 0
        self.pos_begin = 0
        self.cur_flags = 0
        if protocol_label is not None:
            self.state[0:6] = bytes([1, STROBE_R + 2, 1, 0, 1, 96])
            self.state[6:18] = b"STROBEv1.0.2"
            _permute_bytes(self.state)
            self.meta_ad(protocol_label, False)

    def clone(self):
        c = Strobe128()
        c.state = bytearray(self.state)
        c.pos, c.pos_begin, c.cur_flags = self.pos, self.pos_begin, self.cur_flags
        return c

    def to_bytes(self):
        """203-byte snapshot: state || pos || pos_begin || cur_flags (the C ABI's transcript_state)."""
        return bytes(self.state) + bytes([self.pos, self.pos_begin, self.cur_flags])

    def _run_f(self):
        self.state[self.pos] ^= self.pos_begin
        self.state[self.pos + 1] ^= 0x04
        self.state[STROBE_R + 1] ^= 0x80
        _permute_bytes(self.state)
        self.pos = 0
        self.pos_begin = 0

    def _absorb(self, data):
        for b in data:
            self.state[self.pos] ^= b
            self.pos += 1
            if self.pos == STROBE_R:
                self._run_f()

    def _overwrite(self, data):
        for b in data:
            self.state[self.pos] = b
            self.pos += 1
            if self.pos == STROBE_R:
                self._run_f()

    def _squeeze(self, n):
        out = bytearray(n)
        for i in range(n):
            out[i] = self.state[self.pos]
            self.state[self.pos] = 0
            self.pos += 1
            if self.pos == STROBE_R:
                self._run_f()
        return bytes(out)

    def _begin_op(self, flags, more):
        if more:
            assert self.cur_flags == flags
            return
        assert flags & FLAG_T == 0
        old_begin = self.pos_begin
        self.pos_begin = self.pos + 1
        self.cur_flags = flags
        self._absorb(bytes([old_begin, flags]))
        force_f = (flags & (FLAG_C | FLAG_K)) != 0
        if force_f and self.pos != 0:
            self._run_f()

    def meta_ad(self, data, more):
        self._begin_op(FLAG_M | FLAG_A, more)
        self._absorb(data)

    def ad(self, data, more):
        self._begin_op(FLAG_A, more)
        self._absorb(data)

    def prf(self, n, more):
        self._begin_op(FLAG_I | FLAG_A | FLAG_C, more)
        return self._squeeze(n)

    def key(self, data, more):
        self._begin_op(FLAG_A | FLAG_C, more)
        self._overwrite(data)


def _u32le(n):
    return int(n).to_bytes(4, "little")


class Transcript:
    """merlin::Transcript."""

    def __init__(self, label=None):
        if label is None:
            self.strobe = None
        else:
            self.strobe = Strobe128(b"Merlin v1.0")
            self.append_message(b"dom-sep", label)

    def clone(self):
        t = Transcript()
        t.strobe = self.strobe.clone()
        return t

    def append_message(self, label, message):
        self.strobe.meta_ad(label, False)
        self.strobe.meta_ad(_u32le(len(message)), True)
        self.strobe.ad(message, False)

    def append_u64(self, label, x):
        self.append_message(label, int(x).to_bytes(8, "little"))

    def challenge_bytes(self, label, n):
        self.strobe.meta_ad(label, False)
        self.strobe.meta_ad(_u32le(n), True)
        return self.strobe.prf(n, False)

    def build_rng(self):
        return TranscriptRngBuilder(self.strobe.clone())


class TranscriptRngBuilder:
    def __init__(self, strobe):
        self.strobe = strobe

    def rekey_with_witness_bytes(self, label, witness):
        self.strobe.meta_ad(label, False)
        self.strobe.meta_ad(_u32le(len(witness)), True)
        self.strobe.key(witness, False)
        return self

    def finalize(self, rng):
        """rng: object with fill_bytes(n) -> bytes (the external RNG)."""
        random_bytes = rng.fill_bytes(32)
        self.strobe.meta_ad(b"rng", False)
        self.strobe.key(random_bytes, False)
        return TranscriptRng(self.strobe)


class TranscriptRng:
    def __init__(self, strobe):
        self.strobe = strobe

    def fill_bytes(self, n):
        self.strobe.meta_ad(_u32le(n), False)
        return self.strobe.prf(n, False)


class NullRng:
    """src/utils/nullrng.rs:16-40 -- every output byte is zero."""

    def fill_bytes(self, n):
        return bytes(n)


class ByteStreamRng:
    """Deterministic external RNG for tests: hands out a pre-agreed byte string 32 bytes at a time
    (exactly r+3 draws per proof)."""

    def __init__(self, data):
        self.data = data
        self.off = 0

    def fill_bytes(self, n):
        if self.off + n > len(self.data):
            raise ValueError("external rng bytes exhausted")
        out = self.data[self.off:self.off + n]
        self.off += n
        return out
