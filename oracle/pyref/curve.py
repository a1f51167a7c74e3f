"""ORACLE (test infrastructure only -- never imported by the product path).

Big-int restatement of the curve25519-dalek 4.1.3 arithmetic the reference reaches
(dalek is NOT under /root/reference; pinned in supply-chain/config.toml:76-77).
Algorithms follow the published ristretto255 specification (RFC 9496 sections 4.2-4.3.4)
and are pinned by the RFC's known-answer vectors (tests/test_oracle_kats.py) plus,
in the build container only, libsodium's crypto_core_ristretto255_* (tests/golden/make_golden.py).

Reference call sites restated here:
  decompress   src/range_proof.rs:1067-1109  (CompressedRistretto::decompress)
  compress     src/range_proof.rs:348,499-504,587,598-605
  from_uniform src/ristretto.rs:48-52, src/generators/generators_chain.rs:43-49
  point ==     src/range_proof.rs:1057, :686-705
"""

P = 2**255 - 19
L = 2**252 + 27742317777372353535851937790883648493


def finv(x):
    return pow(x, P - 2, P)


D = (-121665 * finv(121666)) % P
SQRT_M1 = pow(2, (P - 1) // 4, P)


def is_negative(x):
    return (x % P) & 1


def fabs(x):
    x %= P
    return (P - x) % P if x & 1 else x


def sqrt_ratio_m1(u, v):
    """RFC 9496 4.2 SQRT_RATIO_M1: returns (was_square, non-negative sqrt(u/v) or sqrt(i*u/v))."""
    u %= P
    v %= P
    v3 = v * v % P * v % P
    v7 = v3 * v3 % P * v % P
    r = u * v3 % P * pow(u * v7 % P, (P - 5) // 8, P) % P
    check = v * r % P * r % P
    correct_sign = check == u
    flipped_sign = check == (P - u) % P
    flipped_sign_i = check == (P - u) * SQRT_M1 % P
    if flipped_sign or flipped_sign_i:
        r = r * SQRT_M1 % P
    r = fabs(r)
    return (correct_sign or flipped_sign), r


def _const_sqrt(x):
    ok, r = sqrt_ratio_m1(x, 1)
    assert ok
    return r


# RFC 9496 4.1 constants (derived here, checked against the RFC decimal values in the KAT test)
ONE_MINUS_D_SQ = (1 - D * D) % P
D_MINUS_ONE_SQ = (D - 1) * (D - 1) % P
SQRT_AD_MINUS_ONE = _const_sqrt((-D - 1) % P)  # a = -1
# the RFC fixes the sign of these two roots by value, not by parity:
if SQRT_AD_MINUS_ONE != 25063068953384623474111414158702152701244531502492656460079210482610430750235:
    SQRT_AD_MINUS_ONE = P - SQRT_AD_MINUS_ONE
INVSQRT_A_MINUS_D = sqrt_ratio_m1(1, (-1 - D) % P)[1]
if INVSQRT_A_MINUS_D != 54469307008909316920995813868745141605393597292927456921205312896311721017578:
    INVSQRT_A_MINUS_D = P - INVSQRT_A_MINUS_D


class Point:
    """Extended twisted Edwards coordinates (X:Y:Z:T), a = -1; equality is Ristretto equality."""

    __slots__ = ("X", "Y", "Z", "T")

    def __init__(self, X, Y, Z, T):
        self.X, self.Y, self.Z, self.T = X % P, Y % P, Z % P, T % P

    @staticmethod
    def identity():
        return Point(0, 1, 1, 0)

    def __add__(self, o):
        # add-2008-hwcd-3 (a = -1), the formula dalek's EdwardsPoint addition implements
        A = (self.Y - self.X) * (o.Y - o.X) % P
        B = (self.Y + self.X) * (o.Y + o.X) % P
        C = self.T * 2 * D % P * o.T % P
        Dd = self.Z * 2 * o.Z % P
        E, F, G, H = B - A, Dd - C, Dd + C, B + A
        return Point(E * F, G * H, F * G, E * H)

    def double(self):
        A = self.X * self.X % P
        B = self.Y * self.Y % P
        C = 2 * self.Z * self.Z % P
        H = A + B
        E = H - (self.X + self.Y) ** 2 % P
        G = A - B
        F = C + G
        return Point(E * F, G * H, F * G, E * H)

    def __neg__(self):
        return Point(-self.X, self.Y, self.Z, -self.T)

    def __sub__(self, o):
        return self + (-o)

    def __mul__(self, k):
        k %= L
        acc = Point.identity()
        if k == 0:
            return acc
        # fixed 4-bit windows, MSB first
        tbl = [Point.identity(), self]
        for _ in range(14):
            tbl.append(tbl[-1] + self)
        nibbles = []
        while k:
            nibbles.append(k & 15)
            k >>= 4
        for nib in reversed(nibbles):
            acc = acc.double().double().double().double()
            if nib:
                acc = acc + tbl[nib]
        return acc

    __rmul__ = __mul__

    def __eq__(self, o):
        # dalek RistrettoPoint::ct_eq: X1*Y2 == Y1*X2  |  X1*X2 == Y1*Y2
        return (self.X * o.Y - self.Y * o.X) % P == 0 or (self.X * o.X - self.Y * o.Y) % P == 0

    def __ne__(self, o):
        return not self.__eq__(o)

    def is_identity(self):
        return self == Point.identity()

    def compress(self):
        """RFC 9496 4.3.2 Encode."""
        x0, y0, z0, t0 = self.X, self.Y, self.Z, self.T
        u1 = (z0 + y0) * (z0 - y0) % P
        u2 = x0 * y0 % P
        _, invsqrt = sqrt_ratio_m1(1, u1 * u2 % P * u2 % P)
        den1 = invsqrt * u1 % P
        den2 = invsqrt * u2 % P
        z_inv = den1 * den2 % P * t0 % P
        ix0 = x0 * SQRT_M1 % P
        iy0 = y0 * SQRT_M1 % P
        enchanted = den1 * INVSQRT_A_MINUS_D % P
        rotate = is_negative(t0 * z_inv)
        if rotate:
            x, y, den_inv = iy0, ix0, enchanted
        else:
            x, y, den_inv = x0, y0, den2
        if is_negative(x * z_inv):
            y = (P - y) % P
        s = fabs(den_inv * (z0 - y))
        return s.to_bytes(32, "little")


def decompress(b):
    """RFC 9496 4.3.1 Decode; None on failure (dalek CompressedRistretto::decompress -> Option)."""
    if len(b) != 32:
        return None
    s = int.from_bytes(b, "little")
    if s >= P or (s & 1):
        return None
    ss = s * s % P
    u1 = (1 - ss) % P
    u2 = (1 + ss) % P
    u2_sqr = u2 * u2 % P
    v = (-(D * u1 % P * u1) - u2_sqr) % P
    was_square, invsqrt = sqrt_ratio_m1(1, v * u2_sqr % P)
    den_x = invsqrt * u2 % P
    den_y = invsqrt * den_x % P * v % P
    x = fabs(2 * s * den_x)
    y = u1 * den_y % P
    t = x * y % P
    if (not was_square) or is_negative(t) or y == 0:
        return None
    return Point(x, y, 1, t)


def _elligator_map(t):
    """RFC 9496 4.3.4 MAP."""
    r = SQRT_M1 * t % P * t % P
    u = (r + 1) * ONE_MINUS_D_SQ % P
    v = (-1 - r * D) % P * ((r + D) % P) % P
    was_square, s = sqrt_ratio_m1(u, v)
    s_prime = (P - fabs(s * t)) % P
    if not was_square:
        s = s_prime
        c = r
    else:
        c = P - 1
    N = (c * ((r - 1) % P) % P * D_MINUS_ONE_SQ - v) % P
    w0 = 2 * s * v % P
    w1 = N * SQRT_AD_MINUS_ONE % P
    w2 = (1 - s * s) % P
    w3 = (1 + s * s) % P
    return Point(w0 * w3, w2 * w1, w1 * w3, w0 * w2)


def from_uniform_bytes(b):
    """RistrettoPoint::from_uniform_bytes (RFC 9496 4.3.4 one-way map): 64 bytes -> point."""
    assert len(b) == 64
    r0 = (int.from_bytes(b[:32], "little") & ((1 << 255) - 1)) % P
    r1 = (int.from_bytes(b[32:], "little") & ((1 << 255) - 1)) % P
    return _elligator_map(r0) + _elligator_map(r1)


# Ed25519 basepoint = RISTRETTO_BASEPOINT_POINT (src/ristretto.rs:70)
_By = 4 * finv(5) % P
_Bx = _const_sqrt((_By * _By - 1) * finv(D * _By * _By + 1) % P)
if _Bx & 1:
    _Bx = P - _Bx
BASEPOINT = Point(_Bx, _By, 1, _Bx * _By)


def multiscalar_mul(scalars, points):
    """Sum of scalar*point; algorithm irrelevant to the result (group element)."""
    acc = Point.identity()
    for s, p in zip(scalars, points):
        s %= L
        if s == 0:
            continue
        acc = acc + p * s
    return acc


# ---- scalars (dalek Scalar) ----

def scalar_from_wide(b):
    """Scalar::from_bytes_mod_order_wide (src/protocols/transcript_protocol.rs:70)."""
    assert len(b) == 64
    return int.from_bytes(b, "little") % L


def scalar_from_canonical(b):
    """Scalar::from_canonical_bytes -> None if not canonical (src/range_proof.rs:1165)."""
    if len(b) != 32:
        return None
    v = int.from_bytes(b, "little")
    return v if v < L else None


def scalar_bytes(s):
    return (s % L).to_bytes(32, "little")


def scalar_inv(s):
    """Scalar::invert; inverse(0) = 0 as a Fermat inversion gives."""
    return pow(s % L, L - 2, L)
