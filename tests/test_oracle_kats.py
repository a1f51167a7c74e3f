"""CPU suite: pins the oracle (pyref) against public known-answer vectors and the committed libsodium fixtures.
(the reference's own tests hold no golden bytes -> these KATs are what pins the oracle.)"""
import hashlib
import json
import os

from oracle.pyref import curve as C
from oracle.pyref import merlin as M
from oracle.pyref import protocol as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_rfc9496_constants():
    assert C.SQRT_M1 == 19681161376707505956807079304988542015446066515923890162744021073123829784752
    assert C.D == 37095705934669439343138083508754565189542113879843219016388785533085940283555
    assert C.ONE_MINUS_D_SQ == 1159843021668779879193775521855586647937357759715417654439879720876111806838
    assert C.D_MINUS_ONE_SQ == 40440834346308536858101042469323190826248399146238708352240133220865137265952
    assert C.SQRT_AD_MINUS_ONE ** 2 % C.P == (-C.D - 1) % C.P
    assert C.INVSQRT_A_MINUS_D ** 2 * (-1 - C.D) % C.P == 1


def test_rfc9496_generator_multiples():
    # RFC 9496 appendix A.1
    want = ["0000000000000000000000000000000000000000000000000000000000000000",
            "e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76",
            "6a493210f7499cd17fecb510ae0cea23a110e8d5b901f8acadd3095c73a3b919",
            "94741f5d5d52755ece4f23f044ee27d5d1ea1e2bd196b462166b16152a9d0259",
            "da80862773358b466ffadfe0b3293ab3d9fd53c5ea6c955358f568322daf6a57",
            "e882b131016b52c1d3337080187cf768423efccbb517bb495ab812c4160ff44e"]
    for k, w in enumerate(want):
        assert (C.BASEPOINT * k).compress().hex() == w
        if k:
            assert C.decompress(bytes.fromhex(w)) == C.BASEPOINT * k


def test_rfc9496_invalid_encodings():
    # RFC 9496 appendix A.2 (a sample of each class: non-canonical, negative, non-square, s = -1 ...)
    bad = ["00ffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff",
           "ffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff7f",
           "f3ffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff7f",
           "edffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff7f",
           "0100000000000000000000000000000000000000000000000000000000000000",
           "01ffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff7f",
           "ed57ffd8c914fb201471d1c3d245ce3c746fcbe63a3679d51b6a516ebebe0e20",
           "c34c4e1826e5d403b78e246e88aa051c36ccf0aafebffe137d148a2bf9104562",
           "26948d35ca62e643e26a83177332e6b6afeb9d08e4268b650f1f5bbd8d81d371",
           "4eac077a713c57b4f4397629a4145982c661f48044dd3f96427d40b147d9742f",
           "ecffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff7f",
           "47cfc5497c53dc8e61c91d17fd626ffb1c49e2bca94eed052281b510b1117a24"]
    for b in bad:
        assert C.decompress(bytes.fromhex(b)) is None, b


def test_rfc9496_one_way_map():
    h = hashlib.sha512(b"Ristretto is traditionally a short shot of espresso coffee").digest()
    assert C.from_uniform_bytes(h).compress().hex() == "3066f82a1a747d45120d1740f14358531a8f04bbffe6a819f86dfe50f44a0a46"


def test_merlin_equivalence_simple():
    # merlin 3.0.0 src/transcript.rs `equivalence_simple`
    t = M.Transcript(b"test protocol")
    t.append_message(b"some label", b"some data")
    assert t.challenge_bytes(b"challenge", 32).hex() == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"


def test_keccak_against_hashlib():
    # rebuild SHA3-256 on the oracle's permutation
    def sha3_256(msg):
        st = bytearray(200)
        rate = 136
        msg = bytearray(msg) + b"\x06"
        msg += bytes(-len(msg) % rate)
        msg[-1] |= 0x80
        for off in range(0, len(msg), rate):
            for i in range(rate):
                st[i] ^= msg[off + i]
            M._permute_bytes(st)
        return bytes(st[:32])
    for m in [b"", b"abc", bytes(range(200)) * 3]:
        assert sha3_256(m) == hashlib.sha3_256(m).digest()


def test_libsodium_fixture():
    k = json.load(open(os.path.join(GOLD, "kat_libsodium.json")))
    for h, p in k["from_hash"]:
        assert C.from_uniform_bytes(bytes.fromhex(h)).compress().hex() == p
    for s, p, q in k["scalarmult"]:
        assert (C.decompress(bytes.fromhex(p)) * int.from_bytes(bytes.fromhex(s), "little")).compress().hex() == q
    for a, b, c in k["add"]:
        assert (C.decompress(bytes.fromhex(a)) + C.decompress(bytes.fromhex(b))).compress().hex() == c


def test_anchor_fixture():
    a = json.load(open(os.path.join(GOLD, "protocol_small.json")))["anchors"]
    assert [p.compress().hex() for p in O.ristretto_masking_basepoints()] == a["masking_basepoints"]
    og = O.BulletproofGens(64, 2)
    assert og.g_vec[0][0].compress().hex() == a["G[0][0]"] and og.h_vec[1][0].compress().hex() == a["H[1][0]"]
    assert C.scalar_bytes(O.nonce(1, "dL", 3, 2)).hex() == a["nonce(1,dL,3,2)"]
    # the prefixes/suffixes SURVEY 8c quotes for these anchors
    assert a["masking_basepoints"][0].startswith("044fad91") and a["masking_basepoints"][5].endswith("5b937b5e")
    assert a["G[0][0]"].startswith("fc3b2580") and a["H[0][63]"].endswith("df4e8263")
    assert a["nonce(1,alpha,None,0)"].startswith("48b5fc57") and a["nonce(1,eta,None,None)"].endswith("cc10230c")
