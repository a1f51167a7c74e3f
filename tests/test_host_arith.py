"""CPU suite: the product's shared host/device arithmetic headers (csrc/*.h), host-compiled, vs the oracle.
The same headers are what hipcc compiles for gfx950; the -m gpu tests then cover the device build."""
import ctypes
import hashlib
import importlib

import pytest

from oracle.pyref import curve as C
from oracle.pyref import merlin as M
from oracle.pyref import protocol as O

P, L = C.P, C.L


@pytest.fixture(scope="module")
def ht():
    pkg = importlib.import_module("bulletproofs-plus_amd")
    return ctypes.CDLL(pkg._build.build_hosttest())


def _r(tag, i, n=32):
    return hashlib.shake_256(b"%s%d" % (tag, i)).digest(n)


def _buf(n=32):
    return ctypes.create_string_buffer(n)


EDGE_FE = [0, 1, 2, P - 1, P - 2, P, P + 1, 2**255 - 1, 2**255 - 20, 2**254, 19, 2**26 - 1, 2**51]
EDGE_SC = [0, 1, L - 1, L - 2, 2, L, L + 1, 2**256 - 1, 2**252, 2**253 - 1]


def test_field(ht):
    vals = [int.from_bytes(_r(b"f", i), "little") & ((1 << 255) - 1) for i in range(60)] + EDGE_FE
    for a in vals:
        ab = a.to_bytes(32, "little")
        for b in vals[:5] + EDGE_FE:
            o = _buf()
            ht.ht_fe_mul(ab, b.to_bytes(32, "little"), o)
            assert int.from_bytes(o.raw, "little") == a * b % P
            ht.ht_fe_addsubmul(ab, b.to_bytes(32, "little"), o)
            assert int.from_bytes(o.raw, "little") == (a + b) * (a - b) % P
        o = _buf()
        ht.ht_fe_sq(ab, o)
        assert int.from_bytes(o.raw, "little") == a * a % P
        ht.ht_fe_invert(ab, o)
        assert int.from_bytes(o.raw, "little") == pow(a, P - 2, P)


def test_scalar(ht):
    vals = [int.from_bytes(_r(b"s", i), "little") for i in range(60)] + EDGE_SC
    for a in vals:
        ab = a.to_bytes(32, "little")
        for b in vals[:4] + EDGE_SC:
            o = _buf()
            ht.ht_sc_mul(ab, b.to_bytes(32, "little"), o)
            assert int.from_bytes(o.raw, "little") == a * b % L
            if a < L and b < L:
                s, d = _buf(), _buf()
                ht.ht_sc_addsub(ab, b.to_bytes(32, "little"), s, d)
                assert int.from_bytes(s.raw, "little") == (a + b) % L and int.from_bytes(d.raw, "little") == (a - b) % L
        for k, b in enumerate(vals[:3] + EDGE_SC[:4]):  # unpacked (nine-limb) products: (a b) c with a lazy middle, a 2^e
            c = vals[(k * 7 + 3) % len(vals)]
            o, o2 = _buf(), _buf()
            e = (a + 5 * k) % 64
            ht.ht_sc9_mul3(ab, b.to_bytes(32, "little"), c.to_bytes(32, "little"), e, o, o2)
            assert int.from_bytes(o.raw, "little") == a * b * c % L
            assert int.from_bytes(o2.raw, "little") == a * (1 << e) % L
            d = vals[(k * 11 + 1) % len(vals)]
            ht.ht_sc9_mul2(ab, b.to_bytes(32, "little"), c.to_bytes(32, "little"), d.to_bytes(32, "little"), o)
            assert int.from_bytes(o.raw, "little") == (a * b + c * d) % L
        assert ht.ht_sc_canonical(ab) == (1 if a < L else 0)
        o = _buf()
        ht.ht_sc_invert((a % L).to_bytes(32, "little"), o)
        assert int.from_bytes(o.raw, "little") == pow(a % L, L - 2, L)
    for w in [_r(b"w", i, 64) for i in range(300)] + [b"\xff" * 64, bytes(64), L.to_bytes(64, "little"),
                                                       (L * L - 1).to_bytes(64, "little"), (L - 1).to_bytes(64, "little"),
                                                       ((L << 256) - 1).to_bytes(64, "little"), (2**256 - 1).to_bytes(64, "little"),
                                                       ((2**256 - 1) << 256).to_bytes(64, "little")]:
        o = _buf()
        ht.ht_sc_wide(w, o)
        assert int.from_bytes(o.raw, "little") == int.from_bytes(w, "little") % L
        ht.ht_host_wide(w, o)  # the host weight chains' four-limb form
        assert int.from_bytes(o.raw, "little") == int.from_bytes(w, "little") % L


def test_ristretto(ht):
    for i in range(24):
        u = _r(b"u", i, 64)
        o = _buf()
        ht.ht_from_uniform(u, o)
        pt = C.from_uniform_bytes(u)
        assert o.raw == pt.compress()
        o2 = _buf()
        assert ht.ht_decompress_compress(o.raw, o2) == 1 and o2.raw == o.raw
        o5 = _buf()
        assert ht.ht_decompress_lean(o.raw, o5) == 1 and o5.raw == o.raw
        o4 = _buf()
        assert ht.ht_from_niels(o.raw, 0, o4) == 1 and o4.raw == (pt * 2).compress()  # ge_from_niels(P) + P
        assert ht.ht_from_niels(o.raw, 1, o4) == 1 and o4.raw == bytes(32)            # ge_from_niels(-P) + P
        # the MSM kernels' way of applying a term's sign (niels_load_swapped + ge_madd_swapped): 2P + P, 2P - P
        assert ht.ht_madd_swapped(o.raw, 0, o4) == 1 and o4.raw == (pt * 3).compress()
        assert ht.ht_madd_swapped(o.raw, 1, o4) == 1 and o4.raw == o.raw
        # a bucket's first term: accumulator = +-P by ge_from_niels_first, then + P
        assert ht.ht_from_niels_first(o.raw, 0, o4) == 1 and o4.raw == (pt * 2).compress()
        assert ht.ht_from_niels_first(o.raw, 1, o4) == 1 and o4.raw == bytes(32)
        k = int.from_bytes(_r(b"k", i), "little") % L
        o3 = _buf()
        assert ht.ht_scalarmult(k.to_bytes(32, "little"), o.raw, o3) == 1 and o3.raw == (pt * k).compress()
    for i in range(120):  # random strings: accept/reject must agree with the oracle
        s = _r(b"bad", i)
        assert (ht.ht_decompress_compress(s, _buf()) == 1) == (C.decompress(s) is not None)
        assert (ht.ht_decompress_lean(s, _buf()) == 1) == (C.decompress(s) is not None)
    for s in [P.to_bytes(32, "little"), (1).to_bytes(32, "little"), b"\xff" * 32, (2**255).to_bytes(32, "little")]:
        assert ht.ht_decompress_compress(s, _buf()) == 0 and ht.ht_decompress_lean(s, _buf()) == 0
    assert ht.ht_is_identity(bytes(32)) == 1 and ht.ht_is_identity(C.BASEPOINT.compress()) == 0


def test_merlin_and_blake2b(ht):
    o, st = _buf(32), _buf(203)
    ht.ht_merlin_kat(b"test protocol", 13, b"some label", 10, b"some data", 9, b"challenge", 9, o, 32, st)
    assert o.raw.hex() == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"
    t = M.Transcript(b"test protocol")
    t.append_message(b"some label", b"some data")
    t.challenge_bytes(b"challenge", 32)
    assert st.raw == t.strobe.to_bytes()

    class R:
        def fill_bytes(self, n):
            return b"\x07" * n
    for wit in [b"", b"w" * 40, b"z" * 200]:
        out = _buf(100)
        ht.ht_merlin_rng(st.raw, wit, len(wit), b"\x07" * 32, out, 100)
        b = t.build_rng()
        if wit:
            b.rekey_with_witness_bytes(b"witness", wit)
        assert b.finalize(R()).fill_bytes(100) == out.raw
    for n in [165, 166, 167, 400]:  # absorbs that straddle the STROBE rate
        msg = (bytes(range(256)) * 2)[:n]
        out = _buf(70)
        ht.ht_merlin_kat(b"x", 1, b"lbl", 3, msg, n, b"c", 1, out, 70, st)
        t2 = M.Transcript(b"x")
        t2.append_message(b"lbl", msg)
        assert t2.challenge_bytes(b"c", 70) == out.raw
    for (sn, lab, j, k) in [(1, "alpha", None, 0), (1, "dL", 3, 2), (1, "eta", None, None), (L - 1, "dR", 7, 5)]:
        key = b"\x00" + sn.to_bytes(32, "little") + (b"j" + j.to_bytes(4, "little") if j is not None else b"") + \
            (b"k" + k.to_bytes(4, "little") if k is not None else b"")
        out = _buf(64)
        ht.ht_blake2b(key, len(key), lab.encode(), len(lab), out)
        assert int.from_bytes(out.raw, "little") % L == O.nonce(sn, lab, j, k)
        # the word-level form the kernels use (blake2b.h: nonce_hash_words -- the key block assembled by shifts, no byte array)
        out2 = _buf(64)
        ht.ht_nonce_words(sn.to_bytes(32, "little"), lab.encode(), len(lab), -1 if j is None else j, -1 if k is None else k, out2)
        assert out2.raw == out.raw
    for (j, k) in [(None, 0x5eadbeef), (0x01020304, 0x70b0c0d0), (0, None), (0x7fffffff, 0x7fffffff)]:  # every byte lane of the two indices
        sn = int.from_bytes(bytes(range(101, 133)), "little") % L
        out2 = _buf(64)
        ht.ht_nonce_words(sn.to_bytes(32, "little"), b"dR", 2, -1 if j is None else j, -1 if k is None else k, out2)
        assert int.from_bytes(out2.raw, "little") % L == O.nonce(sn, "dR", j, k)


def test_weight_chains_lockstep(ht):
    """host batch-weight transcript (src/range_proof.rs:811,849,853,894): scalar and lock-step vector instances of
    chain_host.h against the oracle's Merlin; every chain of a bundle must come out as if it had run alone"""
    from oracle.pyref import protocol as O
    import ctypes
    ht.ht_weight_chains.restype = ctypes.c_int
    for n in (1, 3, 40, 129):  # 129 proofs cross several sponge blocks in both the absorb and the squeeze phase
        for width in (1, 4, 8):
            rng = b"".join(_r(b"chain%d" % width, 1000 * n + i) for i in range(width * n))
            out = ctypes.create_string_buffer(32 * width * n)
            rc = ht.ht_weight_chains(rng, n, width, out)
            if rc == 0:
                pytest.skip("CPU without the vector instruction set of the %d-wide instance" % width)
            assert rc == 1
            for k in range(width):
                t = M.Transcript(b"Bulletproofs+ verifier weights")
                for i in range(n):
                    t.append_message(b"proof", rng[(k * n + i) * 32:(k * n + i + 1) * 32])
                wr = t.build_rng().finalize(O.NullRng())
                want = b"".join(C.scalar_bytes(O.random_not_zero(wr)) for _ in range(n))
                assert out.raw[k * n * 32:(k + 1) * n * 32] == want, (n, width, k)


def test_wide_chains_and_bit_interleaving(ht):
    """round 6: (1) the WIDE host chains (chain_host.h: the sponge's 64 PRF bytes per proof handed to the device, which reduces them:
    option chain = 2) -- single and lock-step forms -- reduced mod l here must be the weights of the ordinary forms, i.e. the oracle's;
    (2) wkeccak.h's bit interleaving: even / odd halves of a word and back, and a rotation by r as two 32-bit rotations (by r / 2; an
    odd r swaps the halves and rotates the new even half one further) -- the identity the one-wavefront Keccak-f rests on"""
    import ctypes
    from oracle.pyref import protocol as O
    ht.ht_wide_chains.restype = ctypes.c_int
    for n in (1, 4, 40, 129):
        for width in (1, 4, 8):
            rng = b"".join(_r(b"wide%d" % width, 1000 * n + i) for i in range(width * n))
            out = ctypes.create_string_buffer(64 * width * n)
            rc = ht.ht_wide_chains(rng, n, width, out)
            if rc == 0:
                continue  # (CPU without that vector instruction set)
            assert rc == 1
            for k in range(width):
                t = M.Transcript(b"Bulletproofs+ verifier weights")
                for i in range(n):
                    t.append_message(b"proof", rng[(k * n + i) * 32:(k * n + i + 1) * 32])
                wr = t.build_rng().finalize(O.NullRng())
                for i in range(n):
                    wide = out.raw[(k * n + i) * 64:(k * n + i + 1) * 64]
                    assert wide == wr.fill_bytes(64), (n, width, k, i)
    ht.ht_wk_halves.argtypes = [ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint32)]
    ht.ht_wk_word.restype = ctypes.c_uint64
    ht.ht_wk_word.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
    rol32 = lambda v, k: ((v << (k % 32)) | (v >> ((32 - k % 32) % 32))) & 0xFFFFFFFF if k % 32 else v
    words = [0, 1, 2, 1 << 63, 0x0123456789ABCDEF, 0xFFFFFFFF00000000, 0xAAAAAAAAAAAAAAAA] + [int.from_bytes(_r(b"wk", i)[:8], "little") for i in range(20)]
    for w in words:
        h = (ctypes.c_uint32 * 2)()
        ht.ht_wk_halves(w, h)
        even = sum(((w >> (2 * j)) & 1) << j for j in range(32))
        odd = sum(((w >> (2 * j + 1)) & 1) << j for j in range(32))
        assert (h[0], h[1]) == (even, odd)
        assert ht.ht_wk_word(h[0], h[1]) == w
        for r in (1, 2, 3, 27, 28, 44, 55, 62, 63):
            rot = ((w << r) | (w >> (64 - r))) & (2**64 - 1)
            k = r // 2
            e2, o2 = (rol32(h[0], k), rol32(h[1], k)) if r % 2 == 0 else (rol32(h[1], k + 1), rol32(h[0], k))
            assert ht.ht_wk_word(e2, o2) == rot, (hex(w), r)


def test_scalar_inversions_agree(ht):
    """divsteps inversion (sc_invert_vartime_plain) and the bit-at-a-time binary GCD it replaced, against pow(a, -1, l):
    random values, small values, values next to l and powers of two (long runs of zero bits stress the divstep batching)"""
    vals = [int.from_bytes(_r(b"inv", i), "little") % L for i in range(300)]
    vals += [1, 2, 3, 5, L - 1, L - 2, (L - 1) // 2, (L + 1) // 2, 2**252, 2**252 - 1, 2**128, 2**200 + 1, 2**30, 2**30 - 1,
             2**60, 2**31, 0]
    for a in vals:
        for which in (0, 1):
            o = _buf()
            ht.ht_sc_invert_plain(a.to_bytes(32, "little"), which, o)
            got = int.from_bytes(o.raw, "little")
            assert got == (pow(a, -1, L) if a else 0), (hex(a), which)


def test_scalar_recodings(ht, monkeypatch):
    """signed-digit recodings used by the MSM (uneven windows of c and c-1 bits, 253 bits in total) and by the prover's
    fixed-base tables: the digits must add up to the scalar, stay in (-2^(w-1), 2^(w-1)] and the widths must sum to 253
    (MSM) -- for every window width, on random scalars and on the extremes of [0, l)"""
    vals = [int.from_bytes(_r(b"rec", i), "little") % L for i in range(40)]
    vals += [0, 1, L - 1, L - 2, 2**252, 2**252 - 1, 2**252 + 1, (1 << 252) - (1 << 200), 0x8080808080808080 << 64]
    for c in range(4, 15):
        # windows whose raw value is exactly half their range pass the carry of the window below on: chains of them, with
        # and without a carry coming in at the bottom (the one-window digit function has to walk down them)
        half_chain = sum(1 << (c * k + c - 1) for k in range(9))
        chains = [half_chain, half_chain + (1 << (c - 1)) - 1, (half_chain << c) + (1 << (c - 1)) + 1, (half_chain << (2 * c)) + (1 << c) - 1]
        for a in vals + [x % L for x in chains]:
            dig = (ctypes.c_int16 * 64)()
            wid = (ctypes.c_uint32 * 64)()
            K = ht.ht_msm_recode(a.to_bytes(32, "little"), c, dig, wid)
            assert sum(wid[:K]) == 253 and all(w in (c, c - 1) for w in wid[:K]) and K == -(-253 // c)
            off, total = 0, 0
            for k in range(K):
                assert -(1 << (wid[k] - 1)) < dig[k] <= (1 << (wid[k] - 1)), (c, k, dig[k])
                total += dig[k] << off
                off += wid[k]
            assert total == a, (c, hex(a))
    # half-scalar plan of small calls: s = lo + 2^126 hi, both halves recoded over 127-bit windows (one spare bit: nothing
    # carries out of the top window; s >= 2^252 makes the high half exactly 2^126: the top window's digit is +half)
    for c in range(4, 12):
        edge = [(1 << 126) - 1, 1 << 126, (1 << 126) + 1, L - 1, L - 2, 1 << 252, (1 << 252) + 1, (1 << 252) - 1,
                ((1 << 126) - 1) << 126, (((1 << 126) - 1) << 126) | ((1 << 126) - 1)]
        for a in vals + edge:
            a %= L
            lo, hi, wid = (ctypes.c_int16 * 64)(), (ctypes.c_int16 * 64)(), (ctypes.c_uint32 * 64)()
            K = ht.ht_msm_recode_split(a.to_bytes(32, "little"), c, lo, hi, wid)
            assert sum(wid[:K]) == 127 and K == -(-127 // c)
            tot_lo = tot_hi = off = 0
            for k in range(K):
                for d in (lo[k], hi[k]):
                    assert -(1 << (wid[k] - 1)) < d <= (1 << (wid[k] - 1))
                tot_lo += lo[k] << off
                tot_hi += hi[k] << off
                off += wid[k]
            assert tot_lo == a % (1 << 126) and tot_hi == a >> 126 and tot_lo + (tot_hi << 126) == a
    for forced, n_gens in [(None, 5), (None, 516), (None, 1026), (None, 5000), ("8", 5), ("9", 5), ("10", 5), ("11", 5)]:
        if forced:
            monkeypatch.setenv("BPP_FB_WBITS", forced)
        else:
            monkeypatch.delenv("BPP_FB_WBITS", raising=False)
        for a in vals:
            dig = (ctypes.c_int16 * 32)()
            wb = ctypes.c_uint32()
            W = ht.ht_fb_recode(a.to_bytes(32, "little"), n_gens, dig, ctypes.byref(wb))
            w = wb.value
            slots = -(-254 // w)  # table slots per generator
            merged = 253 % w == 0  # a full top window stays unsigned instead of carrying into a one-bit digit (recode.h)
            assert 8 <= w <= 11 and W == (slots - 1 if merged else slots) and (forced is None or w == int(forced))
            assert slots * (1 << (w - 1)) * 128 * n_gens <= 1800 << 20 or w == 8
            assert all(-(1 << (w - 1)) < dig[k] <= (1 << (w - 1)) for k in range(W - 1 if merged else W))
            assert not merged or 0 <= dig[W - 1] <= (1 << w)  # entries 1 .. 2^w: its own slot and the spare one after it
            assert sum(dig[k] << (w * k) for k in range(W)) == a


def test_weight_chain_single_forms(ht):
    """the low-latency single chain of chain_host.h (state addressed as bytes, unrolled 64-bit Keccak-f, BMI build) against
    merlin.h's generic sponge and the oracle's Merlin, for lengths that put the block boundary of the sponge at every
    position of the per-proof message and of the 64-byte squeezes"""
    from oracle.pyref import protocol as O
    import ctypes
    ht.ht_weight_chain_single.restype = ctypes.c_int
    for n in list(range(1, 24)) + [37, 129, 500]:
        rng = b"".join(_r(b"single", 1000 * n + i) for i in range(n))
        outs = []
        for form in (0, 1, 2, 3):
            out = ctypes.create_string_buffer(32 * n)
            rc = ht.ht_weight_chain_single(rng, n, form, out)
            assert rc in (0, 1)
            if rc == 1:
                outs.append(out.raw)
        assert len(outs) >= 3 and all(o == outs[0] for o in outs), n
        if n in (1, 7, 37):
            t = M.Transcript(b"Bulletproofs+ verifier weights")
            for i in range(n):
                t.append_message(b"proof", rng[i * 32:(i + 1) * 32])
            wr = t.build_rng().finalize(O.NullRng())
            assert outs[0] == b"".join(C.scalar_bytes(O.random_not_zero(wr)) for _ in range(n))


def test_uniform_access_scalar_multiplication(ht):
    """csrc/ct.h, host-compiled with its table reads recorded (BPP_CT_TOUCH): the form behind bpp_pedersen_commit, the prover's
    witness check and A1 / B -- where the reference is constant-time (src/generators/pedersen_gens.rs:112-122,
    src/range_proof.rs:275-284,572-584).  (1) results equal the oracle's scalar multiplication, edge scalars included;
    (2) the signed radix-16 digits add up to the scalar and stay in range; (3) the SEQUENCE OF TABLE ENTRIES READ is the same
    for every scalar and every point: 64 digit positions x all nine entries, in order -- no read depends on a secret."""
    ht.ht_ct_scalarmul.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t,
                                   ctypes.POINTER(ctypes.c_size_t)]
    scalars = [0, 1, 2, 7, 8, 9, 15, 16, 17, L - 1, L - 2, 2**252, 2**252 - 1, (L - 1) // 2, 0x8888888888888888, 2**128 - 1,
               int("8" * 63, 16) % L, int("7" * 63, 16) % L] + [int.from_bytes(_r(b"ctk", i), "little") % L for i in range(24)]
    traces = set()
    for i, k in enumerate(scalars):
        pt = C.from_uniform_bytes(_r(b"ctp", i % 5, 64))
        out, tr, n = _buf(), ctypes.create_string_buffer(4096), ctypes.c_size_t()
        assert ht.ht_ct_scalarmul(pt.compress(), k.to_bytes(32, "little"), out, tr, 4096, ctypes.byref(n)) == 1
        assert out.raw == (pt * k).compress(), k
        traces.add(tr.raw[:n.value])
        dig = (ctypes.c_int8 * 64)()
        ht.ht_ct_recode16(k.to_bytes(32, "little"), dig)
        assert sum(int(d) << (4 * j) for j, d in enumerate(dig)) == k
        assert all(-8 <= int(d) < 8 for d in dig[:63]) and 0 <= int(dig[63]) <= 2
    assert traces == {bytes(range(9)) * 64}, "a table read depends on the scalar"
    # the fixed-base form (the Pedersen bases: bpp_pedersen_commit, the witness check): 64 positions x all eight lines of the position
    ht.ht_ct_fixed_scalarmul.argtypes = ht.ht_ct_scalarmul.argtypes
    traces = set()
    for i, k in enumerate(scalars):
        pt = C.from_uniform_bytes(_r(b"ctp", i % 3, 64))
        out, tr, n = _buf(), ctypes.create_string_buffer(4096), ctypes.c_size_t()
        assert ht.ht_ct_fixed_scalarmul(pt.compress(), k.to_bytes(32, "little"), out, tr, 4096, ctypes.byref(n)) == 1
        assert out.raw == (pt * k).compress(), k
        traces.add(tr.raw[:n.value])
    assert traces == {bytes(range(8)) * 64}, "a table read depends on the scalar"
    # the digit-parallel form of A1's two folded generators (ct_pos_multiple: no table, all eight multiples of every position
    # computed, one kept under a mask): the same products
    ht.ht_ct_var_scalarmul.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p]
    for i, k in enumerate(scalars):
        pt = C.from_uniform_bytes(_r(b"ctv", i % 4, 64))
        out = _buf()
        assert ht.ht_ct_var_scalarmul(pt.compress(), k.to_bytes(32, "little"), out) == 1
        assert out.raw == (pt * k).compress(), k
