"""CPU suite: the C-ABI library loads and exports every symbol include/bpp.h declares; no compute without a GPU."""
import ctypes
import importlib
import os
import re

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "bpp.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bpp_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    pkg = importlib.import_module("bulletproofs-plus_amd")
    pkg._build.build()
    lib = pkg._lib.load()
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "libbpp_hip.so does not export %s" % n
    assert sorted(n for n, _, _ in pkg._lib.SYMBOLS) == names


def test_host_only_entry_points():
    pkg = importlib.import_module("bulletproofs-plus_amd")
    assert 1 <= pkg.host_threads() <= 256
    packed = importlib.import_module("bulletproofs-plus_amd.packed")  # asserts that its item dtypes match the C structs
    assert packed._VERIFY_ITEM.itemsize == ctypes.sizeof(pkg._lib.VerifyItem)
    from oracle.pyref import merlin as M
    assert pkg.Transcript.new(b"BatchedRangeProofTest").strobe_state() == M.Transcript(b"BatchedRangeProofTest").strobe.to_bytes()


def test_fails_loudly_without_a_gpu():
    if torch.cuda.is_available():
        return
    pkg = importlib.import_module("bulletproofs-plus_amd")
    lib = pkg._lib.load()
    ctx = ctypes.c_void_p()
    assert lib.bpp_ctx_create(ctypes.byref(ctx), 0) < 0  # BPP_ERR_NO_DEVICE: there is no CPU fallback
    try:
        pkg.Engine(0)
    except pkg.EngineError:
        pass
    else:
        raise AssertionError("Engine() must not succeed without a gfx950 device")


def test_from_bytes_mirror_matches_reference_rules():
    """RangeProof::from_bytes host mirror (src/range_proof.rs:1339-1435)"""
    pkg = importlib.import_module("bulletproofs-plus_amd")
    from tests.helpers import make_oracle_batch
    raw = make_oracle_batch(4, [1], 1, seed=b"ser").o_proofs[0].to_bytes()
    K = pkg.ProofErrorKind
    assert pkg.RangeProof.from_bytes(raw).to_bytes() == raw
    for bad, kind in [(b"", K.InvalidLength), (raw[:-1], K.InvalidLength), (raw + b"\0", K.InvalidLength),
                      (raw + bytes(32), K.InvalidLength), (b"\x07" + raw[1:], K.InvalidArgument),
                      (raw[:1 + 32 * 6], K.InvalidLength), (raw[:1] + b"\xff" * 32 + raw[33:], K.InvalidArgument)]:
        try:
            pkg.RangeProof.from_bytes(bad)
        except pkg.ProofError as e:
            assert e.kind == kind
        else:
            raise AssertionError("expected ProofError")
    assert pkg.RangeProof.extension_degree_from_proof_bytes(raw) == pkg.ExtensionDegree.DefaultPedersen


def test_vectorised_weight_chains_equal_the_scalar_chain():
    """bpp_weights_from_chains (lockstep AVX-512 / AVX2 Keccak, threads) == per-group bpp_weights_from_chain == oracle"""
    import hashlib
    pkg = importlib.import_module("bulletproofs-plus_amd")
    from oracle.pyref import curve as C
    from oracle.pyref import merlin as M
    from oracle.pyref import protocol as O
    for groups, per in [(1, 5), (3, 7), (4, 9), (8, 33), (9, 4), (13, 6), (21, 3)]:
        rng = b"".join(hashlib.sha256(b"c%d-%d" % (groups, i)).digest() for i in range(groups * per))
        got = pkg.weights_from_chains(rng, groups)
        want = b"".join(pkg.weights_from_chain(rng[32 * per * g:32 * per * (g + 1)]) for g in range(groups))
        assert got == want
    rng = b"".join(hashlib.sha256(b"o%d" % i).digest() for i in range(40))
    wt = M.Transcript(b"Bulletproofs+ verifier weights")
    for i in range(40):
        wt.append_message(b"proof", rng[32 * i:32 * i + 32])
    r = wt.build_rng().finalize(M.NullRng())
    assert pkg.weights_from_chains(rng, 1) == b"".join(C.scalar_bytes(O.random_not_zero(r)) for _ in range(40))


def test_missing_rccl_library_maps_to_the_comm_error_code():
    """BPP_RCCL_LIB names THE library the sharded entry points load RCCL from; a file that is not there must come back as
    BPP_ERR_COMM (-4) from bpp_comm_unique_id -- no fallback to another copy, no crash (child process: the binding is resolved
    once per process; needs no GPU)"""
    import subprocess
    import sys
    code = ("import ctypes, importlib, sys; sys.path.insert(0, %r); pkg = importlib.import_module('bulletproofs-plus_amd'); "
            "lib = pkg._lib.load(); buf = (ctypes.c_uint8 * 128)(); rc = lib.bpp_comm_unique_id(buf); print('rc', rc); "
            "sys.exit(0 if rc == -4 else 1)" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, BPP_RCCL_LIB="/nonexistent/librccl.so"), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr[-1000:])


def test_rust_ffi_declares_every_symbol_of_the_header():
    """rust/bpp-gpu-shim/src/ffi.rs cannot be compiled here (no cargo): at least its extern block must name exactly the symbols
    include/bpp.h declares, so that a new entry point never goes missing on the Rust side unnoticed"""
    ffi = open(os.path.join(ROOT, "rust", "bpp-gpu-shim", "src", "ffi.rs")).read()
    rust = sorted(set(re.findall(r"pub fn (bpp_[a-z0-9_]+)\s*\(", ffi)))
    assert rust == _declared(), (sorted(set(_declared()) - set(rust)), sorted(set(rust) - set(_declared())))
