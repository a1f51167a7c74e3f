"""The built gfx950 code object must not contain a DPP result consumed as store data by the very next instruction: on the
MI355X that reads stale data (found with a spilled `v_mov_b32_dpp` / `scratch_store_dwordx2` pair) and the compiler does not
insert the wait state.  tools/isa/dpp_hazard_check.py disassembles libbpp_hip.so (no GPU needed)."""
import importlib
import importlib.machinery
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "bulletproofs-plus_amd", "libbpp_hip.so")


def _mod():
    spec = importlib.util.spec_from_file_location("dpp_hazard_check", os.path.join(ROOT, "tools", "isa", "dpp_hazard_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_a_build_that_cannot_be_checked_fails(monkeypatch, tmp_path):
    """no disassembler -> the build refuses (and removes the library) instead of shipping unchecked; only an explicit
    BPP_SKIP_ISA_CHECK=1 lets it through"""
    build = importlib.import_module("bulletproofs-plus_amd._build")
    fake = tmp_path / "libfake.so"
    fake.write_bytes(b"x")
    real = _mod().llvm_bin
    import types
    stub = types.SimpleNamespace(llvm_bin=lambda hipcc=None: None, check=lambda so: (0, []), LLVM=None)
    monkeypatch.setattr(importlib.util, "module_from_spec", lambda spec: stub)
    monkeypatch.setattr(importlib.machinery.SourceFileLoader, "exec_module", lambda self, m: None)
    monkeypatch.delenv("BPP_SKIP_ISA_CHECK", raising=False)
    with pytest.raises(RuntimeError):
        build._check_isa_hazards(str(fake))
    assert not fake.exists()
    fake.write_bytes(b"x")
    monkeypatch.setenv("BPP_SKIP_ISA_CHECK", "1")
    build._check_isa_hazards(str(fake))
    assert fake.exists() and real is not None


def test_no_dpp_result_stored_by_next_instruction():
    mod = _mod()
    assert os.path.exists(SO), "libbpp_hip.so is not built"
    assert mod.llvm_bin() is not None, "no llvm-objdump: the hazard check cannot run (BPP_SKIP_ISA_CHECK=1 only skips it at build time)"
    n_dpp, hits = mod.check(SO)
    assert n_dpp > 100  # the quad kernels are in there
    assert not hits, hits[:5]
