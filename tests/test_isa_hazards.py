"""The built gfx950 code object must not contain a DPP result consumed as store data by the very next instruction: on the
MI355X that reads stale data (found with a spilled `v_mov_b32_dpp` / `scratch_store_dwordx2` pair) and the compiler does not
insert the wait state.  tools/isa/dpp_hazard_check.py disassembles libbpp_hip.so (no GPU needed)."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "bulletproofs-plus_amd", "libbpp_hip.so")


@pytest.mark.skipif(not os.path.exists(SO) or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"), reason="needs the built library and llvm-objdump")
def test_no_dpp_result_stored_by_next_instruction():
    spec = importlib.util.spec_from_file_location("dpp_hazard_check", os.path.join(ROOT, "tools", "isa", "dpp_hazard_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    n_dpp, hits = mod.check(SO)
    assert n_dpp > 100  # the quad kernels are in there
    assert not hits, hits[:5]
