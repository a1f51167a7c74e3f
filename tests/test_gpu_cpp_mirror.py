"""GPU test: the C++ host-side mirror (include/bpp.hpp) runs the reference's integration test shapes end to end
(tests/cpp/ristretto_mirror.cpp = tests/ristretto.rs restated) against libbpp_hip.so."""
import importlib
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_mirror_runs_reference_integration_shapes(tmp_path):
    pkg = importlib.import_module("bulletproofs-plus_amd")
    lib = pkg._build.build()
    exe = str(tmp_path / "ristretto_mirror")
    libdir = os.path.dirname(lib)
    subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "ristretto_mirror.cpp"),
                    "-o", exe, "-L", libdir, "-lbpp_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "all reference integration shapes passed" in out.stdout, out.stderr[-2000:]


def test_compiled_caller_of_the_round3_entry_points(tmp_path):
    """tests/cpp/abi_round3.cpp: packed / pipelined / sharded (wave and grouped) entry points and their structs as declared in
    include/bpp.h, from compiled code (ctypes carries its own declarations and would not notice a wrong header)"""
    pkg = importlib.import_module("bulletproofs-plus_amd")
    lib = pkg._build.build()
    exe = str(tmp_path / "abi_round3")
    libdir = os.path.dirname(lib)
    subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "abi_round3.cpp"),
                    "-o", exe, "-L", libdir, "-lbpp_hip", "-lpthread", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "abi_round3 ok" in out.stdout, (out.stdout[-500:], out.stderr[-2000:])
