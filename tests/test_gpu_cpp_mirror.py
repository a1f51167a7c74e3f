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
