"""A short run of tools/soak.py inside the GPU suite: concurrent contexts, random sub-batches of the fixture with random
single-bit mutations, every verdict (accept / reject and error kind) compared with the CPU oracle's -- through every entry
path of the verifier (item form, packed, pipelined submit / collect, sharded over a one-rank RCCL communicator)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_differential_soak_short():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak.py"), "--seconds", "14", "--threads", "3"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    res = json.loads(out.stdout.strip().split("\n")[-1])
    assert res["mismatch"] == 0 and res["calls"] > 50 and res["rejected"] > 10, res
    assert all(res["calls_" + p] > 3 for p in ("items", "packed", "pipeline", "sharded")), res
