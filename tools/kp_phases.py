#!/usr/bin/env python3
"""Where a prover round's serial kernel spends its cycles: runs configs[4]'s prove call on a library built with -DBPP_KP_PHASES
(tools/gpu_kp_phases.sh builds it as gpurun_in/kp_phases.so and names it in BPP_LIB_PATH) and prints the shader-clock cycles per phase of kp_round, per proof and call."""
import ctypes
import importlib
import json
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NAMES = {0: "encode L,R (2 lanes)", 1: "load + append L,R", 2: "build_rng", 3: "challenge e", 4: "inversion (step 0: + A, y, z)",
         5: "powers, squares, alpha", 6: "draws dL,dR / r,s,d,eta", 7: "store transcript", 8: "fold a,b", 9: "fold cG,cH",
         10: "inner products", 11: "term lists", 12: "final-step term lists", 13: "step-0 vector prep",
         14: "sum of the slices' partial sums (L, R)"}


def main():
    import numpy as np
    import bench
    bpp = importlib.import_module("bulletproofs-plus_amd")
    packed = importlib.import_module("bulletproofs-plus_amd.packed")
    eng = bpp.Engine(0)
    fn = getattr(eng.lib, "bpp_debug_kp_phases", None)
    if fn is None:
        raise SystemExit("this libbpp_hip.so was not built with -DBPP_KP_PHASES")
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
    m, t = int(os.environ.get("KP_M", "4")), int(os.environ.get("KP_T", "3"))
    p5 = bpp.RangeParameters.init(64, m, bpp.create_pedersen_gens_with_extension_degree(t), engine=eng)
    d5 = bench.make_inputs(np, packed, p5, 1024, seed=8675309 + 5)
    args = (p5, d5["values"], d5["blindings"], d5["commitments"], d5["min_values"], d5["min_present"], d5["seeds"], bench.LABEL, d5["ext"])
    for _ in range(2):
        packed.prove(*args)
    buf = (ctypes.c_ulonglong * 32)()
    fn(buf, 1)
    iters = 4
    for _ in range(iters):
        packed.prove(*args)
    fn(buf, 0)
    per = [buf[i] / (1024.0 * iters) for i in range(32)]
    tot = sum(per)
    out = {"shape": {"m": m, "t": t, "proofs": 1024}, "cycles_per_proof_and_call": round(tot),
           "phases": {NAMES.get(i, str(i)): {"cycles": round(per[i]), "share": round(per[i] / tot, 4)} for i in range(15) if per[i]}}
    print(json.dumps(out, indent=1))
    p5.close()
    eng.close()


if __name__ == "__main__":
    main()
