"""What a pooled call of the batcher spends: upload of G x 256 fresh proofs, the first grouped verification (group layout + kernels)
and a second one on the same resident batch, with the stage intervals of the last call."""
import importlib, sys, time, json
sys.path.insert(0, '/root/repo')
import numpy as np
import bench
bpp = importlib.import_module("bulletproofs-plus_amd")
packed = importlib.import_module("bulletproofs-plus_amd.packed")
eng = bpp.Engine(0)
eng.profile(True)
params = bpp.RangeParameters.init(64, 1, bpp.create_pedersen_gens_with_extension_degree(1), engine=eng)
d = bench.make_inputs(np, packed, params, 8192, seed=5)
for G in (8, 32):
    n = G * 256
    bounds = [256 * g for g in range(G + 1)]
    ups, firsts, seconds = [], [], []
    for it in range(12):
        t0 = time.perf_counter()
        rb = packed.ResidentBatch(params, d["proofs"][:n], d["commitments"][:n], d["min_values"][:n], d["min_present"][:n], None, bench.LABEL)
        t1 = time.perf_counter()
        packed.verify_groups(rb, bounds)
        t2 = time.perf_counter()
        packed.verify_groups(rb, bounds)
        t3 = time.perf_counter()
        prof = eng.last_profile()
        rb.close()
        if it >= 2:
            ups.append(t1 - t0); firsts.append(t2 - t1); seconds.append(t3 - t2)
    med = lambda v: sorted(v)[len(v) // 2] * 1e3
    print(json.dumps({"groups": G, "proofs": n, "upload_ms": round(med(ups), 3), "first_verify_ms (layout + verify)": round(med(firsts), 3), "second_verify_ms": round(med(seconds), 3),
                      "stages_ms": {k: round(v, 3) for k, v in prof.items() if k.endswith("_ms")}}))
