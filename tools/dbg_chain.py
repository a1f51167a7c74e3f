import sys, importlib, ctypes
sys.path.insert(0, '.')
import numpy as np
import bench
bpp = importlib.import_module("bulletproofs-plus_amd")
packed = importlib.import_module("bulletproofs-plus_amd.packed")
_lib = importlib.import_module("bulletproofs-plus_amd._lib")
lib = _lib.load()
eng = bpp.Engine(0)
params = bpp.RangeParameters.init(64, 1, bpp.create_pedersen_gens_with_extension_degree(1), engine=eng)
N = 4096 + 452
d = bench.make_inputs(np, packed, params, N, seed=20261005)
def host_chain(rng, n):
    out = (ctypes.c_uint8 * (32 * n))()
    lib.bpp_weights_from_chain((ctypes.c_uint8 * len(rng)).from_buffer_copy(rng), n, out)
    return bytes(out)
for (lo, hi, chunk) in [(0, N, 1024), (0, N, 1000), (4096, N, 0), (0, 452, 0), (0, 2048 + 452, 1024), (0, 900, 448), (0, 130, 64), (0, 130, 65), (0,200,100)]:
    rb = packed.ResidentBatch(params, d["proofs"][lo:hi], d["commitments"][lo:hi], d["min_values"][lo:hi], d["min_present"][lo:hi], None, bench.LABEL)
    eng.set_option("chain", 1)
    rb.verify_only(chunk=chunk)
    wd = rb.trace(3); rng = rb.trace(2)
    eng.set_option("chain", 0)
    rb.verify_only(chunk=chunk)
    wh = rb.trace(3); rng2 = rb.trace(2)
    n = hi - lo
    c = chunk if chunk else n
    bounds = list(range(0, n, c)) + [n]
    res = []
    for g in range(len(bounds) - 1):
        a, b = bounds[g], bounds[g + 1]
        want = host_chain(rng[32 * a:32 * b], b - a)
        res.append((b - a, wd[32 * a:32 * b] == want, wh[32 * a:32 * b] == want))
    print((lo, hi, chunk), "rng equal", rng == rng2, res, flush=True)
    rb.close()
