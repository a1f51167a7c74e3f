#!/usr/bin/env python3
"""Wall time of bpp_weights_from_chains (the engine's own chain scheduler on its host pool, libbpp_hip.so's clang -O3 build) for
G chains of n proofs: what one sharded call (64 x 4096) and one headline step (64 x 1024) spend on the host.  No GPU needed."""
import ctypes, importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("bulletproofs-plus_amd")
lib = pkg._lib.load()
print("host pool:", lib.bpp_host_threads(), "threads")
for (G, n) in [(64, 1024), (64, 4096), (8, 4096), (16, 4096), (1, 4096), (1, 1024)]:
    rng = os.urandom(32 * n * G)
    out = ctypes.create_string_buffer(32 * n * G)
    lib.bpp_weights_from_chains(rng, G, n, out)
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        lib.bpp_weights_from_chains(rng, G, n, out)
    el = (time.perf_counter() - t0) / reps
    print("G=%3d n=%5d: %.3f ms per call, %.3f us per proof and chain of CPU-wall" % (G, n, 1e3 * el, 1e6 * el / n / G))
