#!/usr/bin/env python3
"""Host timing of the batch-weight chains through libbpp_hosttest.so (g++ -O2 build: for the engine's own clang -O3 build use
tools/microbench/chain_forms.cpp): lock-step bundles of 1 / 4 / 8 chains and the three forms of the single chain."""
import ctypes, importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("bulletproofs-plus_amd")
ht = ctypes.CDLL(pkg._build.build_hosttest())
n = 1024
for width in (1, 4, 8):
    rng = os.urandom(32 * n * width)
    out = ctypes.create_string_buffer(32 * n * width)
    ht.ht_weight_chains(rng, n, width, out)
    t0 = time.perf_counter()
    for _ in range(20):
        rc = ht.ht_weight_chains(rng, n, width, out)
    el = (time.perf_counter() - t0) / 20
    print("lock-step width %d: rc %d  %.3f us per proof and chain (%.3f us per proof step)" % (width, rc, 1e6 * el / n / width, 1e6 * el / n))
out = ctypes.create_string_buffer(32 * n)
rng = os.urandom(32 * n)
for form in (0, 1, 2):
    t0 = time.perf_counter()
    for _ in range(20):
        rc = ht.ht_weight_chain_single(rng, n, form, out)
    print("single form %d: rc %d %.3f us per proof" % (form, rc, 1e6 * (time.perf_counter() - t0) / 20 / n))
