import importlib, json, os, sys, threading, time
sys.path.insert(0, '/root/repo')
import numpy as np
import bench
bpp = importlib.import_module("bulletproofs-plus_amd")
packed = importlib.import_module("bulletproofs-plus_amd.packed")
eng0 = bpp.Engine(0)
p0 = bpp.RangeParameters.init(64, 1, bpp.create_pedersen_gens_with_extension_degree(1), engine=eng0)
d = bench.make_inputs(np, packed, p0, 4096, seed=5)
for S in [int(x) for x in os.environ.get("WIDE_PROBE_S", "1,4,6,8").split(",")]:
    engs = [bpp.Engine(0) for _ in range(S)]
    ps = [p0.share(e) for e in engs]
    rbs = [packed.ResidentBatch(ps[k], d["proofs"], d["commitments"], d["min_values"], d["min_present"], None, bench.LABEL) for k in range(S)]
    for rb in rbs:
        for _ in range(5): rb.verify_only(chunk=0)
    cnt = [0] * S
    stop = time.time() + 2.0
    def w(k):
        while time.time() < stop:
            rbs[k].verify_only(chunk=0); cnt[k] += 1
    th = [threading.Thread(target=w, args=(k,)) for k in range(S)]
    t0 = time.perf_counter()
    [x.start() for x in th]; [x.join() for x in th]
    el = time.perf_counter() - t0
    print(json.dumps({"graph": os.environ.get("BPP_GRAPH", "1"), "in_flight": S, "proofs_per_s": 4096 * sum(cnt) / el, "ms_per_batch_per_slot": 1e3 * el * S / max(1, sum(cnt))}))
    for rb in rbs: rb.close()
    for p in ps: p.close()
    for e in engs: e.close()
