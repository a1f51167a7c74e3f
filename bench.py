#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric on MI355X: 64-bit range proofs verified / second (batch).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--mode wide|shard] [--config cfg2|cfg3] [--no-cpu-baseline]

A step = one engine call = one pass of the hot path (RangeProof::verify: transcript replay, decompression, scalar block,
weight chain, final MSM) over one resident input of --batches-per-step (default 64) INDEPENDENT reference batches of
BASELINE.json configs[1]: 1024 non-aggregated 64-bit proofs each, extension degree 1, already resident in HBM
(tests/golden/bench_cfg2.bin, produced by tests/golden/make_golden.py with the recipe of benches/range_proof.rs:206-262).
Every 1024-proof batch keeps its own weight transcript, its own final MSM and its own identity test, exactly as if it had
been passed to the reference's verify() alone; --concurrency steps are in flight per GPU and EXACTLY --steps are timed.
value = proofs verified per second = steps x batches-per-step x 1024 / elapsed.
N>1 (launched by torch.distributed.run, one rank per GPU), weak scaling: mode "shard" (default) = every rank verifies
its own batches exactly as at N=1, verdicts combined by one all_reduce; mode "wide" = the union of all ranks' shards is
ONE reference batch per step (all_gather of transcript-RNG bytes + all_gather of accumulator points over RCCL).
Prints ONE JSON line on rank 0.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--mode", default="shard", choices=["wide", "shard"])
    ap.add_argument("--config", default="cfg2", choices=["cfg2", "cfg3"])
    ap.add_argument("--chunk", type=int, default=0, help="proofs per reference batch at N=1 (0 = whole batch)")
    ap.add_argument("--concurrency", type=int, default=int(os.environ.get("BPP_BENCH_CONCURRENCY", "6")),
                    help="steps in flight per GPU (one engine/stream + one host thread each)")
    ap.add_argument("--batches-per-step", "--batches-per-launch", dest="batches_per_launch", type=int,
                    default=int(os.environ.get("BPP_BENCH_BATCHES_PER_LAUNCH", "64")),
                    help="independent 1024-proof reference batches verified by one step (one engine call); each keeps its "
                         "own weight chain, final MSM and identity test")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    bpp = importlib.import_module("bulletproofs-plus_amd")
    from tests.golden.loader import load_bench

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d bench.py --gpus %d"
                             % (args.gpus, args.gpus))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    use_dist = "RANK" in os.environ  # launched by torch.distributed.run (also with one rank: exercises the RCCL path)
    if use_dist:
        dist.init_process_group("nccl", device_id=device)

    data = load_bench("bench_%s.bin" % args.config)
    items = data["items"]
    if world > 1:  # each rank verifies a different rotation of the fixture, so shards are not byte-identical
        k = (rank * 131) % len(items)
        items = items[k:] + items[:k]
    n_local = len(items)

    import threading
    from concurrent.futures import ThreadPoolExecutor

    wide = use_dist and args.mode == "wide"
    S = 1 if wide else max(1, args.concurrency)  # cross-rank collectives must stay in program order -> no threads
    R = 1 if (wide or args.chunk) else max(1, args.batches_per_launch)
    S = max(1, min(S, args.steps))
    lanes = []  # one engine + stream + resident copy of the batch per in-flight slot
    t_upload = t_marshal = 0.0
    for i in range(S):
        stream = torch.cuda.Stream(device=device)
        eng = bpp.Engine(local_rank, stream=stream.cuda_stream)
        eng.profile(True)
        params = bpp.RangeParameters.init(data["bit_length"], data["m"],
                                          bpp.create_pedersen_gens_with_extension_degree(data["t"]), engine=eng)
        its = []
        for r in range(R):  # R distinct rotations of the fixture = R different 1024-proof batches
            k = ((i * R + r) * 37) % len(items)
            its += items[k:] + items[:k]
        sts = [bpp.RangeStatement.init(params, it["commitments"], it["min_values"], None) for it in its]
        proofs = [bpp.RangeProof.from_bytes(it["proof"]) for it in its]
        trs = [bpp.Transcript.new(data["label"]) for _ in its]
        rb = bpp.ResidentBatch(trs, sts, proofs)
        t_upload, t_marshal = rb.upload_seconds, rb.marshal_seconds
        lanes.append((stream, eng, rb))

    if wide:
        dmod = importlib.import_module("bulletproofs-plus_amd.dist")
        ops = dmod.LocalEngineOps(lanes[0][2])

    def one_step(slot):
        _, eng, rb = lanes[slot]
        t0 = time.perf_counter()
        if wide:
            dmod.verify_sharded(ops, n_local, device, mode="wide")
        else:
            rb.verify(bpp.VerifyAction.VerifyOnly, chunk=(args.chunk or n_local))  # raises on an invalid batch
        return time.perf_counter() - t0, eng.last_profile()

    def run_steps(count):
        """`count` complete steps, at most S in flight; returns (per-step latencies, per-step stage profiles)"""
        if S == 1:
            res = [one_step(0) for _ in range(count)]
        else:
            nxt = [0]
            lock = threading.Lock()

            def worker(slot):
                out = []
                while True:
                    with lock:
                        if nxt[0] >= count:
                            return out
                        nxt[0] += 1
                    out.append(one_step(slot))
            with ThreadPoolExecutor(S) as ex:
                res = [r for part in ex.map(worker, range(S)) for r in part]
        return [r[0] for r in res], [r[1] for r in res]

    def sync():
        torch.cuda.synchronize(device)
        if use_dist:
            dist.barrier()

    run_steps(args.warmup)
    sync()
    t0 = time.perf_counter()
    lat, profs = run_steps(args.steps)
    torch.cuda.synchronize(device)
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ok_all = 1
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        flag = torch.tensor([ok_all], dtype=torch.int32, device=device)  # verdicts of independent shards
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        assert int(flag.item()) == 1
    # calibration after the timed region: the roofline kernel with nothing co-running (in the timed region several steps
    # share the chip, so each launch of it is stretched by its neighbours)
    alone_ms = None
    if not wide:
        sync()
        alone = [one_step(0)[1] for _ in range(3)]
        vals = [a.get("msm_accumulate_ms", 0.0) for a in alone if a]
        if vals and min(vals) > 0:
            alone_ms = sum(vals) / len(vals)
        sync()
    prof_sum = {}
    for pf in profs:
        for k, v in pf.items():
            prof_sum[k] = prof_sum.get(k, 0.0) + v

    if rank == 0:
        total = n_local * R * world * args.steps
        out = {
            "metric": "64-bit range proofs verified/sec (batch)", "value": total / elapsed, "unit": "proofs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[%d]: reference batches of %d x aggregation-%d 64-bit proofs, extension "
                                   "degree %d, VerifyOnly, resident in HBM; one step = %d such batches in one engine call"
                                   % (1 if args.config == "cfg2" else 2, n_local, data["m"], data["t"], R),
                       "proofs_per_reference_batch": n_local, "batches_per_step": R, "proofs_per_step_per_gpu": n_local * R,
                       "mode": ("one reference batch over all ranks (all_gather rng bytes + accumulators)" if wide else
                                ("each %d-proof chunk is a reference batch" % args.chunk if args.chunk else
                                 "each 1024-proof resident batch is one reference batch (private verify())")),
                       "steps_in_flight_per_gpu": S,
                       "parallelism": "proof-sharded x%d" % world},
            "step_latency_ms": 1e3 * sum(lat) / len(lat),
        }
        if prof_sum and prof_sum.get("msm_final_ms", 0) > 0:
            k = len(profs)
            avg = {n: v / k for n, v in prof_sum.items()}
            terms = int(avg["msm_terms"])  # over all R groups of one launch
            K = int(avg["msm_windows"])
            acc_ms = avg["msm_accumulate_ms"]
            msm_bytes = 64 * terms  # 32 B scalar + 32 B compressed point per MSM term
            achieved = msm_bytes / (acc_ms * 1e-3) / 1e9
            # integer roofline of the same kernel: one mixed addition per (term, window) = 7 field multiplications
            # = 700 v_mad_u64_u32; peak = 49 lanes/clk/CU x 256 CU x 2.4 GHz (tools/microbench/int_rates.hip)
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
            if os.path.exists(tpath):  # PMC figure of the same kernel/workload, collected in its own rocprofv3 passes
                tj = json.load(open(tpath))
                traffic = tj["hbm_bytes_per_launch"] * (terms / float(tj["msm_terms_per_launch"]))
            mads = terms * K * 700.0
            out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                               "kernel": "k_msm_accumulate (Pippenger bucket accumulation of the final MSM)",
                               "kernel_ms": acc_ms, "algorithmic_bytes": msm_bytes, "msm_terms_per_launch": terms,
                               "note": "integer-VALU bound, not HBM bound (SURVEY 8d): see valu",
                               "valu": {"achieved_Tmad_per_s": mads / (acc_ms * 1e-3) / 1e12, "peak_Tmad_per_s": 30.1,
                                        "frac": mads / (acc_ms * 1e-3) / 30.1e12}}
            if alone_ms:  # same kernel, same launch, no other step in flight (3 launches after the timed region)
                out["roofline"]["alone"] = {"kernel_ms": alone_ms, "achieved": msm_bytes / (alone_ms * 1e-3) / 1e9,
                                            "frac": msm_bytes / (alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                            "valu_frac": mads / (alone_ms * 1e-3) / 30.1e12}
            out["stages_ms"] = {n: round(v, 4) for n, v in avg.items() if n.endswith("_ms")}
            # bpp_batch_upload alone: host proof/statement buffers -> parsed, packed and resident (R batches of 1024)
            out["pcie_inclusive_upload_ms"] = 1e3 * t_upload
            out["python_marshal_ms"] = 1e3 * t_marshal
        if not args.no_cpu_baseline:
            from oracle import cport  # cpu_baseline leg only
            cp = cport.Params(data["bit_length"], data["m"], data["t"])
            sample = data["items"][:256]  # one reference-sized batch (MAX_RANGE_PROOF_BATCH_SIZE)
            rc, sec1 = cp.verify_timed(sample, 256, 1)
            iters = max(1, int(args.cpu_seconds / max(sec1, 1e-3)))
            rc, sec = cp.verify_timed(sample, 256, iters)
            assert rc == 0
            out["cpu_baseline"] = {"value": len(sample) * iters / sec, "unit": "proofs/s", "cores": 1, "kind": "port",
                                   "sample": "%d x verify of one 256-proof reference batch (first 256 proofs of the workload), "
                                             "single thread, oracle/c port with dalek's algorithms" % iters}
            cp.close()
        print(json.dumps(out))
    for _, eng, rb in lanes:
        rb.close()
        eng.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
