#!/usr/bin/env python3
"""Per-launch view of the prover's kernels from a rocprofv3 kernel trace (csv): the dispatches of the LAST bpp_prove_batch
call in start order -- kernel, workgroups, lanes per workgroup, duration -- and, for k_fb_msm, the additions per second of
every launch when the call's shape is given (--m, --t, --proofs, --windows): which launches of a call run below the kernel's
own best.

  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prov -- python3 bench.py --only prover --no-extra --no-cpu-baseline --no-traffic
  python3 tools/prover_trace.py gpurun_out/prov --m 4 --t 3 --proofs 1024 --windows 23 > profiles/<tag>_prover_launches.txt
"""
import argparse
import csv
import glob
import os
import re


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--m", type=int, default=4)
    ap.add_argument("--t", type=int, default=3)
    ap.add_argument("--bits", type=int, default=64)
    ap.add_argument("--proofs", type=int, default=1024)
    ap.add_argument("--windows", type=int, default=23)
    a = ap.parse_args()
    rows = []
    for f in sorted(glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            name = re.sub(r"^void ", "", r["Kernel_Name"])
            name = re.sub(r"\(.*", "", name).replace("bpp::", "")
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, int(r.get("Grid_Size", r.get("Grid_Size_X", 0))) // max(1, int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", 1)))),
                         int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", 1)))))
    rows.sort()
    # the last call = from the last kp_commit_terms pair (one per sub-batch) on
    starts = [i for i, r in enumerate(rows) if r[2] == "kp_commit_terms"]
    if not starts:
        raise SystemExit("no prover kernels in the trace")
    first = starts[-1]
    while first - 1 >= 0 and rows[first - 1][2] in ("kp_commit_terms",) or (first - 1 in starts):
        first -= 1
    subs = 1
    for i in reversed(starts[:-1]):  # the sub-batches' first kernels lie next to each other
        if rows[starts[-1]][0] - rows[i][0] < 2_000_000:
            first = min(first, i)
            subs += 1
        else:
            break
    call = rows[first:]
    t0 = call[0][0]
    mn = a.m * a.bits
    per_output = {"commit": 1 + a.t, "round": mn + a.t + 1, "final": (2 * mn + a.t + 1 + a.t + 1) / 2.0}
    print("# last bpp_prove_batch call: %d dispatches, %d sub-batch stream(s), %.3f ms from first start to last end"
          % (len(call), subs, (max(r[1] for r in call) - t0) / 1e6))
    print("# %-22s %10s %6s %10s %10s %12s" % ("kernel", "workgroups", "lanes", "start_us", "dur_us", "G_adds/s"))
    fb = [r for r in call if r[2] == "k_fb_msm"]
    n_fb_sub = len(fb) // max(1, subs)
    seen = 0
    tot_adds = tot_ns = 0
    for r in call:
        rate = ""
        if r[2] == "k_fb_msm":
            k = seen // max(1, subs) if subs > 1 and len(fb) % subs == 0 else seen
            kind = "commit" if r[3] == (a.proofs // subs) * a.m else ("final" if k == n_fb_sub - 1 else "round")
            adds = r[3] * per_output[kind] * a.windows
            tot_adds += adds
            tot_ns += r[1] - r[0]
            rate = "%.2f (%s)" % (adds / (r[1] - r[0]), kind)
            seen += 1
        print("  %-22s %10d %6d %10.1f %10.1f %12s" % (r[2], r[3], r[4], (r[0] - t0) / 1e3, (r[1] - r[0]) / 1e3, rate))
    if tot_ns:
        print("# k_fb_msm: %d launches, %.3f ms summed, %.2f G additions/s over the call" % (len(fb), tot_ns / 1e6, tot_adds / tot_ns))


if __name__ == "__main__":
    main()
