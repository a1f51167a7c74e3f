#!/usr/bin/env python3
"""Timeline of a bench run from a rocprofv3 kernel trace: which kernels ran when, how many at once, idle gaps.

usage (GPU box):  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-traffic
                  python3 tools/timeline.py gpurun_out/tl gpurun_out/timeline.txt [steps]
The timed steps are taken to be the last `steps` (+3 calibration) k_transcripts dispatches of the trace."""
import csv, glob, os, sys, collections


def load(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("bpp::", "").replace("void ", ""),
                         r.get("Queue_Id", "?")))
    rows.sort()
    return rows


def main():
    d, out = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    rows = load(d)
    tr = [r for r in rows if r[2].startswith("k_transcripts")]
    first = tr[-(steps + 3)][0]          # first timed step's first kernel
    last_tr = tr[-4]                     # last timed step's first kernel
    fin = [r for r in rows if r[2].startswith("k_msm_final")]
    end = fin[-4][1]
    win = [r for r in rows if r[0] >= first and r[1] <= end]
    T = end - first
    # union busy time / concurrency histogram by sweeping
    ev = []
    for s, e, n, q in win:
        ev.append((s, 1)); ev.append((e, -1))
    ev.sort()
    lvl, prev, hist = 0, first, collections.Counter()
    for t, dlt in ev:
        hist[lvl] += t - prev
        prev = t
        lvl += dlt
    with open(out, "w") as o:
        o.write("timed window %.3f ms, %d kernels, %d timed steps -> %.3f ms per step\n" % (T / 1e6, len(win), steps, T / 1e6 / steps))
        for k in sorted(hist):
            o.write("  %d kernels running: %.3f ms (%.1f %%)\n" % (k, hist[k] / 1e6, 100.0 * hist[k] / T))
        # per-kernel: total time, mean duration
        agg = collections.defaultdict(lambda: [0, 0])
        for s, e, n, q in win:
            agg[n][0] += 1; agg[n][1] += e - s
        for n, (c, t) in sorted(agg.items(), key=lambda x: -x[1][1]):
            o.write("  %-28s %4d x %8.1f us\n" % (n, c, t / c / 1e3))
        # step table: per queue, start of k_transcripts and end of k_msm_final
        o.write("steps (queue, start ms, end ms, latency ms):\n")
        byq = collections.defaultdict(list)
        for s, e, n, q in win:
            byq[q].append((s, e, n))
        for q, lst in sorted(byq.items()):
            cur = None
            for s, e, n in lst:
                if n.startswith("k_transcripts"):
                    cur = s
                if n.startswith("k_msm_final") and cur is not None:
                    o.write("  q%s %8.3f %8.3f %7.3f\n" % (q, (cur - first) / 1e6, (e - first) / 1e6, (e - cur) / 1e6))
                    cur = None
    print(open(out).read())


if __name__ == "__main__":
    main()
