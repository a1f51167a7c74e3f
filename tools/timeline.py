#!/usr/bin/env python3
"""Timeline of a bench run from a rocprofv3 kernel trace: which kernels ran when, how many at once, idle gaps.

usage (GPU box):  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-traffic
                  python3 tools/timeline.py gpurun_out/tl gpurun_out/timeline.txt [steps]
The timed region is the last burst of exactly `steps` steps between two moments with nothing running (bench.py synchronises on both
sides of it).  With a fourth argument every kernel of the region is listed as well (start, end, duration, queue, name)."""
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
    # the timed region: bench.py synchronises on both sides of it, so it is a burst of kernels between two moments with nothing
    # running -- the LAST burst holding exactly `steps` k_transcripts dispatches (the calibration's solo steps and the stage-profile
    # runs behind it are bursts of their own)
    bursts, cur, cur_end = [], [], 0
    for r in rows:
        if cur and r[0] - cur_end > 100000:  # 0.1 ms of nothing
            bursts.append(cur)
            cur, cur_end = [], 0
        cur.append(r)
        cur_end = max(cur_end, r[1])
    if cur:
        bursts.append(cur)
    timed = [b for b in bursts if sum(1 for r in b if r[2].startswith("k_transcripts")) == steps]
    if not timed:
        raise SystemExit("no burst of %d steps in the trace: %r" % (steps, [sum(1 for r in b if r[2].startswith("k_transcripts")) for b in bursts][-12:]))
    first, end = timed[-1][0][0], max(r[1] for r in timed[-1])
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
        if len(sys.argv) > 4:
            o.write("kernels (start ms, end ms, us, queue, name):\n")
            for s, e, n, q in win:
                o.write("  %8.3f %8.3f %8.1f q%s %s\n" % ((s - first) / 1e6, (e - first) / 1e6, (e - s) / 1e3, q, n))
    print(open(out).read() if len(sys.argv) <= 4 else "written: " + out)


if __name__ == "__main__":
    main()
