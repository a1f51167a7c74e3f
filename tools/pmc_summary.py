#!/usr/bin/env python3
"""Summaries of rocprofv3 output directories (csv format) for profiles/.

  pmc_summary.py stats <dir> <out.csv>            per-kernel dispatch count / total / average duration (kernel trace)
  pmc_summary.py counters <dir> <out.csv>         per-kernel sums of every collected counter (+ derived fractions for SQ_*)
  pmc_summary.py traffic <fetch_dir> <write_dir> <kernel substring> <out.json> [calib.json]
                                                  HBM bytes per launch of one kernel from a FETCH_SIZE pass and a WRITE_SIZE pass
  pmc_summary.py calib <fetch_dir> <write_dir> <true_bytes.txt> <out.json>
                                                  reported / true byte ratios of tools/microbench/fetch_calib's kernels
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def _csvs(d, suffix):
    return sorted(glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True))


def _short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"(?:bpp::)?([A-Za-z_0-9:<>, ]+?)\(", name)
    return (m.group(1) if m else name).replace("bpp::", "").strip()


def kernel_stats(d):
    rows = collections.OrderedDict()
    for f in _csvs(d, "kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            k = _short(r["Kernel_Name"])
            dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            e = rows.setdefault(k, [0, 0, None, 0, r.get("VGPR_Count", ""), r.get("SGPR_Count", ""), r.get("Scratch_Size", r.get("Private_Segment_Size", ""))])
            e[0] += 1
            e[1] += dur
            e[2] = dur if e[2] is None else min(e[2], dur)
            e[3] = max(e[3], dur)
    return rows


def cmd_stats(d, out):
    rows = kernel_stats(d)
    tot = sum(e[1] for e in rows.values()) or 1
    with open(out, "w") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "dispatches", "total_us", "avg_us", "min_us", "max_us", "share_of_kernel_time", "vgpr", "sgpr", "scratch"])
        for k, e in sorted(rows.items(), key=lambda kv: -kv[1][1]):
            w.writerow([k, e[0], "%.1f" % (e[1] / 1e3), "%.2f" % (e[1] / e[0] / 1e3), "%.2f" % (e[2] / 1e3), "%.2f" % (e[3] / 1e3),
                        "%.4f" % (e[1] / tot), e[4], e[5], e[6]])


def counters(d):
    """{kernel: {"dispatches": n, "dur_ns": total, counter: sum}}"""
    acc = collections.OrderedDict()
    seen = set()
    for f in _csvs(d, "counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = _short(r["Kernel_Name"])
            e = acc.setdefault(k, collections.defaultdict(float))
            did = (f, r.get("Dispatch_Id"))
            if did not in seen:
                seen.add(did)
                e["dispatches"] += 1
                if r.get("Start_Timestamp") and r.get("End_Timestamp"):
                    e["dur_ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                try:  # workgroups of the dispatch: what "per dispatch" has to be divided by to compare launches of different sizes
                    e["workgroups"] += int(r.get("Grid_Size", 0) or 0) // max(1, int(r.get("Workgroup_Size", 1) or 1))
                except ValueError:
                    pass
            e[r["Counter_Name"]] += float(r["Counter_Value"])
    return acc


def cmd_counters(d, out):
    acc = counters(d)
    names = sorted({c for e in acc.values() for c in e if c not in ("dispatches", "dur_ns", "workgroups")})
    tot_valu = sum(e.get("SQ_INSTS_VALU", 0.0) for e in acc.values()) or 1.0
    with open(out, "w") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "dispatches", "avg_us"] + names + ["wait_inst_frac", "active_valu_frac", "valu_share", "valu_per_dispatch",
                                                               "workgroups_per_dispatch", "valu_per_workgroup"])
        for k, e in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", kv[1].get("dur_ns", 0))):
            wc = e.get("SQ_WAVE_CYCLES", 0.0)
            w.writerow([k, int(e["dispatches"]), "%.1f" % (e["dur_ns"] / max(e["dispatches"], 1) / 1e3)] + ["%d" % e.get(c, 0) for c in names] +
                       ["%.3f" % (e.get("SQ_WAIT_INST_ANY", 0) / wc) if wc else "", "%.3f" % (e.get("SQ_ACTIVE_INST_VALU", 0) / wc) if wc else "",
                        "%.4f" % (e.get("SQ_INSTS_VALU", 0) / tot_valu), "%d" % (e.get("SQ_INSTS_VALU", 0) / max(e["dispatches"], 1)),
                        "%d" % (e.get("workgroups", 0) / max(e["dispatches"], 1)), ("%d" % (e.get("SQ_INSTS_VALU", 0) / e["workgroups"])) if e.get("workgroups") else ""])


def _per_launch(d, counter, sub):
    acc = counters(d)
    for k, e in acc.items():
        if sub in k and e.get(counter):
            return e[counter] / e["dispatches"], int(e["dispatches"]), k
    raise SystemExit("no %s for a kernel matching %r in %s" % (counter, sub, d))


def cmd_traffic(fd, wd, sub, out, calib=None):
    f, nf, k = _per_launch(fd, "FETCH_SIZE", sub)
    w, nw, _ = _per_launch(wd, "WRITE_SIZE", sub)
    j = {"kernel": k, "fetch_size_kb_per_launch": f, "write_size_kb_per_launch": w, "dispatches": [nf, nw],
         "hbm_bytes_per_launch_raw": (f + w) * 1024,
         "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and, in its own pass, --pmc WRITE_SIZE (csv), one step in flight; "
                   "bytes = counter x 1024 averaged over the kernel's dispatches"}
    if calib and os.path.exists(calib):
        c = json.load(open(calib))
        j["calibration"] = c
        rf, rw = c["ratios"]["k_gather<128, 30>"]["FETCH_SIZE"], c["ratios"]["k_store160"]["WRITE_SIZE"]
        j["hbm_bytes_per_launch"] = f * 1024 / rf + w * 1024 / rw
        j["correction"] = "FETCH_SIZE / %.3f (128-byte entry gather), WRITE_SIZE / %.3f (160-byte scattered stores): reported / true ratios " \
                          "measured with tools/microbench/fetch_calib in the same session" % (rf, rw)
    else:
        j["hbm_bytes_per_launch"] = j["hbm_bytes_per_launch_raw"]
    json.dump(j, open(out, "w"), indent=1)


def cmd_calib(fd, wd, true_txt, out):
    txt = open(true_txt).read()
    true = {"k_stream16": int(re.search(r"k_stream16 (\d+)", txt).group(1)), "k_gather<128, 30>": int(re.search(r"k_gather<128> (\d+)", txt).group(1)),
            "k_gather<120, 30>": int(re.search(r"k_gather<120> (\d+)", txt).group(1)), "k_store160": int(re.search(r"k_store160 (\d+)", txt).group(1))}
    fa, wa = counters(fd), counters(wd)
    ratios = {}
    for k, tb in true.items():
        r = {}
        for acc, c in ((fa, "FETCH_SIZE"), (wa, "WRITE_SIZE")):
            for kk, e in acc.items():
                if kk.replace(" ", "") == k.replace(" ", "") or kk.startswith(k.split("<")[0]) and k.split("<")[-1][:3] in kk:
                    r[c] = e.get(c, 0.0) * 1024 / e["dispatches"] / tb
        ratios[k] = r
    json.dump({"true_bytes": true, "ratios": ratios,
               "note": "ratio = counter x 1024 / bytes the kernel really moved (each byte of a 4 GiB buffer touched once); k_stream16 shows the "
                       "documented x0.5 of FETCH_SIZE for wide coalesced reads"}, open(out, "w"), indent=1)


if __name__ == "__main__":
    {"stats": cmd_stats, "counters": cmd_counters, "traffic": cmd_traffic, "calib": cmd_calib}[sys.argv[1]](*sys.argv[2:])
