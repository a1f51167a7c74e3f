#!/usr/bin/env python3
"""What ramps after the load starts?  Runs a command (default: the headline with a 3 s pre-heat) as a child and samples the
amdgpu sysfs files of card 0 every 50 ms beside it: the active DPM level of sclk / mclk / fclk / socclk and the socket power.
  tools/dpm_watch.py [--out FILE] [-- command ...]        (this process never touches the GPU itself)"""
import glob
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def active(path):
    try:
        for line in open(path):
            if line.rstrip().endswith("*"):
                return line.split(":", 1)[1].replace("*", "").strip()
    except OSError:
        return None
    return "-"


def main():
    argv = sys.argv[1:]
    out = os.path.join("gpurun_out", "dpm_watch.txt")
    if argv[:1] == ["--out"]:
        out, argv = argv[1], argv[2:]
    cmd = argv[1:] if argv[:1] == ["--"] else [sys.executable, "bench.py", "--no-extra", "--no-cpu-baseline", "--no-traffic", "--preheat-ms", "3000"]
    devs = [os.path.dirname(x) for x in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))]
    if not devs:
        print("no amdgpu sysfs files visible")
        return 1
    # which card is ours is not known in advance (the box shows every card of the host): a mark file written by the command's first
    # GPU work would need the GPU here; instead every card is sampled and the one whose busy figure follows the command is printed
    p = subprocess.Popen(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    t0 = time.perf_counter()
    samples = {d: [] for d in devs}
    while p.poll() is None:
        t = time.perf_counter() - t0
        for dev in devs:
            row = ["%7.3f" % t] + ["%s %s" % (k, active(os.path.join(dev, "pp_dpm_" + k))) for k in ("sclk", "mclk", "fclk", "socclk")]
            for pw in sorted(glob.glob(os.path.join(dev, "hwmon", "hwmon*", "power1_*")))[:1]:
                try:
                    row.append("power %.0f W" % (int(open(pw).read()) / 1e6))
                except (OSError, ValueError):
                    pass
            try:
                b = int(open(os.path.join(dev, "gpu_busy_percent")).read())
            except (OSError, ValueError):
                b = -1
            row.append("busy %d" % b)
            samples[dev].append((b, "  ".join(row)))
        time.sleep(0.05)
    # ours: idle while the command starts up (imports), busy afterwards
    def follows(dev):
        v = [b for b, _ in samples[dev]]
        n = max(1, len(v) // 10)
        return sum(v[-3 * n:]) / (3 * n) - sum(v[:n]) / n
    dev = max(devs, key=follows)
    rows = [r for _, r in samples[dev]]
    tail = p.stdout.read().strip().splitlines()
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    with open(out, "w") as f:
        f.write("# %s (device %s)\n" % (" ".join(cmd), dev))
        last = None
        for r in rows:  # only the rows where something other than the time stamp changed, and every 10th
            key = r.split("  ", 1)[1]
            if key != last or rows.index(r) % 10 == 0:
                f.write(r + "\n")
            last = key
        f.write("# last line of the command: %s\n" % (tail[-1][:300] if tail else "none"))
    print(open(out).read()[-3000:])
    return 0


if __name__ == "__main__":
    sys.exit(main())
