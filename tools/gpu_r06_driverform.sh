#!/bin/bash
# the driver's command (--steps 20 --warmup 5, no extra legs here) a few times per setting: value, cores
O=gpurun_out; mkdir -p $O; : > $O/r06_driverform.txt
for rep in 1 2 3; do
for setting in "host-wide 3" "host-wide 4" "device 5"; do
  set -- $setting
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-traffic --chain $1 --concurrency $2 > $O/r06_df.json 2> $O/r06_df.err
  python - "$1" "$2" <<'PY' | tee -a gpurun_out/r06_driverform.txt
import json, sys
try:
    d = json.loads(open("gpurun_out/r06_df.json").read().strip().splitlines()[-1])
    print("chain %-9s c%s: %.2f M/s  ms/step %.3f  cores %.2f  clock %.2f host_bound %s" % (sys.argv[1], sys.argv[2], d["value"] / 1e6, d["ms_per_step"], d["host_cores_busy"], d["shader_clock_ghz"] or 0, d["host_bound"]))
except Exception as e:
    print("no line", sys.argv[1:], e)
PY
done; done
