#!/bin/bash
# On the GPU box: headline bench for several --concurrency values, at the driver's 20 timed steps and at 256 (same build, same box).
# usage: tools/gpu_conc_sweep.sh <tag> 2 3 4 5
TAG=$1; shift
mkdir -p gpurun_out
for C in "$@"; do
  for ST in "20 5" "256 32"; do
    set -- $ST
    python3 bench.py --no-extra --no-cpu-baseline --no-traffic --concurrency $C --steps $1 --warmup $2 > gpurun_out/${TAG}_c${C}_s$1.json 2> gpurun_out/${TAG}_c${C}_s$1.err
    python3 - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_c${C}_s$1.json").read().strip().split("\n")[-1])
print("concurrency ${C} steps $1: %.2f M proofs/s, %.3f ms/step, step latency %.2f ms" % (d["value"] / 1e6, d["ms_per_step"], d["step_latency_ms"]))
PY
  done
done
