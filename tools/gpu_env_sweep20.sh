#!/bin/bash
# like gpu_env_sweep.sh, at the driver's 20 timed steps and at the default 256
TAG=$1; shift
mkdir -p gpurun_out
i=0
for ENVS in "$@"; do
  i=$((i+1))
  for ST in "20 5" "256 32"; do
    set -- $ST
    env $ENVS python3 bench.py --no-extra --no-cpu-baseline --no-traffic --steps $1 --warmup $2 > gpurun_out/${TAG}_e${i}_s$1.json 2> gpurun_out/${TAG}_e${i}_s$1.err
    python3 - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_e${i}_s$1.json").read().strip().split("\n")[-1])
print("env '${ENVS}' steps $1: %.2f M proofs/s, %.3f ms/step, latency %.2f" % (d["value"] / 1e6, d["ms_per_step"], d["step_latency_ms"]))
PY
  done
done
