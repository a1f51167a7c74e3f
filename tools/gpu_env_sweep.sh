#!/bin/bash
# On the GPU box: the headline bench under several environment settings (same build, same box).
# usage: tools/gpu_env_sweep.sh <tag> "VAR=x VAR2=y" "..." ...    (an empty string = defaults)
TAG=$1; shift
mkdir -p gpurun_out
i=0
for ENVS in "$@"; do
  i=$((i+1))
  env $ENVS python3 bench.py --no-extra --no-cpu-baseline --no-traffic > gpurun_out/${TAG}_e${i}.json 2> gpurun_out/${TAG}_e${i}.err
  python3 - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_e${i}.json").read().strip().split("\n")[-1])
s = d["stages_ms"]
print("env '${ENVS}': %.2f M proofs/s, %.3f ms/step, acc alone %.3f ms | dec %.2f scal %.2f acc %.2f red %.2f" % (d["value"] / 1e6, d["ms_per_step"], d["roofline"]["alone"]["kernel_ms"], s["decompress_ms"], s["scalars_ms"], s["msm_accumulate_ms"], s["msm_bucket_reduce_ms"]))
PY
done
