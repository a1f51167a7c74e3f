#!/bin/bash
# On the GPU box: rebuild libbpp_hip.so with extra hipcc flags and run the headline bench for each variant.
# usage: tools/gpu_variants.sh <tag> "<flags variant 1>" "<flags variant 2>" ...   (an empty string = default build)
set -e -o pipefail
TAG=$1; shift
mkdir -p gpurun_out
i=0
for FLAGS in "$@"; do
  i=$((i+1))
  echo "== variant $i: '$FLAGS'"
  BPP_HIPCC_FLAGS="$FLAGS" python3 -c "import importlib; p=importlib.import_module('bulletproofs-plus_amd'); p._build.build(force=True)"
  python3 bench.py --no-extra --no-cpu-baseline --no-traffic > gpurun_out/${TAG}_v${i}.json 2> gpurun_out/${TAG}_v${i}.err
  python3 - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_v${i}.json").read().strip().split("\n")[-1])
print("variant ${i} '${FLAGS}': %.2f M proofs/s, %.3f ms/step, acc alone %.3f ms" % (d["value"] / 1e6, d["ms_per_step"], d["roofline"]["alone"]["kernel_ms"]), d["stages_ms"])
PY
done
