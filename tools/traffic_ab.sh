#!/bin/bash
# On the GPU box: FETCH_SIZE / WRITE_SIZE of k_msm_accumulate per launch and its un-shared duration for several builds.
# usage: tools/traffic_ab.sh <tag> "<flags 1>" "<flags 2>" ...
set -e -o pipefail
TAG=$1; shift
export TMPDIR=/tmp
mkdir -p gpurun_out
i=0
for FLAGS in "$@"; do
  i=$((i+1))
  BPP_HIPCC_FLAGS="$FLAGS" python3 -c "import importlib; p=importlib.import_module('bulletproofs-plus_amd'); p._build.build(force=True)"
  P=gpurun_out/prof_${TAG}_$i; rm -rf $P
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/f -- python3 bench.py --no-extra --no-cpu-baseline --no-traffic --concurrency 1 --steps 8 --warmup 2 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/w -- python3 bench.py --no-extra --no-cpu-baseline --no-traffic --concurrency 1 --steps 8 --warmup 2 > /dev/null 2>&1
  python3 tools/pmc_summary.py traffic $P/f $P/w k_msm_accumulate gpurun_out/${TAG}_traffic_v$i.json
  python3 - <<PY
import json
j = json.load(open("gpurun_out/${TAG}_traffic_v$i.json"))
print("variant $i '$FLAGS': FETCH_SIZE %.1f MB raw, WRITE_SIZE %.1f MB raw per launch" % (j["fetch_size_kb_per_launch"] / 1024, j["write_size_kb_per_launch"] / 1024))
PY
  rm -rf $P
done
