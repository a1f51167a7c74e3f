#!/bin/bash
# round 6: headline with the chains on either side, over steps in flight; wait mode (BPP_WAIT) beside it
set -o pipefail
O=gpurun_out
mkdir -p $O
: > $O/r06_sweep.txt
run() {  # mode conc wait
  BPP_WAIT=$3 timeout -k 10 200 python bench.py --steps 300 --warmup 20 --no-extra --no-cpu-baseline --no-traffic --chain $1 --concurrency $2 > $O/r06_sw.json 2> $O/r06_sw.err
  python - "$1" "$2" "$3" <<'PY' | tee -a gpurun_out/r06_sweep.txt
import json, sys
try:
    d = json.loads(open("gpurun_out/r06_sw.json").read().strip().splitlines()[-1])
    print("chain %-6s c%s wait %2s: %.2f M/s  ms/step %.3f  latency %.2f ms  cores %.2f  %s" % (sys.argv[1], sys.argv[2], sys.argv[3], d["value"] / 1e6, d["ms_per_step"], d["step_latency_ms"], d["host_cores_busy"],
          " ".join("%s:%.2f" % (e["thread"], e["cores_busy"]) for e in d["host_cores_busy_by_thread"])))
except Exception as e:
    print("no line", sys.argv[1:], e)
PY
}
for c in 3 4; do run host $c -1; run host-wide $c -1; done
for c in 3 4 5 6 8; do run device $c -1; done
run device 5 -1
