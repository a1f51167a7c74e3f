#!/bin/bash
# round 6, first contact: the wavefront Keccak's latency, the device-chain tests, the headline with the chains on either side
set -o pipefail
O=gpurun_out
mkdir -p $O
timeout -k 10 60 tools/microbench/keccak_wave > $O/r06_keccak_wave.txt 2>&1; echo "keccak_wave rc $?"; cat $O/r06_keccak_wave.txt
timeout -k 10 600 python -m pytest tests/test_gpu_round6.py -x -q > $O/r06_tests_chain.log 2>&1; echo "tests rc $?"; tail -15 $O/r06_tests_chain.log
for mode in host device; do
  for c in 3 5; do
    timeout -k 10 200 python bench.py --steps 300 --warmup 20 --no-extra --no-cpu-baseline --no-traffic --chain $mode --concurrency $c > $O/r06_first_${mode}_c$c.json 2> $O/r06_first_${mode}_c$c.err
    echo "bench $mode c$c rc $?"
    python - <<PY
import json
try:
    d=json.loads(open("$O/r06_first_${mode}_c$c.json").read().strip().splitlines()[-1])
    print("$mode c$c: %.2f M/s  ms/step %.3f  cores %.2f  chain_cpu_ms %.2f proc_cpu_ms %.2f" % (d["value"]/1e6, d["ms_per_step"], d["host_cores_busy"], d["host_chain_cpu_ms_per_step"], d["host_process_cpu_ms_per_step"]))
    print("   ", d["host_cores_busy_by_thread"])
    print("   ", d.get("stages_ms"))
except Exception as e:
    print("no line", e)
PY
  done
done
