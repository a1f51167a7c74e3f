#!/bin/bash
set -o pipefail
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_prove.py tests/test_gpu_round5.py -x -q > $O/r06_tests_prove.log 2>&1; rc=$?; tail -4 $O/r06_tests_prove.log; [ $rc -eq 0 ] || exit 1
: > $O/r06_ct_inflight.jsonl
for rep in 1 2; do for cfg in "1 1" "2 1"; do set -- $cfg
  BPP_CT=$1 BPP_CT_BACK=$2 timeout -k 10 200 python tools/prover_inflight.py 2>/dev/null | grep "^{" | tee -a $O/r06_ct_inflight.jsonl
done; done
