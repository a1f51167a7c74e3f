#!/bin/bash
set -o pipefail
O=gpurun_out; mkdir -p $O
: > $O/r06_ct_inflight.jsonl
for rep in 1 2; do for cfg in "1 1" "2 1" "2 2"; do set -- $cfg
  BPP_CT=$1 BPP_CT_BACK=$2 timeout -k 10 200 python tools/prover_inflight.py 2>/dev/null | grep "^{" | tee -a $O/r06_ct_inflight.jsonl
done; done
