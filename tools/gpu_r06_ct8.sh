#!/bin/bash
set -o pipefail
O=gpurun_out; mkdir -p $O
: > $O/r06_ct_back_ab.txt
for rep in 1 2 3; do for cfg in "1 1" "2 1" "2 2" "2 3"; do set -- $cfg
  BPP_CT=$1 BPP_CT_BACK=$2 timeout -k 10 200 python tools/bench_prover_leg.py 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('ct=$1 back=$2 rep=$rep: %.1f k proofs/s one call at a time, %.3f ms per call, engine %.3f ms, fb %.3f ms' % (d['proofs_per_s'] / 1e3, d['ms_per_call'], d['engine_total_ms'], d['fb_msm_ms']))" | tee -a $O/r06_ct_back_ab.txt
done; done
