#!/bin/bash
set -o pipefail
O=gpurun_out; mkdir -p $O
: > $O/r06_prover_wait_ab.txt
for rep in 1 2; do for w in 0 1; do for ct in 1 2; do
  BPP_WAIT=$w BPP_CT=$ct timeout -k 10 200 python tools/bench_prover_leg.py 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('wait=$w ct=$ct rep=$rep: %.1f k proofs/s one call at a time, %.3f ms per call, engine %.3f ms, fb %.3f ms' % (d['proofs_per_s'] / 1e3, d['ms_per_call'], d['engine_total_ms'], d['fb_msm_ms']))" | tee -a $O/r06_prover_wait_ab.txt
done; done; done
