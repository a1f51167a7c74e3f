#!/bin/bash
# round 6: the prover with ct = 1 / 2 (tests first, then configs[4] one call at a time and four in flight)
set -o pipefail
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_prove.py tests/test_gpu_round5.py -x -q -k "prove or ct or wipe or secret or slice" > $O/r06_tests_prove.log 2>&1; rc=$?; tail -6 $O/r06_tests_prove.log; [ $rc -eq 0 ] || exit 1
: > $O/r06_ct_ab.txt
for rep in 1 2 3; do for ct in 1 2; do
  BPP_CT=$ct timeout -k 10 200 python tools/bench_prover_leg.py 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('ct=$ct rep=$rep: %.1f k proofs/s one call at a time, %.3f ms per call, engine %.3f ms' % (d['proofs_per_s'] / 1e3, d['ms_per_call'], d['engine_total_ms']))" | tee -a $O/r06_ct_ab.txt
done; done
