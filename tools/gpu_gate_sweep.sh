#!/bin/bash
# separate small callers (host buffers in, 256 proofs per call) against the admission gate's limit: one box, alternating
out=${1:-gpurun_out/gate_sweep.jsonl}
: > $out
for lim in 0 4 6 8 12 16; do
  echo "{\"small_call_limit\": $lim}" >> $out
  BPP_SMALL_CALLS_IN_FLIGHT=$lim WIDE_PROBE_N=256 WIDE_PROBE_HOST=1 WIDE_PROBE_S=1,8,16,32,64 python tools/wide_probe.py >> $out 2>&1 || exit 1
done
