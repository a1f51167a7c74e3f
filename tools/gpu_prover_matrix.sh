#!/bin/bash
# prover rate (configs[4], one call at a time) against sub-batch streams per call (and lanes per output of k_fb_msm): one box
out=${1:-gpurun_out/prover_matrix.txt}
: > $out
for rep in 1 2; do
for subs in 1 2 3 4; do
    r=$(BPP_PROVE_SUBS=$subs python bench.py --only prover --no-extra --no-cpu-baseline --no-traffic 2>/dev/null | tail -1)
    echo "rep=$rep subs=$subs $r" >> $out
done
done
cat $out
