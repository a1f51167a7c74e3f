#!/bin/bash
# A/B of one environment variable on the prover leg (configs[4], one call at a time + four calls in flight): alternating, one box
var=$1; a=$2; b=$3; out=${4:-gpurun_out/env_ab_prover.txt}
: > $out
for rep in 1 2 3; do
for v in $a $b; do
  r=$(env $var=$v python tools/bench_prover_leg.py 2>/dev/null | tail -1)
  echo "rep=$rep $var=$v $r" >> $out
done
done
sort -k2,2 $out
