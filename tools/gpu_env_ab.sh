#!/bin/bash
# A/B of one environment variable on the headline leg: gpu_env_ab.sh VAR A B [steps] [out]   (alternating runs on one box)
var=$1; a=$2; b=$3; steps=${4:-256}; out=${5:-gpurun_out/env_ab.txt}
: > $out
for rep in 1 2 3; do
for v in $a $b; do
  r=$(env $var=$v python bench.py --steps $steps --warmup 5 --no-extra --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stages_ms']; print('value %.3f M  ms_per_step %.3f  clock %.3f  scalars %.3f reduce %.3f total %.2f' % (d['value']/1e6, d['ms_per_step'], d['shader_clock_ghz'], s['scalars_ms'], s['reduce_ms'], s['total_ms']))")
  echo "rep=$rep $var=$v steps=$steps $r" >> $out
done
done
sort -k2,2 $out
