#!/bin/bash
# steps in flight, long and short timed regions, alternating on one box
out=${1:-gpurun_out/conc_ab.txt}
: > $out
for rep in 1 2 3; do
for conc in 3 4; do
for steps in 20 256; do
  r=$(python bench.py --steps $steps --warmup 5 --concurrency $conc --no-extra --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('value %.3f M  ms_per_step %.3f  clock %.3f' % (d['value']/1e6, d['ms_per_step'], d['shader_clock_ghz']))")
  echo "rep=$rep steps=$steps concurrency=$conc $r" >> $out
done
done
done
sort -k2,2 -k3,3 $out
