#!/bin/bash
# the driver's form (--steps 20 --warmup 5) against the untimed pre-heat and the steps in flight: one box, alternating
out=${1:-gpurun_out/preheat_sweep.txt}
: > $out
for rep in 1 2; do
for ph in 150 400 1000; do
for conc in 3 4 5; do
  r=$(python bench.py --steps 20 --warmup 5 --preheat-ms $ph --concurrency $conc --no-extra --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('value %.3f M  ms_per_step %.3f  clock %.3f' % (d['value']/1e6, d['ms_per_step'], d['shader_clock_ghz']))")
  echo "rep=$rep preheat_ms=$ph concurrency=$conc $r" >> $out
done
done
done
cat $out
