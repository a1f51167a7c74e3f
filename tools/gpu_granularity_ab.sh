#!/bin/bash
# the same work in flight cut finer: batches per step x steps in flight (64 x 3 is the default); 256 steps' worth of proofs each
out=${1:-gpurun_out/granularity_ab.txt}
: > $out
for rep in 1 2; do
for cfg in "64 3 256" "32 6 512" "32 4 512" "16 12 1024" "16 8 1024" "128 2 128" "96 2 170"; do
  set -- $cfg
  r=$(timeout -k 10 200 python bench.py --steps $3 --warmup 6 --batches-per-step $1 --concurrency $2 --no-extra --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('value %.3f M  ms_per_step %.3f  clock %.3f' % (d['value']/1e6, d['ms_per_step'], d['shader_clock_ghz']))")
  echo "rep=$rep batches_per_step=$1 in_flight=$2 steps=$3 $r" >> $out
done
done
sort -k2,3 -s $out
