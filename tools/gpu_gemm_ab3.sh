#!/bin/bash
# matrix-product columns on / off at three and four steps in flight
out=${1:-gpurun_out/gemm_ab3.txt}
: > $out
for rep in 1 2; do
for c in 3; do
for v in 0 1 2; do
  r=$(BPP_STATIC_GEMM=$v python bench.py --steps 256 --warmup 5 --concurrency $c --no-extra --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stages_ms']; print('value %.3f M  ms_per_step %.3f  clock %.3f  scalars %.3f reduce %.3f total %.2f' % (d['value']/1e6, d['ms_per_step'], d['shader_clock_ghz'], s['scalars_ms'], s['reduce_ms'], s['total_ms']))")
  echo "rep=$rep conc=$c BPP_STATIC_GEMM=$v $r" >> $out
done
done
done
sort -k2,3 -s $out
