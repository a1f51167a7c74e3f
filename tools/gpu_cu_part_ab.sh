#!/bin/bash
# steps in flight on disjoint sets of compute units (BPP_CU_PARTITIONS = N, contexts take turns) against the shared chip
out=${1:-gpurun_out/cu_part_ab.txt}
: > $out
for rep in 1 2; do
for cfg in "0 3" "4 4" "2 2" "2 4" "4 8" "8 8"; do
  set -- $cfg
  r=$(BPP_CU_PARTITIONS=$1 timeout -k 10 200 python bench.py --steps 128 --warmup 5 --concurrency $2 --no-extra --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stages_ms']; print('value %.3f M  ms_per_step %.3f  clock %.3f  total %.2f' % (d['value']/1e6, d['ms_per_step'], d['shader_clock_ghz'], s['total_ms']))")
  echo "rep=$rep partitions=$1 in_flight=$2 $r" >> $out
done
done
sort -k2,3 -s $out
