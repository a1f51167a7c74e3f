#!/bin/bash
# A/B of the runtime's active-wait time (ROC_ACTIVE_WAIT_TIMEOUT, microseconds of spinning before a blocked wait) on the driver form
# of the headline and on the single-call latencies
out=${1:-gpurun_out/wait_ab.txt}
: > $out
for rep in 1 2; do
for v in 0 200 100000; do
  r=$(ROC_ACTIVE_WAIT_TIMEOUT=$v python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('value %.3f M  ms_per_step %.3f  clock %.3f' % (d['value']/1e6, d['ms_per_step'], d['shader_clock_ghz']))")
  echo "rep=$rep wait=$v steps=20 $r" >> $out
  r=$(ROC_ACTIVE_WAIT_TIMEOUT=$v python tools/bench_latency.py --no-cpu --sizes 1,64,256,1024 2>/dev/null | grep '^{' | python -c "
import sys,json
o=[]
for l in sys.stdin:
    d=json.loads(l)
    o.append('%d:%.3f' % (d['batch'], d['gpu_ms_median']))
print(' '.join(o))")
  echo "rep=$rep wait=$v latency $r" >> $out
done
done
sort -k2,2 -s $out
