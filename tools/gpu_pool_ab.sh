#!/bin/bash
# host pool sized by usable CPUs (new default) against the 32 workers hardware_concurrency() gave: headline (256 and 20 steps),
# the pipelined sharded form on one rank, alternating on one box
out=${1:-gpurun_out/pool_ab.txt}
: > $out
for rep in 1 2; do
for v in "BPP_HOST_THREADS=32" "BPP_DUMMY=1"; do
  h=$(env $v python bench.py --steps 256 --warmup 5 --no-extra --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('256: %.3f M %.3f ms chain %.2f host_threads %d' % (d['value']/1e6, d['ms_per_step'], d['stages_ms']['chain_host_ms'], d['host_threads']))")
  h2=$(env $v python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('20: %.3f M %.3f ms chain %.2f' % (d['value']/1e6, d['ms_per_step'], d['stages_ms']['chain_host_ms']))")
  w=$(env $v python tools/wave_probe.py "1p2g64" 20 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    t = d['last_wave_host_ms']
    print('pipe2 %.2f M call %.2f ms chains %.2f' % (d['proofs_per_s'] / 1e6, d['ms_per_wave'], t['chains_ms']))
")
  echo "rep=$rep $v | $h | $h2 | $w" >> $out
done
done
sort -k2,2 $out
