#!/bin/bash
# the pipelined grouped sharded form on one rank (1p2g64) and the grouped form (3g32), with the collective deadline on (default)
# and off (BPP_COMM_TIMEOUT_MS=0: blocking waits), and fused column sums on / off: alternating on one box
out=${1:-gpurun_out/wave_ab.txt}
: > $out
for rep in 1 2; do
for v in "BPP_COMM_TIMEOUT_MS=60000" "BPP_COMM_TIMEOUT_MS=0" "BPP_FUSED_COLUMNS=0"; do
  r=$(env $v python tools/wave_probe.py "1p2g64,3g32" 20 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    t = d['last_wave_host_ms']
    print('%s %.2f M  call %.2f ms chains %.2f gather1 %.2f wait2 %.2f' % (d.get('form'), d['proofs_per_s'] / 1e6, d['ms_per_wave'], t['chains_ms'], t['gather1_ms'], t['wait2_ms']), end=' | ')
")
  echo "rep=$rep $v $r" >> $out
done
done
sort -k2,2 $out
