#!/bin/bash
# parity test of the matrix-product columns, then the A/B on the headline and configs[2], then per-kernel times with one step in flight
set -e -o pipefail
out=${1:-gpurun_out/gemm_round}
timeout -k 10 300 python -m pytest tests/test_gpu_round4.py -x -q -k "matrix_product or column_sums" > $out.test.txt 2>&1 || { tail -30 $out.test.txt; exit 1; }
tail -2 $out.test.txt
timeout -k 10 600 tools/gpu_gemm_ab.sh $out.ab.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/gm1 -- python3 bench.py --steps 40 --warmup 5 --concurrency 1 --no-extra --no-cpu-baseline --no-traffic > gpurun_out/gm1.json 2> gpurun_out/gm1.err
f=$(find gpurun_out/gm1 -name "*kernel_stats.csv" | head -1)
python3 - $f > $out.alone.txt <<'PY'
import csv,re,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=re.sub(r'\(.*','',r['Name']).replace('bpp::','').replace('void ','')
    if n.startswith('k_'): print('%-28s %5s x %9.1f us' % (n[:28], r['Calls'], float(r['AverageNs'])/1e3))
PY
rm -rf gpurun_out/gm1
cat $out.alone.txt
