#!/bin/bash
# On the GPU box: the rocprofv3 passes whose summaries go to profiles/ (run from the repo root; TAG = e.g. r02_v2).
#   1. kernel trace of the default bench command            -> gpurun_out/<TAG>_kernel_stats_cfg2_default.csv (+ the bench line)
#   2. SQ counters, one step in flight                       -> gpurun_out/<TAG>_pmc_sq_summary.csv
#   3. FETCH_SIZE and WRITE_SIZE, separate passes            -> gpurun_out/<TAG>_traffic.json
#   4. the same two counters on tools/microbench/fetch_calib -> gpurun_out/<TAG>_fetch_calib.json (calibration of 3)
# Counter passes use --kernel-trace only (gpurun refuses --pmc together with the hip/hsa/memory trace domains).
set -e -o pipefail
TAG=${1:-r02}
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
P=$O/prof_$TAG
rm -rf $P
rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -- python3 bench.py --no-extra --no-cpu-baseline --no-traffic > $O/${TAG}_bench_profiled_run.json 2> $O/${TAG}_trace.err
python3 tools/pmc_summary.py stats $P/trace $O/${TAG}_kernel_stats_cfg2_default.csv
echo "trace done"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d $P/sq -- python3 bench.py --no-extra --no-cpu-baseline --no-traffic --concurrency 1 --steps 6 --warmup 2 > $O/${TAG}_sq_run.json 2> $O/${TAG}_sq.err
python3 tools/pmc_summary.py counters $P/sq $O/${TAG}_pmc_sq_summary.csv
echo "sq done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/fetch -- python3 bench.py --no-extra --no-cpu-baseline --no-traffic --concurrency 1 --steps 8 --warmup 2 > /dev/null 2> $O/${TAG}_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/write -- python3 bench.py --no-extra --no-cpu-baseline --no-traffic --concurrency 1 --steps 8 --warmup 2 > /dev/null 2> $O/${TAG}_write.err
echo "traffic passes done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/cfetch -- tools/microbench/fetch_calib > $O/${TAG}_fetch_calib_true.txt 2> $O/${TAG}_cfetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/cwrite -- tools/microbench/fetch_calib > /dev/null 2> $O/${TAG}_cwrite.err
python3 tools/pmc_summary.py calib $P/cfetch $P/cwrite $O/${TAG}_fetch_calib_true.txt $O/${TAG}_fetch_calib.json
python3 tools/pmc_summary.py traffic $P/fetch $P/write k_msm_accumulate $O/${TAG}_traffic.json $O/${TAG}_fetch_calib.json
python3 tools/pmc_summary.py counters $P/fetch $O/${TAG}_fetch_by_kernel.csv
python3 tools/pmc_summary.py counters $P/write $O/${TAG}_write_by_kernel.csv
rm -rf $P
echo "profile_round $TAG done"
