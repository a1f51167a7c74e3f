#!/bin/bash
# On the GPU box: every file tools/results_table.py reads, for one build (TAG = e.g. r03_v2): profile set, bench lines, tool
# benches.  Steps are joined so that a failing GPU step stops the script (no further GPU work after a failure).
set -e -o pipefail
TAG=${1:-r03}
PART=${2:-all}   # a: profile set + bench lines, b: the probes (two gpurun calls: together they exceed one call's time limit)
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
if [ "$PART" != "b" ]; then
(cd tools/microbench && (test -x fetch_calib || hipcc -O3 --offload-arch=gfx950 fetch_calib.hip -o fetch_calib))
timeout -k 10 700 bash tools/profile_round.sh $TAG > $O/${TAG}_profile.log 2>&1
echo "profile set done"
timeout -k 10 700 python3 bench.py > $O/${TAG}_bench_full.json 2> $O/${TAG}_bench.err
echo "full bench done"
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-traffic > $O/${TAG}_bench_steps20.json 2>> $O/${TAG}_bench.err
timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/${TAG}_bench_torchrun1.json 2>> $O/${TAG}_bench.err
echo "bench lines done"
fi
if [ "$PART" = "a" ]; then echo "final_round $TAG part a done"; exit 0; fi
timeout -k 10 300 python3 tools/wave_probe.py "1x1,1x16,3x12,1g16,1g64,3g32,1p2g64" 20 4096 2>> $O/${TAG}_bench.err | grep "^{" > $O/${TAG}_wave_probe.jsonl
timeout -k 10 300 python3 tools/wave_probe.py "1x16,1g16,1g64,3g32,1p2g64" 20 512 2>> $O/${TAG}_bench.err | grep "^{" >> $O/${TAG}_wave_probe.jsonl
timeout -k 10 300 python3 tools/chunk_probe.py "128,256,512,1024,2048,4096" 64 2>> $O/${TAG}_bench.err | grep "^{" > $O/${TAG}_chunk_probe.jsonl
WIDE_PROBE_N=256 WIDE_PROBE_S=1,4,8,16 timeout -k 10 300 python3 tools/wide_probe.py 2>> $O/${TAG}_bench.err | grep "^{" > $O/${TAG}_calls_in_flight.jsonl
WIDE_PROBE_HOST=1 WIDE_PROBE_N=256 WIDE_PROBE_S=1,4,8,16,32,64 timeout -k 10 300 python3 tools/wide_probe.py 2>> $O/${TAG}_bench.err | grep "^{" >> $O/${TAG}_calls_in_flight.jsonl
WIDE_PROBE_HOST=2 WIDE_PROBE_N=256 WIDE_PROBE_S=1,4,8,16,32,64 timeout -k 10 300 python3 tools/wide_probe.py 2>> $O/${TAG}_bench.err | grep "^{" >> $O/${TAG}_calls_in_flight.jsonl
# the same separate callers with the admission gate switched off (what round 3 measured)
BPP_SMALL_CALLS_IN_FLIGHT=0 WIDE_PROBE_HOST=1 WIDE_PROBE_N=256 WIDE_PROBE_S=16,32,64 timeout -k 10 300 python3 tools/wide_probe.py 2>> $O/${TAG}_bench.err | grep "^{" | sed 's/"form": "packed"/"form": "packed, gate off"/' >> $O/${TAG}_calls_in_flight.jsonl
timeout -k 10 120 python3 tools/bench_prover_leg.py > $O/${TAG}_prover_leg.json 2>> $O/${TAG}_bench.err
# the prover on configs[4]: kernel timeline of one call; SQ counters per kernel with ONE sub-batch stream (a dispatch = all 1024
# proofs; the headline's counter pass only sees the input generation's dispatches: 4096 non-aggregated proofs each); phase clocks
timeout -k 10 200 bash tools/gpu_prover_trace.sh $O/${TAG}_prover_launches.txt
rm -rf $O/prov_pmc
BPP_PROVE_SUBS=1 PROVER_ITERS=3 timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/prov_pmc -- python3 tools/bench_prover_leg.py > /dev/null 2>> $O/${TAG}_bench.err
python3 tools/pmc_summary.py counters $O/prov_pmc $O/${TAG}_prover_kernels_serial.csv
rm -rf $O/prov_pmc
# the multi-rank code as two PROCESSES on this one GPU (gloo transport through bpp_comm_create_callbacks)
BPP_BENCH_ONE_DEVICE=1 timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic > $O/${TAG}_bench_two_ranks_one_gpu.json 2>> $O/${TAG}_bench.err
timeout -k 10 300 python3 tools/bench_latency.py > $O/${TAG}_bench_latency.jsonl 2>> $O/${TAG}_bench.err
BPP_MSM_SPLIT=0 timeout -k 10 300 python3 tools/bench_latency.py --no-cpu > $O/${TAG}_bench_latency_nosplit.jsonl 2>> $O/${TAG}_bench.err
# last: the round kernel's phase clocks from a measurement build (its own file, loaded through BPP_LIB_PATH: nothing above or
# after it can pick it up by accident)
test -f gpurun_in/kp_phases.so && timeout -k 10 200 bash tools/gpu_kp_phases.sh $O/${TAG}_kp_phases.json > /dev/null 2>> $O/${TAG}_bench.err
echo "final_round $TAG done"
