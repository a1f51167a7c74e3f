#!/bin/bash
# On the GPU box: every number the docs quote, for one build (TAG = e.g. r02_v7): profile set, bench lines, tool benches.
TAG=${1:-r02}
O=gpurun_out
mkdir -p $O
tools/profile_round.sh $TAG > $O/${TAG}_profile.log 2>&1 || echo "profile_round failed"
python3 bench.py > $O/${TAG}_bench_cfg2_default.json 2> $O/${TAG}_bench.err
python3 bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_steps20.json 2>> $O/${TAG}_bench.err
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/${TAG}_bench_torchrun1.json 2>> $O/${TAG}_bench.err
for T in 2 4 8 16; do BPP_HOST_THREADS=$T python3 bench.py --no-extra --no-cpu-baseline --no-traffic 2>/dev/null | tail -1; done > $O/${TAG}_host_threads.jsonl
python3 tools/bench_upload.py --threads 1,4 > $O/${TAG}_bench_upload.jsonl 2>> $O/${TAG}_bench.err
python3 tools/bench_latency.py > $O/${TAG}_bench_latency.jsonl 2>> $O/${TAG}_bench.err
python3 tools/bench_prove.py > $O/${TAG}_bench_prove.jsonl 2>> $O/${TAG}_bench.err
echo "final_round $TAG done"
