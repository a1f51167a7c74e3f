#!/bin/bash
set -o pipefail
O=gpurun_out; mkdir -p $O
bash tools/gpu_r06_ct2.sh
BPP_CT=2 bash tools/gpu_prover_trace.sh $O/r06_prover_launches_ct2.txt && tail -60 $O/r06_prover_launches_ct2.txt
BPP_CT=1 bash tools/gpu_prover_trace.sh $O/r06_prover_launches_ct1.txt && tail -30 $O/r06_prover_launches_ct1.txt
