#!/bin/bash
# lazily reduced column sums in k_scalars_lanes (BPP_LAZY_COLUMNS=1, default) against one Montgomery product per (proof, generator): headline, then aggregation 2 / 4 / 8
out=${1:-gpurun_out/lazy_ab.txt}
bash tools/gpu_env_ab.sh BPP_LAZY_COLUMNS 0 1 256 $out > /dev/null
cat $out
