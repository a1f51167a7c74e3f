#!/bin/bash
# A/B of the generator columns as a matrix product (BPP_STATIC_GEMM=1, default) against the per-proof products (0): headline and configs[2]
out=${1:-gpurun_out/gemm_ab.txt}
bash tools/gpu_env_ab.sh BPP_STATIC_GEMM 0 1 256 $out.headline > /dev/null
cat $out.headline > $out
for rep in 1 2; do
for v in 0 1; do
  r=$(BPP_STATIC_GEMM=$v python bench.py --only cfg3 --steps 48 --warmup 8 2>/dev/null | tail -1)
  echo "rep=$rep BPP_STATIC_GEMM=$v cfg3 $r" >> $out
done
done
cat $out
