#!/bin/bash
# configs[2] (256 x aggregation-8 batches, 64 per step) with the generator columns per proof (0), as workgroup (1) and one-wavefront (2) matrix products
out=${1:-gpurun_out/gemm_ab5.txt}
: > $out
for rep in 1 2 3; do
for v in 0 1 2; do
  r=$(BPP_STATIC_GEMM=$v python bench.py --only cfg3 --steps 48 --warmup 8 2>/dev/null | tail -1)
  echo "rep=$rep BPP_STATIC_GEMM=$v cfg3 $r" >> $out
done
done
sort -k2,2 -s $out
