#!/bin/bash
# A/B of two builds of libbpp_hip.so (gpurun_in/<a>.so, gpurun_in/<b>.so) on one box, alternating: gpu_lib_ab.sh a b "<command>" out
a=$1; b=$2; cmd=$3; out=${4:-gpurun_out/lib_ab.txt}
: > $out
cp bulletproofs-plus_amd/libbpp_hip.so /tmp/libbpp_saved.so
for rep in 1 2 3; do
for v in $a $b; do
  cp gpurun_in/$v.so bulletproofs-plus_amd/libbpp_hip.so; touch bulletproofs-plus_amd/libbpp_hip.so
  r=$(eval "$cmd" 2>/dev/null | tail -1)
  echo "rep=$rep lib=$v $r" >> $out
done
done
cp /tmp/libbpp_saved.so bulletproofs-plus_amd/libbpp_hip.so
sort -k2,2 $out
