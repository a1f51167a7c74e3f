#!/bin/bash
# On the GPU box: the headline and a one-step-in-flight run for several workgroup sizes of k_msm_sort (rebuilds the library).
O=gpurun_out
for T in 256 512 1024; do
  BPP_HIPCC_FLAGS="-DBPP_SORT_THREADS=$T" python3 -c "import importlib,sys; sys.path.insert(0,'.'); b=importlib.import_module('bulletproofs-plus_amd._build'); b.build(force=True)" > $O/sortT_build_$T.log 2>&1 || { echo "build $T failed"; tail -3 $O/sortT_build_$T.log; continue; }
  for rep in 1 2; do
    python3 bench.py --no-extra --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('T=$T 4 in flight', round(d['value']/1e6,2), round(d['ms_per_step'],3), round(d['shader_clock_ghz'],3), d['stages_ms']['msm_sort_ms'])"
  done
  python3 bench.py --no-extra --no-cpu-baseline --no-traffic --concurrency 1 --steps 20 --warmup 4 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('T=$T alone', round(d['value']/1e6,2), d['stages_ms']['msm_sort_ms'])"
done
