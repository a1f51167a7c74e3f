#!/bin/bash
# phase clocks of the prover's round kernel from a MEASUREMENT build of the library: gpurun_in/kp_phases.so = engine.hip with
# -DBPP_KP_PHASES and the product build's flags, rebuilt HERE whenever it is older than any source.  The product's libbpp_hip.so is
# never touched: tools/kp_phases.py loads the measurement build through BPP_LIB_PATH (bulletproofs-plus_amd/_lib.py).
set -e
out=${1:-gpurun_out/kp_phases.json}
so=gpurun_in/kp_phases.so
stale=0
[ -f $so ] || stale=1
if [ $stale = 0 ]; then
  for f in bulletproofs-plus_amd/csrc/* include/bpp.h; do [ "$f" -nt $so ] && stale=1; done
fi
if [ $stale = 1 ]; then
  mkdir -p gpurun_in
  (cd bulletproofs-plus_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC $BPP_HIPCC_FLAGS -DBPP_KP_PHASES -o ../../$so engine.hip)
fi
BPP_LIB_PATH=$so python3 tools/kp_phases.py > $out
cat $out
