#!/bin/bash
# phase clocks of the prover's round kernel: gpurun_in/kp_phases.so = engine.hip built with -DBPP_KP_PHASES (built HERE if missing)
set -e
out=${1:-gpurun_out/kp_phases.json}
if [ ! -f gpurun_in/kp_phases.so ]; then
  mkdir -p gpurun_in
  (cd bulletproofs-plus_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DBPP_KP_PHASES -o ../../gpurun_in/kp_phases.so engine.hip)
fi
cp bulletproofs-plus_amd/libbpp_hip.so /tmp/libbpp_saved.so
cp gpurun_in/kp_phases.so bulletproofs-plus_amd/libbpp_hip.so; touch bulletproofs-plus_amd/libbpp_hip.so
python3 tools/kp_phases.py > $out || true
cp /tmp/libbpp_saved.so bulletproofs-plus_amd/libbpp_hip.so; touch bulletproofs-plus_amd/libbpp_hip.so
cat $out
