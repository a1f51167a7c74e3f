#!/bin/bash
# kernel timeline of the last bpp_prove_batch call of configs[4]'s leg: gpu_prover_trace.sh <out.txt>
set -e -o pipefail
export TMPDIR=/tmp
out=${1:-gpurun_out/prover_launches.txt}
rm -rf gpurun_out/prov_trace
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prov_trace -- python3 bench.py --only prover --no-extra --no-cpu-baseline --no-traffic > /dev/null 2> gpurun_out/prov_trace.err
python3 tools/prover_trace.py gpurun_out/prov_trace --m 4 --t 3 --proofs 1024 --windows 23 > $out
rm -rf gpurun_out/prov_trace
