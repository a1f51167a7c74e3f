#!/bin/bash
# On the GPU box: every GPU test file in a pytest process of its own (the suite as a whole runs in one process and one order;
# this shows whether a file depends on what ran before it).  Stops at the first failing file.
set -e -o pipefail
O=gpurun_out/per_file_tests.log
: > $O
for f in tests/test_gpu_*.py tests/test_ref_golden.py; do
  echo "== $f" >> $O
  timeout -k 10 600 python3 -m pytest $f -x -q -m gpu >> $O 2>&1 || { rc=$?; [ $rc -eq 5 ] || { echo "FAILED $f rc=$rc" >> $O; tail -30 $O; exit 1; }; }
  tail -1 $O
done
echo "all files passed standalone"
