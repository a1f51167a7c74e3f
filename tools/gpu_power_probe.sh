#!/bin/bash
# board power / clocks while the headline leg runs (is the chip power-limited under this load?): one sample per second
out=${1:-gpurun_out/power_probe.txt}
: > $out
rocm-smi --showmaxpower >> $out 2>&1
python bench.py --steps 4000 --warmup 5 --no-extra --no-cpu-baseline --no-traffic > gpurun_out/power_probe_bench.json 2>/dev/null &
BP=$!
i=0
while kill -0 $BP 2>/dev/null; do
  i=$((i+1))
  p=$(rocm-smi --showpower 2>/dev/null | grep -i "Power (W)" | sed 's/.*: //')
  c=$(rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | sed 's/.*(//; s/).*//')
  t=$(rocm-smi --showtemp 2>/dev/null | grep -i "junction" | sed 's/.*: //')
  echo "t=$i power_w=$p sclk=$c junction_c=$t" >> $out
  sleep 1
done
tail -c 300 gpurun_out/power_probe_bench.json >> $out
