for cfg in "16 16 80 20" "16 12 80 20" "16 8 80 20" "32 8 40 10" "32 6 40 10" "64 4 20 5" "16 16 1024 64" "32 8 512 32" "64 4 256 32"; do
  set -- $cfg
  python3 bench.py --no-extra --no-cpu-baseline --no-traffic --batches-per-step $1 --concurrency $2 --steps $3 --warmup $4 > gpurun_out/x.json 2>gpurun_out/x.err
  python3 - <<PY
import json
d = json.loads(open("gpurun_out/x.json").read().strip().split("\n")[-1])
print("batches/step $1 conc $2 steps $3: %.2f M proofs/s, %.3f ms/step, latency %.2f ms" % (d["value"] / 1e6, d["ms_per_step"], d["step_latency_ms"]))
PY
done
