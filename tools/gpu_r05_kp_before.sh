set -e -o pipefail
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
python3 tools/bench_prover_leg.py > $O/r05_prover_before.json 2> $O/r05_prover_before.err
echo "prover leg: $(cat $O/r05_prover_before.json)"
bash tools/gpu_kp_phases.sh $O/r05_kp_phases_before.json > /dev/null 2> $O/r05_kp_phases.err
KP_M=1 KP_T=1 bash tools/gpu_kp_phases.sh $O/r05_kp_phases_before_m1.json > /dev/null 2>> $O/r05_kp_phases.err
echo phases done
rm -rf $O/prof_kp
PROVER_ITERS=3 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/prof_kp/f1 -- python3 tools/bench_prover_leg.py > $O/r05_pmc_f1.json 2> $O/r05_pmc_f1.err
python3 tools/pmc_per_dispatch.py $O/prof_kp/f1 kp_round > $O/r05_kp_round_per_dispatch_before.txt
BPP_PROVE_FUSED=0 PROVER_ITERS=3 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/prof_kp/f0 -- python3 tools/bench_prover_leg.py > $O/r05_pmc_f0.json 2> $O/r05_pmc_f0.err
for k in kp_lane kp_wave k_compress_ge; do python3 tools/pmc_per_dispatch.py $O/prof_kp/f0 $k > $O/r05_${k}_per_dispatch_before.txt; done
python3 tools/pmc_summary.py counters $O/prof_kp/f0 $O/r05_pmc_unfused_before.csv
rm -rf $O/prof_kp
echo done
