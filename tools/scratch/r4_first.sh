#!/bin/bash
mkdir -p gpurun_out
bash tools/profile_round.sh r4_a_swt SW_Transformer MOD > gpurun_out/r4_a_profile.log 2>&1
tail -42 gpurun_out/r4_a_profile.log
cp gpurun_out/r4_a_swt_reference.json profiles/r4_reference_SW_Transformer_MOD.json
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r4_a_bench.json 2> gpurun_out/r4_a_bench.err
FOCAL_BENCH_TRACE_MODE=events python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4_a_bench_events.json 2>> gpurun_out/r4_a_bench.err
bash tools/profile_round.sh r4_a_deepsense DeepSense MOD > gpurun_out/r4_a_profile_ds.log 2>&1
bash tools/profile_round.sh r4_a_har4 SW_Transformer HAR4 > gpurun_out/r4_a_profile_har4.log 2>&1
tail -30 gpurun_out/r4_a_profile_ds.log
