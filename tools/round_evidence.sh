#!/bin/bash
# round evidence: rocprofv3 + PMC + launch trace of the three workloads, then the driver-style bench lines
tag=${1:-r5_z}
mkdir -p gpurun_out
bash tools/profile_round.sh ${tag}_swt SW_Transformer MOD > gpurun_out/${tag}_profile_swt.log 2>&1
cp gpurun_out/${tag}_swt_reference.json profiles/r5_reference_SW_Transformer_MOD.json
bash tools/profile_round.sh ${tag}_deepsense DeepSense MOD > gpurun_out/${tag}_profile_ds.log 2>&1
cp gpurun_out/${tag}_deepsense_reference.json profiles/r5_reference_DeepSense_MOD.json
bash tools/profile_round.sh ${tag}_har4 SW_Transformer HAR4 > gpurun_out/${tag}_profile_har4.log 2>&1
cp gpurun_out/${tag}_har4_reference.json profiles/r5_reference_SW_Transformer_HAR4.json
python3 bench.py > gpurun_out/${tag}_bench_swt.json 2> gpurun_out/${tag}_bench.err
python3 bench.py --model DeepSense --no-secondary > gpurun_out/${tag}_bench_deepsense.json 2>> gpurun_out/${tag}_bench.err
python3 bench.py --dataset HAR4 --no-secondary > gpurun_out/${tag}_bench_har4.json 2>> gpurun_out/${tag}_bench.err
python3 bench.py --views random --no-cpu-baseline > gpurun_out/${tag}_bench_swt_views_random.json 2>> gpurun_out/${tag}_bench.err
tail -25 gpurun_out/${tag}_profile_swt.log
