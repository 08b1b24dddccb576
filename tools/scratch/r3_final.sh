#!/bin/bash
# end-of-round evidence run (GPU box): PMC traffic first (bench.py's roofline.traffic reads it), the bench lines, rocprofv3 kernel
# stats + instance tables, timelines, gradient-error tables, the full GPU test suite.   bash tools/scratch/r3_final.sh [tag]
cd "$(dirname "$0")/../.."
tag=${1:-r3_z}
mkdir -p gpurun_out
for w in "swt:SW_Transformer:MOD:" "deepsense:DeepSense:MOD:--model DeepSense" "har4:SW_Transformer:HAR4:--dataset HAR4"; do
  IFS=: read name model dataset args <<< "$w"
  bash tools/pmc_step_traffic.sh ${tag}_${name} $args > gpurun_out/${tag}_${name}_pmc.log 2>&1
  cp gpurun_out/${tag}_${name}_pmc_groups.json profiles/r3_pmc_groups_${model}_${dataset}.json
  cp profiles/r3_pmc_groups_${model}_${dataset}.json gpurun_out/
  tail -3 gpurun_out/${tag}_${name}_step_traffic.txt
done
python bench.py 2>gpurun_out/${tag}_bench_swt.err | tail -1 > gpurun_out/${tag}_bench_swt.json; cut -c1-220 gpurun_out/${tag}_bench_swt.json; echo
python bench.py --model DeepSense 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_deepsense.json; cut -c1-220 gpurun_out/${tag}_bench_deepsense.json; echo
python bench.py --dataset HAR4 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_har4.json; cut -c1-220 gpurun_out/${tag}_bench_har4.json; echo
python bench.py --views random --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_swt_views_random.json; cut -c1-220 gpurun_out/${tag}_bench_swt_views_random.json; echo
bash tools/profile_step.sh ${tag}_swt > gpurun_out/${tag}_swt_profile.log 2>&1; tail -3 gpurun_out/${tag}_swt_profile.log
bash tools/profile_step.sh ${tag}_deepsense --model DeepSense > gpurun_out/${tag}_deepsense_profile.log 2>&1
bash tools/profile_step.sh ${tag}_har4 --dataset HAR4 > gpurun_out/${tag}_har4_profile.log 2>&1
python tools/timeline.py --no-cpu-baseline --no-roofline > /dev/null 2>&1; cp gpurun_out/timeline_SW_Transformer_MOD.txt gpurun_out/${tag}_timeline_swt.txt
python tools/timeline.py --no-cpu-baseline --no-roofline --model DeepSense > /dev/null 2>&1; cp gpurun_out/timeline_DeepSense_MOD.txt gpurun_out/${tag}_timeline_deepsense.txt
python tools/timeline.py --no-cpu-baseline --no-roofline --dataset HAR4 > /dev/null 2>&1; cp gpurun_out/timeline_SW_Transformer_HAR4.txt gpurun_out/${tag}_timeline_har4.txt
python tools/grad_error_table.py SW_Transformer 256 > gpurun_out/${tag}_grad_error_swt.txt 2>&1
python tools/grad_error_table.py DeepSense 256 > gpurun_out/${tag}_grad_error_deepsense.txt 2>&1
rm -f gpurun_out/observed_parity.json
timeout 2400 python -m pytest tests -m gpu -q -rf > gpurun_out/${tag}_pytest_gpu.log 2>&1; tail -3 gpurun_out/${tag}_pytest_gpu.log
