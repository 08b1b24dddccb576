#!/bin/bash
# end-of-round evidence run (GPU box): full GPU test suite, the three bench lines, rocprofv3 kernel stats + instance table, PMC traffic
cd "$(dirname "$0")/../.."
tag=${1:-r2_z}
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/${tag}_pytest_gpu.log 2>&1; tail -3 gpurun_out/${tag}_pytest_gpu.log
python bench.py 2>&1 | tail -1 > gpurun_out/${tag}_bench_swt.json; cut -c1-200 gpurun_out/${tag}_bench_swt.json; echo
python bench.py --model DeepSense 2>&1 | tail -1 > gpurun_out/${tag}_bench_deepsense.json; cut -c1-200 gpurun_out/${tag}_bench_deepsense.json; echo
python bench.py --dataset HAR4 2>&1 | tail -1 > gpurun_out/${tag}_bench_har4.json; cut -c1-200 gpurun_out/${tag}_bench_har4.json; echo
bash tools/profile_step.sh ${tag}_swt > gpurun_out/${tag}_profile.log 2>&1; tail -5 gpurun_out/${tag}_profile.log
bash tools/pmc_step_traffic.sh ${tag} > gpurun_out/${tag}_pmc.log 2>&1; tail -4 gpurun_out/${tag}_pmc.log
