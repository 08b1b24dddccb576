#!/bin/bash
tag=r6_z
bash tools/profile_round.sh ${tag}_swt SW_Transformer MOD > gpurun_out/${tag}_profile_swt.log 2>&1
cp gpurun_out/${tag}_swt_reference.json profiles/r6_reference_SW_Transformer_MOD.json
bash tools/profile_round.sh ${tag}_deepsense DeepSense MOD > gpurun_out/${tag}_profile_ds.log 2>&1
cp gpurun_out/${tag}_deepsense_reference.json profiles/r6_reference_DeepSense_MOD.json
bash tools/profile_round.sh ${tag}_har4 SW_Transformer HAR4 > gpurun_out/${tag}_profile_har4.log 2>&1
cp gpurun_out/${tag}_har4_reference.json profiles/r6_reference_SW_Transformer_HAR4.json
cp profiles/r6_reference_*.json gpurun_out/
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${tag}_bench_driver_line.json 2> gpurun_out/${tag}_bench.err
python3 bench.py > gpurun_out/${tag}_bench_swt.json 2>> gpurun_out/${tag}_bench.err
python3 bench.py --model DeepSense --no-secondary > gpurun_out/${tag}_bench_deepsense.json 2>> gpurun_out/${tag}_bench.err
python3 bench.py --dataset HAR4 --no-secondary > gpurun_out/${tag}_bench_har4.json 2>> gpurun_out/${tag}_bench.err
python3 bench.py --views random --no-cpu-baseline > gpurun_out/${tag}_bench_swt_views_random.json 2>> gpurun_out/${tag}_bench.err
tail -4 gpurun_out/${tag}_swt_step_traffic.txt
python3 - <<PY
import json
for n in ("driver_line", "swt", "deepsense", "har4", "swt_views_random"):
    d = json.loads(open("gpurun_out/${tag}_bench_%s.json" % n).read().strip().splitlines()[-1])
    r = d.get("roofline") or {}
    print(n, d["value"], d["ms_per_step"], "warmup_effective", d.get("warmup_effective"), "roofline", r.get("kernel"), r.get("frac"), "suspect", r.get("suspect"), (r.get("step") or {}).get("hbm_bytes_per_step_pmc"))
PY
