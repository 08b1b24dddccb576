#!/bin/bash
# which tail products get fp32 operands: the B = 8 fixture's embedding / loss-term errors per variant (tests record them in gpurun_out/observed_parity.json)
for v in none mod_in projector both; do
  rm -f gpurun_out/observed_parity.json
  FOCAL_TAIL_FP32=$v python -m pytest tests/test_swt_parity_gpu.py -q -k "eval_embeddings or train_step_loss" 2>&1 | tail -1
  python3 - <<PY
import json
d = json.load(open("gpurun_out/observed_parity.json"))
keys = ["swt.eval.emb.audio.bf16.max_err_over_max_ref", "swt.eval.emb.seismic.bf16.max_err_over_max_ref", "swt.train.emb.audio.bf16.max_err_over_max_ref", "swt.train.emb.seismic.bf16.max_err_over_max_ref",
        "swt.train.loss.rank.bf16.abs_err_over_max1", "swt.train.loss.shared.bf16.abs_err_over_max1", "swt.train.loss.private.bf16.abs_err_over_max1", "swt.train.loss.orth.bf16.abs_err_over_max1"]
print("$v", " ".join(f"{d.get(k, float('nan')) * 1e2:.3f}e-2" for k in keys))
PY
done
