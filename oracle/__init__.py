"""CPU oracle for the FOCAL pretraining step -- TEST INFRASTRUCTURE ONLY.

This package is a from-scratch, functional (state-dict in, tensors out) fp32/fp64 restatement of the
reference algorithm (tomoyoshki/focal, `src/train_utils/pretrain.py:62-74` and everything it calls).
It exists to *check* the HIP path; it is never the thing shipped or measured:

  * only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it;
  * nothing under `focal_amd/` imports it, and `focal_amd` raises when its HIP library is missing
    instead of falling back to anything here.

Parity status: the reference has no tests, golden vectors or seeds of its own, so the
oracle is pinned against outputs of the reference itself, imported in the build container by
`tests/golden/gen_golden.py` (timm / tsai are absent there and are replaced by minimal stand-ins for
`trunc_normal_`, `DropPath`, the two schedulers and two augmenters that the parity boundary never reaches;
see that script's header).  The committed fixtures under `tests/golden/` are those outputs.
Every function cites the reference file:line it restates.
"""
