// Parameter blocks of the fused Swin MLP kernels (mlp.hip, mlp_bwd.hip).
#pragma once
#include "gemm.hpp"

constexpr int MLP_C = 64, MLP_H = 256;

struct MlpFwdParams {
  int M;
  const bf16_t* a;      // [M][C]   LayerNorm output (norm2)
  const float* resid;   // [M][C]   x_mid
  const bf16_t* w1;     // [H][C]
  const float* b1;      // [H]
  const bf16_t* w2;     // [C][H]
  const float* b2;      // [C]
  float* y;             // [M][C]   x_out
  MaskParams drop_h;    // Mlp.drop after the activation (pair hash over [M][H])
  MaskParams drop_o;    // Mlp.drop after fc2 x DropPath (over [M][C])
  const float* ln_gamma; const float* ln_beta; bf16_t* y_ln; float* ln_stats; float ln_eps;
};

struct MlpBwdParams {
  int M;
  const bf16_t* gm;     // [M][C]   dL/dx_out x (dropout x drop-path mask of the branch), operand dtype
  const bf16_t* a;      // [M][C]   LayerNorm output saved by the forward
  const bf16_t* w1; const float* b1; const bf16_t* w2;
  bf16_t* da;           // [M][C]   dL/da (LN_BWD = false)
  float* dw1; float* db1; float* dw2; float* db2;   // fp32, accumulated (+=)
  MaskParams drop_h;
  // LN_BWD: the norm2 backward fused behind dL/da -- completes the residual-stream gradient g (+= in place) and emits
  // gm_next = bf16(g x next_mask) for the attention branch
  const float* x; const float* stats; const float* ln_gamma; float* g; bf16_t* gm_next; float* dgamma; float* dbeta;
  MaskParams next_mask;
};

int mlp_check_desc(const focal_mlp_desc* d, const char* who);
