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
  uint32_t* mask_bits;  // [M][8] (DROP): the keep bits of drop_h -- bit 4 (T % 8) + e of word 2 g + T / 8 = hidden unit 16 T + 4 g + e
  // PROJ (round 6): the attention branch's tail in front of the MLP, in the same kernel -- x_mid = x + drop_p(o Wp^T + bp), a2 = norm2(x_mid):
  // `a` and `resid` are then OUTPUTS (a2, x_mid: the backward pass reads them), st2 the LayerNorm's {mean, rstd}
  const bf16_t* o;      // [M][C]   attention output
  const float* x;       // [M][C]   the block's input (residual stream)
  const bf16_t* wp;     // [C][C]
  const float* bp;      // [C]
  MaskParams drop_p;    // proj dropout x DropPath (over [M][C])
  const float* g2; const float* bt2; float* st2;   // norm2
#if defined(__HIPCC__)
  __device__ __forceinline__ float* resid_out() const { return const_cast<float*>(resid); }
  __device__ __forceinline__ bf16_t* a_out() const { return const_cast<bf16_t*>(a); }
#endif
};

struct MlpBwdParams {
  int M;
  const bf16_t* gm;     // [M][C]   dL/dx_out x (dropout x drop-path mask of the branch), operand dtype
  const bf16_t* a;      // [M][C]   LayerNorm output saved by the forward
  const bf16_t* w1; const float* b1; const bf16_t* w2;
  bf16_t* da;           // [M][C]   dL/da (LN_BWD = false)
  float* dw1; float* db1; float* dw2; float* db2;   // fp32, accumulated (+=)
  float* partials;      // non-null: every workgroup stores its [dW2 | dW1] image here (32 768 floats each) for mlp_bwd_reduce_kernel
  MaskParams drop_h;
  const uint32_t* mask_bits;  // [M][8]: the forward kernel's keep bits of drop_h (required when p_elem > 0)
  // LN_BWD: the norm2 backward fused behind dL/da -- completes the residual-stream gradient g (+= in place) and emits
  // gm_next = bf16(g x next_mask) for the attention branch
  const float* x; const float* stats; const float* ln_gamma; float* g; bf16_t* gm_next; float* dgamma; float* dbeta;
  MaskParams next_mask;
};

int mlp_check_desc(const focal_mlp_desc* d, const char* who);
bool focal_mlp_proj_width_enabled(int C_);  // FOCAL_MLP_PROJ (mlp.hip)

#if defined(__HIPCC__)
// ---------------------------------------------------------------------------------------------- element math of the fused kernels
// Both kernels are VALU-bound (the matrix cores sit idle ~85 % of the time: ~10 vector instructions per hidden element is the floor
// for bias + erf-GELU + dropout + bf16 packing against 1/16 MFMA per element), so the element math is written for instruction
// count: erf by Abramowitz-Stegun 7.1.26 (|err| < 1.5e-7) in the one-sided form
//   gelu(x) = max(x, 0) - |x| hp(|x|),   hp(a) = 0.5 poly(t) exp(-a^2 / 2),  t = 1 / (1 + 0.3275911 a / sqrt 2)
// (no copysign, no 0.5 + 0.5 erf), exp as a bare v_exp_f32 (base 2, the log2 e folded into the argument), everything else packed
// fp32 (two elements per instruction).
// Packed fp32 (v_pk_fma_f32 ...: two elements per instruction) vs scalar element math: measured equal on these kernels (same-box A/B of
// two builds, profiles/r2_mlp_ab.txt: forward 77.7 vs 79.2 us, backward 182 vs 186 us at the stage-0 audio shape) -- a packed
// instruction occupies the vector pipe about as long as the two scalar ones it replaces.  MLP_PK = 0 selects the scalar form.
#ifndef MLP_PK
#define MLP_PK 1
#endif
#if MLP_PK
typedef gelu_f2 mlp_v;
__device__ __forceinline__ mlp_v mlp_abs(mlp_v x) { return mlp_v{fabsf(x.x), fabsf(x.y)}; }
__device__ __forceinline__ mlp_v mlp_rcp(mlp_v x) { return mlp_v{__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)}; }
__device__ __forceinline__ mlp_v mlp_exp2(mlp_v x) { return mlp_v{__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)}; }
__device__ __forceinline__ mlp_v mlp_relu(mlp_v x) { return mlp_v{fmaxf(x.x, 0.f), fmaxf(x.y, 0.f)}; }
__device__ __forceinline__ mlp_v mlp_copysign(mlp_v m, mlp_v s) { return mlp_v{copysignf(m.x, s.x), copysignf(m.y, s.y)}; }
#else
typedef float mlp_v;
__device__ __forceinline__ mlp_v mlp_abs(mlp_v x) { return fabsf(x); }
__device__ __forceinline__ mlp_v mlp_rcp(mlp_v x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ mlp_v mlp_exp2(mlp_v x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ mlp_v mlp_relu(mlp_v x) { return fmaxf(x, 0.f); }
__device__ __forceinline__ mlp_v mlp_copysign(mlp_v m, mlp_v s) { return copysignf(m, s); }
#endif
struct MlpGelu {
  mlp_v e, poly, ax;  // exp(-x^2/2), 0.5 poly(t), |x|
};
__device__ __forceinline__ MlpGelu mlp_gelu_core(mlp_v x) {
  MlpGelu r;
  r.ax = mlp_abs(x);
  const mlp_v d = r.ax * (0.3275911f * 0.70710678118654752f) + 1.0f;
  const mlp_v t = mlp_rcp(d);
  const mlp_v a2 = (x * -0.72134752044448170f) * x;  // -x^2 / 2 * log2(e)
  r.e = mlp_exp2(a2);
  r.poly = t * (t * (t * (t * (t * (0.5f * 1.061405429f) - (0.5f * 1.453152027f)) + (0.5f * 1.421413741f)) - (0.5f * 0.284496736f)) + (0.5f * 0.254829592f));
  return r;
}
__device__ __forceinline__ mlp_v mlp_gelu_fwd1(mlp_v x) {
  const MlpGelu c = mlp_gelu_core(x);
  return mlp_relu(x) - c.poly * c.e * c.ax;
}
// value and derivative: cdf = 0.5 + copysign(0.5 - hp, x), h = x cdf, h' = cdf + x pdf
__device__ __forceinline__ void mlp_gelu_bwd1(mlp_v x, mlp_v& h, mlp_v& hg) {
  const MlpGelu c = mlp_gelu_core(x);
  const mlp_v om = c.poly * c.e * -1.0f + 0.5f;
  const mlp_v cdf = mlp_copysign(om, x) + 0.5f;
  h = x * cdf;
  hg = x * (c.e * 0.39894228040143268f) + cdf;
}
__device__ __forceinline__ gelu_f2 mlp_mul2(gelu_f2 a, gelu_f2 b) {
#if MLP_PK
  return a * b;
#else
  return gelu_f2{a.x * b.x, a.y * b.y};
#endif
}
// two neighbouring elements (the granularity of the dropout stream and of the bf16 packing)
__device__ __forceinline__ gelu_f2 mlp_gelu_fwd(gelu_f2 x) {
#if MLP_PK
  return mlp_gelu_fwd1(x);
#else
  return gelu_f2{mlp_gelu_fwd1(x.x), mlp_gelu_fwd1(x.y)};
#endif
}
__device__ __forceinline__ void mlp_gelu_bwd(gelu_f2 x, gelu_f2& h, gelu_f2& hg) {
#if MLP_PK
  mlp_gelu_bwd1(x, h, hg);
#else
  float h0, g0, h1, g1;
  mlp_gelu_bwd1(x.x, h0, g0);
  mlp_gelu_bwd1(x.y, h1, g1);
  h = gelu_f2{h0, h1};
  hg = gelu_f2{g0, g1};
#endif
}

// Dropout on the hidden activation inside the fused kernels: a lane owns, for its token row m and its lane group g, the hidden units
// {16 T + 4 g .. + 3 : T = 0..15}, i.e. 32 element pairs that BOTH kernels visit in ascending T.  One full hash per (row, group)
// seeds a xorshift32 stream; every pair takes one step (6 full-rate instructions against a 2-multiply hash per pair) and uses the two
// 16-bit halves of the state against a 16-bit threshold, exactly like the GEMM epilogue's pair hash.  The mask is a pure function of
// (seed word, stream id, row, hidden unit): the backward kernel regenerates it.
struct MlpDropStream {
  uint32_t s, t16;
  float scale;
  __device__ __forceinline__ void init(const MaskParams& mp) {
    const DropCtx c = make_drop(mp.seed, mp.stream_elem, mp.p_elem);
    s = c.key;  // re-seeded per row by start()
    t16 = c.thresh >> 8;
    scale = c.scale;
  }
  __device__ __forceinline__ uint32_t start(uint32_t key, int row, int group) const {
    // (the bijection, not focal_hash24: a launch starts ~6e5 streams, and a hash with 2^24 distinct outputs per key would give ~1e4 pairs
    // of (row, group)s identical 64-unit masks; the seed is drawn once per row, outside the issue-bound loops -- ADVICE r4)
    const uint32_t v = focal_mix32((((uint32_t)row << 2) | (uint32_t)group) ^ key);
    return v ? v : 0x9E3779B9u;
  }
  __device__ __forceinline__ gelu_f2 next(uint32_t& st) const {
    st ^= st << 13;
    st ^= st >> 17;
    st ^= st << 5;
    return gelu_f2{(st & 0xffffu) < t16 ? 0.0f : scale, (st >> 16) < t16 ? 0.0f : scale};
  }
  // the two keep decisions of a pair as bits 0 / 1 (the forward kernel saves them for the backward kernel: mlp_bwd.hip)
  static __device__ __forceinline__ uint32_t keep_bits(gelu_f2 mult) {
    return (mult.x != 0.0f ? 1u : 0u) | (mult.y != 0.0f ? 2u : 0u);
  }
};
#endif
