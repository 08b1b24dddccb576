// Zero-pad + PatchEmbed (Conv2d with kernel = stride = [1, pw]) + LayerNorm, fp32 end to end: this is the only
// layer that sees the raw spectrum (dynamic range ~ sqrt(n) * input), so it is kept out of bf16.  HBM-bound: the
// window is read exactly once, tokens are written once.  64 tokens per workgroup: patches and the transposed
// filter bank are staged in LDS, each thread produces 16 channels of one token, LayerNorm over the 4 threads of a
// token by wave shuffles.
#include <stdlib.h>
#include "common.hpp"
#include "patch_stage.hpp"

#define EMB_TOK 64

// Optional second LayerNorm over the 64 embedded channels (block 0's norm1, which reads exactly these tokens next): y_ln
// (fp32 or bf16) and stats {mean, rstd} come out of the same kernel -- one launch and one read of the token tensor less.
__device__ __forceinline__ void emb_store4(float* p, const float* f) { *reinterpret_cast<float4*>(p) = make_float4(f[0], f[1], f[2], f[3]); }
__device__ __forceinline__ void emb_store4(bf16_t* p, const float* f) {
  bf16x4 t;
  t[0] = (bf16_t)f[0]; t[1] = (bf16_t)f[1]; t[2] = (bf16_t)f[2]; t[3] = (bf16_t)f[3];
  *reinterpret_cast<bf16x4*>(p) = t;
}
struct EmbedLn2 { const float* gamma; const float* beta; void* y; float* stats; float eps; int bf16; };

template <int C0>
__global__ __launch_bounds__(256) void patch_embed_ln_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* __restrict__ tokens,
                                                             focal_embed_desc d, int K, int total_tokens, EmbedLn2 l2) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int KP = K + 1;
  float* wt = smem;                 // [K][C0]   transposed filters
  float* patch = smem + K * C0;     // [EMB_TOK][KP]
  const int tid = threadIdx.x;
  for (int i = tid; i < K * C0; i += 256) {
    const int k = i / C0, n = i % C0;
    wt[i] = w[n * K + k];  // conv weight [C0][cin][1][pw] flattened as [C0][K], K index = c*pw + t
  }
  constexpr int CPT = C0 / 4;  // channels per thread
  const int tl = tid >> 2, q = tid & 3;
  for (int t0 = blockIdx.x * EMB_TOK; t0 < total_tokens; t0 += gridDim.x * EMB_TOK) {
    __syncthreads();
    for (int i = tid; i < EMB_TOK * K; i += 256) {
      const int t = i / K, k = i % K;
      const int tok = t0 + t;
      float v = 0.f;
      if (tok < total_tokens) {
        const int px = tok % d.Wp, r = tok / d.Wp, py = r % d.Hp, b = r / d.Hp;
        const int c = k / d.pw, tt = k % d.pw;
        const int col = px * d.pw + tt;
        if (py < d.I && col < d.S) v = x[(((long)b * d.cin + c) * d.I + py) * d.S + col];  // else: zero padding
      }
      patch[t * KP + k] = v;
    }
    __syncthreads();
    float acc[CPT];
#pragma unroll
    for (int n = 0; n < CPT; ++n) acc[n] = bias[q * CPT + n];
    for (int k = 0; k < K; ++k) {
      const float a = patch[tl * KP + k];
      const float* wr = wt + k * C0 + q * CPT;
#pragma unroll
      for (int n = 0; n < CPT; n += 4) {
        const float4 wv = *reinterpret_cast<const float4*>(wr + n);
        acc[n] += a * wv.x; acc[n + 1] += a * wv.y; acc[n + 2] += a * wv.z; acc[n + 3] += a * wv.w;
      }
    }
    float s = 0.f;
#pragma unroll
    for (int n = 0; n < CPT; ++n) s += acc[n];
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    const float mean = s / C0;
    float v2 = 0.f;
#pragma unroll
    for (int n = 0; n < CPT; ++n) v2 += (acc[n] - mean) * (acc[n] - mean);
    v2 += __shfl_xor(v2, 1, 64);
    v2 += __shfl_xor(v2, 2, 64);
    const float rstd = rsqrtf(v2 / C0 + d.eps);
    const int tok = t0 + tl;
    if (tok < total_tokens) {
      float* dst = tokens + (long)tok * C0 + q * CPT;
#pragma unroll
      for (int n = 0; n < CPT; n += 4) {
        float4 o;
        o.x = (acc[n] - mean) * rstd * gamma[q * CPT + n] + beta[q * CPT + n];
        o.y = (acc[n + 1] - mean) * rstd * gamma[q * CPT + n + 1] + beta[q * CPT + n + 1];
        o.z = (acc[n + 2] - mean) * rstd * gamma[q * CPT + n + 2] + beta[q * CPT + n + 2];
        o.w = (acc[n + 3] - mean) * rstd * gamma[q * CPT + n + 3] + beta[q * CPT + n + 3];
        *reinterpret_cast<float4*>(dst + n) = o;
        acc[n] = o.x; acc[n + 1] = o.y; acc[n + 2] = o.z; acc[n + 3] = o.w;
      }
    }
    if (l2.y != nullptr) {  // (uniform)
      float t = 0.f;
#pragma unroll
      for (int n = 0; n < CPT; ++n) t += acc[n];
      t += __shfl_xor(t, 1, 64);
      t += __shfl_xor(t, 2, 64);
      const float m2 = t / C0;
      float u = 0.f;
#pragma unroll
      for (int n = 0; n < CPT; ++n) u += (acc[n] - m2) * (acc[n] - m2);
      u += __shfl_xor(u, 1, 64);
      u += __shfl_xor(u, 2, 64);
      const float r2 = rsqrtf(u / C0 + l2.eps);
      if (tok < total_tokens) {
#pragma unroll
        for (int n = 0; n < CPT; n += 4) {
          const float4 g2 = *reinterpret_cast<const float4*>(l2.gamma + q * CPT + n), b2 = *reinterpret_cast<const float4*>(l2.beta + q * CPT + n);
          const float ga2[4] = {g2.x, g2.y, g2.z, g2.w}, ba2[4] = {b2.x, b2.y, b2.z, b2.w};
          float y[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) y[e] = (acc[n + e] - m2) * r2 * ga2[e] + ba2[e];
          if (l2.bf16) emb_store4(reinterpret_cast<bf16_t*>(l2.y) + (long)tok * C0 + q * CPT + n, y);
          else emb_store4(reinterpret_cast<float*>(l2.y) + (long)tok * C0 + q * CPT + n, y);
        }
        if (q == 0) *reinterpret_cast<float2*>(l2.stats + 2 * (long)tok) = make_float2(m2, r2);
      }
    }
  }
}

// ---- matrix-core form for K = cin * pw = 80, C0 = 64 (the MOD audio patch embedding): the kernel above spends most of its
// time in per-element index arithmetic (100 us against a 25 us traffic bound).  A 64-token tile of patches is staged with
// 16-byte loads (a token's pw taps are contiguous per input channel), the contraction runs on the exact-fp32 MFMA with the
// filter bank resident in registers as fragments, and LayerNorm is finished in registers: a lane holds 16 channels of one
// token, the other 48 are in the three lanes 16 apart.
typedef float ef4 __attribute__((ext_vector_type(4)));
template <int K>
__global__ __launch_bounds__(256) void patch_embed_ln_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                  const float* __restrict__ bias, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, float* __restrict__ tokens,
                                                                  focal_embed_desc d, int total_tokens, PatchGeom pg, EmbedLn2 l2) {
  constexpr int KP = K + 4, KS = K / 4;
  __shared__ __attribute__((aligned(16))) float patch[EMB_TOK * KP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lm = lane & 15, lg = lane >> 4;
  float wf[4][KS];
  load_filter_fragments<K>(w, patch, tid, wf);  // (the patch tile has the same [64][K + 4] shape)
  float4 bv[4], gv[4], be[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    bv[nt] = *reinterpret_cast<const float4*>(bias + 16 * nt + 4 * lg);
    gv[nt] = *reinterpret_cast<const float4*>(gamma + 16 * nt + 4 * lg);
    be[nt] = *reinterpret_cast<const float4*>(beta + 16 * nt + 4 * lg);
  }
  PatchPlan<K, EMB_TOK> plan;
  patch_stage_plan<K, EMB_TOK, KP>(plan, pg, tid);
  // the NEXT tile's patches are requested while this one is multiplied (20 registers: still two waves per SIMD)
  float4 stage[PatchPlan<K, EMB_TOK>::NIT];
  if ((int)(blockIdx.x * EMB_TOK) < total_tokens) patch_stage_load<K, EMB_TOK>(plan, pg, x, blockIdx.x * EMB_TOK, total_tokens, stage);
  for (int t0 = blockIdx.x * EMB_TOK; t0 < total_tokens; t0 += gridDim.x * EMB_TOK) {
    __syncthreads();
    patch_stage_store<K, EMB_TOK>(plan, patch, stage);
    __syncthreads();
    if (t0 + (int)(gridDim.x * EMB_TOK) < total_tokens) patch_stage_load<K, EMB_TOK>(plan, pg, x, t0 + gridDim.x * EMB_TOK, total_tokens, stage);
    ef4 acc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[nt] = ef4{bv[nt].x, bv[nt].y, bv[nt].z, bv[nt].w};
    const float* pr = patch + (16 * wave + lm) * KP + lg;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const float pf = pr[4 * ks];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[nt][ks], pf, acc[nt], 0, 0, 0);
    }
    float s1 = 0.f;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) s1 += acc[nt][0] + acc[nt][1] + acc[nt][2] + acc[nt][3];
    s1 += __shfl_xor(s1, 16, 64);
    s1 += __shfl_xor(s1, 32, 64);
    const float mean = s1 * (1.0f / 64.0f);
    float s2 = 0.f;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) s2 += (acc[nt][r] - mean) * (acc[nt][r] - mean);
    s2 += __shfl_xor(s2, 16, 64);
    s2 += __shfl_xor(s2, 32, 64);
    const float rstd = rsqrtf(s2 * (1.0f / 64.0f) + d.eps);
    const int tok = t0 + 16 * wave + lm;
    if (tok < total_tokens) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const float ga[4] = {gv[nt].x, gv[nt].y, gv[nt].z, gv[nt].w}, ba[4] = {be[nt].x, be[nt].y, be[nt].z, be[nt].w};
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (acc[nt][r] - mean) * rstd * ga[r] + ba[r];
        *reinterpret_cast<float4*>(tokens + (long)tok * 64 + 16 * nt + 4 * lg) = make_float4(o[0], o[1], o[2], o[3]);
      }
    }
    if (l2.y != nullptr) {  // (uniform) second LayerNorm on the values just stored
      float t = 0.f;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const float ga[4] = {gv[nt].x, gv[nt].y, gv[nt].z, gv[nt].w}, ba[4] = {be[nt].x, be[nt].y, be[nt].z, be[nt].w};
#pragma unroll
        for (int r = 0; r < 4; ++r) { acc[nt][r] = (acc[nt][r] - mean) * rstd * ga[r] + ba[r]; t += acc[nt][r]; }
      }
      t += __shfl_xor(t, 16, 64);
      t += __shfl_xor(t, 32, 64);
      const float m2 = t * (1.0f / 64.0f);
      float u = 0.f;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) u += (acc[nt][r] - m2) * (acc[nt][r] - m2);
      u += __shfl_xor(u, 16, 64);
      u += __shfl_xor(u, 32, 64);
      const float r2 = rsqrtf(u * (1.0f / 64.0f) + l2.eps);
      if (tok < total_tokens) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          // (16-byte loads: as 32 one-dword loads per tile these were the kernel's largest group of memory instructions)
          const float4 g2 = *reinterpret_cast<const float4*>(l2.gamma + 16 * nt + 4 * lg), b2 = *reinterpret_cast<const float4*>(l2.beta + 16 * nt + 4 * lg);
          const float ga2[4] = {g2.x, g2.y, g2.z, g2.w}, ba2[4] = {b2.x, b2.y, b2.z, b2.w};
          float y[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) y[r] = (acc[nt][r] - m2) * r2 * ga2[r] + ba2[r];
          if (l2.bf16) emb_store4(reinterpret_cast<bf16_t*>(l2.y) + (long)tok * 64 + 16 * nt + 4 * lg, y);
          else emb_store4(reinterpret_cast<float*>(l2.y) + (long)tok * 64 + 16 * nt + 4 * lg, y);
        }
        if (lg == 0) *reinterpret_cast<float2*>(l2.stats + 2 * (long)tok) = make_float2(m2, r2);
      }
    }
  }
}

static int embed_launch(const focal_embed_desc* d, const float* x, const float* w, const float* b, const float* gamma, const float* beta,
                        float* tokens, EmbedLn2 l2, void* stream) {
  FOCAL_CHECK_ARG(d && x && w && b && gamma && beta && tokens, "pad_patch_embed_ln: null argument");
  FOCAL_CHECK_ARG(d->C0 == 64 || d->C0 == 128, "pad_patch_embed_ln: embed dim %d not in {64, 128}", d->C0);
  FOCAL_CHECK_ARG(d->Hp >= d->I && d->Wp * d->pw >= d->S && d->pw > 0, "pad_patch_embed_ln: padded grid smaller than the input");
  const int K = d->cin * d->pw;
  const size_t sm = ((size_t)K * d->C0 + (size_t)EMB_TOK * (K + 1)) * sizeof(float);
  FOCAL_CHECK_ARG(sm <= 64 * 1024, "pad_patch_embed_ln: patch of %d values does not fit in LDS", K);
  const int total = d->B * d->Hp * d->Wp;
  hipStream_t st = (hipStream_t)stream;
  if (K == 80 && d->C0 == 64 && d->pw % 4 == 0 && d->S % 4 == 0) {
    int mb = ceil_div(total, EMB_TOK);
    // two resident workgroups per CU (208-230 VGPRs); more only repeats the filter-bank load (measured 61 / 74 / 102 / 178 us at
    // 512 / 1024 / 2048 / 4608 workgroups before the bank went through LDS)
    const int mb_cap = 512;
    if (mb > mb_cap) mb = mb_cap;
    const PatchGeom pg = make_patch_geom(d->Wp, d->Hp, d->I, d->S, d->pw, d->cin);
    FOCAL_LAUNCH((patch_embed_ln_mfma_kernel<80>), dim3(mb), dim3(256), 0, st, x, w, b, gamma, beta, tokens, *d, total, pg, l2);
    FOCAL_LAUNCH_CHECK();
    return FOCAL_OK;
  }
  int blocks = ceil_div(total, EMB_TOK);
  if (blocks > 2048) blocks = 2048;
  if (d->C0 == 64) FOCAL_LAUNCH((patch_embed_ln_kernel<64>), dim3(blocks), dim3(256), sm, st, x, w, b, gamma, beta, tokens, *d, K, total, l2);
  else FOCAL_LAUNCH((patch_embed_ln_kernel<128>), dim3(blocks), dim3(256), sm, st, x, w, b, gamma, beta, tokens, *d, K, total, l2);
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}

extern "C" int focal_pad_patch_embed_ln_fwd(const focal_embed_desc* d, const float* x, const float* w, const float* b,
                                            const float* gamma, const float* beta, float* tokens, void* stream) {
  const EmbedLn2 none = {nullptr, nullptr, nullptr, nullptr, 0.f, 0};
  return embed_launch(d, x, w, b, gamma, beta, tokens, none, stream);
}

extern "C" int focal_pad_patch_embed_ln2_fwd(const focal_embed_desc* d, const float* x, const float* w, const float* b,
                                             const float* gamma, const float* beta, float* tokens, const float* gamma2,
                                             const float* beta2, float eps2, int ln_dtype, void* y_ln, float* stats, void* stream) {
  FOCAL_CHECK_ARG(gamma2 && beta2 && y_ln && stats, "pad_patch_embed_ln2: null argument");
  FOCAL_CHECK_ARG(((uintptr_t)gamma2 | (uintptr_t)beta2) % 16 == 0, "pad_patch_embed_ln2: gamma2 / beta2 are read as float4: 16-byte aligned pointers");
  FOCAL_CHECK_ARG(ln_dtype == FOCAL_F32 || ln_dtype == FOCAL_BF16, "pad_patch_embed_ln2: bad dtype %d", ln_dtype);
  const EmbedLn2 l2 = {gamma2, beta2, y_ln, stats, eps2, ln_dtype == FOCAL_BF16};
  return embed_launch(d, x, w, b, gamma, beta, tokens, l2, stream);
}
