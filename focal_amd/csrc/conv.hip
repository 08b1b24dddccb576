// DeepSense ConvBlock convolutions (models/ConvModules.py:115-216) in channel-last token layout [B*I*S, C]:
//   * in-conv   Conv2d(cin -> C, [1,k], stride [1,s]) straight from the reference's NCHW fp32 spectrum: fp32 FMA
//               kernel (HBM-bound: the audio layer reads the 128 KB window once; AI ~ 32 flop/B), + its weight gradient;
//   * inter-convs / flatten+1x1: MFMA GEMMs whose A (or, for dW, B) operand is a sliding window over tokens
//               (PRO_CONV in gemm.hpp) -- a [1,k] "same" conv over channel-last rows is a GEMM with lda = C_in < K;
//   * the small weight re-layouts between the reference's [Cout][Cin][1][k] storage and the GEMM operand orders.
#include <stdlib.h>
#include "gemm.hpp"
#include "patch_stage.hpp"
#include "conv_ring.hpp"

#define CIN_TOK 64

// ------------------------------------------------------------------------------------------------ in-conv forward
template <int C0>
__global__ __launch_bounds__(256) void conv_in_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ z,
                                                          focal_conv_in_desc d, int K, int total) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int KP = K + 1;
  float* wt = smem;              // [K][C0]
  float* patch = smem + K * C0;  // [CIN_TOK][KP]
  const int tid = threadIdx.x;
  for (int i = tid; i < K * C0; i += 256) wt[i] = w[(i % C0) * K + i / C0];  // w: [C0][cin][1][k] = [C0][K]
  constexpr int CPT = C0 / 4;
  const int tl = tid >> 2, q = tid & 3;
  for (int t0 = blockIdx.x * CIN_TOK; t0 < total; t0 += gridDim.x * CIN_TOK) {
    __syncthreads();
    for (int i = tid; i < CIN_TOK * K; i += 256) {
      const int t = i / K, kk = i % K, tok = t0 + t;
      float v = 0.f;
      if (tok < total) {
        const int so = tok % d.S_out, r = tok / d.S_out, ii = r % d.I, b = r / d.I;
        const int c = kk / d.k, tt = kk % d.k;
        const int col = so * d.stride + tt - d.pad_left;
        if (col >= 0 && col < d.S_in) v = x[(((long)b * d.cin + c) * d.I + ii) * d.S_in + col];
      }
      patch[t * KP + kk] = v;
    }
    __syncthreads();
    float acc[CPT];
#pragma unroll
    for (int n = 0; n < CPT; ++n) acc[n] = bias[q * CPT + n];
    for (int kk = 0; kk < K; ++kk) {
      const float a = patch[tl * KP + kk];
      const float* wr = wt + kk * C0 + q * CPT;
#pragma unroll
      for (int n = 0; n < CPT; n += 4) {
        const float4 wv = *reinterpret_cast<const float4*>(wr + n);
        acc[n] += a * wv.x; acc[n + 1] += a * wv.y; acc[n + 2] += a * wv.z; acc[n + 3] += a * wv.w;
      }
    }
    const int tok = t0 + tl;
    if (tok < total) {
      float* dst = z + (long)tok * C0 + q * CPT;
#pragma unroll
      for (int n = 0; n < CPT; n += 4) *reinterpret_cast<float4*>(dst + n) = make_float4(acc[n], acc[n + 1], acc[n + 2], acc[n + 3]);
    }
  }
}

// dw[n][kk] += sum_tok dz[tok][n] * patch[tok][kk];  dbias[n] += sum_tok dz[tok][n].
// Thread kk owns column kk of dw for all C0 rows (accumulators in registers) across several 64-token chunks, so the
// final fp32 atomics of a wave hit CONSECUTIVE addresses of one dw row (the only shape atomics run fast in).
template <typename TZ, int C0>
__global__ __launch_bounds__(256) void conv_in_bwd_weight_kernel(const float* __restrict__ x, const TZ* __restrict__ dz,
                                                                 float* __restrict__ dw, float* __restrict__ dbias,
                                                                 focal_conv_in_desc d, int K, int total, int chunks_per_wg) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int KP = K + 1, GP = C0 + 4;
  float* patch = smem;              // [CIN_TOK][KP]
  float* g = smem + CIN_TOK * KP;   // [CIN_TOK][GP]
  const int tid = threadIdx.x;
  float acc[C0];
#pragma unroll
  for (int n = 0; n < C0; ++n) acc[n] = 0.f;
  float bsum = 0.f;  // threads K..K+C0-1 double as bias-gradient accumulators when they exist
  const int bn = tid - (256 - C0);  // last C0 threads: bias column
  for (int ch = 0; ch < chunks_per_wg; ++ch) {
    const int t0 = (blockIdx.x * chunks_per_wg + ch) * CIN_TOK;
    if (t0 >= total) break;
    __syncthreads();
    for (int i = tid; i < CIN_TOK * K; i += 256) {
      const int t = i / K, kk = i - t * K, tok = t0 + t;
      float v = 0.f;
      if (tok < total) {
        const int so = tok % d.S_out, r = tok / d.S_out, ii = r % d.I, b = r / d.I;
        const int c = kk / d.k, tt = kk - c * d.k;
        const int col = so * d.stride + tt - d.pad_left;
        if (col >= 0 && col < d.S_in) v = x[(((long)b * d.cin + c) * d.I + ii) * d.S_in + col];
      }
      patch[t * KP + kk] = v;
    }
    for (int i = tid; i < CIN_TOK * C0; i += 256) {
      const int t = i / C0, c = i % C0, tok = t0 + t;
      g[t * GP + c] = tok < total ? to_f32(dz[(long)tok * C0 + c]) : 0.f;
    }
    __syncthreads();
    if (tid < K) {
      for (int t = 0; t < CIN_TOK; ++t) {
        const float pv = patch[t * KP + tid];
        const float* gr = g + t * GP;
#pragma unroll
        for (int n = 0; n < C0; n += 4) {
          const float4 gv = *reinterpret_cast<const float4*>(gr + n);
          acc[n] += pv * gv.x; acc[n + 1] += pv * gv.y; acc[n + 2] += pv * gv.z; acc[n + 3] += pv * gv.w;
        }
      }
    } else if (bn >= 0) {
      for (int t = 0; t < CIN_TOK; ++t) bsum += g[t * GP + bn];
    }
  }
  if (tid < K) {
#pragma unroll
    for (int n = 0; n < C0; ++n) atomicAdd(dw + (long)n * K + tid, acc[n]);
  } else if (bn >= 0 && dbias) {
    atomicAdd(dbias + bn, bsum);
  }
}

// Small filters (K = cin * k <= 16, the seismic [1,3] in-conv: K = 6): the kernel above would keep K of its 256 threads busy.
// Here thread (n = tid & 63, kg = tid >> 6) owns dw[n][kg], dw[n][kg + 4], ...; same staging, same contiguous atomics.
// K = cin * k <= 16 (the seismic [1, 3] in-conv: 20 MFLOP in all).  A wave is the 64 output channels; it walks its own slice of
// tokens straight from global memory -- one coalesced 128/256-byte row of dz per token, the token's K input samples through
// wave-uniform (scalar) loads -- with the K + 1 running sums per channel in registers, no LDS tile and no barrier in the loop.
// 16 waves per workgroup fold their sums through LDS and ONE wave issues the atomics: the 64 x K gradient is 12 cache lines, and
// atomics into one line retire ~3.4 ns apart, so the number of workgroups (64) is what bounds the tail, not the math.
template <typename TZ>
__global__ __launch_bounds__(1024) void conv_in_bwd_weight_tiny_kernel(const float* __restrict__ x, const TZ* __restrict__ dz,
                                                                       float* __restrict__ dw, float* __restrict__ dbias,
                                                                       focal_conv_in_desc d, int K, int total, int tok_per_wave) {
  __shared__ float red[16][17][64];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gw = blockIdx.x * 16 + wave;
  float acc[16];
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) acc[kk] = 0.f;
  float bsum = 0.f;
  const int t_begin = gw * tok_per_wave, t_end = min(total, t_begin + tok_per_wave);
  // lane kk < K fetches tap kk of the current token (one vector load for all K samples); v_readlane hands each to the whole wave
  const int my_c = lane < K ? lane / d.k : 0, my_tt = lane < K ? lane - my_c * d.k : 0;
  const long my_off = (long)my_c * d.I * d.S_in + my_tt - d.pad_left;
  int so = t_begin % d.S_out, r0 = t_begin / d.S_out;  // wave-uniform; advanced incrementally (no division in the loop)
  long row_base = (((long)(r0 / d.I) * d.cin) * d.I + (r0 % d.I)) * d.S_in;
  int ii = r0 % d.I;
#pragma unroll 4
  for (int tok = t_begin; tok < t_end; ++tok) {
    const float g = to_f32(dz[(long)tok * 64 + lane]);
    const int col = so * d.stride + my_tt - d.pad_left;
    float xv = 0.f;
    if (lane < K && col >= 0 && col < d.S_in) xv = x[row_base + (long)so * d.stride + my_off];
    bsum += g;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk)
      if (kk < K) acc[kk] += g * __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xv), kk));
    if (++so == d.S_out) {
      so = 0;
      if (++ii == d.I) { ii = 0; row_base += (long)(d.cin - 1) * d.I * d.S_in; }
      row_base += d.S_in;
    }
  }
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) red[wave][kk][lane] = acc[kk];
  red[wave][16][lane] = bsum;
  __syncthreads();
  if (wave == 0) {
    for (int kk = 0; kk <= 16; ++kk) {
      if (kk < K || kk == 16) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) v += red[w][kk][lane];
        if (kk < 16) atomicAdd(dw + (long)lane * K + kk, v);
        else if (dbias) atomicAdd(dbias + lane, v);
      }
    }
  }
}

template <typename TZ, int C0>
__global__ __launch_bounds__(256) void conv_in_bwd_weight_smallk_kernel(const float* __restrict__ x, const TZ* __restrict__ dz,
                                                                        float* __restrict__ dw, float* __restrict__ dbias,
                                                                        focal_conv_in_desc d, int K, int total, int chunks_per_wg) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int KP = K + 1, GP = C0 + 4;
  float* patch = smem;              // [CIN_TOK][KP]
  float* g = smem + CIN_TOK * KP;   // [CIN_TOK][GP]
  const int tid = threadIdx.x, n = tid & (C0 - 1), kg = tid / C0;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  for (int ch = 0; ch < chunks_per_wg; ++ch) {
    const int t0 = (blockIdx.x * chunks_per_wg + ch) * CIN_TOK;
    if (t0 >= total) break;
    __syncthreads();
    // every load of the chunk is issued before the first LDS write (a load -> wait -> write loop costs one memory latency per
    // iteration: 16 of them for the gradient tile alone, ~80 % of this kernel's time)
    constexpr int EPT = 16 / (int)sizeof(TZ), NG = CIN_TOK * C0 / EPT / 256;  // 16-byte pieces of the dz tile per thread
    float gv[NG][EPT];
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      const int i = tid + 256 * j, t = i / (C0 / EPT), c = (i % (C0 / EPT)) * EPT, tok = t0 + t;
#pragma unroll
      for (int e = 0; e < EPT; ++e) gv[j][e] = 0.f;
      if (tok < total) loadN<EPT>(dz + (long)tok * C0 + c, gv[j]);
    }
    float pv[4];  // K <= 16: at most 4 patch values per thread
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = tid + 256 * j, t = i / K, kk = i - t * K, tok = t0 + t;
      pv[j] = 0.f;
      if (i < CIN_TOK * K && tok < total) {
        const int so = tok % d.S_out, r = tok / d.S_out, ii = r % d.I, b = r / d.I;
        const int c = kk / d.k, tt = kk - c * d.k;
        const int col = so * d.stride + tt - d.pad_left;
        if (col >= 0 && col < d.S_in) pv[j] = x[(((long)b * d.cin + c) * d.I + ii) * d.S_in + col];
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = tid + 256 * j;
      if (i < CIN_TOK * K) patch[(i / K) * KP + (i % K)] = pv[j];
    }
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      const int i = tid + 256 * j, t = i / (C0 / EPT), c = (i % (C0 / EPT)) * EPT;
#pragma unroll
      for (int e = 0; e < EPT; e += 4) *reinterpret_cast<float4*>(g + t * GP + c + e) = make_float4(gv[j][e], gv[j][e + 1], gv[j][e + 2], gv[j][e + 3]);
    }
    __syncthreads();
    for (int t = 0; t < CIN_TOK; ++t) {
      const float gv = g[t * GP + n];
      if (kg == 0) bsum += gv;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kk = kg + 4 * j;
        if (kk < K) acc[j] += gv * patch[t * KP + kk];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int kk = kg + 4 * j;
    if (kk < K) atomicAdd(dw + (long)n * K + kk, acc[j]);
  }
  if (kg == 0 && dbias) atomicAdd(dbias + n, bsum);
}

// ---- matrix-core forms for the patchifying case (stride == k, no padding, S_in == S_out * k: the audio in-conv, K = 160).
// The VALU kernels above spend most of their time in per-element index arithmetic (98 / 127 us against a 10 us traffic
// bound); here a 64-token tile of patches is staged in LDS with 16-byte loads (a token's taps are contiguous per input
// channel) and the contraction runs on the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32), so the layer stays fp32 end to end.
typedef float cf4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ cf4 cmfma(float a, float b, cf4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// (patch tiles are staged by patch_stage.hpp: all of a thread's loads in flight before the first LDS write)
// z[tok][n] = bias[n] + sum_kk patch[tok][kk] * w[n][kk].  Wave w owns tokens 16w..16w+15 of the tile and all 64 channels;
// the filter bank lives in registers as MFMA fragments (K = 160: 160 VGPRs) for the whole (grid-strided) kernel.
template <int K>
__global__ __launch_bounds__(256) void conv_in_fwd_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ bias, float* __restrict__ z,
                                                               focal_conv_in_desc d, int total, PatchGeom pg) {
  constexpr int KP = K + 4, KS = K / 4;
  __shared__ __attribute__((aligned(16))) float patch[CIN_TOK * KP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lm = lane & 15, lg = lane >> 4;
  float wf[4][KS];
  load_filter_fragments<K>(w, patch, tid, wf);  // coalesced, through the (same-shaped) patch tile
  float4 bv[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) bv[nt] = *reinterpret_cast<const float4*>(bias + 16 * nt + 4 * lg);
  PatchPlan<K, CIN_TOK> plan;
  patch_stage_plan<K, CIN_TOK, KP>(plan, pg, tid);
  for (int t0 = blockIdx.x * CIN_TOK; t0 < total; t0 += gridDim.x * CIN_TOK) {
    __syncthreads();
    patch_stage_tile<K, CIN_TOK>(plan, pg, x, patch, t0, total);
    __syncthreads();
    cf4 acc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[nt] = cf4{bv[nt].x, bv[nt].y, bv[nt].z, bv[nt].w};
    const float* pr = patch + (16 * wave + lm) * KP + lg;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const float pf = pr[4 * ks];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[nt] = cmfma(wf[nt][ks], pf, acc[nt]);
    }
    const int tok = t0 + 16 * wave + lm;
    if (tok < total) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        *reinterpret_cast<float4*>(z + (long)tok * 64 + 16 * nt + 4 * lg) = make_float4(acc[nt][0], acc[nt][1], acc[nt][2], acc[nt][3]);
    }
  }
}

// dw[n][kk] += sum_tok dz[tok][n] * patch[tok][kk], dbias[n] += sum_tok dz[tok][n].  40 output tiles (4 x 10 of 16 x 16), ten per
// wave, accumulated in registers over this workgroup's chunks; they leave through an LDS copy of the [64][K] block so that every
// atomic wave-instruction covers 256 contiguous bytes.
template <typename TZ, int K>
__global__ __launch_bounds__(256) void conv_in_bwd_weight_mfma_kernel(const float* __restrict__ x, const TZ* __restrict__ dz,
                                                                      float* __restrict__ dw, float* __restrict__ dbias,
                                                                      focal_conv_in_desc d, int total, int chunks_per_wg, PatchGeom pg) {
  constexpr int KP = K + 16, GP = 80, NKT = K / 16, TPW = 4 * NKT / 4;  // tiles per wave
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* patch = smem;             // [CIN_TOK][KP]   (reused as the [64][K] staging block at the end)
  float* g = smem + CIN_TOK * KP;  // [CIN_TOK][GP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lm = lane & 15, lg = lane >> 4;
  cf4 acc[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i) acc[i] = cf4{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  PatchPlan<K, CIN_TOK> plan;
  patch_stage_plan<K, CIN_TOK, KP>(plan, pg, tid);
  for (int ch = 0; ch < chunks_per_wg; ++ch) {
    const int t0 = (blockIdx.x * chunks_per_wg + ch) * CIN_TOK;
    if (t0 >= total) break;
    __syncthreads();
    float gv[4][4];  // this thread's four 16-byte pieces of the dz tile: requested before the patch loads, written after them
#pragma unroll
    for (int j = 0; j < 4; ++j) {  // 16 float4 per token row
      const int i = tid + 256 * j, t = i >> 4, c4 = (i & 15) * 4, tok = t0 + t;
      gv[j][0] = gv[j][1] = gv[j][2] = gv[j][3] = 0.f;
      if (tok < total) loadN<4>(dz + (long)tok * 64 + c4, gv[j]);
    }
    patch_stage_tile<K, CIN_TOK>(plan, pg, x, patch, t0, total);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = tid + 256 * j, t = i >> 4, c4 = (i & 15) * 4;
      *reinterpret_cast<float4*>(g + t * GP + c4) = make_float4(gv[j][0], gv[j][1], gv[j][2], gv[j][3]);
    }
    __syncthreads();
    if (tid < 64) {
      for (int t = 0; t < CIN_TOK; ++t) bsum += g[t * GP + tid];
    }
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
      const int tile = wave + 4 * i, mt = tile / NKT, kt = tile - mt * NKT;
      const float* ga = g + lg * GP + 16 * mt + lm;
      const float* pb = patch + lg * KP + 16 * kt + lm;
#pragma unroll
      for (int ks = 0; ks < CIN_TOK / 4; ++ks) acc[i] = cmfma(ga[4 * ks * GP], pb[4 * ks * KP], acc[i]);
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int tile = wave + 4 * i, mt = tile / NKT, kt = tile - mt * NKT;
#pragma unroll
    for (int r = 0; r < 4; ++r) patch[(16 * mt + 4 * lg + r) * K + 16 * kt + lm] = acc[i][r];
  }
  __syncthreads();
  for (int i = tid; i < 64 * K; i += 256) atomicAdd(dw + i, patch[i]);
  if (tid < 64 && dbias) atomicAdd(dbias + tid, bsum);
}

static bool conv_in_is_patchify(const focal_conv_in_desc* d) {
  return d->stride == d->k && d->pad_left == 0 && d->S_in == d->S_out * d->k && d->k % 4 == 0 && d->cin * d->k == 160 && d->C == 64;
}

static int conv_in_check(const focal_conv_in_desc* d) {
  FOCAL_CHECK_ARG(d != nullptr, "conv_in: null descriptor");
  FOCAL_CHECK_ARG(d->C == 64, "conv_in: %d output channels unsupported (64 only)", d->C);
  FOCAL_CHECK_ARG(d->cin * d->k <= 192 && d->k >= 1 && d->stride >= 1, "conv_in: kernel too large for the LDS patch buffer");
  return FOCAL_OK;
}

extern "C" int focal_conv_in_fwd(const focal_conv_in_desc* d, const float* x, const float* w, const float* bias, float* z,
                                 void* stream) {
  if (int rc = conv_in_check(d)) return rc;
  FOCAL_CHECK_ARG(x && w && bias && z, "conv_in_fwd: null tensor");
  const int K = d->cin * d->k, total = d->B * d->I * d->S_out;
  if (conv_in_is_patchify(d)) {
    int blocks = ceil_div(total, CIN_TOK);
    // one workgroup per CU: every workgroup pays for the 40 KB filter bank once (measured at 800 tiles: 26 us at 200-256
    // workgroups, 29 at 512, 32.5 at 768)
    const int cap = 256;
    if (blocks > cap) blocks = cap;
    const PatchGeom pg = make_patch_geom(d->S_out, d->I, d->I, d->S_in, d->k, d->cin);
    FOCAL_LAUNCH((conv_in_fwd_mfma_kernel<160>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, w, bias, z, *d, total, pg);
    FOCAL_LAUNCH_CHECK();
    return FOCAL_OK;
  }
  const size_t sm = ((size_t)K * d->C + (size_t)CIN_TOK * (K + 1)) * sizeof(float);
  int blocks = ceil_div(total, CIN_TOK);
  if (blocks > 2048) blocks = 2048;
  FOCAL_LAUNCH((conv_in_fwd_kernel<64>), dim3(blocks), dim3(256), sm, (hipStream_t)stream, x, w, bias, z, *d, K, total);
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}

extern "C" int focal_conv_in_bwd_weight(const focal_conv_in_desc* d, const float* x, const void* dz, int dz_dtype, float* dw,
                                        float* dbias, void* stream) {
  if (int rc = conv_in_check(d)) return rc;
  FOCAL_CHECK_ARG(x && dz && dw, "conv_in_bwd_weight: null tensor");
  const int K = d->cin * d->k, total = d->B * d->I * d->S_out;
  if (conv_in_is_patchify(d)) {
    const int chunks = ceil_div(total, CIN_TOK);
    int cpw = ceil_div(chunks, 256);
    if (cpw < 1) cpw = 1;
    const int blocks = ceil_div(chunks, cpw);
    const size_t smm = ((size_t)CIN_TOK * (160 + 16) + (size_t)CIN_TOK * 80) * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    const PatchGeom pg = make_patch_geom(d->S_out, d->I, d->I, d->S_in, d->k, d->cin);
    if (dz_dtype == FOCAL_F32)
      FOCAL_LAUNCH((conv_in_bwd_weight_mfma_kernel<float, 160>), dim3(blocks), dim3(256), smm, st, x, (const float*)dz, dw, dbias, *d, total, cpw, pg);
    else
      FOCAL_LAUNCH((conv_in_bwd_weight_mfma_kernel<bf16_t, 160>), dim3(blocks), dim3(256), smm, st, x, (const bf16_t*)dz, dw, dbias, *d, total, cpw, pg);
    FOCAL_LAUNCH_CHECK();
    return FOCAL_OK;
  }
  if (K <= 16 && d->C == 64) {
    const int tiny_wg = 64;
    const int waves = tiny_wg * 16;
    const int tpw = ceil_div(total, waves);
    const int blocks = ceil_div(ceil_div(total, tpw), 16);
    hipStream_t st = (hipStream_t)stream;
    if (dz_dtype == FOCAL_F32)
      FOCAL_LAUNCH((conv_in_bwd_weight_tiny_kernel<float>), dim3(blocks), dim3(1024), 0, st, x, (const float*)dz, dw, dbias, *d, K, total, tpw);
    else
      FOCAL_LAUNCH((conv_in_bwd_weight_tiny_kernel<bf16_t>), dim3(blocks), dim3(1024), 0, st, x, (const bf16_t*)dz, dw, dbias, *d, K, total, tpw);
    FOCAL_LAUNCH_CHECK();
    return FOCAL_OK;
  }
  if (K <= 16) {
    const size_t sms = ((size_t)CIN_TOK * (K + 1) + (size_t)CIN_TOK * (d->C + 4)) * sizeof(float);
    const int chunks = ceil_div(total, CIN_TOK);
    const int wg_cap = 512;
    int cpw = ceil_div(chunks, wg_cap);
    if (cpw < 1) cpw = 1;
    const int blocks = ceil_div(chunks, cpw);
    hipStream_t st = (hipStream_t)stream;
    if (dz_dtype == FOCAL_F32)
      FOCAL_LAUNCH((conv_in_bwd_weight_smallk_kernel<float, 64>), dim3(blocks), dim3(256), sms, st, x, (const float*)dz, dw, dbias, *d, K, total, cpw);
    else
      FOCAL_LAUNCH((conv_in_bwd_weight_smallk_kernel<bf16_t, 64>), dim3(blocks), dim3(256), sms, st, x, (const bf16_t*)dz, dw, dbias, *d, K, total, cpw);
    FOCAL_LAUNCH_CHECK();
    return FOCAL_OK;
  }
  FOCAL_CHECK_ARG(K + d->C <= 256, "conv_in_bwd_weight: cin*k + C must be <= 256");
  const size_t sm = ((size_t)CIN_TOK * (K + 1) + (size_t)CIN_TOK * (d->C + 4)) * sizeof(float);
  const int chunks = ceil_div(total, CIN_TOK);
  int cpw = ceil_div(chunks, 256);
  if (cpw < 1) cpw = 1;
  const int blocks = ceil_div(chunks, cpw);
  hipStream_t st = (hipStream_t)stream;
  if (dz_dtype == FOCAL_F32)
    FOCAL_LAUNCH((conv_in_bwd_weight_kernel<float, 64>), dim3(blocks), dim3(256), sm, st, x, (const float*)dz, dw, dbias, *d, K, total, cpw);
  else
    FOCAL_LAUNCH((conv_in_bwd_weight_kernel<bf16_t, 64>), dim3(blocks), dim3(256), sm, st, x, (const bf16_t*)dz, dw, dbias, *d, K, total, cpw);
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}

// ------------------------------------------------------------------------------------------------ weight re-layouts
template <typename TD> __global__ void permute_pack_kernel(const float* __restrict__ src, TD* __restrict__ dst, int A, int Bd, int Cd) {
  const long n = (long)A * Bd * Cd;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
    const int b = e % Bd, c = (e / Bd) % Cd, a = e / ((long)Bd * Cd);  // dst index [a][c][b]
    dst[e] = from_f32<TD>(src[((long)a * Bd + b) * Cd + c]);
  }
}
__global__ void permute_unpack_add_kernel(const float* __restrict__ src, float* __restrict__ dst, int A, int Bd, int Cd) {
  const long n = (long)A * Bd * Cd;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
    const int c = e % Cd, b = (e / Cd) % Bd, a = e / ((long)Bd * Cd);  // dst index [a][b][c]
    dst[e] += src[((long)a * Cd + c) * Bd + b];
  }
}
// w_bwd[ci][t'][co] = w[co][ci][k-1-t']: the data gradient of a "same" conv is the conv with flipped taps
template <typename TD> __global__ void conv_pack_bwd_kernel(const float* __restrict__ w, TD* __restrict__ dst, int Co, int Ci, int k) {
  const long n = (long)Co * Ci * k;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
    const int co = e % Co, t = (e / Co) % k, ci = e / ((long)Co * k);
    dst[e] = from_f32<TD>(w[((long)co * Ci + ci) * k + (k - 1 - t)]);
  }
}

extern "C" int focal_permute_pack(int A, int Bd, int Cd, const float* src, void* dst, int dtype, void* stream) {
  FOCAL_CHECK_ARG(src && dst && A > 0 && Bd > 0 && Cd > 0, "permute_pack: bad argument");
  const int blocks = min(1024, ceil_div((long)A * Bd * Cd, 256));
  if (dtype == FOCAL_F32) FOCAL_LAUNCH((permute_pack_kernel<float>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, (float*)dst, A, Bd, Cd);
  else FOCAL_LAUNCH((permute_pack_kernel<bf16_t>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, A, Bd, Cd);
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}
extern "C" int focal_permute_unpack_add(int A, int Bd, int Cd, const float* src, float* dst, void* stream) {
  FOCAL_CHECK_ARG(src && dst && A > 0 && Bd > 0 && Cd > 0, "permute_unpack_add: bad argument");
  const int blocks = min(1024, ceil_div((long)A * Bd * Cd, 256));
  FOCAL_LAUNCH(permute_unpack_add_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, dst, A, Bd, Cd);
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}
extern "C" int focal_conv_pack_bwd(const focal_conv_desc* d, const float* w, void* w_bwd, void* stream) {
  FOCAL_CHECK_ARG(d && w && w_bwd, "conv_pack_bwd: null argument");
  const int blocks = min(1024, ceil_div((long)d->C_out * d->C_in * d->k, 256));
  if (d->dtype == FOCAL_F32) FOCAL_LAUNCH((conv_pack_bwd_kernel<float>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (float*)w_bwd, d->C_out, d->C_in, d->k);
  else FOCAL_LAUNCH((conv_pack_bwd_kernel<bf16_t>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (bf16_t*)w_bwd, d->C_out, d->C_in, d->k);
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}

// Every weight re-ordering of an encoder in ONE launch (DeepSense re-orders 13 small weights per pass -- four conv filters forward and
// backward, the 1x1 output conv, four GRU W_hh -- and folds 5 packed weight gradients back: 18 launches of ~5 us on each pass's chain).
struct PackTable { int n; focal_pack_entry e[FOCAL_PACK_MAX]; };
template <typename TD> __global__ void pack_multi_kernel(PackTable t) {
  const focal_pack_entry& q = t.e[blockIdx.y];
  const long n = (long)q.A * q.B * q.C;
  TD* dst = reinterpret_cast<TD*>(q.dst);
  const float* src = reinterpret_cast<const float*>(q.src);
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
    if (q.kind == FOCAL_PACK_PERMUTE) {          // dst[a][c][b] = src[a][b][c]
      const int b = e % q.B, c = (e / q.B) % q.C, a = e / ((long)q.B * q.C);
      dst[e] = from_f32<TD>(src[((long)a * q.B + b) * q.C + c]);
    } else if (q.kind == FOCAL_PACK_CONV_BWD) {  // dst[ci][t][co] = src[co][ci][k-1-t]  (A = Co, B = Ci, C = k)
      const int co = e % q.A, tt = (e / q.A) % q.C, ci = e / ((long)q.A * q.C);
      dst[e] = from_f32<TD>(src[((long)co * q.B + ci) * q.C + (q.C - 1 - tt)]);
    } else {
      // FOCAL_PACK_FRAG / _FRAG_T: the [R][Cc] matrix m (= src, or its transpose) in MFMA-fragment order -- e enumerates dst:
      // e = ((tile_r * (Cc / 32) + ks) * 64 + lane) * 8 + j   <->   m[16 tile_r + lane % 16][32 ks + 8 (lane / 16) + j]
      const bool tr = q.kind == FOCAL_PACK_FRAG_T;
      const int R = tr ? q.B : q.A, Cc = tr ? q.A : q.B;
      const int j = e & 7, lane = (e >> 3) & 63;
      const long frag = e >> 9;
      const int ks = (int)(frag % (Cc / 32)), tile_r = (int)(frag / (Cc / 32));
      const int r = 16 * tile_r + (lane & 15), c = 32 * ks + 8 * (lane >> 4) + j;
      (void)R;
      dst[e] = from_f32<TD>(tr ? src[(long)c * q.B + r] : src[(long)r * q.B + c]);
    }
  }
}
__global__ void unpack_add_multi_kernel(PackTable t) {
  const focal_pack_entry& q = t.e[blockIdx.y];
  const long n = (long)q.A * q.B * q.C;
  float* dst = reinterpret_cast<float*>(q.dst);
  const float* src = reinterpret_cast<const float*>(q.src);
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
    const int c = e % q.C, b = (e / q.C) % q.B, a = e / ((long)q.B * q.C);  // dst[a][b][c] += src[a][c][b]
    atomicAdd(dst + e, src[((long)a * q.C + c) * q.B + b]);  // (the two views' passes of a step end on different streams and add into the same gradient)
  }
}
static int pack_table(int n, const focal_pack_entry* e, PackTable* t, long* most) {
  FOCAL_CHECK_ARG(n >= 1 && n <= FOCAL_PACK_MAX && e, "pack_multi: 1 .. %d entries", FOCAL_PACK_MAX);
  t->n = n;
  *most = 0;
  for (int i = 0; i < n; ++i) {
    FOCAL_CHECK_ARG(e[i].src && e[i].dst && e[i].A > 0 && e[i].B > 0 && e[i].C > 0 && e[i].kind >= FOCAL_PACK_PERMUTE && e[i].kind <= FOCAL_PACK_FRAG_T,
                    "pack_multi: bad entry %d", i);
    if (e[i].kind == FOCAL_PACK_FRAG) FOCAL_CHECK_ARG(e[i].C == 1 && e[i].A % 16 == 0 && e[i].B % 32 == 0, "pack_multi: entry %d: fragment order needs [16 m][32 n], C = 1", i);
    if (e[i].kind == FOCAL_PACK_FRAG_T) FOCAL_CHECK_ARG(e[i].C == 1 && e[i].B % 16 == 0 && e[i].A % 32 == 0, "pack_multi: entry %d: transposed fragment order needs [32 m][16 n], C = 1", i);
    t->e[i] = e[i];
    const long cnt = (long)e[i].A * e[i].B * e[i].C;
    if (cnt > *most) *most = cnt;
  }
  return FOCAL_OK;
}
extern "C" int focal_pack_multi(int dtype, int n, const focal_pack_entry* entries, void* stream) {
  FOCAL_CHECK_ARG(dtype == FOCAL_F32 || dtype == FOCAL_BF16, "pack_multi: bad dtype");
  PackTable t;
  long most;
  if (int rc = pack_table(n, entries, &t, &most)) return rc;
  const dim3 grid(min(64, ceil_div(most, 256)), n);
  if (dtype == FOCAL_F32) FOCAL_LAUNCH((pack_multi_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, t);
  else FOCAL_LAUNCH((pack_multi_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, t);
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}
extern "C" int focal_unpack_add_multi(int n, const focal_pack_entry* entries, void* stream) {
  PackTable t;
  long most;
  if (int rc = pack_table(n, entries, &t, &most)) return rc;
  for (int i = 0; i < n; ++i) FOCAL_CHECK_ARG(entries[i].kind == FOCAL_PACK_PERMUTE, "unpack_add_multi: entry %d is not a permutation", i);
  FOCAL_LAUNCH(unpack_add_multi_kernel, dim3(min(64, ceil_div(most, 256)), n), dim3(256), 0, (hipStream_t)stream, t);
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}

// ------------------------------------------------------------------------------------------------ [1,k] convs as GEMMs
static int conv_check(const focal_conv_desc* d) {
  FOCAL_CHECK_ARG(d != nullptr, "conv: null descriptor");
  FOCAL_CHECK_ARG(d->dtype == FOCAL_F32 || d->dtype == FOCAL_BF16, "conv: bad dtype");
  FOCAL_CHECK_ARG(d->k % 2 == 1 && d->k >= 1, "conv: kernel length %d must be odd ('same' padding)", d->k);
  FOCAL_CHECK_ARG(d->rows % d->S == 0 && d->C_in % 8 == 0 && d->C_out % 8 == 0, "conv: rows %% S != 0 or channels not multiples of 8");
  return FOCAL_OK;
}
static MaskParams conv_window(int S, int C, int pad) {
  MaskParams m;
  memset(&m, 0, sizeof(m));
  m.rows_per_sample = S; m.ncols = C; m.stream_elem = (uint32_t)pad;
  return m;
}
static size_t esize(int dtype) { return dtype == FOCAL_F32 ? 4 : 2; }

// focal_conv_fwd + the training-mode statistics of the BatchNorm behind it (focal_bn_stats, FOCAL_BN_TRAIN) from the GEMM's epilogue: the
// pre-BN tensor z is not read back for them and the statistics launch is gone (EPI_STORE_STATS, gemm_body.inc).
extern "C" int focal_conv_fwd_bn(const focal_conv_desc* d, const void* x, const void* w_fwd, const float* bias, float* z,
                                 const focal_bn_desc* bn, float* scratch, float* mean_rstd, float* running_mean, float* running_var,
                                 void* stream) {
  if (int rc = conv_check(d)) return rc;
  FOCAL_CHECK_ARG(x && w_fwd && z && bn && scratch, "conv_fwd_bn: null tensor");
  FOCAL_CHECK_ARG(bn->rows == d->rows && bn->C == d->C_out && (running_mean == nullptr) == (running_var == nullptr),
                  "conv_fwd_bn: the BatchNorm descriptor does not describe the convolution's output");
  if (d->dtype != FOCAL_BF16) {
    focal_set_error("conv_fwd_bn: bf16 operands only (fp32: focal_conv_fwd + focal_bn_stats)");
    return FOCAL_EUNSUPPORTED;
  }
  const int pad = d->k / 2, K = d->k * d->C_in;
  {
    const int G = bn->groups > 1 ? bn->groups : 1;
    if (conv_ring_fits(d, d->C_in, d->C_out, x, w_fwd, G) && (G == 1 || bn->stat_rows <= 0)) {
      ConvRingParams rp;
      memset(&rp, 0, sizeof(rp));
      rp.x = (const bf16_t*)x; rp.w = (const bf16_t*)w_fwd; rp.bias = bias; rp.out = z; rp.rows = d->rows; rp.S = d->S;
      rp.bn_sums = scratch; rp.bn_mean_rstd = mean_rstd; rp.bn_run_mean = running_mean; rp.bn_run_var = running_var;
      rp.bn_rows = G > 1 ? bn->rows / G : (bn->stat_rows > 0 ? bn->stat_rows : bn->rows);
      rp.bn_eps = bn->eps; rp.bn_momentum = bn->momentum; rp.bn_groups = G;
      focal_note_kernel(d->k == 5 ? "conv_ring_kernel<5, stats>" : "conv_ring_kernel<3, stats>");
      hipError_t e = conv_ring_launch<CR_STORE_STATS>(rp, d->k, G, (hipStream_t)stream);
      if (e != hipSuccess) { focal_set_error("conv_fwd_bn (row ring): launch failed: %s", hipGetErrorString(e)); return FOCAL_EHIP; }
      return FOCAL_OK;
    }
  }
  if (mean_rstd == nullptr) {
    focal_set_error("conv_fwd_bn: the sums-only form (mean_rstd = NULL) needs the row-ring kernel (focal_conv_fwd_bn_sums_supported)");
    return FOCAL_EUNSUPPORTED;
  }
  GemmSpec s{d->dtype, d->dtype, d->dtype, FOCAL_F32, false, false, PRO_CONV, PRO_NONE, EPI_STORE_STATS};
  GemmParams p;
  memset(&p, 0, sizeof(p));
  p.M = d->rows; p.N = d->C_out; p.K = K;
  p.A = (const char*)x - (size_t)pad * d->C_in * esize(d->dtype); p.lda = d->C_in;
  p.B = w_fwd; p.ldb = K;
  p.C = z; p.ldc = d->C_out;
  p.batch = 1; p.splits = 1; p.alpha = 1.f; p.bias = bias;
  p.proA = conv_window(d->S, d->C_in, pad);
  p.bn_sums = scratch; p.bn_mean_rstd = mean_rstd; p.bn_run_mean = running_mean; p.bn_run_var = running_var;
  p.bn_rows = bn->stat_rows > 0 ? bn->stat_rows : bn->rows; p.bn_eps = bn->eps; p.bn_momentum = bn->momentum;
  if (bn->groups > 1) {  // (the GEMM tiles are 64 or 128 rows high: a tile then lies inside one group)
    FOCAL_CHECK_ARG(bn->stat_rows <= 0 && bn->rows % bn->groups == 0 && (bn->rows / bn->groups) % 128 == 0,
                    "conv_fwd_bn: %d statistic groups need rows / groups (%d / %d) to be a multiple of 128 and no stat_rows", bn->groups, bn->rows, bn->groups);
    p.bn_groups = bn->groups; p.bn_rows = bn->rows / bn->groups;
  }
  return focal_launch_gemm(s, p, (hipStream_t)stream);
}

// 1: focal_conv_fwd_bn takes mean_rstd = NULL for this convolution (the row-ring kernel runs it) and focal_bn_act_fwd_sums finishes the statistics
extern "C" int focal_conv_fwd_bn_sums_supported(const focal_conv_desc* d, const focal_bn_desc* bn, const void* x, const void* w_fwd) {
  if (d == nullptr || bn == nullptr || d->k % 2 != 1 || d->S <= 0 || d->rows % d->S != 0) return 0;
  const char* sel = getenv("FOCAL_CONV_BN_SUMS");
  if (sel != nullptr && sel[0] == '0') return 0;
  const int G = bn->groups > 1 ? bn->groups : 1;
  return bn->rows == d->rows && bn->C == 64 && conv_ring_fits(d, d->C_in, d->C_out, x, w_fwd, G) && (G == 1 || bn->stat_rows <= 0);
}

extern "C" int focal_conv_fwd(const focal_conv_desc* d, const void* x, const void* w_fwd, const float* bias, float* z, void* stream) {
  if (int rc = conv_check(d)) return rc;
  FOCAL_CHECK_ARG(x && w_fwd && z, "conv_fwd: null tensor");
  const int pad = d->k / 2, K = d->k * d->C_in;
  if (conv_ring_fits(d, d->C_in, d->C_out, x, w_fwd, 1)) {
    ConvRingParams rp;
    memset(&rp, 0, sizeof(rp));
    rp.x = (const bf16_t*)x; rp.w = (const bf16_t*)w_fwd; rp.bias = bias; rp.out = z; rp.rows = d->rows; rp.S = d->S;
    focal_note_kernel(d->k == 5 ? "conv_ring_kernel<5, store>" : "conv_ring_kernel<3, store>");
    hipError_t e = conv_ring_launch<CR_STORE>(rp, d->k, 1, (hipStream_t)stream);
    if (e != hipSuccess) { focal_set_error("conv_fwd (row ring): launch failed: %s", hipGetErrorString(e)); return FOCAL_EHIP; }
    return FOCAL_OK;
  }
  GemmSpec s{d->dtype, d->dtype, d->dtype, FOCAL_F32, false, false, PRO_CONV, PRO_NONE, EPI_STORE};
  GemmParams p;
  memset(&p, 0, sizeof(p));
  p.M = d->rows; p.N = d->C_out; p.K = K;
  p.A = (const char*)x - (size_t)pad * d->C_in * esize(d->dtype); p.lda = d->C_in;   // A[m][kk] = tokens[(m - pad) * C_in + kk]
  p.B = w_fwd; p.ldb = K;
  p.C = z; p.ldc = d->C_out;
  p.batch = 1; p.splits = 1; p.alpha = 1.f; p.bias = bias;
  p.proA = conv_window(d->S, d->C_in, pad);
  return focal_launch_gemm(s, p, (hipStream_t)stream);
}

extern "C" int focal_conv_bwd_data(const focal_conv_desc* d, const void* dz, const void* w_bwd, const float* g_in, float* g_out,
                                   void* stream) {
  if (int rc = conv_check(d)) return rc;
  FOCAL_CHECK_ARG(dz && w_bwd && g_in && g_out, "conv_bwd_data: null tensor");
  const int pad = d->k / 2, K = d->k * d->C_out;
  if (conv_ring_fits(d, d->C_out, d->C_in, dz, w_bwd, 1) && ((uintptr_t)g_in % 16 == 0) && ((uintptr_t)g_out % 16 == 0)) {
    ConvRingParams rp;
    memset(&rp, 0, sizeof(rp));
    rp.x = (const bf16_t*)dz; rp.w = (const bf16_t*)w_bwd; rp.resid = g_in; rp.out = g_out; rp.rows = d->rows; rp.S = d->S;
    focal_note_kernel(d->k == 5 ? "conv_ring_kernel<5, resid>" : "conv_ring_kernel<3, resid>");
    hipError_t e = conv_ring_launch<CR_RESID>(rp, d->k, 1, (hipStream_t)stream);
    if (e != hipSuccess) { focal_set_error("conv_bwd_data (row ring): launch failed: %s", hipGetErrorString(e)); return FOCAL_EHIP; }
    return FOCAL_OK;
  }
  GemmSpec s{d->dtype, d->dtype, d->dtype, FOCAL_F32, false, false, PRO_CONV, PRO_NONE, EPI_RESID};
  GemmParams p;
  memset(&p, 0, sizeof(p));
  p.M = d->rows; p.N = d->C_in; p.K = K;
  p.A = (const char*)dz - (size_t)pad * d->C_out * esize(d->dtype); p.lda = d->C_out;
  p.B = w_bwd; p.ldb = K;
  p.C = g_out; p.ldc = d->C_in;
  p.resid = g_in; p.ldr = d->C_in;
  p.batch = 1; p.splits = 1; p.alpha = 1.f;
  p.proA = conv_window(d->S, d->C_out, pad);
  return focal_launch_gemm(s, p, (hipStream_t)stream);
}

extern "C" int focal_conv_bwd_weight(const focal_conv_desc* d, const void* dz, const void* x, float* dw_packed, float* dbias,
                                     void* stream) {
  if (int rc = conv_check(d)) return rc;
  FOCAL_CHECK_ARG(dz && x && dw_packed, "conv_bwd_weight: null tensor");
  const int pad = d->k / 2, K = d->k * d->C_in;
  GemmSpec s{d->dtype, d->dtype, d->dtype, FOCAL_F32, true, true, PRO_NONE, PRO_CONV, EPI_ATOMIC};
  GemmParams p;
  memset(&p, 0, sizeof(p));
  p.M = d->C_out; p.N = K; p.K = d->rows;
  p.A = dz; p.lda = d->C_out;
  p.B = (const char*)x - (size_t)pad * d->C_in * esize(d->dtype); p.ldb = d->C_in;
  p.C = dw_packed; p.ldc = K;
  p.batch = 1; p.alpha = 1.f;
  p.splits = 1;  // chosen with the tile shape in gemm_dispatch.inc (launch_dw)
  p.dw_target = d->dw_workgroups;
  p.proB = conv_window(d->S, d->C_in, pad);
  p.colsumA = dbias;
  return focal_launch_gemm(s, p, (hipStream_t)stream);
}
