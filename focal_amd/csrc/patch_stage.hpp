// Staging of 64-token patch tiles for the patchifying first layers (Swin PatchEmbed, DeepSense in-conv with kernel = stride):
// token (b, row, px) of an NCHW fp32 input owns `taps` consecutive samples of each input channel.
//
// All of a thread's 16-byte loads for a tile are issued BEFORE the first LDS write (the straightforward `for (u...) { v = load;
// lds = v; }` compiles to one load -> wait -> write round trip per iteration: 5-10 exposed memory latencies per tile, which made
// these kernels 3-4x slower than their traffic), and the token -> (b, row, px) decomposition is one scalar division pair per
// tile plus a multiply-high per load instead of three runtime vector divisions per load.
#pragma once
#include "common.hpp"

struct PatchGeom {
  int tok_per_row;   // tokens along the sample axis (padded grid)
  int rows_per_img;  // token rows per window (padded grid)
  int valid_rows;    // input rows (I); token rows beyond are zero padding
  int valid_cols;    // input samples per row (S); samples beyond are zero padding
  int taps;          // samples per token per channel
  int cin;
  uint32_t magic_tpr, magic_rpi;  // ceil(2^32 / tok_per_row), ceil(2^32 / rows_per_img)
};
static inline PatchGeom make_patch_geom(int tok_per_row, int rows_per_img, int valid_rows, int valid_cols, int taps, int cin) {
  PatchGeom g;
  g.tok_per_row = tok_per_row; g.rows_per_img = rows_per_img; g.valid_rows = valid_rows; g.valid_cols = valid_cols;
  g.taps = taps; g.cin = cin;
  g.magic_tpr = (uint32_t)(((1ull << 32) + tok_per_row - 1) / tok_per_row);
  g.magic_rpi = (uint32_t)(((1ull << 32) + rows_per_img - 1) / rows_per_img);
  return g;
}

// K = cin * taps values per token, TOK tokens per tile, KP = LDS pitch; 256 threads.  `chan_off[j]` / `t_of[j]` / `kq_of[j]` are
// tile-invariant and come from patch_stage_plan (called once per kernel).
template <int K, int TOK> struct PatchPlan {
  static constexpr int UPT = K / 4, NIT = TOK * UPT / 256;
  static_assert(TOK * UPT % 256 == 0, "tile must be a whole number of 256-thread passes");
  int t_of[NIT];      // token slot inside the tile
  int lds_off[NIT];   // float offset inside the patch tile
  long chan_off[NIT]; // c * valid_rows * valid_cols + tt   (element offset of this unit's channel / tap inside a window row)
  int tt[NIT];        // first tap of this unit
};
template <int K, int TOK, int KP>
__device__ __forceinline__ void patch_stage_plan(PatchPlan<K, TOK>& pl, const PatchGeom& g, int tid) {
#pragma unroll
  for (int j = 0; j < PatchPlan<K, TOK>::NIT; ++j) {
    const int u = tid + 256 * j, t = u / PatchPlan<K, TOK>::UPT, kq = u - t * PatchPlan<K, TOK>::UPT;
    const int kk = 4 * kq, c = kk / g.taps, tt = kk - c * g.taps;
    pl.t_of[j] = t;
    pl.lds_off[j] = t * KP + 4 * kq;
    pl.chan_off[j] = (long)c * g.valid_rows * g.valid_cols + tt;
    pl.tt[j] = tt;
  }
}
template <int K, int TOK>
__device__ __forceinline__ void patch_stage_load(const PatchPlan<K, TOK>& pl, const PatchGeom& g, const float* __restrict__ x, int t0,
                                                 int total, float4* v) {
  constexpr int NIT = PatchPlan<K, TOK>::NIT;
  // t0 is uniform: scalar decomposition
  const int t0u = __builtin_amdgcn_readfirstlane(t0);
  const int r0 = t0u / g.tok_per_row, px0 = t0u - r0 * g.tok_per_row;
  const int b0 = r0 / g.rows_per_img, py0 = r0 - b0 * g.rows_per_img;
  const long img = (long)g.cin * g.valid_rows * g.valid_cols;
#pragma unroll
  for (int j = 0; j < NIT; ++j) {
    uint32_t px = (uint32_t)(px0 + pl.t_of[j]);
    const uint32_t q = __umulhi(px, g.magic_tpr);  // exact for px < 2^16
    px -= q * g.tok_per_row;
    uint32_t py = (uint32_t)py0 + q;
    const uint32_t q2 = __umulhi(py, g.magic_rpi);
    py -= q2 * g.rows_per_img;
    const long b = b0 + q2;
    const int col = (int)px * g.taps + pl.tt[j];
    v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t0 + pl.t_of[j] < total && (int)py < g.valid_rows && col < g.valid_cols)  // else: zero padding
      v[j] = *reinterpret_cast<const float4*>(x + b * img + (long)py * g.valid_cols + (long)px * g.taps + pl.chan_off[j]);
  }
}
template <int K, int TOK>
__device__ __forceinline__ void patch_stage_store(const PatchPlan<K, TOK>& pl, float* patch, const float4* v) {
#pragma unroll
  for (int j = 0; j < PatchPlan<K, TOK>::NIT; ++j) *reinterpret_cast<float4*>(patch + pl.lds_off[j]) = v[j];
}
template <int K, int TOK>
__device__ __forceinline__ void patch_stage_tile(const PatchPlan<K, TOK>& pl, const PatchGeom& g, const float* __restrict__ x,
                                                 float* patch, int t0, int total) {
  float4 v[PatchPlan<K, TOK>::NIT];
  patch_stage_load<K, TOK>(pl, g, x, t0, total, v);
  patch_stage_store<K, TOK>(pl, patch, v);
}

// The filter bank as MFMA fragments: wf[nt][ks] = w[16 nt + (lane & 15)][4 ks + (lane >> 4)].  Read straight from global
// memory that is 4 * KS strided dword loads per lane, 16 cache lines per wave-instruction: 13 us per workgroup at K = 80.
// Here the [64][K] bank goes through LDS once (16-byte coalesced loads, pitch K + 4) and fragments are LDS reads.
template <int K>
__device__ __forceinline__ void load_filter_fragments(const float* __restrict__ w, float* lds /* >= 64 * (K + 4) floats */, int tid,
                                                      float (*wf)[K / 4]) {
  constexpr int KP = K + 4;
  for (int u = tid; u < 64 * K / 4; u += 256) {
    const int n = u / (K / 4), k4 = u - n * (K / 4);
    *reinterpret_cast<float4*>(lds + n * KP + 4 * k4) = *reinterpret_cast<const float4*>(w + n * K + 4 * k4);
  }
  __syncthreads();
  const int lane = tid & 63, lm = lane & 15, lg = lane >> 4;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int ks = 0; ks < K / 4; ++ks) wf[nt][ks] = lds[(16 * nt + lm) * KP + 4 * ks + lg];
  __syncthreads();
}
