// Window geometry shared by the VALU (attn.hip) and MFMA (attn_mfma.hip) attention kernels.
#pragma once
#include "common.hpp"

#define ATT_NMAX 16  // max tokens per window

struct AttnGeom {
  int B, H, W, C, heads, hd, wh, ww, sh, sw, N, nWx, nW, shifted;
  float scale;
  // ceil(2^32 / d) for d = heads, nW, nWx: n / d == mulhi(n, magic) exactly while n * d < 2^32 (items, windows: checked on the
  // host).  The MFMA kernels decompose a wave-uniform item index per (window, head): as runtime divisions that was ~150 scalar
  // instructions per item, a third of the forward kernel's instruction stream.
  uint32_t m_heads, m_nW, m_nWx;
};
__host__ __device__ __forceinline__ uint32_t att_div(uint32_t n, uint32_t magic, uint32_t d) {
#if defined(__HIP_DEVICE_COMPILE__)
  return d == 1 ? n : __umulhi(n, magic);
#else
  return n / d;
#endif
}

// Attention-dropout stream (round 4).  The kernels are bound by vector-instruction issue, and a 32-bit hash per probability (two integer
// multiplies at quarter rate) was 24 of the forward kernel's 223 vector instructions per item.  One hash now decides TWO probabilities with its
// 16-bit halves (keep iff half >= p * 2^16: p resolved to 1.5e-5): element (window-head wh, query i, key j) uses half (j & 1) of
//     hash24( ((((wh * 16 + i) * 4 + (j >> 2)) * 2 + ((j >> 1) & 1))) ^ key ),
// i.e. an MFMA lane (query i, keys 4 g .. 4 g + 3) needs the two hashes 2 q, 2 q + 1 of q = (wh * 16 + i) * 4 + g.  All attention kernels
// (fp32 VALU and bf16 MFMA, forward and backward) draw from this one definition.
__device__ __forceinline__ uint32_t att_drop_q(uint32_t wh, int i, int jg) { return ((wh * 16u + (uint32_t)i) * 4u + (uint32_t)jg) * 2u; }
__device__ __forceinline__ void att_drop4(const DropCtx& d, uint32_t q2, float* m) {  // multipliers of keys 4 g .. 4 g + 3
  const uint32_t h0 = focal_hash24(q2 ^ d.key), h1 = focal_hash24((q2 + 1u) ^ d.key), t16 = d.thresh >> 8;
  m[0] = (h0 & 0xffffu) < t16 ? 0.0f : d.scale;
  m[1] = (h0 >> 16) < t16 ? 0.0f : d.scale;
  m[2] = (h1 & 0xffffu) < t16 ? 0.0f : d.scale;
  m[3] = (h1 >> 16) < t16 ? 0.0f : d.scale;
}
__device__ __forceinline__ float att_drop1(const DropCtx& d, uint32_t wh, int i, int j) {
  const uint32_t hh = focal_hash24((att_drop_q(wh, i, j >> 2) + (uint32_t)((j >> 1) & 1)) ^ d.key);
  return ((j & 1) ? (hh >> 16) : (hh & 0xffffu)) < (d.thresh >> 8) ? 0.0f : d.scale;
}

__device__ __forceinline__ int att_token(const AttnGeom& g, int win, int i, int* region) {
  const int b = win / g.nW, wl = win % g.nW;
  const int wy = wl / g.nWx, wx = wl % g.nWx;
  const int Y = wy * g.wh + i / g.ww, X = wx * g.ww + i % g.ww;  // coordinates in the rolled frame
  if (region) {
    const int rh = Y < g.H - g.wh ? 0 : (Y < g.H - g.sh ? 1 : 2);
    const int rw = X < g.W - g.ww ? 0 : (X < g.W - g.sw ? 1 : 2);
    *region = rh * 3 + rw;
  }
  int y = Y, x = X;
  if (g.shifted) {  // rolled[Y] = original[(Y + sh) mod H]   (torch.roll by -shift, SwinModules.py:307)
    y = (Y + g.sh) % g.H;
    x = (X + g.sw) % g.W;
  }
  return (b * g.H + y) * g.W + x;
}


// wqkv / bqkv non-NULL: `qkv` is the LayerNorm output a1 [M][C] and the kernels project q / k / v themselves (C == 64, 4 heads of 16)
int focal_attn_mfma_fwd(const AttnGeom& g, const bf16_t* qkv, const float* bias_table, bf16_t* out, const uint32_t* rng, uint32_t stream_id,
                        float p_attn, hipStream_t st, const bf16_t* wqkv = nullptr, const float* bqkv = nullptr);
int focal_attn_mfma_bwd(const AttnGeom& g, const bf16_t* qkv, const float* bias_table, const bf16_t* dout, bf16_t* dqkv, float* dbias_table,
                        const uint32_t* rng, uint32_t stream_id, float p_attn, hipStream_t st, const bf16_t* wqkv = nullptr,
                        const float* bqkv = nullptr, const bf16_t* wproj = nullptr);  // wproj (with wqkv): dout is the gradient of the proj OUTPUT
