// Windowed multi-head self-attention (W-MSA / SW-MSA) between the qkv and proj Linears of a Swin block.
// The cyclic shift, window partition and window reverse are pure index arithmetic here: a workgroup gathers the
// q|k|v rows of a few windows from their ORIGINAL token positions into LDS (fp32), one thread owns one
// (window, head, query row), and results are scattered straight back to original positions.
// Tiles are 9x9xhead_dim: far too small for MFMA (1.1 % of the model's FLOPs), so this is an LDS + VALU kernel
// whose job is to move qkv once through HBM.
#include "attn_geom.hpp"

template <typename T> __device__ __forceinline__ void load_vec_f32(const T* p, float* f);
template <> __device__ __forceinline__ void load_vec_f32<float>(const float* p, float* f) {
  const float4 t = *reinterpret_cast<const float4*>(p);
  f[0] = t.x; f[1] = t.y; f[2] = t.z; f[3] = t.w;
}
template <> __device__ __forceinline__ void load_vec_f32<bf16_t>(const bf16_t* p, float* f) {
  const bf16x4 t = *reinterpret_cast<const bf16x4*>(p);
  f[0] = (float)t[0]; f[1] = (float)t[1]; f[2] = (float)t[2]; f[3] = (float)t[3];
}
template <typename T> __device__ __forceinline__ void store_vec(T* p, const float* f);
template <> __device__ __forceinline__ void store_vec<float>(float* p, const float* f) {
  *reinterpret_cast<float4*>(p) = make_float4(f[0], f[1], f[2], f[3]);
}
template <> __device__ __forceinline__ void store_vec<bf16_t>(bf16_t* p, const float* f) {
  bf16x4 o;
  o[0] = (bf16_t)f[0]; o[1] = (bf16_t)f[1]; o[2] = (bf16_t)f[2]; o[3] = (bf16_t)f[3];
  *reinterpret_cast<bf16x4*>(p) = o;
}

// scores + softmax for one (window, head, row); q already scaled.  Returns probabilities in p[0..N).
template <int HD, int NT>
__device__ __forceinline__ void att_row_probs(const AttnGeom& g, const float* q, const float* lds_win, int pitch, int koff,
                                              int h, int i, const float* bias_table, const int* regions, float* p) {
  float mx = -3.0e38f;
  const int iy = i / g.ww, ix = i % g.ww;
  _Pragma("unroll") for (int j = 0; j < NT; ++j) if (j < g.N) {
    const float* kj = lds_win + j * pitch + koff + h * HD;
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < HD; d += 4) {
      const float4 kv = *reinterpret_cast<const float4*>(kj + d);
      s += q[d] * kv.x + q[d + 1] * kv.y + q[d + 2] * kv.z + q[d + 3] * kv.w;
    }
    const int rel = (iy - j / g.ww + g.wh - 1) * (2 * g.ww - 1) + (ix - j % g.ww + g.ww - 1);
    s += bias_table[rel * g.heads + h];
    if (g.shifted && regions[i] != regions[j]) s += -100.0f;  // SwinModules.py:287
    p[j] = s;
    mx = fmaxf(mx, s);
  }
  float sum = 0.f;
  _Pragma("unroll") for (int j = 0; j < NT; ++j) if (j < g.N) {
    p[j] = __expf(p[j] - mx);
    sum += p[j];
  }
  const float inv = 1.0f / sum;
  _Pragma("unroll") for (int j = 0; j < NT; ++j) p[j] *= inv;
}

template <typename T, int HD, int NT>
__global__ __launch_bounds__(256) void window_attn_fwd_kernel(const T* __restrict__ qkv, const float* __restrict__ bias_table,
                                                              T* __restrict__ out, AttnGeom g, int wpb, int total_windows,
                                                              const uint32_t* rng, uint32_t stream, float p_attn) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int C = g.C, pitch = 3 * C + 4;
  float* lds = smem;                                          // [wpb][N][pitch]
  int* regs = reinterpret_cast<int*>(smem + wpb * g.N * pitch);  // [wpb][N] region ids
  int* toks = regs + wpb * g.N;                               // [wpb][N] token index
  const int tid = threadIdx.x;
  const DropCtx dc = make_drop(rng, stream, p_attn);
  const bool drop_on = p_attn > 0.f;
  const int ngroups = (total_windows + wpb - 1) / wpb;
  for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int win0 = grp * wpb;
    const int nwin = min(wpb, total_windows - win0);
    for (int t = tid; t < nwin * g.N; t += blockDim.x) {
      int region;
      toks[t] = att_token(g, win0 + t / g.N, t % g.N, &region);
      regs[t] = region;
    }
    __syncthreads();
    // stage q|k|v rows (vector of 4 elements per step), q pre-scaled (SwinModules.py:130)
    const int vec_per_tok = 3 * C / 4;
    for (int v = tid; v < nwin * g.N * vec_per_tok; v += blockDim.x) {
      const int t = v / vec_per_tok, c = (v % vec_per_tok) * 4;
      float f[4];
      load_vec_f32<T>(qkv + (long)toks[t] * 3 * C + c, f);
      if (c < C) { f[0] *= g.scale; f[1] *= g.scale; f[2] *= g.scale; f[3] *= g.scale; }
      *reinterpret_cast<float4*>(lds + t * pitch + c) = make_float4(f[0], f[1], f[2], f[3]);
    }
    __syncthreads();
    const int item = tid;
    const int per_win = g.heads * g.N;
    if (item < nwin * per_win) {
      const int wl = item / per_win, h = (item % per_win) / g.N, i = item % g.N;
      const float* lw = lds + wl * g.N * pitch;
      float q[HD];
#pragma unroll
      for (int d = 0; d < HD; d += 4) {
        const float4 t4 = *reinterpret_cast<const float4*>(lw + i * pitch + h * HD + d);
        q[d] = t4.x; q[d + 1] = t4.y; q[d + 2] = t4.z; q[d + 3] = t4.w;
      }
      float p[NT];
      att_row_probs<HD, NT>(g, q, lw, pitch, C, h, i, bias_table, regs + wl * g.N, p);
      if (drop_on) {
        const uint32_t wh = (uint32_t)(win0 + wl) * g.heads + h;
        _Pragma("unroll") for (int j = 0; j < NT; ++j) if (j < g.N) p[j] *= att_drop1(dc, wh, i, j);
      }
      float o[HD];
#pragma unroll
      for (int d = 0; d < HD; ++d) o[d] = 0.f;
      _Pragma("unroll") for (int j = 0; j < NT; ++j) if (j < g.N) {
        const float* vj = lw + j * pitch + 2 * C + h * HD;
#pragma unroll
        for (int d = 0; d < HD; d += 4) {
          const float4 vv = *reinterpret_cast<const float4*>(vj + d);
          o[d] += p[j] * vv.x; o[d + 1] += p[j] * vv.y; o[d + 2] += p[j] * vv.z; o[d + 3] += p[j] * vv.w;
        }
      }
      T* dst = out + (long)toks[wl * g.N + i] * C + h * HD;
#pragma unroll
      for (int d = 0; d < HD; d += 4) store_vec<T>(dst + d, o + d);
    }
    __syncthreads();
  }
}

template <typename T, int HD, int NT>
__global__ __launch_bounds__(256) void window_attn_bwd_kernel(const T* __restrict__ qkv, const float* __restrict__ bias_table,
                                                              const T* __restrict__ dout, T* __restrict__ dqkv,
                                                              float* __restrict__ dbias_table, AttnGeom g, int wpb,
                                                              int total_windows, const uint32_t* rng, uint32_t stream,
                                                              float p_attn) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int C = g.C, pitch = 4 * C + 4;  // q | k | v | dO per token
  const int NN = g.N * g.N;
  const int table = (2 * g.wh - 1) * (2 * g.ww - 1) * g.heads;
  float* lds = smem;                                   // [wpb][N][pitch]
  float* Pd = lds + wpb * g.N * pitch;                 // [wpb][heads][N][N]  dropped probabilities
  float* dS = Pd + wpb * g.heads * NN;                 // [wpb][heads][N][N]
  float* dbacc = dS + wpb * g.heads * NN;              // [table]
  int* regs = reinterpret_cast<int*>(dbacc + table);   // [wpb][N]
  int* toks = regs + wpb * g.N;
  const int tid = threadIdx.x;
  for (int t = tid; t < table; t += blockDim.x) dbacc[t] = 0.f;
  const DropCtx dc = make_drop(rng, stream, p_attn);
  const bool drop_on = p_attn > 0.f;
  const int ngroups = (total_windows + wpb - 1) / wpb;
  const int per_win = g.heads * g.N;
  for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int win0 = grp * wpb;
    const int nwin = min(wpb, total_windows - win0);
    for (int t = tid; t < nwin * g.N; t += blockDim.x) {
      int region;
      toks[t] = att_token(g, win0 + t / g.N, t % g.N, &region);
      regs[t] = region;
    }
    __syncthreads();
    const int vec_per_tok = 4 * C / 4;
    for (int v = tid; v < nwin * g.N * vec_per_tok; v += blockDim.x) {
      const int t = v / vec_per_tok, c = (v % vec_per_tok) * 4;
      float f[4];
      if (c < 3 * C) {
        load_vec_f32<T>(qkv + (long)toks[t] * 3 * C + c, f);
        if (c < C) { f[0] *= g.scale; f[1] *= g.scale; f[2] *= g.scale; f[3] *= g.scale; }
      } else {
        load_vec_f32<T>(dout + (long)toks[t] * C + (c - 3 * C), f);
      }
      *reinterpret_cast<float4*>(lds + t * pitch + c) = make_float4(f[0], f[1], f[2], f[3]);
    }
    __syncthreads();
    const bool active = tid < nwin * per_win;
    const int wl = tid / per_win, h = (tid % per_win) / g.N, i = tid % g.N;
    const float* lw = lds + wl * g.N * pitch;
    if (active) {
      float q[HD], go[HD];
#pragma unroll
      for (int d = 0; d < HD; d += 4) {
        const float4 a = *reinterpret_cast<const float4*>(lw + i * pitch + h * HD + d);
        const float4 b = *reinterpret_cast<const float4*>(lw + i * pitch + 3 * C + h * HD + d);
        q[d] = a.x; q[d + 1] = a.y; q[d + 2] = a.z; q[d + 3] = a.w;
        go[d] = b.x; go[d + 1] = b.y; go[d + 2] = b.z; go[d + 3] = b.w;
      }
      float p[NT], dp[NT];
      att_row_probs<HD, NT>(g, q, lw, pitch, C, h, i, bias_table, regs + wl * g.N, p);
      const uint32_t drop_wh = (uint32_t)(win0 + wl) * g.heads + h;
      float dot = 0.f;
      _Pragma("unroll") for (int j = 0; j < NT; ++j) if (j < g.N) {
        const float* vj = lw + j * pitch + 2 * C + h * HD;
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < HD; d += 4) {
          const float4 vv = *reinterpret_cast<const float4*>(vj + d);
          s += go[d] * vv.x + go[d + 1] * vv.y + go[d + 2] * vv.z + go[d + 3] * vv.w;
        }
        const float dm = drop_on ? att_drop1(dc, drop_wh, i, j) : 1.0f;
        dp[j] = s * dm;                                             // dL/dP (through the dropout mask)
        Pd[((wl * g.heads + h) * g.N + i) * g.N + j] = p[j] * dm;   // what multiplied V in forward
        dot += p[j] * dp[j];
      }
      float dq[HD];
#pragma unroll
      for (int d = 0; d < HD; ++d) dq[d] = 0.f;
      _Pragma("unroll") for (int j = 0; j < NT; ++j) if (j < g.N) {
        const float ds = p[j] * (dp[j] - dot);  // softmax backward
        dS[((wl * g.heads + h) * g.N + i) * g.N + j] = ds;
        const float* kj = lw + j * pitch + C + h * HD;
#pragma unroll
        for (int d = 0; d < HD; d += 4) {
          const float4 kv = *reinterpret_cast<const float4*>(kj + d);
          dq[d] += ds * kv.x; dq[d + 1] += ds * kv.y; dq[d + 2] += ds * kv.z; dq[d + 3] += ds * kv.w;
        }
      }
      T* dst = dqkv + (long)toks[wl * g.N + i] * 3 * C + h * HD;
#pragma unroll
      for (int d = 0; d < HD; d += 4) {
        float t4[4] = {dq[d] * g.scale, dq[d + 1] * g.scale, dq[d + 2] * g.scale, dq[d + 3] * g.scale};
        store_vec<T>(dst + d, t4);
      }
    }
    __syncthreads();
    // relative-position-bias gradient: sum dS over this group's windows first, then ONE LDS add per (head, i, j)
    for (int t = tid; t < g.heads * NN; t += blockDim.x) {
      const int hh = t / NN, ij = t % NN, ii = ij / g.N, jj = ij % g.N;
      float sacc = 0.f;
      for (int w = 0; w < nwin; ++w) sacc += dS[(w * g.heads + hh) * NN + ij];
      const int rel = (ii / g.ww - jj / g.ww + g.wh - 1) * (2 * g.ww - 1) + (ii % g.ww - jj % g.ww + g.ww - 1);
      atomicAdd(&dbacc[rel * g.heads + hh], sacc);
    }
    if (active) {
      // this thread now owns key/value row j = i:  dK_j = sum_i dS[i][j] q'_i,  dV_j = sum_i Pd[i][j] dO_i
      const int j = i;
      float dk[HD], dv[HD];
#pragma unroll
      for (int d = 0; d < HD; ++d) dk[d] = dv[d] = 0.f;
      _Pragma("unroll") for (int r = 0; r < NT; ++r) if (r < g.N) {
        const float ds = dS[((wl * g.heads + h) * g.N + r) * g.N + j];
        const float pd = Pd[((wl * g.heads + h) * g.N + r) * g.N + j];
        const float* qr = lw + r * pitch + h * HD;
        const float* gr = lw + r * pitch + 3 * C + h * HD;
#pragma unroll
        for (int d = 0; d < HD; d += 4) {
          const float4 a = *reinterpret_cast<const float4*>(qr + d);
          const float4 b = *reinterpret_cast<const float4*>(gr + d);
          dk[d] += ds * a.x; dk[d + 1] += ds * a.y; dk[d + 2] += ds * a.z; dk[d + 3] += ds * a.w;
          dv[d] += pd * b.x; dv[d + 1] += pd * b.y; dv[d + 2] += pd * b.z; dv[d + 3] += pd * b.w;
        }
      }
      T* dst = dqkv + (long)toks[wl * g.N + j] * 3 * C + h * HD;
#pragma unroll
      for (int d = 0; d < HD; d += 4) {
        store_vec<T>(dst + C + d, dk + d);
        store_vec<T>(dst + 2 * C + d, dv + d);
      }
    }
    __syncthreads();
  }
  for (int t = tid; t < table; t += blockDim.x) atomicAdd(dbias_table + t, dbacc[t]);
}

static int attn_geometry(const focal_attn_desc* d, AttnGeom* g) {
  FOCAL_CHECK_ARG(d != nullptr, "window_attn: null descriptor");
  FOCAL_CHECK_ARG(d->dtype == FOCAL_F32 || d->dtype == FOCAL_BF16, "window_attn: bad dtype");
  FOCAL_CHECK_ARG(d->heads > 0 && d->C % d->heads == 0, "window_attn: C %% heads != 0");
  FOCAL_CHECK_ARG(d->wh > 0 && d->ww > 0 && d->H % d->wh == 0 && d->W % d->ww == 0, "window_attn: windows must tile the grid");
  FOCAL_CHECK_ARG(d->wh * d->ww <= ATT_NMAX, "window_attn: window of %d tokens exceeds %d", d->wh * d->ww, ATT_NMAX);
  g->B = d->B; g->H = d->H; g->W = d->W; g->C = d->C; g->heads = d->heads; g->hd = d->C / d->heads;
  g->wh = d->wh; g->ww = d->ww; g->sh = d->sh; g->sw = d->sw; g->N = d->wh * d->ww;
  g->nWx = d->W / d->ww; g->nW = (d->H / d->wh) * g->nWx;
  g->shifted = (d->sh > 0 && d->sw > 0) ? 1 : 0;  // `min(shift_size) > 0`, SwinModules.py:262,305
  auto magic = [](int dv) { return (uint32_t)(((1ull << 32) + dv - 1) / dv); };
  g->m_heads = magic(g->heads); g->m_nW = magic(g->nW); g->m_nWx = magic(g->nWx);
  FOCAL_CHECK_ARG((double)d->B * g->nW * g->heads * (double)(g->heads > g->nW ? g->heads : g->nW) < 4.0e9,
                  "window_attn: too many (window, head) items for the 32-bit index arithmetic");
  g->scale = 1.0f / sqrtf((float)g->hd);
  FOCAL_CHECK_ARG(g->hd == 16 || g->hd == 32 || g->hd == 64, "window_attn: head_dim %d not in {16, 32, 64}", g->hd);
  FOCAL_CHECK_ARG(g->heads * g->N <= 256, "window_attn: heads * window tokens > 256");
  return FOCAL_OK;
}

// LDS budget per workgroup of the exact-fp32 kernels: small enough that several workgroups share a CU and overlap their gather /
// compute / scatter phases
static size_t att_lds_budget(bool bwd) { return (size_t)(bwd ? 40 : 32) * 1024; }

extern "C" int focal_window_attn_fwd(const focal_attn_desc* d, const void* qkv, const float* bias_table, void* out,
                                     void* stream) {
  AttnGeom g;
  if (int rc = attn_geometry(d, &g)) return rc;
  FOCAL_CHECK_ARG(qkv && bias_table && out, "window_attn_fwd: null tensor");
  if (d->dtype == FOCAL_BF16)  // matrix-core path (attn_mfma.hip); fp32 stays on the exact VALU kernel
    return focal_attn_mfma_fwd(g, (const bf16_t*)qkv, bias_table, (bf16_t*)out, d->rng, d->stream, d->p_attn, (hipStream_t)stream);
  const size_t per_win = (size_t)g.N * (3 * g.C + 4) * 4 + 2 * g.N * 4;
  int wpb = 256 / (g.heads * g.N);
  const size_t budget = att_lds_budget(false) > per_win ? att_lds_budget(false) : (per_win <= 60 * 1024 ? per_win : 0);
  if ((size_t)wpb * per_win > budget) wpb = (int)(budget / per_win);
  FOCAL_CHECK_ARG(wpb >= 1, "window_attn_fwd: one window does not fit in LDS");
  const int total = g.B * g.nW;
  int threads = ((wpb * g.heads * g.N + 63) / 64) * 64;
  int blocks = ceil_div(total, wpb);
  if (blocks > 4096) blocks = 4096;
  const size_t sm = wpb * per_win;
  hipStream_t st = (hipStream_t)stream;
#define ATT_FWD_(T, HD, NT) FOCAL_LAUNCH((window_attn_fwd_kernel<T, HD, NT>), dim3(blocks), dim3(threads), sm, st, (const T*)qkv, bias_table, (T*)out, g, wpb, total, d->rng, d->stream, d->p_attn)
#define ATT_FWD(T, HD) do { if (g.N == 9) ATT_FWD_(T, HD, 9); else ATT_FWD_(T, HD, ATT_NMAX); } while (0)
  if (g.hd == 16) ATT_FWD(float, 16); else if (g.hd == 32) ATT_FWD(float, 32); else ATT_FWD(float, 64);
#undef ATT_FWD
#undef ATT_FWD_
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}

// ---- q / k / v projected inside the attention kernels (attn_mfma.hip: qkv_project)
extern "C" int focal_window_attn_qkv_supported(int dtype, int C, int heads, int window_tokens) {
  return dtype == FOCAL_BF16 && C == 64 && heads == 4 && window_tokens >= 1 && window_tokens <= ATT_NMAX;
}

extern "C" int focal_window_attn_qkv_fwd(const focal_attn_desc* d, const void* a1, const void* wqkv, const float* bqkv, const float* bias_table,
                                         void* out, void* stream) {
  AttnGeom g;
  if (int rc = attn_geometry(d, &g)) return rc;
  FOCAL_CHECK_ARG(a1 && wqkv && bqkv && bias_table && out, "window_attn_qkv_fwd: null tensor");
  FOCAL_CHECK_ARG(focal_window_attn_qkv_supported(d->dtype, g.C, g.heads, g.N), "window_attn_qkv_fwd: built for bf16, C = 64, 4 heads (got dtype %d, C = %d, heads = %d)",
                  d->dtype, g.C, g.heads);
  return focal_attn_mfma_fwd(g, (const bf16_t*)a1, bias_table, (bf16_t*)out, d->rng, d->stream, d->p_attn, (hipStream_t)stream, (const bf16_t*)wqkv, bqkv);
}

extern "C" int focal_window_attn_qkv_bwd(const focal_attn_desc* d, const void* a1, const void* wqkv, const float* bqkv, const float* bias_table,
                                         const void* dout, const void* wproj, void* dqkv, float* dbias_table, void* stream) {
  AttnGeom g;
  if (int rc = attn_geometry(d, &g)) return rc;
  FOCAL_CHECK_ARG(a1 && wqkv && bqkv && bias_table && dout && dqkv && dbias_table, "window_attn_qkv_bwd: null tensor");
  FOCAL_CHECK_ARG(focal_window_attn_qkv_supported(d->dtype, g.C, g.heads, g.N), "window_attn_qkv_bwd: built for bf16, C = 64, 4 heads (got dtype %d, C = %d, heads = %d)",
                  d->dtype, g.C, g.heads);
  FOCAL_CHECK_ARG((2 * g.wh - 1) * (2 * g.ww - 1) * g.heads <= 256, "window_attn_qkv_bwd: bias table too large");
  return focal_attn_mfma_bwd(g, (const bf16_t*)a1, bias_table, (const bf16_t*)dout, (bf16_t*)dqkv, dbias_table, d->rng, d->stream, d->p_attn,
                             (hipStream_t)stream, (const bf16_t*)wqkv, bqkv, (const bf16_t*)wproj);
}


extern "C" int focal_window_attn_bwd(const focal_attn_desc* d, const void* qkv, const float* bias_table, const void* dout,
                                     void* dqkv, float* dbias_table, void* stream) {
  AttnGeom g;
  if (int rc = attn_geometry(d, &g)) return rc;
  FOCAL_CHECK_ARG(qkv && bias_table && dout && dqkv && dbias_table, "window_attn_bwd: null tensor");
  if (d->dtype == FOCAL_BF16) {
    FOCAL_CHECK_ARG((2 * g.wh - 1) * (2 * g.ww - 1) * g.heads <= 256, "window_attn_bwd: bias table too large");
    return focal_attn_mfma_bwd(g, (const bf16_t*)qkv, bias_table, (const bf16_t*)dout, (bf16_t*)dqkv, dbias_table, d->rng, d->stream,
                               d->p_attn, (hipStream_t)stream);
  }
  const int table = (2 * g.wh - 1) * (2 * g.ww - 1) * g.heads;
  const size_t per_win = (size_t)g.N * (4 * g.C + 4) * 4 + 2 * (size_t)g.heads * g.N * g.N * 4 + 2 * g.N * 4;
  const size_t fixed = (size_t)table * 4;
  int wpb = 256 / (g.heads * g.N);
  const size_t budget = att_lds_budget(true) > per_win + fixed ? att_lds_budget(true) : (per_win + fixed <= 60 * 1024 ? per_win + fixed : 0);
  if ((size_t)wpb * per_win + fixed > budget) wpb = (int)((budget - fixed) / per_win);
  FOCAL_CHECK_ARG(wpb >= 1, "window_attn_bwd: one window does not fit in LDS");
  const int total = g.B * g.nW;
  int threads = ((wpb * g.heads * g.N + 63) / 64) * 64;
  int blocks = ceil_div(total, wpb);
  if (blocks > 1024) blocks = 1024;  // bounds the atomic fan-in on the 25 x heads bias-table gradient
  const size_t sm = wpb * per_win + fixed;
  hipStream_t st = (hipStream_t)stream;
#define ATT_BWD_(T, HD, NT) FOCAL_LAUNCH((window_attn_bwd_kernel<T, HD, NT>), dim3(blocks), dim3(threads), sm, st, (const T*)qkv, bias_table, (const T*)dout, (T*)dqkv, dbias_table, g, wpb, total, d->rng, d->stream, d->p_attn)
#define ATT_BWD(T, HD) do { if (g.N == 9) ATT_BWD_(T, HD, 9); else ATT_BWD_(T, HD, ATT_NMAX); } while (0)
  if (g.hd == 16) ATT_BWD(float, 16); else if (g.hd == 32) ATT_BWD(float, 32); else ATT_BWD(float, 64);
#undef ATT_BWD
#undef ATT_BWD_
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}
