// LayerNorm forward / backward for the SW_Transformer residual stream (nn.LayerNorm, eps 1e-5, biased variance),
// optionally fused with PatchMerging's 2x2 gather (models/SwinModules.py:388-399).  HBM-bound streaming kernels:
// one float4 per lane, LPR = C/4 lanes per row (up to a full wave), wave-shuffle reductions, no LDS in forward.
#include <stdlib.h>
#include "common.hpp"

// Row addressing.  Plain: row r -> x + r*C.  Gather: row r = (b, y2, x2) of the merged grid; segment s = c / Cin
// comes from token (2*y2 + (s & 1), 2*x2 + (s >> 1)) of the [B, H, W, Cin] input (cat order x00, x10, x01, x11).
struct RowMap {
  int gather, H, W, Cin, C;
  __device__ __forceinline__ long offset(int r, int c) const {
    if (!gather) return (long)r * C + c;
    const int W2 = W >> 1, H2 = H >> 1;
    const int x2 = r % W2, t = r / W2, y2 = t % H2, b = t / H2;
    const int s = c / Cin, cc = c - s * Cin;
    const int y = 2 * y2 + (s & 1), x = 2 * x2 + (s >> 1);
    return (((long)b * H + y) * W + x) * Cin + cc;
  }
};

__device__ __forceinline__ float group_sum(float v, int lpr) {
  if (lpr >= 16) {  // the 16-lane part with DPP row operations (no LDS round trips), whole rows beyond that by shuffle
    v = row16_sum(v);
    for (int o = 16; o < lpr; o <<= 1) v = xadd(v, o);
    return v;
  }
  for (int o = lpr >> 1; o > 0; o >>= 1) v = xadd(v, o);
  return v;
}

template <typename TY, int NV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, TY* __restrict__ y,
                                                     float* __restrict__ stats, int rows, int C, int lpr, float eps,
                                                     RowMap map) {
  const int lane = threadIdx.x & 63;
  const int rpw = 64 / lpr;                       // rows per wave
  const int sub = lane / lpr, li = lane % lpr;    // row slot in the wave, lane within the row
  const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * 4;
  for (int r0 = wave_global * rpw; r0 < rows; r0 += nwaves * rpw) {
    const int r = r0 + sub;
    const bool ok = r < rows;
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int c = (k * lpr + li) * 4;
      v[k] = ok ? *reinterpret_cast<const float4*>(x + map.offset(r, c)) : make_float4(0, 0, 0, 0);
      s += v[k].x + v[k].y + v[k].z + v[k].w;
    }
    const float mean = group_sum(s, lpr) / C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const float a = v[k].x - mean, b = v[k].y - mean, c2 = v[k].z - mean, d = v[k].w - mean;
      q += a * a + b * b + c2 * c2 + d * d;
    }
    const float rstd = rsqrtf(group_sum(q, lpr) / C + eps);
    if (!ok) continue;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int c = (k * lpr + li) * 4;
      const float4 g = *reinterpret_cast<const float4*>(gamma + c);
      const float4 b = *reinterpret_cast<const float4*>(beta + c);
      TY* dst = y + (long)r * C + c;
      const float o0 = (v[k].x - mean) * rstd * g.x + b.x, o1 = (v[k].y - mean) * rstd * g.y + b.y;
      const float o2 = (v[k].z - mean) * rstd * g.z + b.z, o3 = (v[k].w - mean) * rstd * g.w + b.w;
      if (sizeof(TY) == 4) {
        *reinterpret_cast<float4*>(dst) = make_float4(o0, o1, o2, o3);
      } else {
        bf16x4 o;
        o[0] = (bf16_t)o0; o[1] = (bf16_t)o1; o[2] = (bf16_t)o2; o[3] = (bf16_t)o3;
        *reinterpret_cast<bf16x4*>(dst) = o;
      }
    }
    if (li == 0) {
      stats[2 * (long)r] = mean;
      stats[2 * (long)r + 1] = rstd;
    }
  }
}


// The residual-stream gradient g that this kernel completes is consumed next by the two GEMMs of a residual branch as
// g * (dropout x drop-path mask of that branch's output).  Emitting that product here, once, in the operand dtype, lets
// both GEMMs run as plain kernels: regenerating the mask in their operand loaders cost more VALU than the GEMM itself
// (it is recomputed for every column tile) and made them read fp32.  Index convention = the forward epilogue's:
// element (row, col) of the [M, ncols] token tensor -> row * ncols + col; sample = row / rows_per_sample.
struct DropMask {
  DropCtx e, p;
  int on_e, on_p, rps, ncols;
  uint32_t rps_magic;  // ceil(2^32 / rps): row / rps == mulhi(row, magic) while row * rps < 2^32 (checked by the launchers); rps == 1: unused
};
__device__ __forceinline__ DropMask make_mask(const focal_drop_desc& d, int ncols) {
  DropMask m;
  m.on_e = d.p_elem > 0.f;
  m.on_p = d.p_path > 0.f;
  m.rps = d.rows_per_sample > 0 ? d.rows_per_sample : 1;
  m.ncols = ncols;
  m.rps_magic = 0xFFFFFFFFu / (uint32_t)m.rps + 1u;
  m.e = make_drop(d.rng, d.stream_elem, d.p_elem);
  m.p = make_drop(d.rng, d.stream_path, d.p_path);
  return m;
}
// (`row` = off / ncols where the caller knows it: the 64-bit division by a run-time width was ~40 instructions per 16-byte store)
template <typename TY> __device__ __forceinline__ void store_masked4_row(TY* out, long off, uint32_t row, float4 v, const DropMask& m) {
  float a[4] = {v.x, v.y, v.z, v.w};
  const float pm = m.on_p ? drop_mult(m.p, m.rps > 1 ? __umulhi(row, m.rps_magic) : row) : 1.0f;
#pragma unroll
  for (int k = 0; k < 4; ++k) a[k] *= m.on_e ? pm * drop_mult(m.e, (uint32_t)off + k) : pm;
  if (sizeof(TY) == 4) *reinterpret_cast<float4*>(out + off) = make_float4(a[0], a[1], a[2], a[3]);
  else {
    bf16x4 t;
    t[0] = (bf16_t)a[0]; t[1] = (bf16_t)a[1]; t[2] = (bf16_t)a[2]; t[3] = (bf16_t)a[3];
    *reinterpret_cast<bf16x4*>(out + off) = t;
  }
}
template <typename TY> __device__ __forceinline__ void store_masked4(TY* out, long off, float4 v, const DropMask& m) {
  store_masked4_row(out, off, (uint32_t)(off / m.ncols), v, m);
}

template <typename TY> __device__ __forceinline__ float4 load_dy4(const TY* p);
template <> __device__ __forceinline__ float4 load_dy4<float>(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <> __device__ __forceinline__ float4 load_dy4<bf16_t>(const bf16_t* p) {
  bf16x4 t = *reinterpret_cast<const bf16x4*>(p);
  return make_float4((float)t[0], (float)t[1], (float)t[2], (float)t[3]);
}

// dx = rstd * (g - mean_c(g) - xhat * mean_c(g * xhat)),  g = dy * gamma;  dgamma += sum_r dy * xhat; dbeta += sum_r dy.
// Each lane keeps private column partials across its grid-stride rows; they are combined through LDS once per
// workgroup and leave as one atomic per column per workgroup.
template <typename TY, int NV>
__global__ __launch_bounds__(1024) void ln_bwd_kernel(const TY* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ stats, const float* __restrict__ gamma,
                                                     float* __restrict__ dx, int accumulate, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, int rows, int C, int lpr, RowMap map,
                                                     TY* __restrict__ dxm, focal_drop_desc dd, int mcols) {
  const DropMask mk = make_mask(dd, mcols);
  extern __shared__ __attribute__((aligned(16))) float part[];  // [waves per block][2][C]
  const int lane = threadIdx.x & 63;
  const int rpw = 64 / lpr;
  const int sub = lane / lpr, li = lane % lpr;
  const int wpb = blockDim.x >> 6;
  const int wave_global = blockIdx.x * wpb + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * wpb;
  float4 pg[NV], pb[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) pg[k] = pb[k] = make_float4(0, 0, 0, 0);
  for (int r0 = wave_global * rpw; r0 < rows; r0 += nwaves * rpw) {
    const int r = r0 + sub;
    const bool ok = r < rows;
    const float mean = ok ? stats[2 * (long)r] : 0.f, rstd = ok ? stats[2 * (long)r + 1] : 0.f;
    float4 xh[NV], g[NV], old[NV];
    float s1 = 0.f, s2 = 0.f;
    // the residual-stream gradient this row's result is added to: requested together with x and dy (it does not depend on
    // the row reductions; loading it after them exposed a second full memory latency per row)
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int c = (k * lpr + li) * 4;
      old[k] = (ok && accumulate) ? *reinterpret_cast<const float4*>(dx + map.offset(r, c)) : make_float4(0, 0, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int c = (k * lpr + li) * 4;
      float4 xv = ok ? *reinterpret_cast<const float4*>(x + map.offset(r, c)) : make_float4(0, 0, 0, 0);
      float4 d = ok ? load_dy4<TY>(dy + (long)r * C + c) : make_float4(0, 0, 0, 0);
      const float4 gm = *reinterpret_cast<const float4*>(gamma + c);
      xh[k] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
      g[k] = make_float4(d.x * gm.x, d.y * gm.y, d.z * gm.z, d.w * gm.w);
      s1 += g[k].x + g[k].y + g[k].z + g[k].w;
      s2 += g[k].x * xh[k].x + g[k].y * xh[k].y + g[k].z * xh[k].z + g[k].w * xh[k].w;
      pg[k].x += d.x * xh[k].x; pg[k].y += d.y * xh[k].y; pg[k].z += d.z * xh[k].z; pg[k].w += d.w * xh[k].w;
      pb[k].x += d.x; pb[k].y += d.y; pb[k].z += d.z; pb[k].w += d.w;
    }
    const float m1 = group_sum(s1, lpr) / C, m2 = group_sum(s2, lpr) / C;
    if (!ok) continue;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int c = (k * lpr + li) * 4;
      float* dst = dx + map.offset(r, c);
      float4 o = make_float4(rstd * (g[k].x - m1 - xh[k].x * m2), rstd * (g[k].y - m1 - xh[k].y * m2),
                             rstd * (g[k].z - m1 - xh[k].z * m2), rstd * (g[k].w - m1 - xh[k].w * m2));
      o.x += old[k].x; o.y += old[k].y; o.z += old[k].z; o.w += old[k].w;
      *reinterpret_cast<float4*>(dst) = o;
      if (dxm) {  // the token row of this element: the LayerNorm row itself unless the rows are gathered 2 x 2 neighbourhoods (PatchMerging)
        const long off = dst - dx;
        store_masked4_row(dxm, off, map.gather ? (uint32_t)(off / mk.ncols) : (uint32_t)r, o, mk);
      }
    }
  }
  // Column partials: fold the rows that share a wave with xor-shuffles, park one [2][C] row per wave in LDS (plain
  // stores -- LDS float atomics from 16 waves onto 2C addresses cost ~10 us here), sum the waves, then one global atomic
  // per column per workgroup.  16 waves per workgroup and at most one workgroup per CU keeps the number of contended
  // global atomics at gridDim x 2C.
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    for (int o = lpr; o < 64; o <<= 1) {
      pg[k].x = xadd(pg[k].x, o); pg[k].y = xadd(pg[k].y, o);
      pg[k].z = xadd(pg[k].z, o); pg[k].w = xadd(pg[k].w, o);
      pb[k].x = xadd(pb[k].x, o); pb[k].y = xadd(pb[k].y, o);
      pb[k].z = xadd(pb[k].z, o); pb[k].w = xadd(pb[k].w, o);
    }
    if (sub == 0) {
      float* row = part + (threadIdx.x >> 6) * 2 * C;
      const int c = (k * lpr + li) * 4;
      *reinterpret_cast<float4*>(row + c) = pg[k];
      *reinterpret_cast<float4*>(row + C + c) = pb[k];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
    float acc = 0.f;
    for (int w = 0; w < wpb; ++w) acc += part[w * 2 * C + i];
#if defined(LN_LAB_NOATOMIC)   // lab: plain stores into a per-workgroup row (wrong sums) -- what the contended atomics cost
    (i < C ? dgamma + i : dbeta + (i - C))[0] = acc;
#elif defined(LN_LAB_NOEPI)
    if (acc == 12345.678f) dgamma[i] = acc;
#else
    atomicAdd(i < C ? dgamma + i : dbeta + (i - C), acc);
#endif
  }
}

static int ln_geometry(const focal_ln_desc* d, int* lpr, int* nv, RowMap* map) {
  FOCAL_CHECK_ARG(d != nullptr, "layernorm: null descriptor");
  FOCAL_CHECK_ARG(d->dtype == FOCAL_F32 || d->dtype == FOCAL_BF16, "layernorm: bad dtype");
  const int C = d->C;
  FOCAL_CHECK_ARG(C >= 16 && C <= 1024 && (C & (C - 1)) == 0, "layernorm: C=%d must be a power of two in [16, 1024]", C);
  *lpr = C / 4 < 64 ? C / 4 : 64;
  *nv = C / (4 * *lpr);
  map->gather = d->gather; map->H = d->H; map->W = d->W; map->Cin = d->Cin; map->C = C;
  if (d->gather) {
    FOCAL_CHECK_ARG(d->Cin * 4 == C && d->H % 2 == 0 && d->W % 2 == 0 && d->rows == d->B * (d->H / 2) * (d->W / 2) && d->Cin % 4 == 0,
                    "layernorm: inconsistent gather geometry");
  }
  return FOCAL_OK;
}

extern "C" int focal_layernorm_fwd(const focal_ln_desc* d, const float* x, const float* gamma, const float* beta, void* y,
                                   float* stats, void* stream) {
  int lpr, nv;
  RowMap map;
  if (int rc = ln_geometry(d, &lpr, &nv, &map)) return rc;
  FOCAL_CHECK_ARG(x && gamma && beta && y && stats, "layernorm_fwd: null tensor");
  const int rpw = 64 / lpr;
  int blocks = ceil_div(d->rows, rpw * 4);
  if (blocks > 2048) blocks = 2048;
  hipStream_t st = (hipStream_t)stream;
#define LN_FWD(TY, NV) FOCAL_LAUNCH((ln_fwd_kernel<TY, NV>), dim3(blocks), dim3(256), 0, st, x, gamma, beta, (TY*)y, stats, d->rows, d->C, lpr, d->eps, map)
  if (d->dtype == FOCAL_F32) { if (nv == 1) LN_FWD(float, 1); else if (nv == 2) LN_FWD(float, 2); else LN_FWD(float, 4); }
  else { if (nv == 1) LN_FWD(bf16_t, 1); else if (nv == 2) LN_FWD(bf16_t, 2); else LN_FWD(bf16_t, 4); }
#undef LN_FWD
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}

extern "C" int focal_layernorm_bwd(const focal_ln_desc* d, const void* dy, const float* x, const float* stats,
                                   const float* gamma, float* dx, int accumulate_dx, float* dgamma, float* dbeta,
                                   void* dx_masked, const focal_drop_desc* mask, void* stream) {
  int lpr, nv;
  RowMap map;
  if (int rc = ln_geometry(d, &lpr, &nv, &map)) return rc;
  FOCAL_CHECK_ARG(dy && x && stats && gamma && dx && dgamma && dbeta, "layernorm_bwd: null tensor");
  const int rpw = 64 / lpr;
  // 16 waves per workgroup up to 512 channels (the PatchMerging norms: 4 waves per CU left every row's memory latency exposed -- rows
  // 18 432 x 512: 66 -> 39 us cold, step +0.9 %, profiles/r3_ln_bwd_wide_tpb.txt); 1024 channels (4 float4 per lane and array) stay at 4
  const int wide_tpb = 1024;  // (16 waves per workgroup at 512 channels: 66 -> 39 us, profiles/r3_ln_bwd_wide_tpb.txt)
  const int tpb = d->C >= 1024 ? 256 : (d->C >= 512 ? wide_tpb : 1024), maxb = 256;
  focal_drop_desc dd;
  memset(&dd, 0, sizeof(dd));
  if (mask) dd = *mask;
  FOCAL_CHECK_ARG(!mask || dx_masked, "layernorm_bwd: mask without dx_masked");
  {  // (the masked copy divides token rows by rows_per_sample with a 32-bit magic multiply)
    const long token_rows = d->gather ? (long)d->B * d->H * d->W : (long)d->rows;
    FOCAL_CHECK_ARG(!mask || mask->rows_per_sample <= 1 || token_rows * (long)mask->rows_per_sample < (1L << 32),
                    "layernorm_bwd: %ld token rows x %d rows per sample exceed the mask's 32-bit row arithmetic", token_rows, mask->rows_per_sample);
  }
  const int mcols = d->gather ? d->Cin : d->C;  // columns of the token tensor dx lives in
  int blocks = ceil_div(d->rows, rpw * (tpb / 64) * 2);  // ~2 rows per wave
  if (blocks > maxb) blocks = maxb;
  hipStream_t st = (hipStream_t)stream;
  const size_t sm = (size_t)(tpb / 64) * 2 * d->C * sizeof(float);
#define LN_BWD(TY, NV) FOCAL_LAUNCH((ln_bwd_kernel<TY, NV>), dim3(blocks), dim3(tpb), sm, st, (const TY*)dy, x, stats, gamma, dx, accumulate_dx, dgamma, dbeta, d->rows, d->C, lpr, map, (TY*)dx_masked, dd, mcols)
  if (d->dtype == FOCAL_F32) { if (nv == 1) LN_BWD(float, 1); else if (nv == 2) LN_BWD(float, 2); else LN_BWD(float, 4); }
  else { if (nv == 1) LN_BWD(bf16_t, 1); else if (nv == 2) LN_BWD(bf16_t, 2); else LN_BWD(bf16_t, 4); }
#undef LN_BWD
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}

// out[r][c] = dtype(g[r][c] * mask(r, c)): the same product for a gradient that no LayerNorm backward completes (the one
// entering the last block of an encoder).
template <typename TY>
__global__ __launch_bounds__(256) void mask_cast_kernel(const float* __restrict__ g, TY* __restrict__ out, long n4, focal_drop_desc dd, int ncols) {
  const DropMask mk = make_mask(dd, ncols);
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n4; e += (long)gridDim.x * 256)
    store_masked4(out, e * 4, reinterpret_cast<const float4*>(g)[e], mk);
}

extern "C" int focal_mask_cast(int dtype, int rows, int C, const float* g, const focal_drop_desc* mask, void* out, void* stream) {
  FOCAL_CHECK_ARG(dtype == FOCAL_F32 || dtype == FOCAL_BF16, "mask_cast: bad dtype");
  FOCAL_CHECK_ARG(rows > 0 && C > 0 && C % 4 == 0 && g && out, "mask_cast: bad arguments");
  FOCAL_CHECK_ARG(!mask || mask->rows_per_sample <= 1 || (long)rows * mask->rows_per_sample < (1L << 32),
                  "mask_cast: %d rows x %d rows per sample exceed the mask's 32-bit row arithmetic", rows, mask->rows_per_sample);
  focal_drop_desc dd;
  memset(&dd, 0, sizeof(dd));
  if (mask) dd = *mask;
  const long n4 = (long)rows * C / 4;
  long blocks = (n4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == FOCAL_F32) FOCAL_LAUNCH((mask_cast_kernel<float>), dim3(blocks), dim3(256), 0, st, g, (float*)out, n4, dd, C);
  else FOCAL_LAUNCH((mask_cast_kernel<bf16_t>), dim3(blocks), dim3(256), 0, st, g, (bf16_t*)out, n4, dd, C);
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}
