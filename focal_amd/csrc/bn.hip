// BatchNorm2d (train: batch statistics, eval: running statistics) + GELU + Dropout2d + residual for the DeepSense
// ConvLayer2D stack (models/ConvModules.py:98-112, 203-204), on channel-last tokens z[rows][C] (C = 64):
//   forward   stats(z) -> y = resid + mask * gelu(gamma * (z - mean) * rstd + beta)       (+ a dtype copy for the next GEMM)
//   backward  da = g * mask * gelu'(.)  ->  s1 = sum da, s2 = sum da * zhat  (= dbeta, dgamma)
//             dz = gamma * rstd * (da - s1/n - zhat * s2/n)
// All streaming, HBM-bound; per-channel reductions: registers -> wave shuffles -> LDS rows -> one atomic per channel per workgroup.
#include <stdlib.h>
#include "common.hpp"

// Per-channel partial sums -> global: lanes that hold the same channels are folded with xor-shuffles, each wave parks one
// [2][C] row in LDS (plain stores: LDS float atomics serialise per lane), the 4 waves are summed and every channel leaves
// as one atomic per workgroup (16-wave workgroups, at most one per CU: the atomics onto one address are a serial chain).
// `red` is [waves][2][C] floats; a[k] belongs to channel c + k of the first half, b[k] of the second.
__device__ __forceinline__ void bn_commit_sums(float* red, float* __restrict__ sums, float (&a)[4], float (&b)[4], int C, int c,
                                               bool wait_performed = false) {
  const int lpr = C / 4, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int i = threadIdx.x; i < nw * 2 * C; i += blockDim.x) red[i] = 0.f;
  for (int o = lpr; o < 64; o <<= 1) {
#pragma unroll
    for (int k = 0; k < 4; ++k) { a[k] = xadd(a[k], o); b[k] = xadd(b[k], o); }
  }
  __syncthreads();
  if (lpr >= 64 || lane < lpr) {
    float* row = red + wave * 2 * C;
    *reinterpret_cast<float4*>(row + c) = make_float4(a[0], a[1], a[2], a[3]);
    *reinterpret_cast<float4*>(row + C + c) = make_float4(b[0], b[1], b[2], b[3]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
    float t = 0.f;
    for (int w = 0; w < nw; ++w) t += red[w * 2 * C + i];
    if (wait_performed) {
      const float old = __hip_atomic_fetch_add(sums + i, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("" ::"v"(old));  // keep the returning form: its result arriving means the add is done at L2
    } else {
      atomicAdd(sums + i, t);
    }
  }
}

// sums[0..C) = sum z, sums[C..2C) = sum z^2   (sums must be zeroed by the caller-side launcher)
// TRAIN mode (fin.mean_rstd != null): the LAST workgroup to commit its sums (arrival counter at sums[2C]) finalises the statistics
// -- one launch less per BatchNorm layer; nobody waits, so this is safe whatever else shares the chip.
struct BnFinalize { float* mean_rstd; float* run_mean; float* run_var; long stat_rows; float eps, momentum; };
__global__ __launch_bounds__(1024) void bn_partial_kernel(const float* __restrict__ z, float* __restrict__ sums, long rows, int C,
                                                          BnFinalize fin) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [4][2][C]
  // (blockIdx.y = statistic group, focal_bn_desc.groups: `rows` rows each, every per-call array one copy per group)
  z += (size_t)blockIdx.y * rows * C;
  sums += (size_t)blockIdx.y * (2 * C + 1);
  if (fin.mean_rstd) fin.mean_rstd += (size_t)blockIdx.y * 2 * C;
  if (fin.run_mean) { fin.run_mean += (size_t)blockIdx.y * C; fin.run_var += (size_t)blockIdx.y * C; }
  const int lpr = C / 4;                 // lanes per row
  const int rpb = blockDim.x / lpr;      // rows per block-iteration
  const int li = threadIdx.x % lpr, sub = threadIdx.x / lpr;
  float4 s = make_float4(0, 0, 0, 0), q = make_float4(0, 0, 0, 0);
  const long step = (long)gridDim.x * rpb;
  long r = (long)blockIdx.x * rpb + sub;
  for (; r + 3 * step < rows; r += 4 * step) {  // four independent 16-byte loads in flight per lane
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(z + (r + u * step) * C + li * 4);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w;
      q.x += v[u].x * v[u].x; q.y += v[u].y * v[u].y; q.z += v[u].z * v[u].z; q.w += v[u].w * v[u].w;
    }
  }
  for (; r < rows; r += step) {
    const float4 v = *reinterpret_cast<const float4*>(z + r * C + li * 4);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    q.x += v.x * v.x; q.y += v.y * v.y; q.z += v.z * v.z; q.w += v.w * v.w;
  }
  float a[4] = {s.x, s.y, s.z, s.w}, b[4] = {q.x, q.y, q.z, q.w};
  bn_commit_sums(red, sums, a, b, C, li * 4, fin.mean_rstd != nullptr);
  if (fin.mean_rstd == nullptr) return;
  __shared__ int is_last;
  // (no __threadfence(): a device-scope fence writes back / invalidates L2 on this part and cost 25 % of the DeepSense step.
  //  The sums were added with RETURNING atomics, i.e. they are performed at L2 before the barrier below lets thread 0 draw
  //  the ticket; the last workgroup reads them back with atomics as well.)
  __syncthreads();
  if (threadIdx.x == 0) is_last = atomicAdd(reinterpret_cast<unsigned int*>(sums + 2 * C), 1u) == gridDim.x - 1;
  __syncthreads();
  if (!is_last) return;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const float n = (float)fin.stat_rows;
    const float mean = atomicAdd(sums + c, 0.f) / n;  // (coherent read of the accumulated sums)
    float var = atomicAdd(sums + C + c, 0.f) / n - mean * mean;
    var = fmaxf(var, 0.f);
    fin.mean_rstd[c] = mean;
    fin.mean_rstd[C + c] = rsqrtf(var + fin.eps);
    if (fin.run_mean) {
      fin.run_mean[c] = (1.f - fin.momentum) * fin.run_mean[c] + fin.momentum * mean;
      fin.run_var[c] = (1.f - fin.momentum) * fin.run_var[c] + fin.momentum * var * (n / fmaxf(n - 1.f, 1.f));
    }
  }
}

// mean_rstd[0..C) = mean, [C..2C) = rstd; running stats: momentum update with the UNBIASED variance (torch semantics)
__global__ void bn_finalize_kernel(const float* __restrict__ sums, float* __restrict__ mean_rstd, float* __restrict__ run_mean,
                                   float* __restrict__ run_var, long rows, int C, float eps, float momentum) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float n = (float)rows;
  const float mean = sums[c] / n;
  float var = sums[C + c] / n - mean * mean;
  var = fmaxf(var, 0.f);
  mean_rstd[c] = mean;
  mean_rstd[C + c] = rsqrtf(var + eps);
  if (run_mean) {
    run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mean;
    run_var[c] = (1.f - momentum) * run_var[c] + momentum * var * (n / fmaxf(n - 1.f, 1.f));
  }
}
__global__ void bn_eval_stats_kernel(const float* __restrict__ run_mean, const float* __restrict__ run_var,
                                     float* __restrict__ mean_rstd, int C, float eps) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  mean_rstd[c] = run_mean[c];
  mean_rstd[C + c] = rsqrtf(run_var[c] + eps);
}

// The statistics of a convolution that left only its per-channel sums behind (focal_conv_fwd_bn with mean_rstd = NULL: 16 slots of {sum[C],
// sum of squares[C]} per group): every workgroup of the BatchNorm launch sums the slots itself (8 KB from L2) -- the arithmetic of the
// convolution kernels' last-arriver finalisation, bit for bit -- and the first one of a group publishes mean / rstd for the backward pass and
// updates the running buffers.  The convolution then ends on fire-and-forget atomics: no returning adds, no arrival ticket, no slot read-back
// (three dependent memory-side round trips, ~9 us per launch: tools/prof_conv.sh).
struct BnFromSums {
  const float* sums;   // NULL: mean_rstd is an input
  float* mean_rstd_out; float* run_mean; float* run_var;
  float n, eps, momentum;
};

template <typename TY>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const float* __restrict__ z, const float* __restrict__ mean_rstd,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const float* __restrict__ resid, float* __restrict__ y, TY* __restrict__ ya,
                                                         long rows, int C, int rows_per_sample, const uint32_t* rng,
                                                         uint32_t stream, float p, BnFromSums fs) {
  __shared__ float mr_lds[2 * 64];
  if (fs.sums != nullptr) {  // (C = 64: the launcher checks)
    const float* sums = fs.sums + (size_t)blockIdx.y * (FOCAL_BN_STAT_SLOTS * 2 * C + 1);
    if (threadIdx.x < C) {
      const int c = threadIdx.x;
      float sm = 0.f, sq = 0.f;
      for (int sl = 0; sl < FOCAL_BN_STAT_SLOTS; ++sl) { sm += sums[sl * 2 * C + c]; sq += sums[sl * 2 * C + C + c]; }
      const float mean = sm / fs.n;
      float var = sq / fs.n - mean * mean;
      var = fmaxf(var, 0.f);
      const float rstd = rsqrtf(var + fs.eps);
      mr_lds[c] = mean;
      mr_lds[C + c] = rstd;
      if (blockIdx.x == 0) {
        float* const mr = fs.mean_rstd_out + (size_t)blockIdx.y * 2 * C;
        mr[c] = mean;
        mr[C + c] = rstd;
        if (fs.run_mean) {
          float* const rm = fs.run_mean + (size_t)blockIdx.y * C;
          float* const rv = fs.run_var + (size_t)blockIdx.y * C;
          rm[c] = (1.f - fs.momentum) * rm[c] + fs.momentum * mean;
          rv[c] = (1.f - fs.momentum) * rv[c] + fs.momentum * var * (fs.n / fmaxf(fs.n - 1.f, 1.f));
        }
      }
    }
    __syncthreads();
  }
  const DropCtx dc = make_drop(rng, stream, p);
  const bool drop_on = p > 0.f;
  const long n4 = rows * C / 4;
  const int cshift = (C & (C - 1)) == 0 ? __builtin_ctz(C) : -1;  // power-of-two channel count: shift / mask, no 64-bit division
  {  // blockIdx.y = statistic group: its `rows` rows, its mean / rstd
    const size_t off = (size_t)blockIdx.y * rows * C;
    z += off; y += off;
    if (resid) resid += off;
    if (ya) ya += off;
    mean_rstd = fs.sums != nullptr ? mr_lds : mean_rstd + (size_t)blockIdx.y * 2 * C;
  }
  const long row0 = (long)blockIdx.y * rows;  // (Dropout2d samples are numbered through the whole tensor)
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n4; e += (long)gridDim.x * 256) {
    const long r = row0 + (cshift >= 0 ? (e * 4) >> cshift : (e * 4) / C);
    const int c = cshift >= 0 ? (int)((e * 4) & (C - 1)) : (int)((e * 4) % C);
    const float4 v = reinterpret_cast<const float4*>(z)[e];
    const float4 mu = *reinterpret_cast<const float4*>(mean_rstd + c), rs = *reinterpret_cast<const float4*>(mean_rstd + C + c);
    const float4 ga = *reinterpret_cast<const float4*>(gamma + c), be = *reinterpret_cast<const float4*>(beta + c);
    float o[4] = {(v.x - mu.x) * rs.x * ga.x + be.x, (v.y - mu.y) * rs.y * ga.y + be.y, (v.z - mu.z) * rs.z * ga.z + be.z,
                  (v.w - mu.w) * rs.w * ga.w + be.w};
    const uint32_t sample = drop_on ? ((uint32_t)r / (uint32_t)rows_per_sample) : 0u;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      o[k] = gelu_f(o[k]);
      if (drop_on) o[k] *= drop_mult(dc, sample * (uint32_t)C + (uint32_t)(c + k));  // Dropout2d: one draw per (sample, channel)
    }
    if (resid) {
      const float4 rr = reinterpret_cast<const float4*>(resid)[e];
      o[0] += rr.x; o[1] += rr.y; o[2] += rr.z; o[3] += rr.w;
    }
    reinterpret_cast<float4*>(y)[e] = make_float4(o[0], o[1], o[2], o[3]);
    if (ya) {
      if (sizeof(TY) == 4) reinterpret_cast<float4*>(ya)[e] = make_float4(o[0], o[1], o[2], o[3]);
      else {
        bf16x4 t;
        t[0] = (bf16_t)o[0]; t[1] = (bf16_t)o[1]; t[2] = (bf16_t)o[2]; t[3] = (bf16_t)o[3];
        reinterpret_cast<bf16x4*>(ya)[e] = t;
      }
    }
  }
}

// da (recomputed, never stored) and its two per-channel sums: sums[0..C) += sum da (= dbeta), sums[C..2C) += sum da*zhat (= dgamma)
__device__ __forceinline__ void bn_da4(const float4 v, const float4 g, const float* mean_rstd, const float* gamma, const float* beta,
                                       int c, int C, const DropCtx& dc, bool drop_on, uint32_t sample, float* da, float* zh) {
  const float vv[4] = {v.x, v.y, v.z, v.w}, gg[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    zh[k] = (vv[k] - mean_rstd[c + k]) * mean_rstd[C + c + k];
    const float pre = zh[k] * gamma[c + k] + beta[c + k];
    float m = 1.f;
    if (drop_on) m = drop_mult(dc, sample * (uint32_t)C + (uint32_t)(c + k));
    da[k] = gg[k] * m * gelu_grad_f(pre);
  }
}

__global__ __launch_bounds__(1024) void bn_bwd_reduce_kernel(const float* __restrict__ z, const float* __restrict__ g,
                                                            const float* __restrict__ mean_rstd, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ sums, long rows, int C,
                                                            int rows_per_sample, const uint32_t* rng, uint32_t stream, float p) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [4][2][C]
  const DropCtx dc = make_drop(rng, stream, p);
  const bool drop_on = p > 0.f;
  const int lpr = C / 4, rpb = blockDim.x / lpr;
  const int li = threadIdx.x % lpr, sub = threadIdx.x / lpr, c = li * 4;
  // blockIdx.y = statistic group
  z += (size_t)blockIdx.y * rows * C; g += (size_t)blockIdx.y * rows * C;
  mean_rstd += (size_t)blockIdx.y * 2 * C;
  sums += (size_t)blockIdx.y * (2 * C + 1);
  const long row0 = (long)blockIdx.y * rows;
  float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  for (long r = (long)blockIdx.x * rpb + sub; r < rows; r += (long)gridDim.x * rpb) {
    float da[4], zh[4];
    bn_da4(*reinterpret_cast<const float4*>(z + r * C + c), *reinterpret_cast<const float4*>(g + r * C + c), mean_rstd, gamma, beta, c, C,
           dc, drop_on, drop_on ? ((uint32_t)(row0 + r) / (uint32_t)rows_per_sample) : 0u, da, zh);
#pragma unroll
    for (int k = 0; k < 4; ++k) { s1[k] += da[k]; s2[k] += da[k] * zh[k]; }
  }
  bn_commit_sums(red, sums, s1, s2, C, c);
}

__global__ void bn_param_grad_kernel(const float* __restrict__ sums, float* __restrict__ dgamma, float* __restrict__ dbeta, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < C) { dbeta[i] += sums[i]; dgamma[i] += sums[C + i]; }
}

template <typename TD>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ z, const float* __restrict__ g,
                                                           const float* __restrict__ mean_rstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ sums,
                                                           TD* __restrict__ dz, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                           long rows, long stat_rows, int C, int rows_per_sample, const uint32_t* rng,
                                                           uint32_t stream, float p) {
  const DropCtx dc = make_drop(rng, stream, p);
  const bool drop_on = p > 0.f;
  const float inv_n = 1.0f / (float)stat_rows;
  if (blockIdx.x == 0 && blockIdx.y == 0 && dgamma) {  // the parameter gradients: the sums of every statistic group
    for (int i = threadIdx.x; i < C; i += 256) {
      float sb = 0.f, sg = 0.f;
      for (int gr = 0; gr < (int)gridDim.y; ++gr) { sb += sums[(size_t)gr * (2 * C + 1) + i]; sg += sums[(size_t)gr * (2 * C + 1) + C + i]; }
      dbeta[i] += sb; dgamma[i] += sg;
    }
  }
  {  // blockIdx.y = statistic group
    const size_t off = (size_t)blockIdx.y * rows * C;
    z += off; g += off; dz += off;
    mean_rstd += (size_t)blockIdx.y * 2 * C;
    sums += (size_t)blockIdx.y * (2 * C + 1);
  }
  const long row0 = (long)blockIdx.y * rows;
  const long n4 = rows * C / 4;
  const int cshift = (C & (C - 1)) == 0 ? __builtin_ctz(C) : -1;
  if (cshift >= 0 && C <= 1024) {
    // A power-of-two channel count divides the 1024 elements a workgroup covers per sweep: a thread's four channels never change, so
    // everything per channel -- mean, rstd, gamma, beta, the two sums -- is loaded ONCE (round 5: the kernel issued 32 one-dword loads of
    // them per 16-byte element group and ran at 2.2 TB/s against the forward kernel's 5.6).
    const int c = (threadIdx.x * 4) & (C - 1);
    float mu[4], rs[4], ga[4], be[4], m1[4], m2[4], gr[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      mu[k] = mean_rstd[c + k]; rs[k] = mean_rstd[C + c + k]; ga[k] = gamma[c + k]; be[k] = beta[c + k];
      m1[k] = sums[c + k] * inv_n; m2[k] = sums[C + c + k] * inv_n; gr[k] = ga[k] * rs[k];
    }
    const long stride = (long)gridDim.x * 256;
    for (long e0 = (long)blockIdx.x * 256 + threadIdx.x; e0 < n4; e0 += 2 * stride) {  // two element groups in flight per thread
      const long e1 = e0 + stride;
      const bool two = e1 < n4;
      const float4 zv0 = reinterpret_cast<const float4*>(z)[e0], gv0 = reinterpret_cast<const float4*>(g)[e0];
      const float4 zv1 = two ? reinterpret_cast<const float4*>(z)[e1] : zv0, gv1 = two ? reinterpret_cast<const float4*>(g)[e1] : gv0;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (u == 1 && !two) break;
        const long e = u ? e1 : e0;
        const float4 zv = u ? zv1 : zv0, gv = u ? gv1 : gv0;
        const float vv[4] = {zv.x, zv.y, zv.z, zv.w}, gg[4] = {gv.x, gv.y, gv.z, gv.w};
        const uint32_t sample = drop_on ? ((uint32_t)(row0 + ((e * 4) >> cshift)) / (uint32_t)rows_per_sample) : 0u;
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float zh = (vv[k] - mu[k]) * rs[k];
          const float pre = zh * ga[k] + be[k];
          float m = 1.f;
          if (drop_on) m = drop_mult(dc, sample * (uint32_t)C + (uint32_t)(c + k));
          const float da = gg[k] * m * gelu_grad_f(pre);
          o[k] = gr[k] * (da - m1[k] - zh * m2[k]);
        }
        if (sizeof(TD) == 4) reinterpret_cast<float4*>(dz)[e] = make_float4(o[0], o[1], o[2], o[3]);
        else {
          bf16x4 t;
          t[0] = (bf16_t)o[0]; t[1] = (bf16_t)o[1]; t[2] = (bf16_t)o[2]; t[3] = (bf16_t)o[3];
          reinterpret_cast<bf16x4*>(dz)[e] = t;
        }
      }
    }
    return;
  }
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n4; e += (long)gridDim.x * 256) {
    const long r = row0 + (cshift >= 0 ? (e * 4) >> cshift : (e * 4) / C);
    const int c = cshift >= 0 ? (int)((e * 4) & (C - 1)) : (int)((e * 4) % C);
    float da[4], zh[4], o[4];
    bn_da4(reinterpret_cast<const float4*>(z)[e], reinterpret_cast<const float4*>(g)[e], mean_rstd, gamma, beta, c, C, dc, drop_on,
           drop_on ? ((uint32_t)r / (uint32_t)rows_per_sample) : 0u, da, zh);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      o[k] = gamma[c + k] * mean_rstd[C + c + k] * (da[k] - sums[c + k] * inv_n - zh[k] * sums[C + c + k] * inv_n);
    if (sizeof(TD) == 4) reinterpret_cast<float4*>(dz)[e] = make_float4(o[0], o[1], o[2], o[3]);
    else {
      bf16x4 t;
      t[0] = (bf16_t)o[0]; t[1] = (bf16_t)o[1]; t[2] = (bf16_t)o[2]; t[3] = (bf16_t)o[3];
      reinterpret_cast<bf16x4*>(dz)[e] = t;
    }
  }
}

static int bn_check(const focal_bn_desc* d) {
  FOCAL_CHECK_ARG(d != nullptr, "bn: null descriptor");
  FOCAL_CHECK_ARG(d->dtype == FOCAL_F32 || d->dtype == FOCAL_BF16, "bn: bad dtype");
  FOCAL_CHECK_ARG(d->C >= 4 && d->C <= 1024 && d->C % 4 == 0 && 256 % (d->C / 4) == 0, "bn: unsupported channel count %d", d->C);
  FOCAL_CHECK_ARG(d->rows > 0 && d->rows_per_sample > 0 && d->rows % d->rows_per_sample == 0, "bn: rows %% rows_per_sample != 0");
  FOCAL_CHECK_ARG(d->groups >= 0 && d->groups <= 16, "bn: %d statistic groups (0 .. 16)", d->groups);
  if (d->groups > 1)
    FOCAL_CHECK_ARG(d->rows % d->groups == 0 && (d->rows / d->groups) % d->rows_per_sample == 0 && d->stat_rows <= 0,
                    "bn: %d statistic groups need rows (%d) to split into whole samples and stat_rows = 0 (one rank)", d->groups, d->rows);
  return FOCAL_OK;
}
static int bn_groups(const focal_bn_desc* d) { return d->groups > 1 ? d->groups : 1; }
static int stream_blocks(long rows, int C) {
  long b = (rows * C / 4 + 255) / 256;
  return (int)(b > 1024 ? 1024 : (b < 1 ? 1 : b));
}

extern "C" int focal_bn_stats(const focal_bn_desc* d, const float* z, float* scratch, float* mean_rstd, float* running_mean,
                              float* running_var, int training, void* stream) {
  if (int rc = bn_check(d)) return rc;
  const bool prezeroed = (training & FOCAL_BN_SCRATCH_ZEROED) != 0;
  training &= ~FOCAL_BN_SCRATCH_ZEROED;
  FOCAL_CHECK_ARG(training >= FOCAL_BN_EVAL && training <= FOCAL_BN_FINALIZE, "bn_stats: bad mode %d", training);
  hipStream_t st = (hipStream_t)stream;
  const int C = d->C, G = bn_groups(d);
  FOCAL_CHECK_ARG(G == 1 || training == FOCAL_BN_TRAIN, "bn_stats: statistic groups are a training-mode, one-rank feature (mode %d)", training);
  const long rows_g = d->rows / G;
  if (training == FOCAL_BN_EVAL) {
    FOCAL_CHECK_ARG(mean_rstd && running_mean && running_var, "bn_stats: null tensor");
    FOCAL_LAUNCH(bn_eval_stats_kernel, dim3(ceil_div(C, 64)), dim3(64), 0, st, running_mean, running_var, mean_rstd, C, d->eps);
    FOCAL_LAUNCH_CHECK();
    return FOCAL_OK;
  }
  FOCAL_CHECK_ARG(scratch, "bn_stats: null scratch");
  const long n_stat = d->stat_rows > 0 ? d->stat_rows : d->rows;
  if (training != FOCAL_BN_FINALIZE) {
    FOCAL_CHECK_ARG(z, "bn_stats: null tensor");
    if (!prezeroed) (void)hipMemsetAsync(scratch, 0, (size_t)G * (2 * C + 1) * sizeof(float), st);
    int blocks = ceil_div(rows_g * C / 4, 1024 * 2);
    if (blocks > 256 / G) blocks = 256 / G;
    BnFinalize fin = {nullptr, nullptr, nullptr, G > 1 ? rows_g : n_stat, d->eps, d->momentum};
    if (training == FOCAL_BN_TRAIN) {
      FOCAL_CHECK_ARG(mean_rstd && running_mean && running_var, "bn_stats: null tensor");
      fin.mean_rstd = mean_rstd; fin.run_mean = running_mean; fin.run_var = running_var;
    }
    FOCAL_LAUNCH(bn_partial_kernel, dim3(blocks, G), dim3(1024), 32 * C * sizeof(float), st, z, scratch, rows_g, C, fin);
  }
  if (training == FOCAL_BN_FINALIZE) {
    FOCAL_CHECK_ARG(mean_rstd && running_mean && running_var, "bn_stats: null tensor");
    FOCAL_LAUNCH(bn_finalize_kernel, dim3(ceil_div(C, 64)), dim3(64), 0, st, scratch, mean_rstd, running_mean, running_var, n_stat, C,
                       d->eps, d->momentum);
  }
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}

extern "C" int focal_bn_act_fwd(const focal_bn_desc* d, const float* z, const float* mean_rstd, const float* gamma, const float* beta,
                                const float* resid, float* y, void* y_cast, void* stream) {
  if (int rc = bn_check(d)) return rc;
  FOCAL_CHECK_ARG(z && mean_rstd && gamma && beta && y, "bn_act_fwd: null tensor");
  hipStream_t st = (hipStream_t)stream;
  const int G = bn_groups(d);
  const long rows_g = d->rows / G;
  const int blocks = ceil_div(stream_blocks(d->rows, d->C), G);
  BnFromSums fs;
  memset(&fs, 0, sizeof(fs));
  if (d->dtype == FOCAL_F32)
    FOCAL_LAUNCH((bn_act_fwd_kernel<float>), dim3(blocks, G), dim3(256), 0, st, z, mean_rstd, gamma, beta, resid, y, (float*)y_cast,
                       rows_g, d->C, d->rows_per_sample, d->rng, d->stream, d->p_drop, fs);
  else
    FOCAL_LAUNCH((bn_act_fwd_kernel<bf16_t>), dim3(blocks, G), dim3(256), 0, st, z, mean_rstd, gamma, beta, resid, y, (bf16_t*)y_cast,
                       rows_g, d->C, d->rows_per_sample, d->rng, d->stream, d->p_drop, fs);
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}

extern "C" int focal_bn_act_fwd_sums(const focal_bn_desc* d, const float* z, const float* sums, float* mean_rstd, float* running_mean,
                                     float* running_var, const float* gamma, const float* beta, const float* resid, float* y, void* y_cast,
                                     void* stream) {
  if (int rc = bn_check(d)) return rc;
  FOCAL_CHECK_ARG(z && sums && mean_rstd && gamma && beta && y, "bn_act_fwd_sums: null tensor");
  FOCAL_CHECK_ARG(d->C == 64 && (running_mean == nullptr) == (running_var == nullptr), "bn_act_fwd_sums: 64 channels (got %d), both running buffers or none", d->C);
  hipStream_t st = (hipStream_t)stream;
  const int G = bn_groups(d);
  FOCAL_CHECK_ARG(G == 1 || d->stat_rows <= 0, "bn_act_fwd_sums: statistic groups and stat_rows do not combine");
  const long rows_g = d->rows / G;
  const int blocks = ceil_div(stream_blocks(d->rows, d->C), G);
  BnFromSums fs;
  fs.sums = sums; fs.mean_rstd_out = mean_rstd; fs.run_mean = running_mean; fs.run_var = running_var;
  fs.n = (float)(G > 1 ? rows_g : (d->stat_rows > 0 ? d->stat_rows : d->rows));
  fs.eps = d->eps; fs.momentum = d->momentum;
  if (d->dtype == FOCAL_F32)
    FOCAL_LAUNCH((bn_act_fwd_kernel<float>), dim3(blocks, G), dim3(256), 0, st, z, nullptr, gamma, beta, resid, y, (float*)y_cast,
                       rows_g, d->C, d->rows_per_sample, d->rng, d->stream, d->p_drop, fs);
  else
    FOCAL_LAUNCH((bn_act_fwd_kernel<bf16_t>), dim3(blocks, G), dim3(256), 0, st, z, nullptr, gamma, beta, resid, y, (bf16_t*)y_cast,
                       rows_g, d->C, d->rows_per_sample, d->rng, d->stream, d->p_drop, fs);
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}

extern "C" int focal_bn_act_bwd(const focal_bn_desc* d, const float* z, const float* g, const float* mean_rstd, const float* gamma,
                                const float* beta, float* scratch, void* dz, float* dgamma, float* dbeta, int phase, void* stream) {
  if (int rc = bn_check(d)) return rc;
  const bool prezeroed = (phase & FOCAL_BN_SCRATCH_ZEROED) != 0;
  phase &= ~FOCAL_BN_SCRATCH_ZEROED;
  FOCAL_CHECK_ARG(phase == FOCAL_BN_TRAIN || phase == FOCAL_BN_PARTIAL || phase == FOCAL_BN_FINALIZE, "bn_act_bwd: bad phase %d", phase);
  FOCAL_CHECK_ARG(z && g && mean_rstd && gamma && beta && scratch, "bn_act_bwd: null tensor");
  hipStream_t st = (hipStream_t)stream;
  const int C = d->C, G = bn_groups(d);
  FOCAL_CHECK_ARG(G == 1 || phase == FOCAL_BN_TRAIN, "bn_act_bwd: statistic groups are a one-rank feature (phase %d)", phase);
  const long rows_g = d->rows / G;
  if (phase != FOCAL_BN_FINALIZE) {
    if (!prezeroed) (void)hipMemsetAsync(scratch, 0, (size_t)G * (2 * C + 1) * sizeof(float), st);
    int rb = ceil_div(rows_g * C / 4, 1024 * 2);
    if (rb > 256 / G) rb = 256 / G;
    FOCAL_LAUNCH(bn_bwd_reduce_kernel, dim3(rb, G), dim3(1024), 32 * C * sizeof(float), st, z, g, mean_rstd, gamma, beta, scratch,
                       rows_g, C, d->rows_per_sample, d->rng, d->stream, d->p_drop);
  }
  if (phase == FOCAL_BN_PARTIAL) {  // the parameter gradients are the LOCAL sums (the gradient all-reduce adds the other ranks')
    FOCAL_CHECK_ARG(dgamma && dbeta, "bn_act_bwd: null tensor");
    FOCAL_LAUNCH(bn_param_grad_kernel, dim3(ceil_div(C, 64)), dim3(64), 0, st, scratch, dgamma, dbeta, C);
  } else {
    FOCAL_CHECK_ARG(dz && (phase == FOCAL_BN_FINALIZE || (dgamma && dbeta)), "bn_act_bwd: null tensor");
    float* dgm = phase == FOCAL_BN_FINALIZE ? nullptr : dgamma;
    const long stat_rows = (phase == FOCAL_BN_FINALIZE && d->stat_rows > 0) ? d->stat_rows : rows_g;
    const int blocks = ceil_div(stream_blocks(d->rows, C), G);
    if (d->dtype == FOCAL_F32)
      FOCAL_LAUNCH((bn_bwd_apply_kernel<float>), dim3(blocks, G), dim3(256), 0, st, z, g, mean_rstd, gamma, beta, scratch, (float*)dz,
                         dgm, dbeta, rows_g, stat_rows, C, d->rows_per_sample, d->rng, d->stream, d->p_drop);
    else
      FOCAL_LAUNCH((bn_bwd_apply_kernel<bf16_t>), dim3(blocks, G), dim3(256), 0, st, z, g, mean_rstd, gamma, beta, scratch, (bf16_t*)dz,
                         dgm, dbeta, rows_g, stat_rows, C, d->rows_per_sample, d->rng, d->stream, d->p_drop);
  }
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}

// ------------------------------------------------------------------------------------------------ two views, one update
// The reference runs the two augmented views of a step through the backbone one after the other, so every BatchNorm's running buffers
// see view 1's batch statistics, then view 2's:  r <- (1 - m) ((1 - m) r + m s1) + m s2.  Passes that each update the buffers
// themselves must therefore run in that order; passes that only record their statistics (focal_bn_stats with momentum 1 into a
// per-pass sink) can run side by side, and this kernel applies the same two updates afterwards -- one launch for every running buffer
// of an encoder.
struct BnCombineTable { int n; float* run[FOCAL_BN_COMBINE_MAX]; const float* s1[FOCAL_BN_COMBINE_MAX]; const float* s2[FOCAL_BN_COMBINE_MAX]; };
__global__ void bn_running_combine_kernel(BnCombineTable t, int C, float m) {
  const int e = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const float r1 = (1.f - m) * t.run[e][c] + m * t.s1[e][c];
    t.run[e][c] = (1.f - m) * r1 + m * t.s2[e][c];
  }
}
extern "C" int focal_bn_running_combine(int n, float* const* running, const float* const* view1, const float* const* view2, int C,
                                        float momentum, void* stream) {
  FOCAL_CHECK_ARG(n >= 1 && n <= FOCAL_BN_COMBINE_MAX && running && view1 && view2 && C > 0, "bn_running_combine: 1 .. %d buffers of C > 0 values", FOCAL_BN_COMBINE_MAX);
  BnCombineTable t;
  t.n = n;
  for (int i = 0; i < n; ++i) {
    FOCAL_CHECK_ARG(running[i] && view1[i] && view2[i], "bn_running_combine: null buffer %d", i);
    t.run[i] = running[i]; t.s1[i] = view1[i]; t.s2[i] = view2[i];
  }
  FOCAL_LAUNCH(bn_running_combine_kernel, dim3(n), dim3(64), 0, (hipStream_t)stream, t, C, momentum);
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}
