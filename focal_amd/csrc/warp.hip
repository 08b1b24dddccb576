// TimeWarp / MagWarp view augmentations (reference: data_augmenter/TimeWarpAugmenter.py:18,44 and MagWarpAugmenter.py:18,44, which
// wrap tsai 0.3.7's TSTimeWarp / TSMagWarp): ONE smooth random curve per call over the flattened (interval x sample) axis of a
// [B, C, I, S] tensor, shared by batch and channels.
//   magnitude warp  y[r][n] = x[r][n] * mult[n]
//   time warp       y[r][n] = cubic-spline interpolant of x[r][.] at position pos[n] (monotone, pos[0] = 0, pos[L-1] = L-1)
// The host draws the curve (a dozen knots) and hands over per-position tables.  The time warp evaluates the interpolating cubic
// spline in its cardinal form: y[n] = sum_t w[n][t] x[k0[n] + t] with `taps` position-dependent weights (B-spline basis x the
// truncated recursive prefilter sqrt(3) (sqrt(3) - 2)^|j|, |j| <= 10: truncation error 5e-7; near the two ends the host puts the
// exact not-a-knot basis weights of an end window into the same table, see focal_amd/warp.py).
#include "common.hpp"

__global__ __launch_bounds__(256) void mag_warp_kernel(int rows, int L, const float* __restrict__ x, const float* __restrict__ mult,
                                                       float* __restrict__ y) {
  const long total = (long)rows * L;
  for (long e = ((long)blockIdx.x * 256 + threadIdx.x) * 4; e < total; e += (long)gridDim.x * 1024) {
    const int n = e % L;  // L % 4 == 0 (checked by the launcher): the 4 elements share a row
    const float4 v = *reinterpret_cast<const float4*>(x + e);
    const float4 m = *reinterpret_cast<const float4*>(mult + n);
    *reinterpret_cast<float4*>(y + e) = make_float4(v.x * m.x, v.y * m.y, v.z * m.z, v.w * m.w);
  }
}

template <int TAPS, int RPT>
__global__ __launch_bounds__(256) void time_warp_kernel(int rows, int L, const float* __restrict__ x, const int* __restrict__ k0,
                                                        const float* __restrict__ w, float* __restrict__ y) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= L) return;
  float wt[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; t += 4) {
    const float4 v = *reinterpret_cast<const float4*>(w + (long)n * TAPS + t);
    wt[t] = v.x; wt[t + 1] = v.y; wt[t + 2] = v.z; wt[t + 3] = v.w;
  }
  const int base = k0[n];
  int idx[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t) idx[t] = min(max(base + t, 0), L - 1);
  const int r0 = blockIdx.y * RPT;
#pragma unroll 2
  for (int r = r0; r < min(rows, r0 + RPT); ++r) {
    const float* src = x + (long)r * L;
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < TAPS; ++t) acc += wt[t] * src[idx[t]];
    y[(long)r * L + n] = acc;
  }
}

extern "C" int focal_warp_fwd(int rows, int L, const float* x, const float* mult, const int* k0, const float* w, int taps, float* y,
                              void* stream) {
  FOCAL_CHECK_ARG(rows > 0 && L > 0 && x && y, "warp: bad arguments");
  FOCAL_CHECK_ARG((mult != nullptr) != (k0 != nullptr && w != nullptr), "warp: give either the multiplier table or the (k0, w) tables");
  hipStream_t st = (hipStream_t)stream;
  if (mult) {
    FOCAL_CHECK_ARG(L % 4 == 0, "warp: row length %d must be a multiple of 4", L);
    long blocks = ((long)rows * L / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    FOCAL_LAUNCH(mag_warp_kernel, dim3((int)blocks), dim3(256), 0, st, rows, L, x, mult, y);
  } else {
    FOCAL_CHECK_ARG(taps == 24, "warp: built for 24 taps (got %d)", taps);
    constexpr int RPT = 16;
    FOCAL_LAUNCH((time_warp_kernel<24, RPT>), dim3((L + 255) / 256, (rows + RPT - 1) / RPT), dim3(256), 0, st, rows, L, x, k0, w, y);
  }
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}

// Mixup / CutMix of the supervised `fixed` augmentation pipeline (reference: data_augmenter/MixupAugmenter.py ->
// input_utils/mixup_utils.py:252-281, mode "random_batch"): ONE permutation of the batch, shared by every (location, modality);
//   mixup : y[b] = lam x[b] + (1 - lam) x[perm[b]]
//   cutmix: y[b][:, yl:yh, xl:xh] = x[perm[b]][:, yl:yh, xl:xh], elsewhere y[b] = x[b]     (one box per tensor)
__global__ __launch_bounds__(256) void mixup_kernel(int B, long per_sample, int I, int S, const float* __restrict__ x, const int* __restrict__ perm,
                                                    float lam, int cut, int yl, int yh, int xl, int xh, float* __restrict__ y) {
  const long total = (long)B * per_sample;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int b = e / per_sample;
    const long r = e - (long)b * per_sample;
    const float a = x[e], o = x[(long)perm[b] * per_sample + r];
    if (cut) {
      const int s = r % S, i = (r / S) % I;
      y[e] = (i >= yl && i < yh && s >= xl && s < xh) ? o : a;
    } else {
      y[e] = lam * a + (1.0f - lam) * o;
    }
  }
}

extern "C" int focal_mixup_fwd(int B, int C_, int I, int S, const float* x, const int* perm, float lam, int cut, int yl, int yh, int xl,
                               int xh, float* y, void* stream) {
  FOCAL_CHECK_ARG(B > 0 && C_ > 0 && I > 0 && S > 0 && x && perm && y && x != y, "mixup: bad arguments (in-place is not supported)");
  const long per = (long)C_ * I * S;
  long blocks = ((long)B * per + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  FOCAL_LAUNCH(mixup_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, B, per, I, S, x, perm, lam, cut, yl, yh, xl, xh, y);
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}
