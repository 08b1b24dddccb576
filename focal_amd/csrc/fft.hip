// Real-input full-spectrum DFT along the last axis, packed as [B, 2C, I, n] (Re / Im channel pairs), fp32.
// n = n1 * n2 four-step DFT held entirely in LDS by one workgroup per row:
//   X[k1 + n1*k2] = sum_{m2} W_{n2}^{m2 k2} * ( W_n^{m2 k1} * sum_{m1} x[n2*m1 + m2] W_{n1}^{m1 k1} )
// (m = n2*m1 + m2, k = k1 + n1*k2).  For the MOD audio rows n = 1600 = 40 x 40; n2 = 1 degenerates to the direct
// DFT used for the 20-point seismic rows.  The twiddle table holds {cos, -sin}(2 pi j / n), j < n, computed in
// fp64 on the host.
#include <stdlib.h>
#include "common.hpp"

// View augmentations that commute with (or fold into) the transform, applied while the rows are staged / stored instead of as
// separate passes over the window (the reference data_augmenter package): scaling / negation (x * a), horizontal flip (intervals and samples
// reversed), interval permutation, and the frequency-domain phase shift (every bin rotated by one angle).
// plan != NULL (round 5): the values come from that device record instead -- focal_view_draw wrote it earlier in the same stream, so
// a captured step draws a fresh view on every replay -- and when the record says the view is warped the rows are read from x_alt.
struct AugParams {
  float scale, pc, ps;
  int flip, use_perm;
  int perm[FOCAL_AUG_MAX_INTERVALS];
  const focal_view_plan* plan;
  const float* x_alt;
};
static AugParams aug_identity() {
  AugParams a;
  memset(&a, 0, sizeof(a));
  a.scale = 1.f; a.pc = 1.f;
  return a;
}
// The augmentation a kernel applies, resolved once per workgroup: from the kernel arguments, or from the device plan.  The interval
// permutation goes to LDS (s_perm, FOCAL_AUG_MAX_INTERVALS ints; the caller's next __syncthreads() publishes it).
struct AugLive {
  float scale, pc, ps;
  int flip, use_perm;
  const float* x;
};
__device__ __forceinline__ AugLive aug_resolve(const AugParams& a, const float* x, int* s_perm, int tid) {
  AugLive v{a.scale, a.pc, a.ps, a.flip, a.use_perm, x};
  if (a.plan != nullptr) {  // (uniform)
    const focal_view_plan* pl = a.plan;
    v.scale = pl->aug.scale; v.pc = pl->aug.phase_cos; v.ps = pl->aug.phase_sin;
    v.flip = pl->aug.flip != 0; v.use_perm = pl->aug.use_perm != 0;
    if (focal_plan_warp(pl) != 0) v.x = a.x_alt;
    if (tid < FOCAL_AUG_MAX_INTERVALS) s_perm[tid] = pl->aug.perm[tid];
  } else if (tid < FOCAL_AUG_MAX_INTERVALS) {
    s_perm[tid] = a.perm[tid];
  }
  return v;
}
// source row of output row `row` = (bc, i)
__device__ __forceinline__ int aug_src_row(const AugLive& a, const int* s_perm, int row, int I) {
  const int i = row % I, bc = row / I;
  int j = a.use_perm ? s_perm[i] : i;
  if (a.flip) j = I - 1 - j;
  return bc * I + j;
}

__global__ __launch_bounds__(256) void fft_realpack_kernel(const float* __restrict__ x_arg, const float* __restrict__ tw,
                                                           float* __restrict__ out, focal_fft_desc d, int rows, AugParams aug_arg) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ int s_perm[FOCAL_AUG_MAX_INTERVALS];
  const AugLive aug = aug_resolve(aug_arg, x_arg, s_perm, threadIdx.x);
  const float* x = aug.x;
  const int n = d.n, n1 = d.n1, n2 = d.n2;
  float* xs = smem;            // [n]
  float* yr = smem + n;        // [n2][n1]  stage-1 output, real
  float* yi = yr + n;          //           imag
  float* twc = yi + n;         // [n] cos
  float* tws = twc + n;        // [n] -sin
  const int tid = threadIdx.x;
  for (int i = tid; i < n; i += 256) {
    twc[i] = tw[2 * i];
    tws[i] = tw[2 * i + 1];
  }
  for (int row = blockIdx.x; row < rows; row += gridDim.x) {
    __syncthreads();
    const float* xr = x + (long)aug_src_row(aug, s_perm, row, d.I) * n;
    for (int i = tid; i < n; i += 256) xs[i] = aug.scale * xr[aug.flip ? n - 1 - i : i];
    __syncthreads();
    // stage 1: for each m2, n1-point DFT over m1, then twiddle W_n^{m2 k1}
    for (int o = tid; o < n; o += 256) {
      const int m2 = o / n1, k1 = o % n1;
      float re = 0.f, im = 0.f;
      int ph = 0;  // (m1 * k1 mod n1) * n2 indexes W_{n1} inside the n-point table
      for (int m1 = 0; m1 < n1; ++m1) {
        const float v = xs[n2 * m1 + m2];
        re += v * twc[ph * n2];
        im += v * tws[ph * n2];
        ph += k1;
        if (ph >= n1) ph -= n1;
      }
      const int t = (m2 * k1) % n;
      const float c = twc[t], s = tws[t];
      yr[o] = re * c - im * s;
      yi[o] = re * s + im * c;
    }
    __syncthreads();
    // stage 2: for each k1, n2-point DFT over m2 -> X[k1 + n1*k2]
    const int ci = row % d.I, bc = row / d.I;  // row = (b*C + c)*I + i
    float* ore = out + ((long)(2 * bc) * d.I + ci) * n;
    float* oim = out + ((long)(2 * bc + 1) * d.I + ci) * n;
    for (int o = tid; o < n; o += 256) {
      const int k = o, k1 = k % n1, k2 = k / n1;
      float re = 0.f, im = 0.f;
      int ph = 0;  // (m2 * k2 mod n2) * n1 indexes W_{n2}
      for (int m2 = 0; m2 < n2; ++m2) {
        const float a = yr[m2 * n1 + k1], b = yi[m2 * n1 + k1];
        const float c = twc[ph * n1], s = tws[ph * n1];
        re += a * c - b * s;
        im += a * s + b * c;
        ph += k2;
        if (ph >= n2) ph -= n2;
      }
      ore[k] = re * aug.pc - im * aug.ps;
      oim[k] = re * aug.ps + im * aug.pc;
    }
  }
}

// ---- matrix-core form of the same four-step DFT (n1, n2 <= 48, multiples of 8): both stages are small real GEMMs, so
// they run on the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32) instead of 6 LDS reads per complex MAC on the VALU path above.
// Two rows per workgroup iteration (2*n2 and 2*n1 are multiples of the 16-row MFMA tile):
//   stage 1   Y[(row, m2)][k1] = sum_m1 x[row][n2*m1 + m2] * W_n1[m1][k1]         A straight from global, K = n1
//   twiddle   Y *= W_n^(m2 k1), in registers; Y -> LDS as Yr / Yi [row][m2][k1]
//   stage 2   X[(row, k1)][k2] = sum_m2 Y[row][m2][k1] * W_n2[m2][k2] (complex)    K = 2*n2 over (Yr | Yi)
// Each wave owns whole 16-row tiles and walks the three 16-column pairs (re tile, im tile), so a lane always holds the
// real and imaginary part of the same element.  Output rows leave as 16-byte stores (4 consecutive k1 per lane).
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f4 mfma4(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// N1 / N2 are compile-time: the index arithmetic below is full of divisions by them (runtime divisors cost ~40
// instructions each and dominated the first version of this kernel).
template <int N1, int N2>
__global__ __launch_bounds__(256) void fft_realpack_mfma_kernel(const float* __restrict__ x_arg, const float* __restrict__ tw,
                                                                float* __restrict__ out, focal_fft_desc d, int rows, AugParams aug_arg) {
  constexpr int P = 48;  // padded tile pitch (3 x 16)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ int s_perm[FOCAL_AUG_MAX_INTERVALS];
  const AugLive aug = aug_resolve(aug_arg, x_arg, s_perm, threadIdx.x);
  const float* x = aug.x;
  constexpr int n = N1 * N2, n1 = N1, n2 = N2;
  float* twc = smem;                 // [n]  cos(2 pi j / n)
  float* tws = twc + n;              // [n] -sin
  float* w1c = tws + n;              // [n1][P]  W_n1[m1][k1]  (zero beyond k1 >= n1)
  float* w1s = w1c + n1 * P;
  constexpr bool same = n1 == n2;    // square factorisation (MOD audio: 40 x 40): one table serves both stages
  float* w2c = same ? w1c : w1s + n1 * P;  // [n2][P]  W_n2[m2][k2]
  float* w2s = same ? w1s : w2c + n2 * P;
  float* yr = (same ? w1s : w2s) + n2 * P;  // [2][n2][P]
  float* yi = yr + 2 * n2 * P;
  float* xs = yi + 2 * n2 * P;       // [2][n]  the two input rows of this pass
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lj = lane & 15, lg = lane >> 4;
  for (int i = tid; i < n; i += 256) { twc[i] = tw[2 * i]; tws[i] = tw[2 * i + 1]; }
  for (int i = tid; i < n1 * P; i += 256) {
    const int m = i / P, k = i % P;
    const int t = ((m * k) % n1) * n2;
    w1c[i] = k < n1 ? tw[2 * t] : 0.f;
    w1s[i] = k < n1 ? tw[2 * t + 1] : 0.f;
  }
  for (int i = tid; i < (same ? 0 : n2 * P); i += 256) {
    const int m = i / P, k = i % P;
    const int t = ((m * k) % n2) * n1;
    w2c[i] = k < n2 ? tw[2 * t] : 0.f;
    w2s[i] = k < n2 ? tw[2 * t + 1] : 0.f;
  }
  constexpr int mt1 = 2 * n2 / 16, mt2 = 2 * n1 / 16;  // 16-row tiles of stage 1 / stage 2
  // the next pass's rows are fetched (coalesced, 16 B per lane) while stage 2 of the current pass runs
  constexpr int XV = (2 * n / 4 + 255) / 256;  // float4 per thread
  float4 xn[XV];
  auto fetch = [&](int pair) {
#pragma unroll
    for (int i = 0; i < XV; ++i) {
      const int e = tid + 256 * i;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < 2 * n / 4 && pair < rows / 2) {
        const int rl = e / (n / 4), q = e - rl * (n / 4);
        const float4* src = reinterpret_cast<const float4*>(x + (long)aug_src_row(aug, s_perm, pair * 2 + rl, d.I) * n);
        if (aug.flip) {
          const float4 t = src[n / 4 - 1 - q];
          v = make_float4(t.w, t.z, t.y, t.x);
        } else {
          v = src[q];
        }
        v.x *= aug.scale; v.y *= aug.scale; v.z *= aug.scale; v.w *= aug.scale;
      }
      xn[i] = v;
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int i = 0; i < XV; ++i) {
      const int e = tid + 256 * i;
      if (e < 2 * n / 4) reinterpret_cast<float4*>(xs)[e] = xn[i];
    }
  };
  __syncthreads();  // s_perm
  fetch(blockIdx.x);
  stash();
  for (int pair = blockIdx.x; pair < rows / 2; pair += gridDim.x) {
    __syncthreads();  // tables and xs ready; previous pass done with yr / yi
    const float* x0 = xs;
    // ---- stage 1
    constexpr int np1 = (n1 + 15) / 16, np2 = (n2 + 15) / 16;  // 16-column (re, im) tile pairs
    for (int u = wave; u < mt1 * np1; u += 4) {  // unit = (16-row tile, column pair)
      const int mt = u / np1, p = u - mt * np1;
      const int f = 16 * mt + lj, row = f / n2, m2 = f - row * n2;  // this lane's A row
      const float* xa = x0 + row * n + m2;
      float a[12];
#pragma unroll
      for (int ks = 0; ks < 12; ++ks) {
        const int m1 = 4 * ks + lg;
        a[ks] = m1 < n1 ? xa[n2 * m1] : 0.f;
      }
      {
        // four independent accumulator chains (even / odd k-steps): back-to-back dependent MFMAs would wait out the
        // matrix pipe's latency on every step
        f4 re = {0.f, 0.f, 0.f, 0.f}, im = re, re2 = re, im2 = re;
#pragma unroll
        for (int ks = 0; ks < 12; ks += 2) {
          if (4 * ks >= n1) break;
          const int bi = (4 * ks + lg) * P + 16 * p + lj;
          re = mfma4(a[ks], w1c[bi], re);
          im = mfma4(a[ks], w1s[bi], im);
          if (4 * (ks + 1) < n1) {
            re2 = mfma4(a[ks + 1], w1c[bi + 4 * P], re2);
            im2 = mfma4(a[ks + 1], w1s[bi + 4 * P], im2);
          }
        }
        re += re2;
        im += im2;
        const int k1 = 16 * p + lj;
        if (k1 < n1) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int fo = 16 * mt + 4 * lg + r, ro = fo / n2, mo = fo - ro * n2;
            const int t = (mo * k1) % n;
            const float c = twc[t], sn = tws[t];
            yr[(ro * n2 + mo) * P + k1] = re[r] * c - im[r] * sn;
            yi[(ro * n2 + mo) * P + k1] = re[r] * sn + im[r] * c;
          }
        }
      }
    }
    __syncthreads();
    fetch(pair + gridDim.x);
    // ---- stage 2
    for (int u = wave; u < mt2 * np2; u += 4) {
      const int mt = u / np2, p = u - mt * np2;
      const int f = 16 * mt + lj, row = f / n1, k1 = f - row * n1;
      const float* ya = yr + row * n2 * P + k1;
      const float* yb = yi + row * n2 * P + k1;
      const int fo = 16 * mt + 4 * lg, ro = fo / n1, ko = fo - ro * n1;  // this lane's 4 output k1: ko .. ko + 3
      const int grow = pair * 2 + ro;
      const int ci = grow % d.I, bc = grow / d.I;  // row = (b*C + c)*I + i
      float* ore = out + ((long)(2 * bc) * d.I + ci) * n + ko;
      float* oim = out + ((long)(2 * bc + 1) * d.I + ci) * n + ko;
      {
        f4 re = {0.f, 0.f, 0.f, 0.f}, im = re, re2 = re, im2 = re;
#pragma unroll
        for (int ks = 0; ks < 12; ++ks) {
          if (4 * ks >= n2) break;
          const int m2 = 4 * ks + lg;
          const float ar = ya[m2 * P], ai = yb[m2 * P];
          const int bi = m2 * P + 16 * p + lj;
          const float c = w2c[bi], sn = w2s[bi];
          re = mfma4(ar, c, re);
          im = mfma4(ar, sn, im);
          re2 = mfma4(ai, -sn, re2);
          im2 = mfma4(ai, c, im2);
        }
        re += re2;
        im += im2;
        const int k2 = 16 * p + lj;
        if (k2 < n2) {
          const f4 orr = re * aug.pc - im * aug.ps, oii = re * aug.ps + im * aug.pc;
          *reinterpret_cast<float4*>(ore + n1 * k2) = make_float4(orr[0], orr[1], orr[2], orr[3]);
          *reinterpret_cast<float4*>(oim + n1 * k2) = make_float4(oii[0], oii[1], oii[2], oii[3]);
        }
      }
    }
    stash();  // xs is free since the barrier after stage 1
  }
}

static int fft_launch(const focal_fft_desc* d, const AugParams& aug, const float* x, const float* twiddle, float* out, void* stream) {
  FOCAL_CHECK_ARG(d && x && twiddle && out, "fft_realpack: null argument");
  FOCAL_CHECK_ARG(d->n1 >= 1 && d->n2 >= 1 && d->n1 * d->n2 == d->n, "fft_realpack: n1*n2 != n");
  const size_t sm = (size_t)5 * d->n * sizeof(float);
  FOCAL_CHECK_ARG(sm <= 64 * 1024, "fft_realpack: n=%d too long for the LDS-resident DFT", d->n);
  const int rows = d->B * d->C * d->I;
  const bool mfma_shape = (d->n1 == d->n2) && (d->n1 == 8 || d->n1 == 16 || d->n1 == 24 || d->n1 == 32 || d->n1 == 40 || d->n1 == 48);
  if (mfma_shape && rows % 2 == 0) {
    const size_t smm = (size_t)(2 * d->n + 2 * 48 * d->n1 + (d->n1 == d->n2 ? 0 : 2 * 48 * d->n2) + 4 * 48 * d->n2 + 2 * d->n) * sizeof(float);
    static size_t lds_granted = 48 * 1024;
    if (smm > lds_granted) {  // above the default dynamic-LDS grant: raise it for this kernel (160 KB per CU on gfx950)
      hipError_t e = hipSuccess;
#define FFT_ATTR(N_) if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(fft_realpack_mfma_kernel<N_, N_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smm)
      FFT_ATTR(8); FFT_ATTR(16); FFT_ATTR(24); FFT_ATTR(32); FFT_ATTR(40); FFT_ATTR(48);
#undef FFT_ATTR
      if (e != hipSuccess) {
        focal_set_error("fft_realpack: cannot reserve %zu bytes of LDS: %s", smm, hipGetErrorString(e));
        return FOCAL_EHIP;
      }
      lds_granted = smm;
    }
    const int maxb = 1024;
    int blocks = rows / 2 < maxb ? rows / 2 : maxb;
#define FFT_GO(N_) case N_: FOCAL_LAUNCH((fft_realpack_mfma_kernel<N_, N_>), dim3(blocks), dim3(256), smm, (hipStream_t)stream, x, twiddle, out, *d, rows, aug); break
    switch (d->n1) { FFT_GO(8); FFT_GO(16); FFT_GO(24); FFT_GO(32); FFT_GO(40); default: FFT_GO(48); }
#undef FFT_GO
    FOCAL_LAUNCH_CHECK();
    return FOCAL_OK;
  }
  int blocks = rows < 4096 ? rows : 4096;
  FOCAL_LAUNCH(fft_realpack_kernel, dim3(blocks), dim3(256), sm, (hipStream_t)stream, x, twiddle, out, *d, rows, aug);
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}

// ---- several short transforms in one launch.  The sensor modalities' rows are 20 samples long (100 Hz x 0.2 s): a direct DFT of 400
// MACs per row, for which fft_realpack_kernel's one-workgroup-per-row form keeps 20 of 256 lanes busy, and of which a step has one
// launch per (view, modality) on its serial head -- 8 launches of ~20 us + gaps for the four-modality config.  Here a thread owns one
// output bin of one row (256 / n rows per workgroup pass), and a table in the kernel arguments maps blockIdx ranges to problems.
#define FFT_MULTI_MAX 8
struct FftSmallProblem { const float* x; const float* tw; float* out; int rows, I, n, rpb; AugParams aug; };  // (aug.plan / aug.x_alt as above)
struct FftSmallTable { int nprob; int wg_end[FFT_MULTI_MAX]; FftSmallProblem p[FFT_MULTI_MAX]; };

__global__ __launch_bounds__(256) void fft_small_multi_kernel(const FftSmallTable t) {
  __shared__ float xs[256], twc[64], tws[64];
  const int b = blockIdx.x, tid = threadIdx.x;
  int pi = 0;
#pragma unroll
  for (int q = 0; q < FFT_MULTI_MAX - 1; ++q) pi += (q < t.nprob - 1 && b >= t.wg_end[q]) ? 1 : 0;
  const FftSmallProblem& P = t.p[pi];
  __shared__ int s_perm[FOCAL_AUG_MAX_INTERVALS];
  const AugLive aug = aug_resolve(P.aug, P.x, s_perm, tid);
  const int start = pi > 0 ? t.wg_end[pi - 1] : 0, nb = t.wg_end[pi] - start;
  const int n = P.n, rpb = P.rpb;
  if (tid < n) { twc[tid] = P.tw[2 * tid]; tws[tid] = P.tw[2 * tid + 1]; }
  const int r = tid / n, k = tid - r * n;
  const bool active = r < rpb;
  for (int row0 = (b - start) * rpb; row0 < P.rows; row0 += nb * rpb) {
    const int row = row0 + r;
    const bool ok = active && row < P.rows;
    __syncthreads();
    if (ok) xs[tid] = aug.scale * aug.x[(long)aug_src_row(aug, s_perm, row, P.I) * n + (aug.flip ? n - 1 - k : k)];
    __syncthreads();
    if (!ok) continue;
    const float* xr = xs + r * n;
    float re = 0.f, im = 0.f;
    int ph = 0;  // m * k mod n
    for (int m = 0; m < n; ++m) {
      const float v = xr[m];
      re += v * twc[ph];
      im += v * tws[ph];
      ph += k;
      if (ph >= n) ph -= n;
    }
    const int ci = row % P.I, bc = row / P.I;  // row = (b*C + c)*I + i
    P.out[((long)(2 * bc) * P.I + ci) * n + k] = re * aug.pc - im * aug.ps;
    P.out[((long)(2 * bc + 1) * P.I + ci) * n + k] = re * aug.ps + im * aug.pc;
  }
}

static int aug_params(const focal_fft_desc* d, const focal_aug_desc* a, AugParams* p) {
  *p = aug_identity();
  if (a == nullptr) return FOCAL_OK;
  p->scale = a->scale;
  p->flip = a->flip != 0;
  p->use_perm = a->use_perm != 0;
  p->pc = a->phase_cos;
  p->ps = a->phase_sin;
  if (p->use_perm) {
    FOCAL_CHECK_ARG(d->I <= FOCAL_AUG_MAX_INTERVALS, "augment_fft: %d intervals exceed the permutation table (%d)", d->I, FOCAL_AUG_MAX_INTERVALS);
    for (int i = 0; i < d->I; ++i) {
      FOCAL_CHECK_ARG(a->perm[i] >= 0 && a->perm[i] < d->I, "augment_fft: permutation entry %d out of range", a->perm[i]);
      p->perm[i] = a->perm[i];
    }
  }
  return FOCAL_OK;
}

extern "C" int focal_fft_realpack_multi(int n, const focal_fft_problem* probs, void* stream) {
  FOCAL_CHECK_ARG(n >= 1 && probs, "fft_realpack_multi: no problems");
  const bool no_multi = false;
  FftSmallTable t;
  memset(&t, 0, sizeof(t));
  auto flush = [&]() -> int {
    if (t.nprob == 0) return FOCAL_OK;
    FOCAL_LAUNCH(fft_small_multi_kernel, dim3(t.wg_end[t.nprob - 1]), dim3(256), 0, (hipStream_t)stream, t);
    FOCAL_LAUNCH_CHECK();
    memset(&t, 0, sizeof(t));
    return FOCAL_OK;
  };
  for (int i = 0; i < n; ++i) {
    const focal_fft_problem& q = probs[i];
    FOCAL_CHECK_ARG(q.x && q.twiddle && q.out, "fft_realpack_multi: null tensor in problem %d", i);
    FOCAL_CHECK_ARG(q.d.B > 0 && q.d.C > 0 && q.d.I > 0 && q.d.n > 0, "fft_realpack: bad shape");
    AugParams ap;
    if (int rc = aug_params(&q.d, q.has_aug ? &q.aug : nullptr, &ap)) return rc;
    if (q.plan != nullptr) {
      FOCAL_CHECK_ARG(q.x_warped != nullptr && q.d.I <= FOCAL_AUG_MAX_INTERVALS, "fft_realpack_multi: a device plan needs x_warped and at most %d intervals", FOCAL_AUG_MAX_INTERVALS);
      ap.plan = q.plan;
      ap.x_alt = q.x_warped;
    }
    if (!no_multi && q.d.n <= 64 && q.d.n2 == 1 && q.d.n1 == q.d.n) {  // short rows: the shared launch
      FftSmallProblem& P = t.p[t.nprob];
      P.x = q.x; P.tw = q.twiddle; P.out = q.out;
      P.rows = q.d.B * q.d.C * q.d.I; P.I = q.d.I; P.n = q.d.n; P.rpb = 256 / q.d.n;
      P.aug = ap;
      int blocks = ceil_div(P.rows, P.rpb);
      if (blocks > 1024) blocks = 1024;
      t.wg_end[t.nprob] = (t.nprob ? t.wg_end[t.nprob - 1] : 0) + blocks;
      if (++t.nprob == FFT_MULTI_MAX)
        if (int rc = flush()) return rc;
    } else if (int rc = fft_launch(&q.d, ap, q.x, q.twiddle, q.out, stream)) {
      return rc;
    }
  }
  return flush();
}

extern "C" int focal_fft_realpack_fwd(const focal_fft_desc* d, const float* x, const float* twiddle, float* out, void* stream) {
  return fft_launch(d, aug_identity(), x, twiddle, out, stream);
}

extern "C" int focal_augment_fft_fwd(const focal_fft_desc* d, const focal_aug_desc* a, const float* x, const float* twiddle, float* out,
                                     void* stream) {
  FOCAL_CHECK_ARG(d && a, "augment_fft: null descriptor");
  AugParams p = aug_identity();
  p.scale = a->scale;
  p.flip = a->flip != 0;
  p.use_perm = a->use_perm != 0;
  p.pc = a->phase_cos;
  p.ps = a->phase_sin;
  if (p.use_perm) {
    FOCAL_CHECK_ARG(d->I <= FOCAL_AUG_MAX_INTERVALS, "augment_fft: %d intervals exceed the permutation table (%d)", d->I, FOCAL_AUG_MAX_INTERVALS);
    for (int i = 0; i < d->I; ++i) {
      FOCAL_CHECK_ARG(a->perm[i] >= 0 && a->perm[i] < d->I, "augment_fft: permutation entry %d out of range", a->perm[i]);
      p.perm[i] = a->perm[i];
    }
  }
  return fft_launch(d, p, x, twiddle, out, stream);
}
