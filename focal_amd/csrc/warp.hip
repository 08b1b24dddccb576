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

// ------------------------------------------------------------------------------------------------ view draws on the device (round 5)
// Everything the reference's augmenter classes draw on the host per view (include/focal_hip.h: focal_view_draw), from a counter RNG:
// draw i of a (view, slot) key = focal_mix32(key + i * odd constant); the key mixes the device seed word (advanced by the optimizer every
// step), the caller's stream id, the view and the slot.  One thread per (view, slot); the kernel is a handful of scalar operations.
__device__ __forceinline__ float vd_uniform(uint32_t key, uint32_t i) { return (focal_mix32(key + i * 0x85EBCA6BU) >> 8) * (1.0f / 16777216.0f); }  // [0, 1)
__device__ __forceinline__ float vd_normal(uint32_t key, uint32_t i) {  // Box-Muller on draws 2 i, 2 i + 1
  const float u1 = 1.0f - vd_uniform(key, 2 * i), u2 = vd_uniform(key, 2 * i + 1);
  return sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
}

// `advance` (focal_view_draw_shared): the seed word is the draw's OWN state -- one every data-parallel rank holds a copy of, so that the global
// batch gets one augmenter / coin / permutation / scale / phase per view as the reference's batch does -- and the kernel moves it on after
// everybody has read it (the word the optimizer advances keys the per-rank dropout streams and stays per-rank).
__global__ __launch_bounds__(64) void view_draw_kernel(const focal_view_pool pool, int n_views, int n_slots, uint32_t* seed,
                                                       uint32_t stream_id, focal_view_plan* __restrict__ plans, int advance) {
  __shared__ int s_perm[64][FOCAL_AUG_MAX_INTERVALS + 1];
  const int t = threadIdx.x;
  const uint32_t s0 = seed ? seed[0] : 0u;
  if (advance) {
    __syncthreads();
    if (t == 0) {
      seed[0] = focal_mix32(s0 + 0x9E3779B9U);
      seed[1] += 1u;
    }
  }
  if (t >= n_views * n_slots) return;
  const int view = t / n_slots, slot = t - view * n_slots;
  // ONE pool entry per view (Augmenter.py:86 np.random.randint), then per slot the augmenter's own coin and parameters
  const uint32_t kv = focal_mix32(s0 * 0x9E3779B9U + stream_id * 0x85EBCA6BU + (uint32_t)view * 0xC2B2AE35U + 0x27D4EB2FU);
  int k = (int)(vd_uniform(kv, 0) * (float)pool.n_aug);
  k = k < pool.n_aug ? k : pool.n_aug - 1;
  const uint32_t ks = focal_mix32(kv ^ ((uint32_t)(slot + 1) * 0x9E3779B9U));
  const bool hit = vd_uniform(ks, 0) < pool.prob[k];
  const int kind = hit ? pool.kind[k] : FOCAL_VIEW_NONE;
  focal_view_plan* pl = plans + t;
  float scale = 1.0f, pc = 1.0f, ps = 0.0f;
  int flip = 0, use_perm = 0, warp = 0, nk = 0;
  const int I = pool.intervals[slot] < FOCAL_AUG_MAX_INTERVALS ? pool.intervals[slot] : FOCAL_AUG_MAX_INTERVALS;
  for (int i = 0; i < FOCAL_AUG_MAX_INTERVALS; ++i) s_perm[t][i] = i;
  if (kind == FOCAL_VIEW_NEGATION) {
    scale = -1.0f;
  } else if (kind == FOCAL_VIEW_SCALING) {
    scale = 1.0f + pool.scaling_std * vd_normal(ks, 1);
  } else if (kind == FOCAL_VIEW_HFLIP) {
    flip = 1;
  } else if (kind == FOCAL_VIEW_PERMUTATION) {  // a uniformly random order of the slot's intervals (Fisher-Yates)
    use_perm = 1;
    for (int i = I - 1; i > 0; --i) {
      int j = (int)(vd_uniform(ks, 8 + i) * (float)(i + 1));
      j = j <= i ? j : i;
      const int a = s_perm[t][i];
      s_perm[t][i] = s_perm[t][j];
      s_perm[t][j] = a;
    }
  } else if (kind == FOCAL_VIEW_PHASE_SHIFT) {
    const float ang = (vd_uniform(ks, 1) - 0.5f) * 6.283185307179586f;
    pc = cosf(ang); ps = sinf(ang);
  } else if (kind == FOCAL_VIEW_MAG_WARP || kind == FOCAL_VIEW_TIME_WARP) {
    warp = kind;
    const int ord = kind == FOCAL_VIEW_MAG_WARP ? pool.mag_order : pool.time_order;
    const float mag = kind == FOCAL_VIEW_MAG_WARP ? pool.mag_magnitude : pool.time_magnitude;
    nk = 3 * (ord - 1) + 1;
    nk = nk < FOCAL_VIEW_MAX_KNOTS ? nk : FOCAL_VIEW_MAX_KNOTS;
    for (int i = 0; i < nk; ++i) pl->knots[i] = 1.0f + mag * vd_normal(ks, 4 + i);
  }
  for (int i = nk; i < FOCAL_VIEW_MAX_KNOTS; ++i) pl->knots[i] = 1.0f;
  pl->aug.scale = scale; pl->aug.flip = flip; pl->aug.use_perm = use_perm; pl->aug.phase_cos = pc; pl->aug.phase_sin = ps;
  for (int i = 0; i < FOCAL_AUG_MAX_INTERVALS; ++i) pl->aug.perm[i] = s_perm[t][i];
  pl->kind = kind; pl->pool_index = k; pl->warp = warp; pl->nknots = nk;
}

static int view_draw_launch(const focal_view_pool* pool, int n_views, int n_slots, uint32_t* seed, uint32_t stream_id,
                            focal_view_plan* plans, int advance, void* stream) {
  FOCAL_CHECK_ARG(pool && plans && n_views >= 1 && n_slots >= 1 && n_slots <= FOCAL_VIEW_MAX_SLOTS && n_views * n_slots <= 64,
                  "view_draw: 1 .. %d slots, at most 64 (view, slot) pairs", FOCAL_VIEW_MAX_SLOTS);
  FOCAL_CHECK_ARG(pool->n_aug >= 1 && pool->n_aug <= FOCAL_VIEW_MAX_POOL, "view_draw: pool of 1 .. %d augmenters", FOCAL_VIEW_MAX_POOL);
  for (int i = 0; i < pool->n_aug; ++i) {
    FOCAL_CHECK_ARG(pool->kind[i] >= FOCAL_VIEW_NONE && pool->kind[i] <= FOCAL_VIEW_TIME_WARP, "view_draw: unknown augmenter kind %d", pool->kind[i]);
    const int ord = pool->kind[i] == FOCAL_VIEW_MAG_WARP ? pool->mag_order : pool->kind[i] == FOCAL_VIEW_TIME_WARP ? pool->time_order : 2;
    FOCAL_CHECK_ARG(ord >= 2 && 3 * (ord - 1) + 1 <= FOCAL_VIEW_MAX_KNOTS, "view_draw: spline order %d needs more than %d knots", ord, FOCAL_VIEW_MAX_KNOTS);
  }
  FOCAL_LAUNCH(view_draw_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, *pool, n_views, n_slots, seed, stream_id, plans, advance);
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}

extern "C" int focal_view_draw(const focal_view_pool* pool, int n_views, int n_slots, const uint32_t* seed, uint32_t stream_id,
                               focal_view_plan* plans, void* stream) {
  return view_draw_launch(pool, n_views, n_slots, const_cast<uint32_t*>(seed), stream_id, plans, 0, stream);
}

extern "C" int focal_view_draw_shared(const focal_view_pool* pool, int n_views, int n_slots, uint32_t* view_state, uint32_t stream_id,
                                      focal_view_plan* plans, void* stream) {
  FOCAL_CHECK_ARG(view_state != nullptr, "view_draw_shared: null view state (4 words: {seed, draw count, 0, 0})");
  return view_draw_launch(pool, n_views, n_slots, view_state, stream_id, plans, 1, stream);
}

// The warp curves of focal_amd/warp.py on the device, one workgroup per problem: the not-a-knot cubic spline through the plan's knots
// (_natural_cubic_through: a dense solve for the knot slopes, n <= 16, by one thread in fp64), evaluated at 0 .. L - 1 by everybody;
// time warp: the curve's running sum (per-thread chunks + an LDS scan, fp64 as numpy's cumsum), re-based to positions in [0, L - 1].
// tables: magnitude warp [L] multipliers; time warp [L] floor(position) (int) | [L] fractional part -- the 24 resampling weights of a
// position (time_warp_tables: cubic B-spline x truncated prefilter in the interior, the end window's exact basis weights within 12
// samples of either end) are formed from these two by the pass itself, per thread, instead of travelling through a 1.5 MB table.
constexpr int WT_TAPS = 24, WT_RADIUS = 10, WT_END_WIN = 48, WT_THREADS = 1024;
struct WarpTable { int n; focal_warp_problem p[8]; };
__global__ __launch_bounds__(WT_THREADS) void warp_curve_kernel(const WarpTable wt) {
  const focal_warp_problem& P = wt.p[blockIdx.x];
  const focal_view_plan* plan = P.plan;
  const int L = P.L;
  float* tables = P.tables;
  const int warp = focal_plan_warp(plan);
  if (warp == 0) return;
  __shared__ double xk[FOCAL_VIEW_MAX_KNOTS], yk[FOCAL_VIEW_MAX_KNOTS], sk[FOCAL_VIEW_MAX_KNOTS], c2[FOCAL_VIEW_MAX_KNOTS], c3[FOCAL_VIEW_MAX_KNOTS];
  __shared__ double Am[FOCAL_VIEW_MAX_KNOTS][FOCAL_VIEW_MAX_KNOTS + 1];
  __shared__ double part[WT_THREADS];
  __shared__ double hk[FOCAL_VIEW_MAX_KNOTS], slk[FOCAL_VIEW_MAX_KNOTS], fcol[FOCAL_VIEW_MAX_KNOTS];
  __shared__ int pivot_row;
  const int tid = threadIdx.x, n = plan->nknots;
  if (tid == 0) {
    // knot abscissae: numpy.linspace(-L, 2 L - 1, n, dtype=int) = floor(arange(n) * step + start) (integer linspace floors since
    // numpy 1.20; focal_amd/warp.py's knot_positions is that call), the last one exactly the stop
    const double start = -(double)L, step = (double)(3 * L - 1) / (double)(n - 1);
    for (int i = 0; i < n; ++i) {
      xk[i] = floor((double)i * step + start);
      yk[i] = (double)plan->knots[i];
    }
    xk[n - 1] = (double)(2 * L - 1);
    double h[FOCAL_VIEW_MAX_KNOTS], slope[FOCAL_VIEW_MAX_KNOTS];
    for (int i = 0; i < n - 1; ++i) { h[i] = xk[i + 1] - xk[i]; slope[i] = (yk[i + 1] - yk[i]) / h[i]; }
    for (int i = 0; i < n; ++i)
      for (int j = 0; j <= n; ++j) Am[i][j] = 0.0;
    for (int i = 1; i < n - 1; ++i) {  // continuity of the second derivative; unknowns = the first derivatives at the knots
      Am[i][i - 1] = h[i]; Am[i][i] = 2.0 * (h[i - 1] + h[i]); Am[i][i + 1] = h[i - 1];
      Am[i][n] = 3.0 * (h[i] * slope[i - 1] + h[i - 1] * slope[i]);
    }
    double d = xk[2] - xk[0];  // not-a-knot at both ends
    Am[0][0] = h[1]; Am[0][1] = d;
    Am[0][n] = ((h[0] + 2.0 * d) * h[1] * slope[0] + h[0] * h[0] * slope[1]) / d;
    d = xk[n - 1] - xk[n - 3];
    Am[n - 1][n - 1] = h[n - 3]; Am[n - 1][n - 2] = d;
    Am[n - 1][n] = (h[n - 2] * h[n - 2] * slope[n - 3] + (2.0 * d + h[n - 2]) * h[n - 3] * slope[n - 2]) / d;
    for (int i = 0; i < n - 1; ++i) { hk[i] = h[i]; slk[i] = slope[i]; }
  }
  __syncthreads();
  // Gaussian elimination with partial pivoting, the row operations spread over the workgroup (thread -> (row, column) of the augmented
  // matrix): the same arithmetic per element as one thread walking the LDS-resident matrix, which took 25-35 us of dependent LDS round
  // trips on the serial head of every step with random views (round 5: warp_curve_kernel 37-53 us -> see profiles/r5_views_random.txt).
  {
    constexpr int W = FOCAL_VIEW_MAX_KNOTS + 1;
    const int er = tid / W, ej = tid % W;
    for (int c = 0; c < n; ++c) {
      if (tid == 0) {
        int piv = c;
        for (int r = c + 1; r < n; ++r)
          if (fabs(Am[r][c]) > fabs(Am[piv][c])) piv = r;
        pivot_row = piv;
      }
      __syncthreads();
      const int piv = pivot_row;
      if (piv != c && tid >= c && tid <= n) { const double tmp = Am[c][tid]; Am[c][tid] = Am[piv][tid]; Am[piv][tid] = tmp; }
      __syncthreads();
      if (tid > c && tid < n) fcol[tid] = Am[tid][c] / Am[c][c];
      __syncthreads();
      if (er > c && er < n && ej >= c && ej <= n) Am[er][ej] -= fcol[er] * Am[c][ej];
      __syncthreads();
    }
  }
  if (tid == 0) {
    for (int r = n - 1; r >= 0; --r) {
      double acc = Am[r][n];
      for (int j = r + 1; j < n; ++j) acc -= Am[r][j] * sk[j];
      sk[r] = acc / Am[r][r];
    }
    for (int i = 0; i < n - 1; ++i) {
      c2[i] = (3.0 * slk[i] - 2.0 * sk[i] - sk[i + 1]) / hk[i];
      c3[i] = (sk[i] + sk[i + 1] - 2.0 * slk[i]) / (hk[i] * hk[i]);
    }
  }
  __syncthreads();
  auto curve = [&](int q) {
    int idx = 0;
    for (int i = 1; i < n - 1; ++i) idx += (xk[i] <= (double)q) ? 1 : 0;
    const double t = (double)q - xk[idx];
    return yk[idx] + t * (sk[idx] + t * (c2[idx] + t * c3[idx]));
  };
  const int CH = (L + WT_THREADS - 1) / WT_THREADS;
  const int q0 = min(L, tid * CH), q1 = min(L, q0 + CH);
  if (warp == FOCAL_VIEW_MAG_WARP) {
    for (int q = q0; q < q1; ++q) tables[q] = (float)curve(q);
    return;
  }
  double loc = 0.0;
  for (int q = q0; q < q1; ++q) loc += curve(q);
  part[tid] = loc;
  __syncthreads();
  for (int o = 1; o < WT_THREADS; o <<= 1) {  // inclusive scan of the chunk sums
    const double add = tid >= o ? part[tid - o] : 0.0;
    __syncthreads();
    part[tid] += add;
    __syncthreads();
  }
  const double total = part[WT_THREADS - 1], first = curve(0);
  const double denom = total - first;
  double run = part[tid] - loc;  // exclusive prefix of this chunk
  int* kf = reinterpret_cast<int*>(tables);
  float* ff = tables + L;
  for (int q = q0; q < q1; ++q) {
    run += curve(q);
    double c = (run - first) / denom;
    c = c < 0.0 ? 0.0 : (c > 1.0 ? 1.0 : c);
    const double pos = c * (double)(L - 1), kd = floor(pos);
    kf[q] = (int)kd;
    ff[q] = (float)(pos - kd);
  }
}

// sqrt(3) z1^|j|, j = -10 .. 10 (z1 = sqrt(3) - 2): the truncated recursive prefilter of the cardinal cubic spline
__device__ __forceinline__ float wt_prefilter(int i) {
  constexpr double S3 = 1.7320508075688772, Z1 = -0.2679491924311228;
  constexpr double P1 = S3 * Z1, P2 = P1 * Z1, P3 = P2 * Z1, P4 = P3 * Z1, P5 = P4 * Z1, P6 = P5 * Z1, P7 = P6 * Z1, P8 = P7 * Z1, P9 = P8 * Z1,
                   P10 = P9 * Z1;
  constexpr float T[11] = {(float)S3, (float)P1, (float)P2, (float)P3, (float)P4, (float)P5, (float)P6, (float)P7, (float)P8, (float)P9, (float)P10};
  const int a = i < WT_RADIUS ? WT_RADIUS - i : i - WT_RADIUS;
  return T[a];
}

// y = warp(x) when the plan asks for one (magnitude: x * mult; time: the 24-tap resampling of time_warp_kernel with the weights formed
// here from floor / fraction of the position); nothing otherwise
template <int TAPS, int RPT>
__global__ __launch_bounds__(256) void warp_plan_apply_kernel(const WarpTable tab, const float* __restrict__ end_coef) {
  const focal_warp_problem& P = tab.p[blockIdx.z];
  const int warp = focal_plan_warp(P.plan);
  if (warp == 0) return;
  const int rows = P.rows, L = P.L;
  const float* __restrict__ x = P.x;
  const float* __restrict__ tables = P.tables;
  float* __restrict__ y = P.y;
  if ((int)(blockIdx.y * RPT) >= rows) return;
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= L) return;
  const int r0 = blockIdx.y * RPT, r1 = min(rows, r0 + RPT);
  if (warp == FOCAL_VIEW_MAG_WARP) {
    const float m = tables[n];
    for (int r = r0; r < r1; ++r) y[(long)r * L + n] = x[(long)r * L + n] * m;
    return;
  }
  const int k = reinterpret_cast<const int*>(tables)[n];
  const float f = tables[L + n];
  const double pos = (double)k + (double)f;
  const int reach = TAPS / 2;
  const bool lo = L >= WT_END_WIN && pos < (double)reach, hi = L >= WT_END_WIN && pos > (double)(L - 1 - reach);
  float wt[TAPS];
  int base;
  if (lo || hi) {  // the end windows: exact not-a-knot basis weights (warp.py: _end_weights)
    const double qq = lo ? pos : pos - (double)(L - WT_END_WIN);
    int seg = (int)floor(qq);
    seg = seg < 0 ? 0 : (seg > WT_END_WIN - 2 ? WT_END_WIN - 2 : seg);
    const float t = (float)(qq - (double)seg);
    const float* cf = end_coef + (long)seg * 4 * WT_END_WIN + (lo ? 0 : WT_END_WIN - TAPS);
#pragma unroll
    for (int tp = 0; tp < TAPS; ++tp)
      wt[tp] = cf[tp] + t * (cf[WT_END_WIN + tp] + t * (cf[2 * WT_END_WIN + tp] + t * cf[3 * WT_END_WIN + tp]));
    base = lo ? 0 : L - TAPS;
  } else {
    const float f2 = f * f, f3 = f2 * f, omf = 1.0f - f;
    const float b[4] = {omf * omf * omf * (1.0f / 6.0f), (3.0f * f3 - 6.0f * f2 + 4.0f) * (1.0f / 6.0f),
                        (-3.0f * f3 + 3.0f * f2 + 3.0f * f + 1.0f) * (1.0f / 6.0f), f3 * (1.0f / 6.0f)};
#pragma unroll
    for (int tp = 0; tp < TAPS; ++tp) {  // coefficient k - 1 + j gathers samples through taps j .. j + 2 RADIUS
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int gi = tp - j;
        if (gi >= 0 && gi <= 2 * WT_RADIUS) acc += b[j] * wt_prefilter(gi);
      }
      wt[tp] = acc;
    }
    base = k - 1 - WT_RADIUS;
  }
  int idx[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t) idx[t] = min(max(base + t, 0), L - 1);
#pragma unroll 2
  for (int r = r0; r < r1; ++r) {
    const float* src = x + (long)r * L;
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < TAPS; ++t) acc += wt[t] * src[idx[t]];
    y[(long)r * L + n] = acc;
  }
}

extern "C" int focal_warp_plan_multi(int n, const focal_warp_problem* problems, const float* end_coef, void* stream) {
  FOCAL_CHECK_ARG(n >= 1 && n <= 8 && problems && end_coef, "warp_plan_multi: 1 .. 8 problems");
  WarpTable wt;
  memset(&wt, 0, sizeof(wt));
  wt.n = n;
  int max_l = 0, max_rows = 0;
  for (int i = 0; i < n; ++i) {
    const focal_warp_problem& q = problems[i];
    FOCAL_CHECK_ARG(q.rows > 0 && q.L >= 4 && q.L % 4 == 0 && q.x && q.plan && q.tables && q.y && q.x != q.y, "warp_plan_multi: bad problem %d", i);
    FOCAL_CHECK_ARG(((uintptr_t)q.tables % 16) == 0, "warp_plan_multi: the table workspace of problem %d must be 16-byte aligned", i);
    wt.p[i] = q;
    max_l = q.L > max_l ? q.L : max_l;
    max_rows = q.rows > max_rows ? q.rows : max_rows;
  }
  hipStream_t st = (hipStream_t)stream;
  FOCAL_LAUNCH(warp_curve_kernel, dim3(n), dim3(WT_THREADS), 0, st, wt);
  constexpr int RPT = 16;
  FOCAL_LAUNCH((warp_plan_apply_kernel<WT_TAPS, RPT>), dim3((max_l + 255) / 256, (max_rows + RPT - 1) / RPT, n), dim3(256), 0, st, wt, end_coef);
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
