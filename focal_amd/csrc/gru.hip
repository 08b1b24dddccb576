// Bidirectional 2-layer GRU of DeepSense's RecurrentBlock (models/RecurrentModule.py:5-31; torch nn.GRU semantics:
// gates (r, z, n), n = tanh(W_in x + b_in + r * (W_hn h + b_hn)), h' = (1 - z) n + z h).
// The matrix products (input projections for all steps at once, one [B,H]x[H,3H] recurrent product per step, and
// after the loop ONE weight-gradient GEMM per matrix over all steps) run on the MFMA GEMM family; the kernels here
// are the per-step gate math forward / backward, the time mean, and two tiny element-wise helpers.
#include "common.hpp"

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }

// gi: [B*T, 3H] rows (b, t) incl. b_ih; gh: [B, 3H] incl. b_hh; save: [4][B][H] = r, z, n, (W_hn h + b_hn)
__global__ __launch_bounds__(256) void gru_gate_fwd_kernel(focal_gru_desc d, int t, int dir_off, const float* __restrict__ gi,
                                                           const float* __restrict__ gh, const float* __restrict__ h_prev,
                                                           float* __restrict__ h_new, float* __restrict__ out, float* __restrict__ save) {
  const int H = d.H, n = d.B * H;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
    const int b = e / H, j = e % H;
    const float* gir = gi + ((long)b * d.T + t) * 3 * H;
    const float* ghr = gh + (long)b * 3 * H;
    const float r = sigmoid_f(gir[j] + ghr[j]);
    const float z = sigmoid_f(gir[H + j] + ghr[H + j]);
    const float ghn = ghr[2 * H + j];
    const float nn = tanhf(gir[2 * H + j] + r * ghn);
    const float hp = h_prev ? h_prev[e] : 0.f;
    const float h = (1.f - z) * nn + z * hp;
    h_new[e] = h;
    out[((long)b * d.T + t) * 2 * H + dir_off + j] = h;
    save[e] = r; save[n + e] = z; save[2 * n + e] = nn; save[3 * n + e] = ghn;
  }
}

__global__ __launch_bounds__(256) void gru_gate_bwd_kernel(focal_gru_desc d, int t, int dir_off, const float* __restrict__ dout, long ld_b,
                                                           long ld_t, float scale, const float* __restrict__ dh_rec,
                                                           const float* __restrict__ dhz_in, const float* __restrict__ save,
                                                           const float* __restrict__ h_prev, float* __restrict__ dgi,
                                                           float* __restrict__ dgh, float* __restrict__ dhz_out) {
  const int H = d.H, n = d.B * H;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
    const int b = e / H, j = e % H;
    float dh = scale * dout[(long)b * ld_b + (long)t * ld_t + dir_off + j];
    if (dh_rec) dh += dh_rec[e];
    if (dhz_in) dh += dhz_in[e];
    const float r = save[e], z = save[n + e], nn = save[2 * n + e], ghn = save[3 * n + e];
    const float hp = h_prev ? h_prev[e] : 0.f;
    const float dn_pre = dh * (1.f - z) * (1.f - nn * nn);
    const float dz_pre = dh * (hp - nn) * z * (1.f - z);
    const float dr_pre = dn_pre * ghn * r * (1.f - r);
    float* gir = dgi + ((long)b * d.T + t) * 3 * H;
    gir[j] = dr_pre; gir[H + j] = dz_pre; gir[2 * H + j] = dn_pre;
    float* ghr = dgh + (long)b * 3 * H;
    ghr[j] = dr_pre; ghr[H + j] = dz_pre; ghr[2 * H + j] = dn_pre * r;
    dhz_out[e] = dh * z;
  }
}

__global__ __launch_bounds__(256) void mean_time_kernel(int B, int T, int D, const float* __restrict__ x, float* __restrict__ y) {
  const int n = B * D;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
    const int b = e / D, j = e % D;
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += x[((long)b * T + t) * D + j];
    y[e] = s / (float)T;
  }
}

__global__ __launch_bounds__(256) void dropout_kernel(long n, const float* __restrict__ x, float* __restrict__ y, const uint32_t* rng,
                                                      uint32_t stream, float p) {
  const DropCtx dc = make_drop(rng, stream, p);
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) y[e] = x[e] * drop_mult(dc, (uint32_t)e);
}

__global__ __launch_bounds__(256) void axpy_kernel(long n, float a, const float* __restrict__ x, float* __restrict__ y) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) y[e] += a * x[e];
}

static int gblocks(long n) { long b = (n + 255) / 256; return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b)); }

extern "C" int focal_gru_gate_fwd(const focal_gru_desc* d, int t, int dir_offset, const float* gi, const float* gh, const float* h_prev,
                                  float* h_new, float* out, float* save, void* stream) {
  FOCAL_CHECK_ARG(d && gi && gh && h_new && out && save && t >= 0 && t < d->T, "gru_gate_fwd: bad argument");
  hipLaunchKernelGGL(gru_gate_fwd_kernel, dim3(gblocks((long)d->B * d->H)), dim3(256), 0, (hipStream_t)stream, *d, t, dir_offset, gi, gh,
                     h_prev, h_new, out, save);
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}

extern "C" int focal_gru_gate_bwd(const focal_gru_desc* d, int t, int dir_offset, const float* dout, long ld_b, long ld_t, float scale,
                                  const float* dh_rec, const float* dhz_in, const float* save, const float* h_prev, float* dgi,
                                  float* dgh, float* dhz_out, void* stream) {
  FOCAL_CHECK_ARG(d && dout && save && dgi && dgh && dhz_out && t >= 0 && t < d->T, "gru_gate_bwd: bad argument");
  hipLaunchKernelGGL(gru_gate_bwd_kernel, dim3(gblocks((long)d->B * d->H)), dim3(256), 0, (hipStream_t)stream, *d, t, dir_offset, dout,
                     ld_b, ld_t, scale, dh_rec, dhz_in, save, h_prev, dgi, dgh, dhz_out);
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}

extern "C" int focal_mean_time(int B, int T, int D, const float* x, float* y, void* stream) {
  FOCAL_CHECK_ARG(x && y && B > 0 && T > 0 && D > 0, "mean_time: bad argument");
  hipLaunchKernelGGL(mean_time_kernel, dim3(gblocks((long)B * D)), dim3(256), 0, (hipStream_t)stream, B, T, D, x, y);
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}

extern "C" int focal_dropout(long n, const float* x, float* y, const uint32_t* rng, uint32_t stream_id, float p, void* stream) {
  FOCAL_CHECK_ARG(x && y && n >= 0 && p >= 0.f && p < 1.f, "dropout: bad argument");
  hipLaunchKernelGGL(dropout_kernel, dim3(gblocks(n)), dim3(256), 0, (hipStream_t)stream, n, x, y, rng, stream_id, p);
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}

extern "C" int focal_axpy(long n, float a, const float* x, float* y, void* stream) {
  FOCAL_CHECK_ARG(x && y && n >= 0, "axpy: bad argument");
  hipLaunchKernelGGL(axpy_kernel, dim3(gblocks(n)), dim3(256), 0, (hipStream_t)stream, n, a, x, y);
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}
