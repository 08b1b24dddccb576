// Classifier head of the finetuning path: the attention core of TransformerFusionBlock
// (models/FusionModules.py:122-140: ONE query -- the mean of the fused tokens -- attending over the M modality tokens with
// nn.MultiheadAttention) and nn.CrossEntropyLoss.  Tensors here are [B, M <= 8, E]: kilobytes; the matrix products around
// them (in_proj, out_proj, class layer) run on the GEMM family, LayerNorm on norm.hip.  One workgroup per sample, one wave per
// head (head_dim = 64): dot products are wave reductions, the M-way softmax lives in registers.
#include "common.hpp"

#define FUSE_MAX_M 8

// q [B, E]; kv [B*M, 2E] rows (b, j) = {k | v}; out [B, E]; probs / weights [B, H, M] = softmax and softmax * dropout mask
__global__ __launch_bounds__(256) void fusion_attn_fwd_kernel(const float* __restrict__ q, const float* __restrict__ kv,
                                                              float* __restrict__ out, float* __restrict__ probs,
                                                              float* __restrict__ weights, int M, int E, int H, float scale,
                                                              const uint32_t* rng, uint32_t stream, float p_drop) {
  const int b = blockIdx.x, e = threadIdx.x, h = e >> 6;
  if (e >= E) return;
  const DropCtx dc = make_drop(rng, stream, p_drop);
  const float qe = q[(long)b * E + e] * scale;
  float s[FUSE_MAX_M], mx = -3.0e38f;
  for (int j = 0; j < M; ++j) {
    s[j] = wave_sum(qe * kv[((long)b * M + j) * 2 * E + e]);
    mx = fmaxf(mx, s[j]);
  }
  float den = 0.f;
  for (int j = 0; j < M; ++j) { s[j] = __expf(s[j] - mx); den += s[j]; }
  float acc = 0.f;
  for (int j = 0; j < M; ++j) {
    const float pj = s[j] / den;
    const float wj = p_drop > 0.f ? pj * drop_mult(dc, ((uint32_t)b * H + h) * M + j) : pj;
    acc += wj * kv[((long)b * M + j) * 2 * E + E + e];
    if ((e & 63) == 0) { probs[((long)b * H + h) * M + j] = pj; weights[((long)b * H + h) * M + j] = wj; }
  }
  out[(long)b * E + e] = acc;
}

// dq [B, E], dkv [B*M, 2E] from dout [B, E]
__global__ __launch_bounds__(256) void fusion_attn_bwd_kernel(const float* __restrict__ q, const float* __restrict__ kv,
                                                              const float* __restrict__ probs, const float* __restrict__ weights,
                                                              const float* __restrict__ dout, float* __restrict__ dq,
                                                              float* __restrict__ dkv, int M, int E, int H, float scale) {
  const int b = blockIdx.x, e = threadIdx.x, h = e >> 6;
  if (e >= E) return;
  const float go = dout[(long)b * E + e], qe = q[(long)b * E + e];
  float dw[FUSE_MAX_M], dot = 0.f;
  for (int j = 0; j < M; ++j) {
    const float pj = probs[((long)b * H + h) * M + j], wj = weights[((long)b * H + h) * M + j];
    const float mj = pj > 0.f ? wj / pj : 0.f;  // dropout multiplier (0 or 1 / (1 - p))
    dkv[((long)b * M + j) * 2 * E + E + e] = wj * go;                      // dV_j
    dw[j] = wave_sum(go * kv[((long)b * M + j) * 2 * E + E + e]) * mj;     // dL / d p_j
    dot += pj * dw[j];
  }
  float dqe = 0.f;
  for (int j = 0; j < M; ++j) {
    const float pj = probs[((long)b * H + h) * M + j];
    const float ds = pj * (dw[j] - dot) * scale;                           // dL / d(q . k_j)
    dqe += ds * kv[((long)b * M + j) * 2 * E + e];
    dkv[((long)b * M + j) * 2 * E + e] = ds * qe;                          // dK_j
  }
  dq[(long)b * E + e] = dqe;
}

extern "C" int focal_fusion_attn_fwd(int B, int M, int E, int heads, const float* q, const float* kv, float* out, float* probs,
                                     float* weights, const uint32_t* rng, uint32_t stream_id, float p_drop, void* stream) {
  FOCAL_CHECK_ARG(q && kv && out && probs && weights, "fusion_attn_fwd: null tensor");
  FOCAL_CHECK_ARG(B > 0 && M >= 1 && M <= FUSE_MAX_M && heads >= 1 && E == heads * 64 && E <= 256,
                  "fusion_attn: need head_dim 64, E <= 256, M <= %d (got M=%d E=%d heads=%d)", FUSE_MAX_M, M, E, heads);
  FOCAL_LAUNCH(fusion_attn_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, q, kv, out, probs, weights, M, E, heads,
                     0.125f, rng, stream_id, p_drop);  // 1 / sqrt(64)
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}

extern "C" int focal_fusion_attn_bwd(int B, int M, int E, int heads, const float* q, const float* kv, const float* probs,
                                     const float* weights, const float* dout, float* dq, float* dkv, void* stream) {
  FOCAL_CHECK_ARG(q && kv && probs && weights && dout && dq && dkv, "fusion_attn_bwd: null tensor");
  FOCAL_CHECK_ARG(B > 0 && M >= 1 && M <= FUSE_MAX_M && heads >= 1 && E == heads * 64 && E <= 256, "fusion_attn_bwd: bad geometry");
  FOCAL_LAUNCH(fusion_attn_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, q, kv, probs, weights, dout, dq, dkv, M, E,
                     heads, 0.125f);
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}

// nn.CrossEntropyLoss() (mean reduction): loss[0] = mean_b (logsumexp(logits_b) - logits_b[label_b]); dlogits = (softmax - onehot) / B
__global__ __launch_bounds__(256) void cross_entropy_kernel(const float* __restrict__ logits, const long* __restrict__ labels,
                                                            float* __restrict__ loss, float* __restrict__ dlogits, int B, int C) {
  __shared__ float part[256];
  float acc = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) {
    const float* row = logits + (long)b * C;
    float mx = row[0];
    for (int c = 1; c < C; ++c) mx = fmaxf(mx, row[c]);
    float den = 0.f;
    for (int c = 0; c < C; ++c) den += expf(row[c] - mx);
    const long y = labels[b];
    acc += logf(den) + mx - row[y];
    for (int c = 0; c < C; ++c) dlogits[(long)b * C + c] = (expf(row[c] - mx) / den - (c == y ? 1.f : 0.f)) / (float)B;
  }
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = part[0] / (float)B;
}

extern "C" int focal_cross_entropy(int B, int C, const float* logits, const long* labels, float* loss, float* dlogits, void* stream) {
  FOCAL_CHECK_ARG(B > 0 && C > 0 && logits && labels && loss && dlogits, "cross_entropy: bad argument");
  FOCAL_LAUNCH(cross_entropy_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, labels, loss, dlogits, B, C);
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}

// ---- class layer: nn.Linear(K -> n_cls) with a handful of outputs (7 vehicle classes): too narrow for the 16-column MFMA
// tiles (and n_cls is not a multiple of the GEMM family's vector width), and a few hundred KFLOP in any case.  fp32 VALU.
// y[b][n] = bias[n] + sum_k x[b][k] w[n][k]: one wave per (b, n), lanes over k.
__global__ __launch_bounds__(256) void small_linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ bias, float* __restrict__ y, int B, int N, int K) {
  const int item = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (item >= B * N) return;
  const int b = item / N, n = item - b * N;
  float acc = 0.f;
  for (int k = lane; k < K; k += 64) acc += x[(long)b * K + k] * w[(long)n * K + k];
  acc = wave_sum(acc);
  if (lane == 0) y[item] = acc + (bias ? bias[n] : 0.f);
}
// dw[n][k] += sum_b dy[b][n] x[b][k] (thread per (n, k));  dbias[n] += sum_b dy[b][n] (first K-column's threads)
__global__ __launch_bounds__(256) void small_linear_bwd_weight_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                      float* __restrict__ dw, float* __restrict__ dbias, int B, int N, int K) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long)N * K) return;
  const int n = (int)(e / K), k = (int)(e - (long)n * K);
  float acc = 0.f, bs = 0.f;
  for (int b = 0; b < B; ++b) {
    const float g = dy[(long)b * N + n];
    acc += g * x[(long)b * K + k];
    bs += g;
  }
  dw[e] += acc;
  if (k == 0 && dbias) dbias[n] += bs;
}
// dx[b][k] = sum_n dy[b][n] w[n][k]
__global__ __launch_bounds__(256) void small_linear_bwd_data_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                                    float* __restrict__ dx, int B, int N, int K) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long)B * K) return;
  const int b = (int)(e / K), k = (int)(e - (long)b * K);
  float acc = 0.f;
  for (int n = 0; n < N; ++n) acc += dy[(long)b * N + n] * w[(long)n * K + k];
  dx[e] = acc;
}

extern "C" int focal_small_linear_fwd(int B, int N, int K, const float* x, const float* w, const float* bias, float* y, void* stream) {
  FOCAL_CHECK_ARG(B > 0 && N > 0 && K > 0 && x && w && y, "small_linear_fwd: bad argument");
  FOCAL_LAUNCH(small_linear_fwd_kernel, dim3(ceil_div((long)B * N, 4)), dim3(256), 0, (hipStream_t)stream, x, w, bias, y, B, N, K);
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}
extern "C" int focal_small_linear_bwd(int B, int N, int K, const float* dy, const float* x, const float* w, float* dw, float* dbias,
                                      float* dx, void* stream) {
  FOCAL_CHECK_ARG(B > 0 && N > 0 && K > 0 && dy && x && w && dw, "small_linear_bwd: bad argument");
  hipStream_t st = (hipStream_t)stream;
  FOCAL_LAUNCH(small_linear_bwd_weight_kernel, dim3(ceil_div((long)N * K, 256)), dim3(256), 0, st, dy, x, dw, dbias, B, N, K);
  if (dx) FOCAL_LAUNCH(small_linear_bwd_data_kernel, dim3(ceil_div((long)B * K, 256)), dim3(256), 0, st, dy, w, dx, B, N, K);
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}
