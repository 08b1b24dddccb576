// Fused Swin MLP branch at 64 channels (Swin stage 0: 2/3 of the model's 4C-wide activation bytes live here).
//
//   forward   x_out = x_mid + drop_o( drop_h(gelu(a2 W1^T + b1)) W2^T + b2 )   (+ the LayerNorm that follows, optional)
//   backward  (mlp_bwd.hip)
//
// replaces models/SwinModules.py:18-34 (Mlp.forward: fc1 -> GELU -> Dropout -> fc2 -> Dropout) + the residual add and
// DropPath of SwinTransformerBlock.forward (:339-341).  The hidden activation h [M, 256] never exists in HBM: a wave owns 16
// token rows, computes 64 hidden columns at a time (fc1 accumulators [hidden][token]), applies bias + GELU + dropout in
// registers and feeds the result STRAIGHT back to the matrix cores as the B operand of fc2 -- the accumulator layout of
// v_mfma_f32_16x16x32_bf16 (lane = token, 4 consecutive hidden units per 16-wide tile) is a legal operand layout once the
// contraction index is re-ordered, and the fc2 weight image in LDS is stored in that order.  Both weights live in LDS for the
// whole workgroup (64 KB, XOR-swizzled so that every ds_read_b128 fragment read is conflict-free); workgroups are persistent and
// waves never synchronise after the weights are staged.
#include "gemm.hpp"
#include "mlp.hpp"

namespace focal_mlp_kernels {

constexpr int C = MLP_C, H = MLP_H;
constexpr int LDS_W1 = 0, LDS_W2 = 32768, LDS_B1 = 65536, LDS_B2 = LDS_B1 + 1024, LDS_G = LDS_B2 + 256, LDS_BT = LDS_G + 256,
              LDS_FWD_BYTES = LDS_BT + 256;

template <bool LN, bool DROP>
__global__ __launch_bounds__(512, 4) void mlp_fwd_kernel(const MlpFwdParams p) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- stage the weights (all loads in flight before the first LDS write)
  {
    uint4 v1[4];
    uint2 v2[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) v1[i] = reinterpret_cast<const uint4*>(p.w1)[tid + 512 * i];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = tid + 512 * i, c = q >> 5, dch = q & 31, h0 = 32 * (dch >> 2) + 4 * (dch & 3);
      v2[2 * i] = *reinterpret_cast<const uint2*>(p.w2 + c * H + h0);
      v2[2 * i + 1] = *reinterpret_cast<const uint2*>(p.w2 + c * H + h0 + 16);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = tid + 512 * i, row = q >> 3, ch = q & 7;
      *reinterpret_cast<uint4*>(lds + LDS_W1 + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4)) = v1[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = tid + 512 * i, c = q >> 5, dch = q & 31;
      *reinterpret_cast<uint4*>(lds + LDS_W2 + c * 512 + ((dch ^ (c & 15)) << 4)) = make_uint4(v2[2 * i].x, v2[2 * i].y, v2[2 * i + 1].x, v2[2 * i + 1].y);
    }
    if (tid < H) reinterpret_cast<float*>(lds + LDS_B1)[tid] = p.b1[tid];
    if (tid < C) {
      reinterpret_cast<float*>(lds + LDS_B2)[tid] = p.b2[tid];
      if (LN) {
        reinterpret_cast<float*>(lds + LDS_G)[tid] = p.ln_gamma[tid];
        reinterpret_cast<float*>(lds + LDS_BT)[tid] = p.ln_beta[tid];
      }
    }
  }
  __syncthreads();

  MaskEval meO;
  meO.init(p.drop_o);
  MlpDropStream ds;
  ds.init(p.drop_h);
  const uint32_t ds_key = ds.s;

  // fragment addresses inside the images
  const int w1_row = l15 * 128, w1_sw = (l15 >> 1) & 7;           // + tile * 2048; chunk (4 kk + g) ^ w1_sw
  const int w2_row = l15 * 512;                                    // + ctile * 8192; chunk (4 s + g) ^ l15
  const int nwt = (p.M + 15) >> 4;
  for (int wt = blockIdx.x * 8 + wave; wt < nwt; wt += gridDim.x * 8) {
    const int m = wt * 16 + l15;
    const bool mok = m < p.M;
    const long mrow = mok ? m : p.M - 1;
    bf16x8 xa[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) xa[kk] = *reinterpret_cast<const bf16x8*>(p.a + mrow * C + kk * 32 + 8 * g);
    f32x4 yacc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) yacc[j] = f32x4{0.f, 0.f, 0.f, 0.f};

    uint32_t dst = DROP ? ds.start(ds_key, m, g) : 0u;
    uint32_t mb0 = 0u, mb1 = 0u;  // keep bits of this lane's 64 hidden units {16 T + 4 g + e}: bit 4 T + e
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {  // 64 hidden units at a time
      f32x4 u[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        u[t] = *reinterpret_cast<const f32x4*>(lds + LDS_B1 + (q * 64 + t * 16 + 4 * g) * 4);  // the accumulator starts at the bias
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const bf16x8 w = *reinterpret_cast<const bf16x8*>(lds + LDS_W1 + (q * 4 + t) * 2048 + w1_row + (((4 * kk + g) ^ w1_sw) << 4));
          u[t] = mma16(w, xa[kk], u[t]);
        }
      }
      bf16x8 hf[2];
      uint32_t kb = 0u;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
          gelu_f2 hh = mlp_gelu_fwd(gelu_f2{u[t][e], u[t][e + 1]});
          if (DROP) {
            const gelu_f2 mult = ds.next(dst);
            kb |= MlpDropStream::keep_bits(mult) << (4 * t + e);
            hh = mlp_mul2(hh, mult);
          }
          hf[t >> 1][(t & 1) * 4 + e] = (bf16_t)hh.x;
          hf[t >> 1][(t & 1) * 4 + e + 1] = (bf16_t)hh.y;
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const bf16x8 w = *reinterpret_cast<const bf16x8*>(lds + LDS_W2 + j * 8192 + w2_row + (((4 * (2 * q + s) + g) ^ l15) << 4));
          yacc[j] = mma16(w, hf[s], yacc[j]);
        }
      }
      if (DROP) {  // hidden tiles T = 4 q .. 4 q + 3: 16 bits of word T / 8
        const uint32_t sh = kb << ((q & 1) * 16);
        if (q < 2) mb0 |= sh; else mb1 |= sh;
      }
    }
    if (DROP && mok && p.mask_bits) *reinterpret_cast<uint2*>(p.mask_bits + (long)m * 8 + 2 * g) = make_uint2(mb0, mb1);

    // ---- epilogue: bias, dropout x drop-path, residual; optionally the LayerNorm that reads x_out next
    f32x4 res[4];  // (requested here, not before the chunk loop: 16 registers less across it; three other waves per SIMD cover the latency)
#pragma unroll
    for (int j = 0; j < 4; ++j) res[j] = load4(p.resid + mrow * C + 16 * j + 4 * g);
    const float rowm = meO.row_mult(m);
    float s1 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c0 = 16 * j + 4 * g;
      const f32x4 b2 = *reinterpret_cast<const f32x4*>(lds + LDS_B2 + c0 * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        yacc[j][e] = res[j][e] + (yacc[j][e] + b2[e]) * rowm * meO.elem_mult(m, c0 + e);
        s1 += yacc[j][e];
      }
      if (mok) store4(p.y + (long)m * C + c0, yacc[j]);
    }
    if (LN) {
      s1 = xadd16(s1);
      s1 = xadd32(s1);
      const float mean = s1 * (1.0f / C);
      float s2 = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) s2 += (yacc[j][e] - mean) * (yacc[j][e] - mean);
      s2 = xadd16(s2);
      s2 = xadd32(s2);
      const float rstd = rsqrtf(s2 * (1.0f / C) + p.ln_eps);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c0 = 16 * j + 4 * g;
        const f32x4 gm = *reinterpret_cast<const f32x4*>(lds + LDS_G + c0 * 4);
        const f32x4 bt = *reinterpret_cast<const f32x4*>(lds + LDS_BT + c0 * 4);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (yacc[j][e] - mean) * rstd * gm[e] + bt[e];
        if (mok) store4(p.y_ln + (long)m * C + c0, o);
      }
      if (mok && g == 0) *reinterpret_cast<float2*>(p.ln_stats + 2 * (long)m) = make_float2(mean, rstd);
    }
  }
}

}  // namespace focal_mlp_kernels
using namespace focal_mlp_kernels;

static MaskParams mlp_mask(const focal_drop_desc& d, int ncols) {
  MaskParams m;
  m.seed = d.rng;
  m.stream_elem = d.stream_elem;
  m.p_elem = d.p_elem;
  m.stream_path = d.stream_path;
  m.p_path = d.p_path;
  m.rows_per_sample = d.rows_per_sample;
  m.ncols = ncols;
  return m;
}

int mlp_check_desc(const focal_mlp_desc* d, const char* who) {
  FOCAL_CHECK_ARG(d != nullptr, "%s: null descriptor", who);
  FOCAL_CHECK_ARG(d->dtype == FOCAL_BF16, "%s: bf16 operands only (dtype %d); the exact-fp32 mode runs the unfused linear kernels", who, d->dtype);
  FOCAL_CHECK_ARG(d->C == MLP_C && d->hidden == MLP_H, "%s: C = %d, hidden = %d (the fused kernel is built for %d -> %d -> %d)", who, d->C, d->hidden, MLP_C, MLP_H, MLP_C);
  FOCAL_CHECK_ARG(d->M > 0, "%s: M = %d", who, d->M);
  return FOCAL_OK;
}

extern "C" int focal_mlp_supported(int dtype, int C_, int hidden) { return dtype == FOCAL_BF16 && C_ == MLP_C && hidden == MLP_H; }

extern "C" int focal_mlp_fwd(const focal_mlp_desc* d, const void* a, const float* resid, const void* w1, const float* b1, const void* w2,
                             const float* b2, float* y, const float* ln_gamma, const float* ln_beta, void* y_ln, float* ln_stats,
                             uint32_t* mask_bits, void* stream) {
  if (int rc = mlp_check_desc(d, "mlp_fwd")) return rc;
  FOCAL_CHECK_ARG(a && resid && w1 && b1 && w2 && b2 && y, "mlp_fwd: null tensor");
  const bool ln = y_ln != nullptr;
  if (ln) FOCAL_CHECK_ARG(ln_gamma && ln_beta && ln_stats, "mlp_fwd: the fused LayerNorm needs gamma, beta and a statistics buffer");
  MlpFwdParams p;
  p.M = d->M;
  p.a = reinterpret_cast<const bf16_t*>(a);
  p.resid = resid;
  p.w1 = reinterpret_cast<const bf16_t*>(w1);
  p.b1 = b1;
  p.w2 = reinterpret_cast<const bf16_t*>(w2);
  p.b2 = b2;
  p.y = y;
  p.drop_h = mlp_mask(d->drop_hidden, MLP_H);
  p.drop_o = mlp_mask(d->drop_out, MLP_C);
  p.ln_gamma = ln_gamma; p.ln_beta = ln_beta; p.y_ln = reinterpret_cast<bf16_t*>(y_ln); p.ln_stats = ln_stats; p.ln_eps = d->ln_eps;
  p.mask_bits = mask_bits;
  const bool drop = d->drop_hidden.p_elem > 0.f;
  void (*kern)(const MlpFwdParams) = ln ? (drop ? mlp_fwd_kernel<true, true> : mlp_fwd_kernel<true, false>)
                                        : (drop ? mlp_fwd_kernel<false, true> : mlp_fwd_kernel<false, false>);
  static std::atomic<bool> attr_set[4] = {{false}, {false}, {false}, {false}};
  const int ki = (ln ? 2 : 0) + (drop ? 1 : 0);
  if (!attr_set[ki].load(std::memory_order_acquire)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_FWD_BYTES) != hipSuccess) {
      focal_set_error("mlp_fwd: cannot reserve %d bytes of LDS", LDS_FWD_BYTES);
      return FOCAL_EHIP;
    }
    attr_set[ki].store(true, std::memory_order_release);
  }
  const int nwg = (d->M + 127) / 128;
  const int grid = nwg < 512 ? nwg : 512;  // two 8-wave workgroups per CU, persistent over 16-row wave tiles
  FOCAL_LAUNCH(kern, dim3(grid), dim3(512), LDS_FWD_BYTES, (hipStream_t)stream, p);
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}
