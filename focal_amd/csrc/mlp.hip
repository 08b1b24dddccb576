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
              LDS_FWD_BYTES = LDS_BT + 256,
              // PROJ: + the proj weight image (64 rows x 128 B, rows permuted: see the kernel), its bias, norm2's gamma / beta
              LDS_WP = LDS_FWD_BYTES, LDS_BP = LDS_WP + 8192, LDS_G2 = LDS_BP + 256, LDS_BT2 = LDS_G2 + 256, LDS_FWD_PROJ_BYTES = LDS_BT2 + 256;

// PROJ (round 6): the 64-channel proj Linear + residual + DropPath + norm2 of the attention branch (SwinModules.py:147, :336-339 -- what
// focal_linear_resid_ln_fwd runs as a launch of its own) in front of the MLP, per 16-row wave tile: o's fragments x the proj weight (rows
// permuted so that a lane's 2 x 4 accumulators are 8 consecutive channels), the residual epilogue and the LayerNorm with that kernel's
// arithmetic and summation tree (bit-identical with the masks off), x_mid / a2 / statistics stored for the backward pass, and a2 goes on to
// fc1 in registers.  4 launches fewer per step, x_mid and a2 not re-read.
template <bool LN, bool DROP, bool PROJ>
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
    if constexpr (PROJ) {
      // LDS row 16 T + r of the proj image <- output channel 32 (T / 2) + 8 (r / 4) + 4 (T % 2) + r % 4; 8 chunks of 16 B per row, swizzled
      const int row = tid >> 3, ch = tid & 7, T = row >> 4, r = row & 15;
      const int oc = 32 * (T >> 1) + 8 * (r >> 2) + 4 * (T & 1) + (r & 3);
      *reinterpret_cast<uint4*>(lds + LDS_WP + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4)) = *reinterpret_cast<const uint4*>(p.wp + oc * C + ch * 8);
      if (tid < C) {
        reinterpret_cast<float*>(lds + LDS_BP)[tid] = p.bp[tid];
        reinterpret_cast<float*>(lds + LDS_G2)[tid] = p.g2[tid];
        reinterpret_cast<float*>(lds + LDS_BT2)[tid] = p.bt2[tid];
      }
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

  MaskEval meO, meP;
  meO.init(p.drop_o);
  if constexpr (PROJ) meP.init(p.drop_p);
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
    if constexpr (!PROJ) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) xa[kk] = *reinterpret_cast<const bf16x8*>(p.a + mrow * C + kk * 32 + 8 * g);
    } else {
      // ---- x_mid = x + drop_p(o Wp^T + bp); a2 = norm2(x_mid).  Lane = token, channels 32 s + 8 g .. + 7 for s = 0, 1.
      bf16x8 oa[2];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) oa[kk] = *reinterpret_cast<const bf16x8*>(p.o + mrow * C + kk * 32 + 8 * g);
      float xr[2][8];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const f32x4 r0 = load4(p.x + mrow * C + 32 * s + 8 * g), r1 = load4(p.x + mrow * C + 32 * s + 8 * g + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { xr[s][e] = r0[e]; xr[s][4 + e] = r1[e]; }
      }
      f32x4 pacc[4];
#pragma unroll
      for (int T = 0; T < 4; ++T) {
        pacc[T] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const bf16x8 w = *reinterpret_cast<const bf16x8*>(lds + LDS_WP + T * 2048 + w1_row + (((4 * kk + g) ^ w1_sw) << 4));
          pacc[T] = mma16(w, oa[kk], pacc[T]);
        }
      }
      const float rowm = meP.row_mult(m);
      float xm[2][8];
      float part[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int c0 = 32 * s + 8 * g;
        const f32x4 bq0 = *reinterpret_cast<const f32x4*>(lds + LDS_BP + c0 * 4), bq1 = *reinterpret_cast<const f32x4*>(lds + LDS_BP + (c0 + 4) * 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float v = (e < 4 ? pacc[2 * s][e] : pacc[2 * s + 1][e - 4]) + (e < 4 ? bq0[e] : bq1[e - 4]);
          float t;
          {
#pragma clang fp contract(off)
            t = v * rowm;
          }
          xm[s][e] = __builtin_fmaf(t, meP.elem_mult(m, c0 + e), xr[s][e]);
        }
        if (mok) {
          store4(p.resid_out() + (long)m * C + c0, f32x4{xm[s][0], xm[s][1], xm[s][2], xm[s][3]});
          store4(p.resid_out() + (long)m * C + c0 + 4, f32x4{xm[s][4], xm[s][5], xm[s][6], xm[s][7]});
        }
      }
      // the LayerNorm of focal_linear_resid_ln_fwd's epilogue: there lane li of a 16-lane row holds channels 4 li .. + 3 and row16_sum joins
      // li's bits 0, 1, 2, 3 in that order; here li = 8 s + 2 g + h (h = which half of the lane's 8 channels): bit 0 = h, bits 1 / 2 = lane
      // bits 4 / 5, bit 3 = s
      auto tree = [&](float (&q)[2][2]) __attribute__((always_inline)) {
        float t0 = q[0][0] + q[0][1], t1 = q[1][0] + q[1][1];
        t0 = xadd32(xadd16(t0));
        t1 = xadd32(xadd16(t1));
        return t0 + t1;
      };
      float q1[2][2], q2[2][2];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int h = 0; h < 2; ++h) q1[s][h] = ((xm[s][4 * h] + xm[s][4 * h + 1]) + xm[s][4 * h + 2]) + xm[s][4 * h + 3];
      const float mean = tree(q1) * (1.0f / 64.0f);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma clang fp contract(off)
          const float d0 = xm[s][4 * h] - mean, d1 = xm[s][4 * h + 1] - mean, d2 = xm[s][4 * h + 2] - mean, d3 = xm[s][4 * h + 3] - mean;
          q2[s][h] = ((d0 * d0 + d1 * d1) + d2 * d2) + d3 * d3;
        }
      const float rstd = rsqrtf(__builtin_fmaf(tree(q2), 1.0f / 64.0f, p.ln_eps));
      (void)part;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int c0 = 32 * s + 8 * g;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float t;
          {
#pragma clang fp contract(off)
            t = (xm[s][e] - mean) * rstd;
          }
          xa[s][e] = (bf16_t)__builtin_fmaf(reinterpret_cast<const float*>(lds + LDS_G2)[c0 + e], t, reinterpret_cast<const float*>(lds + LDS_BT2)[c0 + e]);
        }
        if (mok) *reinterpret_cast<bf16x8*>(p.a_out() + (long)m * C + c0) = xa[s];
      }
      if (mok && g == 0) *reinterpret_cast<float2*>(p.st2 + 2 * (long)m) = make_float2(mean, rstd);
      // (x_mid's rows are read back by OTHER lanes of this wave in the epilogue below -- a lane's eight channels are not the four per column
      // tile it adds there: the wave's vector-memory operations stay in program order, this keeps the compiler from moving the loads up)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
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

struct MlpProjArgs { const void* o; const float* x; const void* wp; const float* bp; const focal_drop_desc* drop; const float* g2; const float* bt2; float* st2; };

static int mlp_fwd_launch(const char* who, const focal_mlp_desc* d, const void* a, const float* resid, const void* w1, const float* b1, const void* w2,
                          const float* b2, float* y, const float* ln_gamma, const float* ln_beta, void* y_ln, float* ln_stats,
                          uint32_t* mask_bits, const MlpProjArgs* pj, void* stream) {
  if (int rc = mlp_check_desc(d, who)) return rc;
  FOCAL_CHECK_ARG(a && resid && w1 && b1 && w2 && b2 && y, "%s: null tensor", who);
  const bool ln = y_ln != nullptr;
  if (ln) FOCAL_CHECK_ARG(ln_gamma && ln_beta && ln_stats, "%s: the fused LayerNorm needs gamma, beta and a statistics buffer", who);
  MlpFwdParams p;
  memset(&p, 0, sizeof(p));
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
  if (pj) {
    FOCAL_CHECK_ARG(pj->o && pj->x && pj->wp && pj->bp && pj->g2 && pj->bt2 && pj->st2, "%s: null proj / norm2 tensor", who);
    FOCAL_CHECK_ARG(((uintptr_t)pj->wp | (uintptr_t)pj->o | (uintptr_t)pj->x | (uintptr_t)a | (uintptr_t)resid) % 16 == 0, "%s: 16-byte aligned operands", who);
    p.o = reinterpret_cast<const bf16_t*>(pj->o);
    p.x = pj->x;
    p.wp = reinterpret_cast<const bf16_t*>(pj->wp);
    p.bp = pj->bp;
    focal_drop_desc dd;
    memset(&dd, 0, sizeof(dd));
    if (pj->drop) dd = *pj->drop;
    p.drop_p = mlp_mask(dd, MLP_C);
    p.g2 = pj->g2; p.bt2 = pj->bt2; p.st2 = pj->st2;
  }
  const bool drop = d->drop_hidden.p_elem > 0.f;
  void (*kern)(const MlpFwdParams);
  if (pj) kern = ln ? (drop ? mlp_fwd_kernel<true, true, true> : mlp_fwd_kernel<true, false, true>) : (drop ? mlp_fwd_kernel<false, true, true> : mlp_fwd_kernel<false, false, true>);
  else kern = ln ? (drop ? mlp_fwd_kernel<true, true, false> : mlp_fwd_kernel<true, false, false>) : (drop ? mlp_fwd_kernel<false, true, false> : mlp_fwd_kernel<false, false, false>);
  const int lds_bytes = pj ? LDS_FWD_PROJ_BYTES : LDS_FWD_BYTES;
  static std::atomic<bool> attr_set[8] = {{false}, {false}, {false}, {false}, {false}, {false}, {false}, {false}};
  const int ki = (pj ? 4 : 0) + (ln ? 2 : 0) + (drop ? 1 : 0);
  if (!attr_set[ki].load(std::memory_order_acquire)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) {
      focal_set_error("%s: cannot reserve %d bytes of LDS", who, lds_bytes);
      return FOCAL_EHIP;
    }
    attr_set[ki].store(true, std::memory_order_release);
  }
  const int nwg = (d->M + 127) / 128;
  const int grid = nwg < 512 ? nwg : 512;  // two 8-wave workgroups per CU, persistent over 16-row wave tiles
  FOCAL_LAUNCH(kern, dim3(grid), dim3(512), lds_bytes, (hipStream_t)stream, p);
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}

extern "C" int focal_mlp_fwd(const focal_mlp_desc* d, const void* a, const float* resid, const void* w1, const float* b1, const void* w2,
                             const float* b2, float* y, const float* ln_gamma, const float* ln_beta, void* y_ln, float* ln_stats,
                             uint32_t* mask_bits, void* stream) {
  return mlp_fwd_launch("mlp_fwd", d, a, resid, w1, b1, w2, b2, y, ln_gamma, ln_beta, y_ln, ln_stats, mask_bits, nullptr, stream);
}

// FOCAL_MLP_PROJ: unset = every width; "0" = none; otherwise the comma-separated widths that keep the fold ("64", "64,128", ...)
bool focal_mlp_proj_width_enabled(int C_) {
  const char* sel = getenv("FOCAL_MLP_PROJ");
  if (sel == nullptr || sel[0] == 0) return true;
  for (const char* p = sel; *p != 0;) {
    if (atoi(p) == C_ && C_ != 0) return true;
    while (*p != 0 && *p != ',') ++p;
    if (*p == ',') ++p;
  }
  return false;
}

extern "C" int focal_mlp_proj_supported(int dtype, int C_, int hidden) {
  return focal_mlp_supported(dtype, C_, hidden) && focal_mlp_proj_width_enabled(C_);
}

extern "C" int focal_mlp_proj_fwd(const focal_mlp_desc* d, const void* o, const float* x, const void* wp, const float* bp, const focal_drop_desc* drop_proj,
                                  const float* g2, const float* bt2, float* x_mid, void* a2, float* st2, const void* w1, const float* b1, const void* w2,
                                  const float* b2, float* y, const float* ln_gamma, const float* ln_beta, void* y_ln, float* ln_stats,
                                  uint32_t* mask_bits, void* stream) {
  const MlpProjArgs pj = {o, x, wp, bp, drop_proj, g2, bt2, st2};
  return mlp_fwd_launch("mlp_proj_fwd", d, a2, x_mid, w1, b1, w2, b2, y, ln_gamma, ln_beta, y_ln, ln_stats, mask_bits, &pj, stream);
}
