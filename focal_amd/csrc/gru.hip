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

// ---------------------------------------------------------------------------------------------- whole-sequence kernels
// The recurrence is 10 strictly sequential steps of a tiny product ([B,H] x [H,3H]) plus gate math: launched per step it is
// pure launch latency (~370 launches of 4-12 us per DeepSense step).  Here one launch runs the whole sequence of one layer,
// both directions (blockIdx.y): a workgroup owns 16 samples for all T steps; W_hh (bf16, 3H x H = 393 KB at H = 256) is
// read once per workgroup as MFMA operand fragments (16 B per lane, straight from its [3H][H] storage) and kept in
// registers, 49 KB per wave; the hidden
// state lives in registers (fp32, the lane that produces h[m][j] is the lane that needs it next step) with a bf16 copy in
// LDS for the other operand.  D = W-fragment x h-fragment, so a lane holds 4 consecutive hidden units of ONE sample for all
// three gates: every global access (gi, hs, out, save) is a 16-byte vector.  8 waves; wave w owns hidden units [w*H/8, (w+1)*H/8).
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 gbf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 gbf16x4;
typedef float gf32x4 __attribute__((ext_vector_type(4)));
#ifndef GRU_NW
#define GRU_NW 8
#endif
struct GruDirFwd { const float* gi; const bf16_t* whh; const float* bhh; float* hs; float* save; };
struct GruFwdArgs { GruDirFwd d[2]; };
struct GruDirBwd { const bf16_t* whh_t; const float* hs; const float* save; float* dgi; float* dgh; };
struct GruBwdArgs { GruDirBwd d[2]; };

// Gate nonlinearities for the whole-sequence kernels: raw v_rcp / v_exp (1 ulp) instead of the IEEE divide and libm tanh of
// the per-step fp32-parity kernels -- with two waves per SIMD the libm forms were ~2 us of VALU per step.
__device__ __forceinline__ float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_fast(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. waits for every global store of the
// step (hs / out / save, ~1-2 us of write latency) twice per step; nothing here communicates through global memory.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, const float* v) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }

template <int H, int NW>
__global__ __launch_bounds__(NW * 64) void gru_seq_fwd_kernel(focal_gru_desc gd, GruFwdArgs args, float* __restrict__ out) {
  constexpr int QT = H / (16 * NW), PH = H + 8;  // NW waves: wave w owns hidden units [w*H/NW, (w+1)*H/NW)
  // dynamic LDS: [n-gate W fragments: NW*QT*(H/32)*64 x 16 B][hb: 16 x PH bf16][bh: 3H f32]
  extern __shared__ __attribute__((aligned(16))) unsigned char gru_lds[];
  constexpr int KS = H / 32;
  constexpr int XS = QT > 1 ? 1 : 0;  // trailing k-steps of the z gate that also live in LDS (register budget, see below)
  gbf16x8* wl = reinterpret_cast<gbf16x8*>(gru_lds);
  gbf16x8* wx = wl + NW * QT * KS * 64;
  bf16_t* hb = reinterpret_cast<bf16_t*>(gru_lds + (size_t)NW * QT * (KS + XS) * 64 * 16);
  float* bh = reinterpret_cast<float*>(hb + 16 * PH);
  const int dir = blockIdx.y;
  const GruDirFwd p = args.d[dir];
  const int B = gd.B, T = gd.T;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lm = lane & 15, lg = lane >> 4;
  const int b = blockIdx.x * 16 + lm;
  const bool ok = b < B;
  const int jw = wave * 16 * QT;
  const long n = (long)B * H;
  float hprev[QT][4];
#pragma unroll
  for (int q = 0; q < QT; ++q)
#pragma unroll
    for (int r = 0; r < 4; ++r) hprev[q][r] = 0.f;
  // This wave's slice of W_hh (3 gates x 16*QT rows x H) is loaded ONCE and stays on chip for all T steps: the r and z
  // gates in registers (H = 256: 30 fragments = 120 VGPRs), the n gate and the z gate's last k-step in LDS in fragment
  // order.  The split is set by the register budget: ONE spilled fragment costs an `s_waitcnt vmcnt(0)` per step, i.e. the
  // acknowledgement of the previous step's 12 result stores (spill reloads are vector-memory loads): 7 us / step.
  // Streaming W_hh from L2 every step instead left the kernel latency-bound at 17 us / step.
  gbf16x8 wreg[2][QT][KS];
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int q = 0; q < QT; ++q)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const gbf16x8 w = *reinterpret_cast<const gbf16x8*>(p.whh + (long)(g * H + jw + 16 * q + lm) * H + 32 * ks + 8 * lg);
        if (g == 2) wl[((wave * QT + q) * KS + ks) * 64 + lane] = w;
        else if (g == 1 && ks >= KS - XS) wx[((wave * QT + q) * XS + ks - (KS - XS)) * 64 + lane] = w;
        else wreg[g][q][ks] = w;
      }
  // Vector-memory results return in issue order, loads and stores alike: a load issued AFTER a step's 12 result stores cannot
  // be consumed before those stores are acknowledged (microseconds).  So the input projections of step s+1 are requested
  // BEFORE the stores of step s, and b_hh sits in LDS (ds_read does not queue behind vector memory).
  for (int i = threadIdx.x; i < 3 * H; i += NW * 64) bh[i] = p.bhh[i];
  float4 gr4[QT], gz4[QT], gn4[QT];
  auto fetch_gi = [&](int s_) {
    const int t_ = dir ? T - 1 - s_ : s_;
#pragma unroll
    for (int q = 0; q < QT; ++q) {
      const float* gir = p.gi + ((long)b * T + t_) * 3 * H + jw + 16 * q + 4 * lg;
      const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
      const bool go = ok && s_ < T;
      gr4[q] = go ? ld4(gir) : z4; gz4[q] = go ? ld4(gir + H) : z4; gn4[q] = go ? ld4(gir + 2 * H) : z4;
    }
  };
  fetch_gi(0);
  lds_barrier();
  for (int s = 0; s < T; ++s) {
    const int t = dir ? T - 1 - s : s;
    gf32x4 acc[3][QT];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int q = 0; q < QT; ++q) acc[g][q] = gf32x4{0.f, 0.f, 0.f, 0.f};
    if (s > 0) {  // h_0 = 0: the first step's recurrent product is just the bias
#pragma unroll
      for (int ks = 0; ks < H / 32; ++ks) {
        const gbf16x8 hf = *reinterpret_cast<const gbf16x8*>(hb + lm * PH + 32 * ks + 8 * lg);
#pragma unroll
        for (int q = 0; q < QT; ++q) {
          acc[0][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[0][q][ks], hf, acc[0][q], 0, 0, 0);
          acc[1][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ks >= KS - XS ? wx[((wave * QT + q) * XS + (ks >= KS - XS ? ks - (KS - XS) : 0)) * 64 + lane] : wreg[1][q][ks], hf, acc[1][q], 0, 0, 0);
          acc[2][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[((wave * QT + q) * KS + ks) * 64 + lane], hf, acc[2][q], 0, 0, 0);
        }
      }
    }
    float hnew[QT][4], rr[QT][4], zz[QT][4], nn[QT][4], gh[QT][4];
#pragma unroll
    for (int q = 0; q < QT; ++q) {
      const int j0 = jw + 16 * q + 4 * lg;
      const float4 gr = gr4[q], gz = gz4[q], gn = gn4[q];
      const float4 br = ld4(bh + j0), bz = ld4(bh + H + j0), bn = ld4(bh + 2 * H + j0);
      const float gra[4] = {gr.x, gr.y, gr.z, gr.w}, gza[4] = {gz.x, gz.y, gz.z, gz.w}, gna[4] = {gn.x, gn.y, gn.z, gn.w};
      const float bra[4] = {br.x, br.y, br.z, br.w}, bza[4] = {bz.x, bz.y, bz.z, bz.w}, bna[4] = {bn.x, bn.y, bn.z, bn.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        rr[q][r] = sigmoid_fast(gra[r] + acc[0][q][r] + bra[r]);
        zz[q][r] = sigmoid_fast(gza[r] + acc[1][q][r] + bza[r]);
        gh[q][r] = acc[2][q][r] + bna[r];
        nn[q][r] = tanh_fast(gna[r] + rr[q][r] * gh[q][r]);
        hnew[q][r] = (1.f - zz[q][r]) * nn[q][r] + zz[q][r] * hprev[q][r];
        hprev[q][r] = hnew[q][r];
      }
    }
    fetch_gi(s + 1);
    if (ok) {
#pragma unroll
      for (int q = 0; q < QT; ++q) {
        const int j0 = jw + 16 * q + 4 * lg;
        const long e = (long)b * H + j0;
        st4(p.hs + (long)(s + 1) * n + e, hnew[q]);
        st4(out + ((long)b * T + t) * 2 * H + dir * H + j0, hnew[q]);
        float* sv = p.save + (long)s * 4 * n + e;
        st4(sv, rr[q]); st4(sv + n, zz[q]); st4(sv + 2 * n, nn[q]); st4(sv + 3 * n, gh[q]);
      }
    }
    lds_barrier();  // every wave has read this step's h fragments
#pragma unroll
    for (int q = 0; q < QT; ++q) {
      gbf16x4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = (__bf16)(ok ? hnew[q][r] : 0.f);
      *reinterpret_cast<gbf16x4*>(hb + lm * PH + jw + 16 * q + 4 * lg) = v;
    }
    lds_barrier();
  }
}

template <int H, int NW>
__global__ __launch_bounds__(NW * 64) void gru_seq_bwd_kernel(focal_gru_desc gd, GruBwdArgs args, const float* __restrict__ dout, long ld_b,
                                                              long ld_t, float scale) {
  constexpr int QT = H / (16 * NW), PG = 3 * H + 8;
  constexpr int KS = 3 * H / 32, KR = 2 * KS / 3;  // k-steps of the product; the first KR live in registers, the rest in LDS
  extern __shared__ __attribute__((aligned(16))) unsigned char gru_lds[];
  gbf16x8* wl = reinterpret_cast<gbf16x8*>(gru_lds);
  bf16_t* gb = reinterpret_cast<bf16_t*>(gru_lds + (size_t)NW * QT * (KS - KR) * 64 * 16);
  const int dir = blockIdx.y;
  const GruDirBwd p = args.d[dir];
  const int B = gd.B, T = gd.T;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lm = lane & 15, lg = lane >> 4;
  const int b = blockIdx.x * 16 + lm;
  const bool ok = b < B;
  const int jw = wave * 16 * QT;
  const long n = (long)B * H;
  float dhz[QT][4], dhrec[QT][4];
#pragma unroll
  for (int q = 0; q < QT; ++q)
#pragma unroll
    for (int r = 0; r < 4; ++r) dhz[q][r] = dhrec[q][r] = 0.f;
  gbf16x8 wreg[QT][KR];  // this wave's rows of W_hh^T, resident for all steps (registers + LDS, as in the forward kernel)
#pragma unroll
  for (int q = 0; q < QT; ++q)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const gbf16x8 w = *reinterpret_cast<const gbf16x8*>(p.whh_t + (long)(jw + 16 * q + lm) * 3 * H + 32 * ks + 8 * lg);
      if (ks < KR) wreg[q][ks] = w;
      else wl[((wave * QT + q) * (KS - KR) + ks - KR) * 64 + lane] = w;
    }
  for (int s = T - 1; s >= 0; --s) {
    const int t = dir ? T - 1 - s : s;
#pragma unroll
    for (int q = 0; q < QT; ++q) {
      const int j0 = jw + 16 * q + 4 * lg;
      const long e = (long)b * H + j0;
      const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 d4 = ok ? ld4(dout + (long)b * ld_b + (long)t * ld_t + dir * H + j0) : z4;
      const float* sv = p.save + (long)s * 4 * n + e;
      const float4 r4 = ok ? ld4(sv) : z4, zz4 = ok ? ld4(sv + n) : z4, n4 = ok ? ld4(sv + 2 * n) : z4, g4 = ok ? ld4(sv + 3 * n) : z4;
      const float4 h4 = ok ? ld4(p.hs + (long)s * n + e) : z4;
      const float da[4] = {d4.x, d4.y, d4.z, d4.w}, ra[4] = {r4.x, r4.y, r4.z, r4.w}, za[4] = {zz4.x, zz4.y, zz4.z, zz4.w};
      const float na[4] = {n4.x, n4.y, n4.z, n4.w}, ga[4] = {g4.x, g4.y, g4.z, g4.w}, ha[4] = {h4.x, h4.y, h4.z, h4.w};
      float dr[4], dz[4], dn[4], dnr[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float dh = scale * da[r] + dhrec[q][r] + dhz[q][r];
        dn[r] = dh * (1.f - za[r]) * (1.f - na[r] * na[r]);
        dz[r] = dh * (ha[r] - na[r]) * za[r] * (1.f - za[r]);
        dr[r] = dn[r] * ga[r] * ra[r] * (1.f - ra[r]);
        dnr[r] = dn[r] * ra[r];
        dhz[q][r] = dh * za[r];
      }
      if (ok) {
        float* gir = p.dgi + ((long)b * T + t) * 3 * H + j0;
        st4(gir, dr); st4(gir + H, dz); st4(gir + 2 * H, dn);
        float* ghr = p.dgh + ((long)s * B + b) * 3 * H + j0;
        st4(ghr, dr); st4(ghr + H, dz); st4(ghr + 2 * H, dnr);
      }
      gbf16x4 v0, v1, v2;
#pragma unroll
      for (int r = 0; r < 4; ++r) { v0[r] = (__bf16)dr[r]; v1[r] = (__bf16)dz[r]; v2[r] = (__bf16)dnr[r]; }
      bf16_t* row = gb + lm * PG + j0;
      *reinterpret_cast<gbf16x4*>(row) = v0;
      *reinterpret_cast<gbf16x4*>(row + H) = v1;
      *reinterpret_cast<gbf16x4*>(row + 2 * H) = v2;
    }
    if (s > 0) {  // dh_{s-1} += dgh_s . W_hh  (through the [H][3H] transposed copy: 16 contiguous bytes per lane again)
      lds_barrier();
      gf32x4 acc[QT];
#pragma unroll
      for (int q = 0; q < QT; ++q) acc[q] = gf32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const gbf16x8 gf = *reinterpret_cast<const gbf16x8*>(gb + lm * PG + 32 * ks + 8 * lg);
#pragma unroll
        for (int q = 0; q < QT; ++q) {
          const gbf16x8 w = ks < KR ? wreg[q][ks < KR ? ks : 0] : wl[((wave * QT + q) * (KS - KR) + (ks < KR ? 0 : ks - KR)) * 64 + lane];
          acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, gf, acc[q], 0, 0, 0);
        }
      }
#pragma unroll
      for (int q = 0; q < QT; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) dhrec[q][r] = acc[q][r];
      lds_barrier();
    }
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
  FOCAL_LAUNCH(gru_gate_fwd_kernel, dim3(gblocks((long)d->B * d->H)), dim3(256), 0, (hipStream_t)stream, *d, t, dir_offset, gi, gh,
                     h_prev, h_new, out, save);
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}

extern "C" int focal_gru_gate_bwd(const focal_gru_desc* d, int t, int dir_offset, const float* dout, long ld_b, long ld_t, float scale,
                                  const float* dh_rec, const float* dhz_in, const float* save, const float* h_prev, float* dgi,
                                  float* dgh, float* dhz_out, void* stream) {
  FOCAL_CHECK_ARG(d && dout && save && dgi && dgh && dhz_out && t >= 0 && t < d->T, "gru_gate_bwd: bad argument");
  FOCAL_LAUNCH(gru_gate_bwd_kernel, dim3(gblocks((long)d->B * d->H)), dim3(256), 0, (hipStream_t)stream, *d, t, dir_offset, dout,
                     ld_b, ld_t, scale, dh_rec, dhz_in, save, h_prev, dgi, dgh, dhz_out);
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}

extern "C" int focal_mean_time(int B, int T, int D, const float* x, float* y, void* stream) {
  FOCAL_CHECK_ARG(x && y && B > 0 && T > 0 && D > 0, "mean_time: bad argument");
  FOCAL_LAUNCH(mean_time_kernel, dim3(gblocks((long)B * D)), dim3(256), 0, (hipStream_t)stream, B, T, D, x, y);
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}

extern "C" int focal_dropout(long n, const float* x, float* y, const uint32_t* rng, uint32_t stream_id, float p, void* stream) {
  FOCAL_CHECK_ARG(x && y && n >= 0 && p >= 0.f && p < 1.f, "dropout: bad argument");
  FOCAL_LAUNCH(dropout_kernel, dim3(gblocks(n)), dim3(256), 0, (hipStream_t)stream, n, x, y, rng, stream_id, p);
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}

__global__ __launch_bounds__(256) void mul_kernel(long n, const float* __restrict__ a, float* __restrict__ y) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] *= a[i];
}
extern "C" int focal_mul(long n, const float* a, float* y, void* stream) {
  FOCAL_CHECK_ARG(a && y && n >= 0, "mul: bad argument");
  FOCAL_LAUNCH(mul_kernel, dim3(gblocks(n)), dim3(256), 0, (hipStream_t)stream, n, a, y);
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}

extern "C" int focal_axpy(long n, float a, const float* x, float* y, void* stream) {
  FOCAL_CHECK_ARG(x && y && n >= 0, "axpy: bad argument");
  FOCAL_LAUNCH(axpy_kernel, dim3(gblocks(n)), dim3(256), 0, (hipStream_t)stream, n, a, x, y);
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}

extern "C" int focal_gru_seq_fwd(const focal_gru_desc* d, int n_dir, const float* const* gi, const void* const* whh, const float* const* bhh,
                                 float* const* hs, float* const* save, float* out, void* stream) {
  FOCAL_CHECK_ARG(d && gi && whh && bhh && hs && save && out && n_dir >= 1 && n_dir <= 2, "gru_seq_fwd: bad argument");
  if (d->H != 128 && d->H != 256) {
    focal_set_error("gru_seq_fwd: hidden size %d not in {128, 256} (use the per-step kernels)", d->H);
    return FOCAL_EUNSUPPORTED;
  }
  GruFwdArgs a;
  memset(&a, 0, sizeof(a));
  for (int i = 0; i < n_dir; ++i) {
    FOCAL_CHECK_ARG(gi[i] && whh[i] && bhh[i] && hs[i] && save[i], "gru_seq_fwd: null tensor");
    a.d[i] = GruDirFwd{gi[i], (const bf16_t*)whh[i], bhh[i], hs[i], save[i]};
  }
  const dim3 grid(ceil_div(d->B, 16), n_dir);
  const int H = d->H;
  const size_t lds = (size_t)(H / 16) * (H / 32 + (H > 128 ? 1 : 0)) * 64 * 16 + (size_t)16 * (H + 8) * 2 + (size_t)3 * H * 4;
  static bool granted = false;
  if (!granted) {  // up to 143 KB of the CU's 160 KB: above the default dynamic-LDS grant
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gru_seq_fwd_kernel<256, GRU_NW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(gru_seq_fwd_kernel<128, GRU_NW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { focal_set_error("gru_seq_fwd: cannot reserve LDS: %s", hipGetErrorString(e)); return FOCAL_EHIP; }
    granted = true;
  }
  if (H == 256) FOCAL_LAUNCH((gru_seq_fwd_kernel<256, GRU_NW>), grid, dim3(GRU_NW * 64), lds, (hipStream_t)stream, *d, a, out);
  else FOCAL_LAUNCH((gru_seq_fwd_kernel<128, GRU_NW>), grid, dim3(GRU_NW * 64), lds, (hipStream_t)stream, *d, a, out);
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}

extern "C" int focal_gru_seq_bwd(const focal_gru_desc* d, int n_dir, const float* dout, long ld_b, long ld_t, float scale,
                                 const void* const* whh_t, const float* const* hs, const float* const* save, float* const* dgi,
                                 float* const* dgh, void* stream) {
  FOCAL_CHECK_ARG(d && dout && whh_t && hs && save && dgi && dgh && n_dir >= 1 && n_dir <= 2, "gru_seq_bwd: bad argument");
  FOCAL_CHECK_ARG(ld_b % 4 == 0 && ld_t % 4 == 0, "gru_seq_bwd: dout strides must be multiples of 4");
  if (d->H != 128 && d->H != 256) {
    focal_set_error("gru_seq_bwd: hidden size %d not in {128, 256} (use the per-step kernels)", d->H);
    return FOCAL_EUNSUPPORTED;
  }
  GruBwdArgs a;
  memset(&a, 0, sizeof(a));
  for (int i = 0; i < n_dir; ++i) {
    FOCAL_CHECK_ARG(whh_t[i] && hs[i] && save[i] && dgi[i] && dgh[i], "gru_seq_bwd: null tensor");
    a.d[i] = GruDirBwd{(const bf16_t*)whh_t[i], hs[i], save[i], dgi[i], dgh[i]};
  }
  const dim3 grid(ceil_div(d->B, 16), n_dir);
  const int H = d->H;
  const size_t lds = (size_t)(H / 16) * (H / 32) * 64 * 16 + (size_t)16 * (3 * H + 8) * 2;
  static bool granted = false;
  if (!granted) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gru_seq_bwd_kernel<256, GRU_NW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(gru_seq_bwd_kernel<128, GRU_NW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { focal_set_error("gru_seq_bwd: cannot reserve LDS: %s", hipGetErrorString(e)); return FOCAL_EHIP; }
    granted = true;
  }
  if (H == 256) FOCAL_LAUNCH((gru_seq_bwd_kernel<256, GRU_NW>), grid, dim3(GRU_NW * 64), lds, (hipStream_t)stream, *d, a, dout, ld_b, ld_t, scale);
  else FOCAL_LAUNCH((gru_seq_bwd_kernel<128, GRU_NW>), grid, dim3(GRU_NW * 64), lds, (hipStream_t)stream, *d, a, dout, ld_b, ld_t, scale);
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}
