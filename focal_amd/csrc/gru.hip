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
#define GRU_NW_128 (GRU_NW > 8 ? 8 : GRU_NW)  // H = 128: a wave owns at least one 16-wide tile of hidden units
#define GRU_TW (GRU_NW == 8 ? 2 : 1)          // H = 256 at 8 waves: two tiles per wave, which two lanes per sample can split
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
// Sequence kernels: a uniform base (scalar registers: everything that depends on the step) + a loop-invariant 32-bit lane offset.  Six 64-bit
// lane pointers per step were 14 vector registers -- the difference between fitting W_hh's resident fragments and spilling one, and a spill
// reload is a vector-memory load: its `s_waitcnt vmcnt(0)` waits for the acknowledgement of the step's 12 result stores (2.7 us per step).
__device__ __forceinline__ float4 ld4u(const void* base, uint32_t off) { return *reinterpret_cast<const float4*>(static_cast<const char*>(base) + off); }
__device__ __forceinline__ void st4u(void* base, uint32_t off, const float* v) {
  *reinterpret_cast<float4*>(static_cast<char*>(base) + off) = make_float4(v[0], v[1], v[2], v[3]);
}

// TW (1 or 2) lanes per sample and tile column -- "twins".  The 16 columns of the MFMA tile hold 16 / TW samples, each TW times; the copies cost
// nothing on the matrix cores (16 columns is the minimum tile) and the grid grows TW-fold.  What that buys: the step is bound by the CU's
// vector-memory front end, not by bytes -- a 16-byte-per-lane access whose samples lie 1 KB apart is 16 requests whatever the lanes hold, 18 such
// instructions per wave and step took 2.6 of the 4.1 us per step, and the time per launch is flat from 8 to 128 workgroups
// (profiles/r4_gru_step_ablation.txt).  With TW = 2 (two tiles of hidden units per wave) a lane finishes ONE of the wave's two tiles for its
// sample -- the twin the other: half the gate math, half the loads and stores per CU, twice the CUs.  Nothing is masked or branched: every lane
// of every memory instruction does useful work.
template <int TW> __device__ __forceinline__ gf32x4 gru_own(const gf32x4& a0, const gf32x4& a1, int twin) {
  if constexpr (TW == 1) return a0;
  gf32x4 o;
#pragma unroll
  for (int r = 0; r < 4; ++r) o[r] = twin ? a1[r] : a0[r];
  return o;
}

template <int H, int NW, int TW>
__global__ __launch_bounds__(NW * 64) void gru_seq_fwd_kernel(focal_gru_desc gd, GruFwdArgs args, float* __restrict__ out) {
  constexpr int QT = H / (16 * NW), PH = H + 8;  // NW waves: wave w owns hidden units [w*H/NW, (w+1)*H/NW)
  constexpr int QO = QT / TW, SPB = 16 / TW;       // tiles a lane finishes; samples per workgroup
  static_assert(TW == 1 || QT == 2, "twins split the wave's two tiles");
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
  const int twin = TW == 2 ? lm >> 3 : 0;
  // No vector-memory instruction of the step loop sits under a condition: a ragged last tile clamps the sample index, its spare lanes then
  // recompute sample B - 1 and store the same values to the same addresses.  A branch around the step's stores makes the compiler's
  // wait-count pass assume the shorter path at the join -- `s_waitcnt vmcnt(0)` before the next step's prefetched input projections are
  // consumed, i.e. a wait for the acknowledgement of all the step's result stores (microseconds) in every step.
  const int b = min((int)blockIdx.x * SPB + (lm & (SPB - 1)), B - 1);
  const int jw = wave * 16 * QT;
  const int jo = jw + 16 * twin + 4 * lg;  // first of this lane's 4 hidden units in its (first) own tile
  const long n = (long)B * H;
  float hprev[QO][4];
#pragma unroll
  for (int q = 0; q < QO; ++q)
#pragma unroll
    for (int r = 0; r < 4; ++r) hprev[q][r] = 0.f;
  // This wave's slice of W_hh (3 gates x 16*QT rows x H) is loaded ONCE and stays on chip for all T steps: the r and z
  // gates in registers (H = 256: 30 fragments = 120 VGPRs), the n gate and the z gate's last k-step in LDS in fragment
  // order.  The split is set by the register budget: ONE spilled fragment costs an `s_waitcnt vmcnt(0)` per step, i.e. the
  // acknowledgement of the previous step's result stores (spill reloads are vector-memory loads).
  // Streaming W_hh from L2 every step instead left the kernel latency-bound at 17 us / step.
  gbf16x8 wreg[2][QT][KS];
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int q = 0; q < QT; ++q)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        // (fragment order, focal_gru_desc.whh_frag: row tile g H / 16 + wave QT + q, k-step ks -> one contiguous KB per load instruction)
        const gbf16x8 w = gd.whh_frag ? *reinterpret_cast<const gbf16x8*>(p.whh + (((long)(g * (H / 16) + wave * QT + q) * KS + ks) * 64 + lane) * 8)
                                      : *reinterpret_cast<const gbf16x8*>(p.whh + (long)(g * H + jw + 16 * q + lm) * H + 32 * ks + 8 * lg);
        if (g == 2) wl[((wave * QT + q) * KS + ks) * 64 + lane] = w;
        else if (g == 1 && ks >= KS - XS) wx[((wave * QT + q) * XS + ks - (KS - XS)) * 64 + lane] = w;
        else wreg[g][q][ks] = w;
      }
  // Vector-memory results return in issue order, loads and stores alike: a load issued AFTER a step's result stores cannot
  // be consumed before those stores are acknowledged (microseconds).  So the input projections of step s+1 are requested
  // BEFORE the stores of step s, and b_hh sits in LDS (ds_read does not queue behind vector memory).
  for (int i = threadIdx.x; i < 3 * H; i += NW * 64) bh[i] = p.bhh[i];
  float4 gr4[QO], gz4[QO], gn4[QO];
  // (a uniform base + a loop-invariant 32-bit lane offset per tensor; a second own tile, TW = 1: + 64 B)
  const uint32_t eoff = ((uint32_t)b * H + jo) * 4u;                          // this lane in a [B][H] plane (hs, save)
  const uint32_t ooff = ((uint32_t)b * T * 2 * H + dir * H + jo) * 4u;        // ... in out [B][T][2H] at t = 0
  const uint32_t goff = ((uint32_t)b * T * 3 * H + jo) * 4u;                  // ... in gi [B][T][3H] at t = 0
  auto fetch_gi = [&](int s_) {
    const int sc = s_ < T ? s_ : T - 1;  // (past the last step: a redundant reload instead of a branch)
    const int t_ = dir ? T - 1 - sc : sc;
    const char* gbase = reinterpret_cast<const char*>(p.gi) + (long)t_ * 3 * H * 4;
#pragma unroll
    for (int q = 0; q < QO; ++q) {
      gr4[q] = ld4u(gbase, goff + 64u * q); gz4[q] = ld4u(gbase + H * 4, goff + 64u * q); gn4[q] = ld4u(gbase + 2 * H * 4, goff + 64u * q);
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
    float hnew[QO][4], rr[QO][4], zz[QO][4], nn[QO][4], gh[QO][4];
#pragma unroll
    for (int q = 0; q < QO; ++q) {
      const int j0 = jo + 16 * q;
      const gf32x4 ar = TW == 2 ? gru_own<TW>(acc[0][0], acc[0][QT - 1], twin) : acc[0][q];
      const gf32x4 az = TW == 2 ? gru_own<TW>(acc[1][0], acc[1][QT - 1], twin) : acc[1][q];
      const gf32x4 an = TW == 2 ? gru_own<TW>(acc[2][0], acc[2][QT - 1], twin) : acc[2][q];
      const float4 gr = gr4[q], gz = gz4[q], gn = gn4[q];
      const float4 br = ld4(bh + j0), bz = ld4(bh + H + j0), bn = ld4(bh + 2 * H + j0);
      const float gra[4] = {gr.x, gr.y, gr.z, gr.w}, gza[4] = {gz.x, gz.y, gz.z, gz.w}, gna[4] = {gn.x, gn.y, gn.z, gn.w};
      const float bra[4] = {br.x, br.y, br.z, br.w}, bza[4] = {bz.x, bz.y, bz.z, bz.w}, bna[4] = {bn.x, bn.y, bn.z, bn.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        rr[q][r] = sigmoid_fast(gra[r] + ar[r] + bra[r]);
        zz[q][r] = sigmoid_fast(gza[r] + az[r] + bza[r]);
        gh[q][r] = an[r] + bna[r];
        nn[q][r] = tanh_fast(gna[r] + rr[q][r] * gh[q][r]);
        hnew[q][r] = (1.f - zz[q][r]) * nn[q][r] + zz[q][r] * hprev[q][r];
        hprev[q][r] = hnew[q][r];
      }
    }
    fetch_gi(s + 1);
    {
      char* hs_s = reinterpret_cast<char*>(p.hs + (long)(s + 1) * n);
      char* out_t = reinterpret_cast<char*>(out) + (long)t * 2 * H * 4;
      char* sv = reinterpret_cast<char*>(p.save + (long)s * 4 * n);
#pragma unroll
      for (int q = 0; q < QO; ++q) {
        st4u(hs_s, eoff + 64u * q, hnew[q]);
        st4u(out_t, ooff + 64u * q, hnew[q]);
        st4u(sv, eoff + 64u * q, rr[q]); st4u(sv + n * 4, eoff + 64u * q, zz[q]); st4u(sv + 2 * n * 4, eoff + 64u * q, nn[q]); st4u(sv + 3 * n * 4, eoff + 64u * q, gh[q]);
      }
    }
    lds_barrier();  // every wave has read this step's h fragments
#pragma unroll
    for (int q = 0; q < QO; ++q) {
      gbf16x4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = (__bf16)hnew[q][r];
      *reinterpret_cast<gbf16x4*>(hb + lm * PH + jo + 16 * q) = v;
      if constexpr (TW == 2) *reinterpret_cast<gbf16x4*>(hb + (lm ^ 8) * PH + jo) = v;  // the twin's row holds the same sample: it gets this tile from here
    }
    lds_barrier();
  }
}

template <int H, int NW, int TW>
__global__ __launch_bounds__(NW * 64) void gru_seq_bwd_kernel(focal_gru_desc gd, GruBwdArgs args, const float* __restrict__ dout, long ld_b,
                                                              long ld_t, float scale) {
  constexpr int QT = H / (16 * NW), PG = 3 * H + 8;
  constexpr int QO = QT / TW, SPB = 16 / TW;  // (twins: see gru_seq_fwd_kernel)
  static_assert(TW == 1 || QT == 2, "twins split the wave's two tiles");
  constexpr int KS = 3 * H / 32, KR = 2 * KS / 3;  // k-steps of the product; the first KR live in registers, the rest in LDS
  extern __shared__ __attribute__((aligned(16))) unsigned char gru_lds[];
  gbf16x8* wl = reinterpret_cast<gbf16x8*>(gru_lds);
  bf16_t* gb = reinterpret_cast<bf16_t*>(gru_lds + (size_t)NW * QT * (KS - KR) * 64 * 16);
  const int dir = blockIdx.y;
  const GruDirBwd p = args.d[dir];
  const int B = gd.B, T = gd.T;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lm = lane & 15, lg = lane >> 4;
  const int twin = TW == 2 ? lm >> 3 : 0;
  const int b = min((int)blockIdx.x * SPB + (lm & (SPB - 1)), B - 1);  // (clamping, no conditions around memory instructions: see gru_seq_fwd_kernel)
  const int jw = wave * 16 * QT;
  const int jo = jw + 16 * twin + 4 * lg;
  const long n = (long)B * H;
  float dhz[QO][4], dhrec[QO][4];
#pragma unroll
  for (int q = 0; q < QO; ++q)
#pragma unroll
    for (int r = 0; r < 4; ++r) dhz[q][r] = dhrec[q][r] = 0.f;
  gbf16x8 wreg[QT][KR];  // this wave's rows of W_hh^T, resident for all steps (registers + LDS, as in the forward kernel)
#pragma unroll
  for (int q = 0; q < QT; ++q)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const gbf16x8 w = gd.whh_frag ? *reinterpret_cast<const gbf16x8*>(p.whh_t + (((long)(wave * QT + q) * KS + ks) * 64 + lane) * 8)
                                    : *reinterpret_cast<const gbf16x8*>(p.whh_t + (long)(jw + 16 * q + lm) * 3 * H + 32 * ks + 8 * lg);
      if (ks < KR) wreg[q][ks] = w;
      else wl[((wave * QT + q) * (KS - KR) + ks - KR) * 64 + lane] = w;
    }
  const uint32_t eoff = ((uint32_t)b * H + jo) * 4u;                      // this lane in a [B][H] plane (hs, save); a second own tile: + 64 B
  const uint32_t doff = (uint32_t)((long)b * ld_b + dir * H + jo) * 4u;   // ... in dout at t = 0
  const uint32_t gioff = ((uint32_t)b * T * 3 * H + jo) * 4u;             // ... in dgi [B][T][3H] at t = 0
  const uint32_t ghoff = ((uint32_t)b * 3 * H + jo) * 4u;                 // ... in a step's dgh [B][3H]
  // The step's operands (upstream gradient, the four saved gate values, h_{s-1}) do not depend on the recurrence.  TW = 2 (24 registers a set):
  // they are requested one step ahead -- after the gate math has consumed the previous set (same registers), BEFORE the step's result stores
  // (results return in issue order: a load behind the stores waits for their acknowledgement) -- and land behind the barrier / MFMA phase.
  // TW = 1: a second set of 48 registers next to W_hh^T's 128 resident ones spills 4-5 fragments (measured 80 vs 72 us); loaded at the top
  // of the step, latency exposed.
  constexpr bool AHEAD = TW == 2;
  float4 d4[QO], r4[QO], zz4[QO], n4[QO], g4[QO], h4[QO];
  auto fetch = [&](int s_) {
    const int sc = s_ > 0 ? s_ : 0;  // (before the first step: a redundant reload instead of a branch)
    const int t_ = dir ? T - 1 - sc : sc;
    const char* dout_t = reinterpret_cast<const char*>(dout + (long)t_ * ld_t);
    const char* sv = reinterpret_cast<const char*>(p.save + (long)sc * 4 * n);
    const char* hs_s = reinterpret_cast<const char*>(p.hs + (long)sc * n);
#pragma unroll
    for (int q = 0; q < QO; ++q) {
      d4[q] = ld4u(dout_t, doff + 64u * q);
      r4[q] = ld4u(sv, eoff + 64u * q); zz4[q] = ld4u(sv + n * 4, eoff + 64u * q); n4[q] = ld4u(sv + 2 * n * 4, eoff + 64u * q); g4[q] = ld4u(sv + 3 * n * 4, eoff + 64u * q);
      h4[q] = ld4u(hs_s, eoff + 64u * q);
    }
  };
  if constexpr (AHEAD) fetch(T - 1);
  for (int s = T - 1; s >= 0; --s) {
    const int t = dir ? T - 1 - s : s;
    if constexpr (!AHEAD) fetch(s);
    char* dgi_t = reinterpret_cast<char*>(p.dgi + (long)t * 3 * H);
    char* dgh_s = reinterpret_cast<char*>(p.dgh + (long)s * B * 3 * H);
    float dr[QO][4], dz[QO][4], dn[QO][4], dnr[QO][4];
#pragma unroll
    for (int q = 0; q < QO; ++q) {
      const float da[4] = {d4[q].x, d4[q].y, d4[q].z, d4[q].w}, ra[4] = {r4[q].x, r4[q].y, r4[q].z, r4[q].w}, za[4] = {zz4[q].x, zz4[q].y, zz4[q].z, zz4[q].w};
      const float na[4] = {n4[q].x, n4[q].y, n4[q].z, n4[q].w}, ga[4] = {g4[q].x, g4[q].y, g4[q].z, g4[q].w}, ha[4] = {h4[q].x, h4[q].y, h4[q].z, h4[q].w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float dh = scale * da[r] + dhrec[q][r] + dhz[q][r];
        dn[q][r] = dh * (1.f - za[r]) * (1.f - na[r] * na[r]);
        dz[q][r] = dh * (ha[r] - na[r]) * za[r] * (1.f - za[r]);
        dr[q][r] = dn[q][r] * ga[r] * ra[r] * (1.f - ra[r]);
        dnr[q][r] = dn[q][r] * ra[r];
        dhz[q][r] = dh * za[r];
      }
      gbf16x4 v0, v1, v2;
#pragma unroll
      for (int r = 0; r < 4; ++r) { v0[r] = (__bf16)dr[q][r]; v1[r] = (__bf16)dz[q][r]; v2[r] = (__bf16)dnr[q][r]; }
      bf16_t* row = gb + lm * PG + jo + 16 * q;
      *reinterpret_cast<gbf16x4*>(row) = v0;
      *reinterpret_cast<gbf16x4*>(row + H) = v1;
      *reinterpret_cast<gbf16x4*>(row + 2 * H) = v2;
      if constexpr (TW == 2) {  // the twin's row holds the same sample: it gets this tile's columns from here
        bf16_t* row2 = gb + (lm ^ 8) * PG + jo;
        *reinterpret_cast<gbf16x4*>(row2) = v0;
        *reinterpret_cast<gbf16x4*>(row2 + H) = v1;
        *reinterpret_cast<gbf16x4*>(row2 + 2 * H) = v2;
      }
    }
    if constexpr (AHEAD) {
      asm volatile("" ::: "memory");  // (the requests stay below the gate math: hoisted above it, both operand sets are live at once)
      fetch(s - 1);
    }
#pragma unroll
    for (int q = 0; q < QO; ++q) {
      st4u(dgi_t, gioff + 64u * q, dr[q]); st4u(dgi_t + H * 4, gioff + 64u * q, dz[q]); st4u(dgi_t + 2 * H * 4, gioff + 64u * q, dn[q]);
      st4u(dgh_s, ghoff + 64u * q, dr[q]); st4u(dgh_s + H * 4, ghoff + 64u * q, dz[q]); st4u(dgh_s + 2 * H * 4, ghoff + 64u * q, dnr[q]);
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
      for (int q = 0; q < QO; ++q) {
        const gf32x4 a = TW == 2 ? gru_own<TW>(acc[0], acc[QT - 1], twin) : acc[q];
#pragma unroll
        for (int r = 0; r < 4; ++r) dhrec[q][r] = a[r];
      }
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

// Twins (lanes per sample, see gru_seq_fwd_kernel) of a sequence launch: 2 -- 8 samples per workgroup, twice the grid -- while that grid stays
// within 128 workgroups: the passes of a DeepSense step run their recurrences side by side on 256 CUs, one workgroup per CU (LDS) -- four
// passes of 256 windows (view x modality: 64 workgroups each) or, with both views of a modality in one pass (round 5), two of 512 (128 each).
// H = 128 has one tile per wave: nothing to split.
#ifndef GRU_GRID_TARGET
#define GRU_GRID_TARGET 128
#endif
static int gru_twins(int B, int H, int n_dir) { return (H == 256 && GRU_NW == 8 && ceil_div(B, 8) * n_dir <= GRU_GRID_TARGET) ? 2 : 1; }

extern "C" int focal_gru_seq_fwd(const focal_gru_desc* d, int n_dir, const float* const* gi, const void* const* whh, const float* const* bhh,
                                 float* const* hs, float* const* save, float* out, void* stream) {
  FOCAL_CHECK_ARG(d && gi && whh && bhh && hs && save && out && n_dir >= 1 && n_dir <= 2, "gru_seq_fwd: bad argument");
  if (d->H != 128 && d->H != 256) {
    focal_set_error("gru_seq_fwd: hidden size %d not in {128, 256} (use the per-step kernels)", d->H);
    return FOCAL_EUNSUPPORTED;
  }
  if ((long)d->B * d->T * 3 * d->H * 4 >= (1L << 32)) {  // (the kernels address gi / out / save planes with 32-bit byte offsets)
    focal_set_error("gru_seq_fwd: B x T x 3H = %ld elements exceed the kernel's 32-bit addressing", (long)d->B * d->T * 3 * d->H);
    return FOCAL_EUNSUPPORTED;
  }
  GruFwdArgs a;
  memset(&a, 0, sizeof(a));
  for (int i = 0; i < n_dir; ++i) {
    FOCAL_CHECK_ARG(gi[i] && whh[i] && bhh[i] && hs[i] && save[i], "gru_seq_fwd: null tensor");
    a.d[i] = GruDirFwd{gi[i], (const bf16_t*)whh[i], bhh[i], hs[i], save[i]};
  }
  const int H = d->H, tw = gru_twins(d->B, H, n_dir);
  const dim3 grid(ceil_div(d->B, 16 / tw), n_dir);
  const size_t lds = (size_t)(H / 16) * (H / 32 + (H > 128 ? 1 : 0)) * 64 * 16 + (size_t)16 * (H + 8) * 2 + (size_t)3 * H * 4;
  static std::atomic<bool> granted{false};
  if (!granted.load(std::memory_order_acquire)) {  // up to 143 KB of the CU's 160 KB: above the default dynamic-LDS grant
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gru_seq_fwd_kernel<256, GRU_NW, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(gru_seq_fwd_kernel<256, GRU_NW, GRU_TW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(gru_seq_fwd_kernel<128, GRU_NW_128, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { focal_set_error("gru_seq_fwd: cannot reserve LDS: %s", hipGetErrorString(e)); return FOCAL_EHIP; }
    granted.store(true, std::memory_order_release);
  }
  if (H == 256 && tw == 2) FOCAL_LAUNCH((gru_seq_fwd_kernel<256, GRU_NW, GRU_TW>), grid, dim3(GRU_NW * 64), lds, (hipStream_t)stream, *d, a, out);
  else if (H == 256) FOCAL_LAUNCH((gru_seq_fwd_kernel<256, GRU_NW, 1>), grid, dim3(GRU_NW * 64), lds, (hipStream_t)stream, *d, a, out);
  else FOCAL_LAUNCH((gru_seq_fwd_kernel<128, GRU_NW_128, 1>), grid, dim3(GRU_NW_128 * 64), lds, (hipStream_t)stream, *d, a, out);
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
  if ((long)d->B * d->T * 3 * d->H * 4 >= (1L << 32) || (long)d->B * ld_b * 4 >= (1L << 32)) {
    focal_set_error("gru_seq_bwd: tensors exceed the kernel's 32-bit addressing (B %d, T %d, H %d, ld_b %ld)", d->B, d->T, d->H, ld_b);
    return FOCAL_EUNSUPPORTED;
  }
  GruBwdArgs a;
  memset(&a, 0, sizeof(a));
  for (int i = 0; i < n_dir; ++i) {
    FOCAL_CHECK_ARG(whh_t[i] && hs[i] && save[i] && dgi[i] && dgh[i], "gru_seq_bwd: null tensor");
    a.d[i] = GruDirBwd{(const bf16_t*)whh_t[i], hs[i], save[i], dgi[i], dgh[i]};
  }
  const int H = d->H, tw = gru_twins(d->B, H, n_dir);
  const dim3 grid(ceil_div(d->B, 16 / tw), n_dir);
  const size_t lds = (size_t)(H / 16) * (H / 32) * 64 * 16 + (size_t)16 * (3 * H + 8) * 2;
  static std::atomic<bool> granted{false};
  if (!granted.load(std::memory_order_acquire)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gru_seq_bwd_kernel<256, GRU_NW, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(gru_seq_bwd_kernel<256, GRU_NW, GRU_TW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(gru_seq_bwd_kernel<128, GRU_NW_128, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { focal_set_error("gru_seq_bwd: cannot reserve LDS: %s", hipGetErrorString(e)); return FOCAL_EHIP; }
    granted.store(true, std::memory_order_release);
  }
  if (H == 256 && tw == 2) FOCAL_LAUNCH((gru_seq_bwd_kernel<256, GRU_NW, GRU_TW>), grid, dim3(GRU_NW * 64), lds, (hipStream_t)stream, *d, a, dout, ld_b, ld_t, scale);
  else if (H == 256) FOCAL_LAUNCH((gru_seq_bwd_kernel<256, GRU_NW, 1>), grid, dim3(GRU_NW * 64), lds, (hipStream_t)stream, *d, a, dout, ld_b, ld_t, scale);
  else FOCAL_LAUNCH((gru_seq_bwd_kernel<128, GRU_NW_128, 1>), grid, dim3(GRU_NW_128 * 64), lds, (hipStream_t)stream, *d, a, dout, ld_b, ld_t, scale);
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}
