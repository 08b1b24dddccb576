// The MLP branch of a Swin block at 128 / 256 channels as ONE launch per direction (round 6).
//
//   forward    h, h' = drop_h(gelu(a2 W1^T + b1)) and its derivative      [M, 4C], written once for the backward pass, never read back
//              x_out = x_mid + drop_o(h W2^T + b2)                         (+ the LayerNorm that reads x_out next, at 128 channels)
//
// replaces models/SwinModules.py:18-34 (Mlp.forward: fc1 -> GELU -> Dropout -> fc2 -> Dropout) + the residual add and DropPath of
// SwinTransformerBlock.forward (:339-341) for stages 1-2, which ran as two GEMM launches (gemm_ring.hpp): fc1 wrote h and h', fc2 read h
// back -- 8C of the ~72C bytes per token a block's forward pass moves -- and the hidden tensor's tile left the chip between the products.
//
// The 64-channel kernel (mlp.hip) keeps both weights in LDS; at 128 / 256 channels they are 256 KB / 1 MB, so they STREAM: a persistent
// workgroup (one per CU) owns 128 token rows at a time -- 8 consumer waves x 16 rows, a2 fragments and the [16, C] fc2 accumulators in
// registers -- and walks the hidden dimension 64 units per step.  Two loader waves fill a 2-slot ring with LDS-DMA
// (global_load_lds_dwordx4, counted vmcnt, one s_barrier per step: the roles of gemm_ring.hpp): per step the 64 rows of W1 and the
// 64-column slice of W2, 32 KB (C = 128) / 64 KB (C = 256), all of it L2-resident (every workgroup streams the same 256 KB / 1 MB).
// A consumer wave multiplies its rows against the slice (fc1 accumulators [hidden][token]), applies bias + GELU + dropout in registers,
// stores h / h' (16 B per lane) and feeds the SAME registers back to the matrix cores as the B operand of fc2: the accumulator layout of
// v_mfma_f32_16x16x32_bf16 (lane = token, 4 consecutive rows per 16-row tile) is a legal operand layout once the W1 rows of a tile are
// chosen so that a lane's 2 x 4 accumulators are 8 CONSECUTIVE hidden units -- the loader permutes the rows when it fills the image
// (LDS row 16 T + r of a step <- hidden unit 32 (T / 2) + 8 (r / 4) + 4 (T % 2) + r % 4), free: LDS-DMA source addresses are per lane.
// Waves never exchange data; the only synchronisation is the ring's barrier.
//
// Results are BIT-IDENTICAL to focal_linear_fwd(GELU) + focal_linear_fwd / focal_linear_resid_ln_fwd: same k order in both products, the
// element math of EPI_GELU_FWD (gemm_pipe.hpp) on the same (row, column) mask indices, and fc2's epilogue is the shared code itself
// (pipe_epilogue_finish, 128 channels) or its per-element arithmetic (256 channels: EPI_RESID straight from the accumulators).
// tests/test_mlp_wide_gpu.py asserts torch.equal on every output, dropout on and off.
#include "gemm_ring.hpp"
#include "mlp.hpp"

// Lab-only in-kernel stamps (-DWIDE_STAMPS: tools/mb_mlp_wide.py --stamps builds a variant and passes a u64 buffer in g2.colsumA)
#ifdef WIDE_STAMPS
#define WS_DECL(n) unsigned long long n = 0
#define WS_T0() const unsigned long long ws_t0_ = __builtin_amdgcn_s_memtime()
#define WS_ADD(n) n += __builtin_amdgcn_s_memtime() - ws_t0_
#define WS_NOW(t) const unsigned long long t = __builtin_amdgcn_s_memtime()
#define WS_ACC(n, a, b) n += (b) - (a)
#define WS_OUT(i, n) if (lane == 0 && p.g2.colsumA) reinterpret_cast<unsigned long long*>(p.g2.colsumA)[((long)blockIdx.x * 16 + wave) * 8 + (i)] = n
#else
#define WS_DECL(n)
#define WS_T0()
#define WS_ADD(n)
#define WS_NOW(t)
#define WS_ACC(n, a, b)
#define WS_OUT(i, n)
#endif

namespace focal_mlp_wide {

struct WideFwdParams {
  int M;
  const bf16_t* a;    // [M][C]   LayerNorm output (norm2)
  const bf16_t* w1;   // [H][C]
  const float* b1;    // [H]
  const bf16_t* w2;   // [C][H]
  bf16_t* h;          // [M][H]   drop(gelu(.))
  bf16_t* hg;         // [M][H]   d gelu x mask
  MaskParams drop_h;  // over [M][H]
  GemmParams g2;      // the fc2 product as focal_linear_fwd / focal_linear_resid_ln_fwd would launch it: bias, resid, C, epi, ln_*, aux_out
  // PROJ: the attention branch's tail in front of the MLP (x_mid = x + drop_p(o Wp^T + bp), a2 = norm2(x_mid)): `a` and g2.resid are then
  // OUTPUTS (a2, x_mid), st2 the LayerNorm's {mean, rstd}
  const bf16_t* o; const float* x; const bf16_t* wp; const float* bp; MaskParams drop_p; const float* ng2; const float* nbt2; float* st2; float ln2_eps;
};

constexpr int NLOAD = 2, R = 2;
// consumer waves per workgroup: 6 -- 8 waves in all, two per SIMD, 256 registers: at 256 channels a wave's 64 accumulator + 32 operand-fragment
// registers do not fit in the 168 of a 10-wave workgroup; at 128 channels 8 consumers fit (166) but 6 with the epilogue's residual rows
// requested before the steps measured faster (96-row tiles: 768 / 384 tiles = whole rounds of 256 CUs)
#ifndef WIDE_NW128
#define WIDE_NW128 6
#endif
#ifndef WIDE_NO_HOIST128
#define WIDE_HOIST128 1
#endif
#ifndef WIDE_NW256
#define WIDE_NW256 6
#endif
// NWX != 0: that many consumer waves instead (the forward kernel at 256 channels runs 4 -- 64-row tiles, one consumer per SIMD -- where the 64-row
// tiles still fit one round of the chip: a workgroup is ONE tile's chain of 12 ring steps, its time is the slowest SIMD's element math, and
// with 6 consumers two SIMDs carry two of them; focal_mlp_wide_fwd's launcher)
template <int C, int NWX = 0> struct WideWaves { static constexpr int NW = NWX != 0 ? NWX : (C == 128 ? WIDE_NW128 : WIDE_NW256), BM = 16 * NW; };

template <int C, int NWX = 0> struct WideLayout {
  static constexpr int NW = WideWaves<C, NWX>::NW, BM = WideWaves<C, NWX>::BM;
  static constexpr int H = 4 * C, KK = C / 32, NIMG1 = C / 64, CT = C / 16, NSTEP = H / 64;
  static constexpr int W1_BYTES = NIMG1 * 8192, W2_BYTES = C * 128, SLOT_BYTES = W1_BYTES + W2_BYTES;
  static constexpr int NP1 = NIMG1 * 8, NP2 = C / 8, L1 = NP1 / NLOAD, L2 = NP2 / NLOAD, LSTEP = L1 + L2;
  static constexpr int WPITCH = C + 4;
  static constexpr int STG_BYTES = C == 128 ? NW * 16 * WPITCH * 4 : 0;  // 128 channels: the shared epilogue's transposition region per wave
  static constexpr int B1_BYTES = H * 4 + 6 * C * 4;  // + fc2's bias + (256 channels) the next LayerNorm's gamma / beta + (PROJ) proj bias, norm2's gamma / beta  // fc1's bias, staged once (a global load inside the step loop would wait, through the in-order vmcnt, for the
                                          // previous step's h / h' stores)
  static constexpr int LDS_BYTES = R * SLOT_BYTES + STG_BYTES + B1_BYTES;
};

template <int N> __device__ __forceinline__ void tie4(bf16x8 (&v)[N]) {
#pragma unroll
  for (int i = 0; i + 3 < N; i += 4) asm volatile("" : "+v"(v[i]), "+v"(v[i + 1]), "+v"(v[i + 2]), "+v"(v[i + 3]));
}
__device__ __forceinline__ void lds_wait8(bf16x8 (&v)[8]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
}
__device__ __forceinline__ f32x4 gload4(const float* p) {
  typedef float f32x4g __attribute__((ext_vector_type(4)));
  return *((const __attribute__((address_space(1))) f32x4g*)(p));  // global_load (a generic pointer would be a flat_load)
}

// EPI2: EPI_RESID or EPI_RESID_LN.  DROP: the hidden dropout is on (its mask hash then runs unconditionally: MaskEval's run-time `on` test is
// a scalar branch per four elements); FULL: M is a multiple of the tile height (every shape of the step: no row guards).  Both keep the
// step loop ONE basic block, so that hipcc can interleave the second half's element math with the first half's fc2 MFMAs.
// PROJ (round 6): per tile, NP = C / 64 ring steps in front of the MLP's carry the proj weight 64 output channels at a time (the same image
// geometry and row permutation as a W1 slice); the wave multiplies its o rows against them, finishes x_mid = x + drop_p(. + bp) in
// registers and stores it, then norm2 on the row -- with the summation tree and roundings of the launch it replaces (focal_linear_resid_ln_fwd's
// epilogue at 128 channels, ln_fwd_kernel at 256) -- stores a2 / statistics and keeps a2 as fc1's operand.  12 proj launches and 8 norm2
// launches fewer per step; x_mid and a2 are not re-read.
template <int C, int EPI2, bool DROP, bool FULL, bool PROJ, int NWX = 0>
__global__ __launch_bounds__(64 * (WideWaves<C, NWX>::NW + NLOAD)) void mlp_wide_fwd_kernel(const WideFwdParams p) {
  using L = WideLayout<C, NWX>;
  constexpr int NW = L::NW, BM = L::BM;
  constexpr int H = L::H, KK = L::KK, CT = L::CT, NSTEP = L::NSTEP, SLOT_BYTES = L::SLOT_BYTES, W1_BYTES = L::W1_BYTES;
  constexpr int NP = PROJ ? C / 64 : 0, TSTEP = NSTEP + NP;   // ring steps per tile
  static_assert(C == 128 || C == 256, "128 or 256 channels");
  static_assert(EPI2 == EPI_RESID || EPI2 == EPI_RESID_LN, "fc2 epilogue");
  extern __shared__ __attribute__((aligned(1024))) char wide_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntiles = (p.M + BM - 1) / BM, G = gridDim.x;
  const int ntl = (ntiles - (int)blockIdx.x + G - 1) / G;  // tiles of this workgroup: blockIdx.x, + G, ...
  const int total = ntl * TSTEP;
  {
    float* b1s = reinterpret_cast<float*>(wide_lds + R * SLOT_BYTES + L::STG_BYTES);
    if constexpr (PROJ) {
      for (int i = tid; i < C; i += 64 * (NW + NLOAD)) {
        b1s[H + 3 * C + i] = p.bp[i];
        b1s[H + 4 * C + i] = p.ng2[i];
        b1s[H + 5 * C + i] = p.nbt2[i];
      }
    }
    for (int i = tid; i < H; i += 64 * (NW + NLOAD)) b1s[i] = p.b1[i];
    for (int i = tid; i < C; i += 64 * (NW + NLOAD)) b1s[H + i] = p.g2.bias[i];
    if (EPI2 == EPI_RESID_LN && C == 256) {
      for (int i = tid; i < C; i += 64 * (NW + NLOAD)) {
        b1s[H + C + i] = p.g2.ln_gamma[i];
        b1s[H + 2 * C + i] = p.g2.ln_beta[i];
      }
    }
    __syncthreads();
  }

  if (wave >= NW) {
    // ================================================================================================ loader waves
    const int lw = wave - NW;
    constexpr int L1 = L::L1, L2 = L::L2, LSTEP = L::LSTEP;
    uint32_t off1[L1], off2[L2];
#pragma unroll
    for (int t = 0; t < L1; ++t) {
      const int i = lw + NLOAD * t, kc = i >> 3, q = i & 7;
      const int lrow = 8 * q + (lane >> 3), pos = lane & 7, chunk = pos ^ ((lrow >> 1) & 7);
      const int T = lrow >> 4, r = lrow & 15, hl = 32 * (T >> 1) + 8 * (r >> 2) + 4 * (T & 1) + (r & 3);  // the row permutation (see top)
      off1[t] = (uint32_t)((hl * C + kc * 64 + chunk * 8) * 2);
    }
#pragma unroll
    for (int t = 0; t < L2; ++t) {
      const int q = lw + NLOAD * t, c = 8 * q + (lane >> 3), pos = lane & 7, chunk = pos ^ ((c >> 1) & 7);
      off2[t] = (uint32_t)((c * H + chunk * 8) * 2);
    }
    const char* w1b = reinterpret_cast<const char*>(p.w1);
    const char* w2b = reinterpret_cast<const char*>(p.w2);
    const char* wpb = reinterpret_cast<const char*>(p.wp);
    int issued = 0, f_st = 0;
    auto issue_next = [&]() __attribute__((always_inline)) {
      const uint32_t slot = (uint32_t)(issued & (R - 1)) * SLOT_BYTES;
      const bool proj_step = PROJ && f_st < NP;   // (wave-uniform) the first NP steps of a tile: 64 rows of the proj weight, no W2 part
      const int mst = f_st - NP;
      const char* s1 = proj_step ? wpb + (long)f_st * (64 * C * 2) : w1b + (long)mst * (64 * C * 2);
      const char* s2 = w2b + (long)mst * 128;
#pragma unroll
      for (int t = 0; t < L1; ++t) {
        const int i = lw + NLOAD * t;
        __builtin_amdgcn_global_load_lds((pipe_glb_ptr)(s1 + off1[t]), (pipe_lds_ptr)(wide_lds + slot + (i >> 3) * 8192 + (i & 7) * 1024), 16, 0, 0);
      }
      if (!proj_step) {
#pragma unroll
        for (int t = 0; t < L2; ++t) {
          const int q = lw + NLOAD * t;
          __builtin_amdgcn_global_load_lds((pipe_glb_ptr)(s2 + off2[t]), (pipe_lds_ptr)(wide_lds + slot + W1_BYTES + q * 1024), 16, 0, 0);
        }
      }
      ++issued;
      if (++f_st == TSTEP) f_st = 0;
    };
    static_assert(R == 2, "the wait below assumes one step in flight behind the one awaited");
    WS_DECL(ws_wait); WS_DECL(ws_bar); WS_DECL(ws_issue); WS_DECL(ws_all);
#ifdef WIDE_STAMPS
    const unsigned long long ws_begin = __builtin_amdgcn_s_memtime();
#endif
    if (total > 0) issue_next();
    for (int g = 0; g < total; ++g) {
      { WS_T0(); ring_vmcnt<0>(); WS_ADD(ws_wait); }   // step g has landed (issued == g + 1: nothing younger is in flight)
      { WS_T0(); ring_barrier(); WS_ADD(ws_bar); }     // B_g: the consumers are past step g - 1 -> its slot is free
      { WS_T0(); if (issued < total) issue_next(); WS_ADD(ws_issue); }
    }
#ifdef WIDE_STAMPS
    ws_all = __builtin_amdgcn_s_memtime() - ws_begin;
#endif
    WS_OUT(0, ws_wait); WS_OUT(1, ws_bar); WS_OUT(2, ws_issue); WS_OUT(3, ws_all); WS_OUT(4, 1ull);
    (void)LSTEP;
    return;
  }

  // ==================================================================================================== consumer waves
  const int g4 = lane >> 4, l15 = lane & 15, swz = (lane >> 1) & 7;
  const uint32_t lds0 = pipe_lds_addr(wide_lds);
  const uint32_t fo0 = lds0 + l15 * 128 + ((g4 ^ swz) << 4), fo1 = lds0 + l15 * 128 + (((4 + g4) ^ swz) << 4);
  MaskEval meH, meO, meP;
  meH.init(p.drop_h);
  meO.init(p.g2.epi);
  if constexpr (PROJ) meP.init(p.drop_p);
  float* est = reinterpret_cast<float*>(wide_lds + R * SLOT_BYTES) + wave * 16 * L::WPITCH;
  const float* b1s = reinterpret_cast<const float*>(wide_lds + R * SLOT_BYTES + L::STG_BYTES);
  const float* b2s = b1s + H;
  float* Cout = reinterpret_cast<float*>(p.g2.C);
  float pgd[4] = {0.f, 0.f, 0.f, 0.f}, pbd[4] = {0.f, 0.f, 0.f, 0.f}, lngd[4] = {0.f, 0.f, 0.f, 0.f};  // (EPI_LN_BWD operands of the shared epilogue: unused)

  int slot = 0;
  WS_DECL(ws_cbar); WS_DECL(ws_fc1); WS_DECL(ws_gelu); WS_DECL(ws_fc2); WS_DECL(ws_epi); WS_DECL(ws_call);
#ifdef WIDE_STAMPS
  const unsigned long long ws_cbegin = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll 1
  for (int i = 0, tile = blockIdx.x; i < ntl; ++i, tile += G) {
    const int m0 = tile * BM, mbase = m0 + wave * 16, m = mbase + l15;
    const bool mok = FULL || m < p.M;
    const long mrow = mok ? m : p.M - 1;
    bf16x8 xa[KK];
    if constexpr (!PROJ) {
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) xa[kk] = *reinterpret_cast<const bf16x8*>(p.a + mrow * C + kk * 32 + 8 * g4);
    } else {
      // ---- x_mid = x + drop_p(o Wp^T + bp); a2 = norm2(x_mid).  Step pp: output channels 64 pp + 32 s + 8 g .. + 7 (s = 0, 1) of this lane's token
      bf16x8 oa[KK];
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) oa[kk] = *reinterpret_cast<const bf16x8*>(p.o + mrow * C + kk * 32 + 8 * g4);
      const float* bps = b2s + 3 * C;
      const float* g2s = b2s + 4 * C;
      const float* bt2s = b2s + 5 * C;
      float* xmid = const_cast<float*>(p.g2.resid);
      bf16_t* a2o = const_cast<bf16_t*>(p.a);
      const float rowmp = meP.row_mult(m);
      float xm[NP][2][8];
      pipe_static_for<0, NP>([&](auto pp_) {
        constexpr int pp = decltype(pp_)::value;
        f32x4 xr[2][2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          xr[s][0] = gload4(p.x + mrow * C + 64 * pp + 32 * s + 8 * g4);
          xr[s][1] = gload4(p.x + mrow * C + 64 * pp + 32 * s + 8 * g4 + 4);
        }
        ring_barrier();
        const uint32_t sb = (uint32_t)slot * SLOT_BYTES;
        f32x4 pu[4];
#pragma unroll
        for (int T = 0; T < 4; ++T) pu[T] = f32x4{0.f, 0.f, 0.f, 0.f};
        pipe_static_for<0, KK / 2>([&](auto kc_) {
          constexpr int kc = decltype(kc_)::value;
          bf16x8 w[8];
          pipe_static_for<0, 4>([&](auto T_) {
            constexpr int T = decltype(T_)::value;
            w[T] = pipe_lds_read128<kc * 8192 + T * 2048>(fo0 + sb);
            w[4 + T] = pipe_lds_read128<kc * 8192 + T * 2048>(fo1 + sb);
          });
          lds_wait8(w);
#pragma unroll
          for (int T = 0; T < 4; ++T) pu[T] = mma16(w[T], oa[2 * kc], pu[T]);
#pragma unroll
          for (int T = 0; T < 4; ++T) pu[T] = mma16(w[4 + T], oa[2 * kc + 1], pu[T]);
        });
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int c0 = 64 * pp + 32 * s + 8 * g4;
          const f32x4 bq0 = *reinterpret_cast<const f32x4*>(bps + c0), bq1 = *reinterpret_cast<const f32x4*>(bps + c0 + 4);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float v = (e < 4 ? pu[2 * s][e] : pu[2 * s + 1][e - 4]) * p.g2.alpha + (e < 4 ? bq0[e] : bq1[e - 4]);
            float t;
            {
#pragma clang fp contract(off)
              t = v * rowmp;
            }
            xm[pp][s][e] = __builtin_fmaf(t, meP.elem_mult(m, c0 + e), e < 4 ? xr[s][0][e] : xr[s][1][e - 4]);
          }
          if (mok) {
            store4(xmid + (long)m * C + c0, f32x4{xm[pp][s][0], xm[pp][s][1], xm[pp][s][2], xm[pp][s][3]});
            store4(xmid + (long)m * C + c0 + 4, f32x4{xm[pp][s][4], xm[pp][s][5], xm[pp][s][6], xm[pp][s][7]});
          }
        }
        slot ^= 1;
      });
      // norm2 with the replaced launch's summation tree: there lane li of a row holds channels 4 li .. + 3 and the partial sums meet pairwise over
      // li's bits 0, 1, 2 ... (row16_sum, then xadd 16 [, xadd 32]); here li = 16 pp + 8 s + 2 g + h: bit 0 = h (the halves of the lane's 8
      // channels), bits 1 / 2 = lane bits 4 / 5, bit 3 = s, bits 4 [, 5] = pp
      auto tree = [&](float (&q)[NP][2][2]) __attribute__((always_inline)) {
        float t[NP][2];
#pragma unroll
        for (int pp = 0; pp < NP; ++pp)
#pragma unroll
          for (int s = 0; s < 2; ++s) t[pp][s] = xadd32(xadd16(q[pp][s][0] + q[pp][s][1]));
        float u2[NP];
#pragma unroll
        for (int pp = 0; pp < NP; ++pp) u2[pp] = t[pp][0] + t[pp][1];
#pragma unroll
        for (int w = 1; w < NP; w <<= 1)
#pragma unroll
          for (int pp = 0; pp < NP; pp += 2 * w) u2[pp] = u2[pp] + u2[pp + w];
        return u2[0];
      };
      float q1[NP][2][2];
#pragma unroll
      for (int pp = 0; pp < NP; ++pp)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int h = 0; h < 2; ++h) q1[pp][s][h] = ((xm[pp][s][4 * h] + xm[pp][s][4 * h + 1]) + xm[pp][s][4 * h + 2]) + xm[pp][s][4 * h + 3];
      const float mean = C == 128 ? tree(q1) * (1.0f / 128.0f) : tree(q1) / C;
#pragma unroll
      for (int pp = 0; pp < NP; ++pp)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
#pragma clang fp contract(off)
            const float d0 = xm[pp][s][4 * h] - mean, d1 = xm[pp][s][4 * h + 1] - mean, d2 = xm[pp][s][4 * h + 2] - mean, d3 = xm[pp][s][4 * h + 3] - mean;
            q1[pp][s][h] = ((d0 * d0 + d1 * d1) + d2 * d2) + d3 * d3;
          }
      const float rstd = rsqrtf(__builtin_fmaf(tree(q1), 1.0f / C, p.ln2_eps));
#pragma unroll
      for (int pp = 0; pp < NP; ++pp)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int c0 = 64 * pp + 32 * s + 8 * g4;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float t;
            {
#pragma clang fp contract(off)
              t = (xm[pp][s][e] - mean) * rstd;
            }
            xa[2 * pp + s][e] = (bf16_t)__builtin_fmaf(g2s[c0 + e], t, bt2s[c0 + e]);
          }
          if (mok) *reinterpret_cast<bf16x8*>(a2o + (long)m * C + c0) = xa[2 * pp + s];
        }
      if (mok && g4 == 0) *reinterpret_cast<float2*>(p.st2 + 2 * (long)m) = make_float2(mean, rstd);
      // (x_mid's rows are read back by other lanes of this wave in fc2's epilogue: see mlp.hip)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    f32x4 yacc[CT];
#pragma unroll
    for (int j = 0; j < CT; ++j) yacc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#ifdef WIDE_HOIST128
    PipePre<float, EPI2, C, 1> pre;
    if constexpr (C == 128) pipe_epilogue_prefetch<float, EPI2, BM, C, NW, 1>(p.g2, p.g2.resid, m0, 0, mbase, 0, lane, Cout, pre);
#endif
    bf16_t* hrow = p.h + (long)m * H + 8 * g4;
    bf16_t* hgrow = p.hg + (long)m * H + 8 * g4;

#pragma unroll 1
    for (int st = 0; st < NSTEP; ++st) {
      // this lane's 2 x 8 biases of the step (hidden 64 st + 32 s + 8 g .. + 7), from the LDS copy
      f32x4 bq[2][2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bq[s][0] = *reinterpret_cast<const f32x4*>(b1s + 64 * st + 32 * s + 8 * g4);
        bq[s][1] = *reinterpret_cast<const f32x4*>(b1s + 64 * st + 32 * s + 8 * g4 + 4);
      }
      { WS_T0(); ring_barrier(); WS_ADD(ws_cbar); }  // B_g: the step is in LDS (the loaders waited for it)
      const uint32_t sb = (uint32_t)slot * SLOT_BYTES;
      WS_NOW(ws_a);
      // ---- fc1: u[T] = W1 rows of tile T (16 hidden units, permuted) x this wave's 16 tokens
      f32x4 u[4];
#pragma unroll
      for (int T = 0; T < 4; ++T) u[T] = f32x4{0.f, 0.f, 0.f, 0.f};
      pipe_static_for<0, KK / 2>([&](auto kc_) {
        constexpr int kc = decltype(kc_)::value;
        bf16x8 w[8];
        pipe_static_for<0, 4>([&](auto T_) {
          constexpr int T = decltype(T_)::value;
          w[T] = pipe_lds_read128<kc * 8192 + T * 2048>(fo0 + sb);
          w[4 + T] = pipe_lds_read128<kc * 8192 + T * 2048>(fo1 + sb);
        });
        lds_wait8(w);
#pragma unroll
        for (int T = 0; T < 4; ++T) u[T] = mma16(w[T], xa[2 * kc], u[T]);
#pragma unroll
        for (int T = 0; T < 4; ++T) u[T] = mma16(w[4 + T], xa[2 * kc + 1], u[T]);
      });
      WS_NOW(ws_b);
      // ---- bias + GELU (+ derivative) + dropout: the element math of EPI_GELU_FWD on columns 64 st + 32 s + 8 g .. + 7 (u[2 s] | u[2 s + 1]),
      // one half s at a time; the fc2 products of half 0 are issued between the two halves' element math (same accumulation order per
      // output: half 0's 32 hidden units, then half 1's)
      bf16x8 hf[2];
      auto gelu_half = [&](auto s_) __attribute__((always_inline)) {
        constexpr int s = decltype(s_)::value;
        float v[8], gq[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = u[2 * s][e] + bq[s][0][e];
          v[4 + e] = u[2 * s + 1][e] + bq[s][1][e];
        }
        const int n = 64 * st + 32 * s + 8 * g4;
#pragma unroll
        for (int e = 0; e < 8; e += 4) {
          gelu_f2 mult[2];
          if constexpr (DROP) {  // MaskEval::elem_mult_quad without its `on` test
            const uint32_t idx = (__umul24((uint32_t)m, (uint32_t)meH.ncols) + (uint32_t)(n + e)) >> 2;
            const uint32_t hh_ = focal_hash24(idx ^ meH.e.key), t16 = meH.e.thresh >> 8;
            uint32_t g2 = hh_ ^ (hh_ >> 13);
            g2 = __umul24(g2, 0xC2B2AFu) + 0x165667B1u;
            g2 ^= g2 >> 15;
            mult[0] = gelu_f2{(hh_ & 0xffffu) < t16 ? 0.0f : meH.e.scale, (hh_ >> 16) < t16 ? 0.0f : meH.e.scale};
            mult[1] = gelu_f2{(g2 & 0xffffu) < t16 ? 0.0f : meH.e.scale, (g2 >> 16) < t16 ? 0.0f : meH.e.scale};
          } else {
            mult[0] = gelu_f2{1.0f, 1.0f};
            mult[1] = gelu_f2{1.0f, 1.0f};
          }
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            const gelu_f2 x = {v[e + 2 * h2], v[e + 2 * h2 + 1]};
#ifndef WIDE_LAB_NO_GELU
            gelu_f2 cdf, pdf;
            gelu_parts2(x, cdf, pdf);
#else
            const gelu_f2 cdf = x, pdf = x;
#endif
            const gelu_f2 gg = (x * pdf + cdf) * mult[h2], hh = x * cdf * mult[h2];
            gq[e + 2 * h2] = gg.x; gq[e + 2 * h2 + 1] = gg.y;
            v[e + 2 * h2] = hh.x; v[e + 2 * h2 + 1] = hh.y;
          }
        }
        bf16x8 hgv;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          hf[s][e] = (bf16_t)v[e];
          hgv[e] = (bf16_t)gq[e];
        }
#ifndef WIDE_LAB_NO_STORES
        if (mok) {
          *reinterpret_cast<bf16x8*>(hrow + 64 * st + 32 * s) = hf[s];
          *reinterpret_cast<bf16x8*>(hgrow + 64 * st + 32 * s) = hgv;
        }
#else
        if (m == -12345) *reinterpret_cast<bf16x8*>(hgrow) = hgv;
#endif
      };
      // ---- fc2, half s: yacc[j] += W2 rows 16 j .. + 15 (hidden columns 32 s .. + 31 of this step's 64) x h
      auto fc2_half = [&](auto s_) __attribute__((always_inline)) {
        constexpr int s = decltype(s_)::value;
        pipe_static_for<0, CT / 8>([&](auto jq_) {
          constexpr int jq = decltype(jq_)::value;
          bf16x8 w[8];
          pipe_static_for<0, 8>([&](auto jj_) {
            constexpr int jj = decltype(jj_)::value, j = 8 * jq + jj;
            w[jj] = pipe_lds_read128<W1_BYTES + j * 2048>((s ? fo1 : fo0) + sb);
          });
          lds_wait8(w);
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) yacc[8 * jq + jj] = mma16(w[jj], hf[s], yacc[8 * jq + jj]);
        });
      };
      gelu_half(std::integral_constant<int, 0>{});
      WS_NOW(ws_c);
      fc2_half(std::integral_constant<int, 0>{});
      gelu_half(std::integral_constant<int, 1>{});
      fc2_half(std::integral_constant<int, 1>{});
      slot ^= 1;
      WS_NOW(ws_d);
      WS_ACC(ws_fc1, ws_a, ws_b); WS_ACC(ws_gelu, ws_b, ws_c); WS_ACC(ws_fc2, ws_c, ws_d);
    }
    WS_NOW(ws_e0);

    // ---- fc2's epilogue
    if constexpr (C == 128) {
      // (what the epilogue reads -- residual rows, bias -- is requested here, not before the steps: 36 registers less across the loop; the
      // SIMD's other waves cover the latency)
#ifndef WIDE_HOIST128
      PipePre<float, EPI2, C, 1> pre;
      pipe_epilogue_prefetch<float, EPI2, BM, C, NW, 1>(p.g2, p.g2.resid, m0, 0, mbase, 0, lane, Cout, pre);
#endif
      const PipeBias<4> bias2 = pipe_epilogue_bias<float, EPI2, C, 1>(p.g2.bias, 0, 0, lane);
      pipe_epilogue_finish<float, EPI2, BM, C, NW, 1>(p.g2, p.g2.alpha, yacc, est, meO, m0, 0, mbase, 0, lane, Cout, bias2.v, pgd, pbd, lngd, pre);
    } else {
      // EPI_RESID, element by element as pipe_epilogue_finish does it (v = acc + bias; y = resid + v * row mask * element mask), straight
      // from the accumulators: lane = token, columns 16 j + 4 g .. + 3
      const float rowm = meO.row_mult(m);
      const float* rrow = p.g2.resid + mrow * p.g2.ldr + 4 * g4;
      float* yrow = Cout + (long)m * p.g2.ldc + 4 * g4;
      // (the residual rows four column tiles ahead: the loads of batch q + 1 are issued BEFORE the stores of batch q, so waiting for them
      // never waits for a store -- vmcnt counts both, in order)
      f32x4 res[2][4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) res[0][jj] = gload4(rrow + 16 * jj);
      pipe_static_for<0, CT / 4>([&](auto q_) {
        constexpr int q = decltype(q_)::value;
        if constexpr (q + 1 < CT / 4) {
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) res[(q + 1) & 1][jj] = gload4(rrow + 16 * (4 * (q + 1) + jj));
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int j = 4 * q + jj;
          const f32x4 b2 = *reinterpret_cast<const f32x4*>(b2s + 16 * j + 4 * g4);
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = yacc[j][e] * p.g2.alpha;
            v += b2[e];
            o[e] = res[q & 1][jj][e] + v * rowm * meO.elem_mult(m, 16 * j + 4 * g4 + e);
          }
          if (mok) store4(yrow + 16 * j, o);
          if constexpr (EPI2 == EPI_RESID_LN) yacc[j] = o;  // the finished row stays in registers for the LayerNorm below
        }
      });
      if constexpr (EPI2 == EPI_RESID_LN) {
        // The LayerNorm that reads x_out next (the next block's norm1), on the row in registers, with ln_fwd_kernel's arithmetic AND its
        // summation tree, so that the result is the stand-alone kernel's bit for bit: there lane li of a row holds columns 4 li .. + 3 and
        // the 64 partial sums meet pairwise over li's bits 0, 1, 2 ... 5 (row16_sum, then xadd 16, xadd 32); here li = 4 j + g: bits 0 / 1 are
        // lane bits 4 / 5 (xadd16 / xadd32), bits 2 - 5 are the column tile j.
        const float* lngs = b2s + C;
        const float* lnbs = b2s + 2 * C;
        auto tree = [&](float (&t)[CT]) __attribute__((always_inline)) {
#pragma unroll
          for (int j = 0; j < CT; ++j) t[j] = xadd32(xadd16(t[j]));
#pragma unroll
          for (int w = 1; w < CT; w <<= 1)
#pragma unroll
            for (int j = 0; j < CT; j += 2 * w) t[j] = t[j] + t[j + w];
          return t[0];
        };
        float part[CT];
#pragma unroll
        for (int j = 0; j < CT; ++j) part[j] = yacc[j][0] + yacc[j][1] + yacc[j][2] + yacc[j][3];
        const float mean = tree(part) / C;
#pragma unroll
        for (int j = 0; j < CT; ++j) {
          const float a = yacc[j][0] - mean, b = yacc[j][1] - mean, c2 = yacc[j][2] - mean, d = yacc[j][3] - mean;
          // (explicit roundings: ln_fwd_kernel's squares are packed multiplies followed by three adds -- hipcc contracts the same expression
          // to an FMA chain here, one rounding less per term)
          {
#pragma clang fp contract(off)
            part[j] = a * a + b * b + c2 * c2 + d * d;
          }
        }
        const float rstd = rsqrtf(tree(part) / C + p.g2.ln_eps);
        bf16_t* arow = reinterpret_cast<bf16_t*>(p.g2.aux_out) + (long)m * p.g2.ldc + 4 * g4;
#pragma unroll
        for (int j = 0; j < CT; ++j) {
          const f32x4 gq = *reinterpret_cast<const f32x4*>(lngs + 16 * j + 4 * g4), bq = *reinterpret_cast<const f32x4*>(lnbs + 16 * j + 4 * g4);
          // ((x - mean) rstd, rounded, then ONE fused multiply-add with gamma and beta: what ln_fwd_kernel compiles to)
          float t0, t1, t2, t3;
          {
#pragma clang fp contract(off)
            t0 = (yacc[j][0] - mean) * rstd; t1 = (yacc[j][1] - mean) * rstd; t2 = (yacc[j][2] - mean) * rstd; t3 = (yacc[j][3] - mean) * rstd;
          }
          const float o0 = __builtin_fmaf(t0, gq[0], bq[0]), o1 = __builtin_fmaf(t1, gq[1], bq[1]);
          const float o2 = __builtin_fmaf(t2, gq[2], bq[2]), o3 = __builtin_fmaf(t3, gq[3], bq[3]);
          bf16x4 ob;
          ob[0] = (bf16_t)o0; ob[1] = (bf16_t)o1; ob[2] = (bf16_t)o2; ob[3] = (bf16_t)o3;
          if (mok) *reinterpret_cast<bf16x4*>(arow + 16 * j) = ob;
        }
        if (mok && g4 == 0) {
          p.g2.ln_stats[2 * (long)m] = mean;
          p.g2.ln_stats[2 * (long)m + 1] = rstd;
        }
      }
    }
    WS_NOW(ws_e1);
    WS_ACC(ws_epi, ws_e0, ws_e1);
  }
#ifdef WIDE_STAMPS
  ws_call = __builtin_amdgcn_s_memtime() - ws_cbegin;
#endif
  WS_OUT(0, ws_cbar); WS_OUT(1, ws_fc1); WS_OUT(2, ws_gelu); WS_OUT(3, ws_fc2); WS_OUT(5, ws_epi); WS_OUT(6, ws_call); WS_OUT(4, 2ull);
}


// ------------------------------------------------------------------------------------------------------------------ backward (data path)
//   du   = (gm W2) x h'                      [M, 4C]   gm = dL/dx_out x the branch's mask (operand dtype), h' = the forward pass's hg
//   dc   = du W1                             [M, C]    = dL/da2: stored (256 channels) or finished as norm2's backward on the row (128:
//                                                       g += dLN, gm_attn = dtype(g x mask), dgamma / dbeta -- EPI_LN_BWD, the shared epilogue)
// The mirror image of the forward kernel: same tiles, same ring, same two weight slices per step -- now read through the hardware
// transpose (ds_read_b64_tr_b16: both weights are [k][n] for these products; the loader fills the [64 k][BN] images of gemm_ring.hpp's TRB
// path) -- and du's accumulators feed the second product from registers.  The first product reads its W2 columns in the forward kernel's
// permuted order (a lane's 2 x 4 accumulators = 8 consecutive hidden units: 16-byte hg loads and du stores, natural k order for product 2).
// du is written once (the fc1 weight gradient reads it) and not read back; replaces focal_linear_bwd_data(MUL_AUX) + focal_linear_bwd_data /
// focal_linear_bwd_data_ln.  Same sums in the same k order; not bit-identical (the permuted k slots of product 1 change the order inside an MFMA).
struct WideBwdParams {
  int M;
  const bf16_t* gm;   // [M][C]
  const bf16_t* hg;   // [M][H]
  const bf16_t* w1;   // [H][C]
  const bf16_t* w2;   // [C][H]
  bf16_t* du;         // [M][H]
  bf16_t* dc;         // [M][C]  (EPI_STORE)
  GemmParams g1;      // the dX-of-fc1 product's epilogue operands as focal_linear_bwd_data_ln fills them (EPI_LN_BWD)
};

template <int C> struct WideBwdLayout {
  static constexpr int NW = WideWaves<C>::NW, BM = WideWaves<C>::BM;
  static constexpr int H = 4 * C, KK = C / 32, KT1 = C / 64, CT = C / 16, NSTEP = H / 64;
  static constexpr int W2_BYTES = KT1 * 8192;   // KT1 images [64 k = c][64 n = hidden], 128-byte rows
  static constexpr int W1_BYTES = 64 * C * 2;   // one image [64 k = hidden][C], 2C-byte rows
  static constexpr int SLOT_BYTES = W1_BYTES + W2_BYTES;
  static constexpr int NP2 = KT1 * 8, NP1 = W1_BYTES / 1024, L2 = NP2 / NLOAD, L1 = NP1 / NLOAD;
  static constexpr int WPITCH = C + 4;
  static constexpr int STG_BYTES = C == 128 ? NW * 16 * WPITCH * 4 + NW * 2 * C * 4 : 0;  // shared epilogue's staging + the dgamma / dbeta fold
  static constexpr int LDS_BYTES = R * SLOT_BYTES + STG_BYTES;
};

template <int C, int EPI1>
__global__ __launch_bounds__(64 * (WideWaves<C>::NW + NLOAD)) void mlp_wide_bwd_kernel(const WideBwdParams p) {
  using L = WideBwdLayout<C>;
  constexpr int NW = L::NW, BM = L::BM;
  constexpr int H = L::H, KK = L::KK, KT1 = L::KT1, CT = L::CT, NSTEP = L::NSTEP, SLOT_BYTES = L::SLOT_BYTES, W2_BYTES = L::W2_BYTES;
  static_assert(EPI1 == EPI_STORE || (EPI1 == EPI_LN_BWD && C == 128), "dX-of-fc1 epilogue");
  extern __shared__ __attribute__((aligned(1024))) char wide_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntiles = (p.M + BM - 1) / BM, G = gridDim.x;
  const int ntl = (ntiles - (int)blockIdx.x + G - 1) / G;
  const int total = ntl * NSTEP;

  if (wave >= NW) {
    // ================================================================================================ loader waves (TRB images: gemm_ring.hpp)
    const int lw = wave - NW;
    constexpr int L1 = L::L1, L2 = L::L2;
    constexpr int CPR1 = C / 8, RPP1 = 64 / CPR1;   // the W1 image: chunks per row, rows per 1-KB piece
    uint32_t off1[L1], off2[L2];
#pragma unroll
    for (int t = 0; t < L2; ++t) {  // W2: image kt = i / 8, piece q = i % 8: k-rows (c) 8 q .. + 7 of that image, 8 chunks of this step's 64 hidden columns
      const int i = lw + NLOAD * t, kt = i >> 3, q = i & 7;
      const int krow = 8 * q + (lane >> 3), pos = lane & 7, chunk = pos ^ ((krow & 3) << 1);
      off2[t] = (uint32_t)(((kt * 64 + krow) * H + chunk * 8) * 2);
    }
#pragma unroll
    for (int t = 0; t < L1; ++t) {  // W1: piece q: k-rows (hidden) RPP1 q .. of the step, all C columns
      const int q = lw + NLOAD * t;
      const int krow = q * RPP1 + lane / CPR1, pos = lane % CPR1, chunk = pos ^ ((krow & 3) << 1);
      off1[t] = (uint32_t)((krow * C + chunk * 8) * 2);
    }
    const char* w1b = reinterpret_cast<const char*>(p.w1);
    const char* w2b = reinterpret_cast<const char*>(p.w2);
    int issued = 0, f_st = 0;
    auto issue_next = [&]() __attribute__((always_inline)) {
      const uint32_t slot = (uint32_t)(issued & (R - 1)) * SLOT_BYTES;
      const char* s2 = w2b + (long)f_st * 128;            // this step's 64 hidden columns of W2 [C][H]
      const char* s1 = w1b + (long)f_st * (64 * C * 2);   // this step's 64 hidden rows of W1 [H][C]
#pragma unroll
      for (int t = 0; t < L2; ++t) {
        const int i = lw + NLOAD * t;
        __builtin_amdgcn_global_load_lds((pipe_glb_ptr)(s2 + off2[t]), (pipe_lds_ptr)(wide_lds + slot + i * 1024), 16, 0, 0);
      }
#pragma unroll
      for (int t = 0; t < L1; ++t) {
        const int q = lw + NLOAD * t;
        __builtin_amdgcn_global_load_lds((pipe_glb_ptr)(s1 + off1[t]), (pipe_lds_ptr)(wide_lds + slot + W2_BYTES + q * 1024), 16, 0, 0);
      }
      ++issued;
      if (++f_st == NSTEP) f_st = 0;
    };
    if (total > 0) issue_next();
    for (int g = 0; g < total; ++g) {
      ring_vmcnt<0>();
      ring_barrier();
      if (issued < total) issue_next();
    }
    if (EPI1 == EPI_LN_BWD) ring_barrier();  // the consumers' __syncthreads() in pipe_ln_bwd_flush
    return;
  }

  // ==================================================================================================== consumer waves
  const int g4 = lane >> 4, l15 = lane & 15, tq = l15 >> 2, tp = lane & 3;
  const uint32_t lds0 = pipe_lds_addr(wide_lds);
  const int tr_krow = 8 * g4 + tq, tr_swz = tq << 1;
  // product 1: tile T = (s, jj) of the step's 64 hidden columns, permuted: this lane's 4 columns are hidden 32 s + 8 tp + 4 jj .. + 3
  uint32_t a1[4];
#pragma unroll
  for (int T = 0; T < 4; ++T) a1[T] = lds0 + tr_krow * 128 + (((4 * (T >> 1) + tp) ^ tr_swz) << 4) + 8 * (T & 1);
  // product 2: c-tile ct of the [64 hidden][C] image, natural order
  const int tr_in = (tp >> 1) * 16 + (tp & 1) * 8;
  const uint32_t a2 = lds0 + W2_BYTES + tr_krow * (C * 2) + tr_in;
  MaskEval meE;
  meE.init(p.g1.epi);
  float* est = reinterpret_cast<float*>(wide_lds + R * SLOT_BYTES) + wave * 16 * L::WPITCH;
  float* Gout = reinterpret_cast<float*>(p.g1.C);
  float pg[4] = {0.f, 0.f, 0.f, 0.f}, pb[4] = {0.f, 0.f, 0.f, 0.f}, lng[4] = {0.f, 0.f, 0.f, 0.f};
  if constexpr (EPI1 == EPI_LN_BWD) loadN<4>(p.g1.ln_gamma + (lane % (C / 4)) * 4, lng);
  const float zero_bias[4] = {0.f, 0.f, 0.f, 0.f};

  int slot = 0;
#pragma unroll 1
  for (int i = 0, tile = blockIdx.x; i < ntl; ++i, tile += G) {
    const int m0 = tile * BM, mbase = m0 + wave * 16, m = mbase + l15;
    const bool mok = m < p.M;
    const long mrow = mok ? m : p.M - 1;
    bf16x8 xa[KK];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) xa[kk] = *reinterpret_cast<const bf16x8*>(p.gm + mrow * C + kk * 32 + 8 * g4);
    f32x4 dacc[CT];
#pragma unroll
    for (int j = 0; j < CT; ++j) dacc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16_t* hgrow = p.hg + mrow * H + 8 * g4;
    bf16_t* durow = p.du + (long)m * H + 8 * g4;
    // h' of the NEXT step is requested before this step's du stores are issued: waiting for it never waits for a store (in-order vmcnt)
    bf16x8 hgn[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) hgn[s] = *reinterpret_cast<const bf16x8*>(hgrow + 32 * s);

#pragma unroll 1
    for (int st = 0; st < NSTEP; ++st) {
      bf16x8 hgc[2] = {hgn[0], hgn[1]};
      if (st + 1 < NSTEP) {
#pragma unroll
        for (int s = 0; s < 2; ++s) hgn[s] = *reinterpret_cast<const bf16x8*>(hgrow + 64 * (st + 1) + 32 * s);
      }
      ring_barrier();
      const uint32_t sb = (uint32_t)slot * SLOT_BYTES;
      // ---- product 1: v[T][e] = sum_c W2[c][hidden(T, 4 g + e)] gm[m][c]
      f32x4 v[4];
#pragma unroll
      for (int T = 0; T < 4; ++T) v[T] = f32x4{0.f, 0.f, 0.f, 0.f};
      pipe_static_for<0, KT1>([&](auto kt_) {
        constexpr int kt = decltype(kt_)::value;
        pipe_static_for<0, 2>([&](auto kk_) {
          constexpr int kk = decltype(kk_)::value;
          bf16x4 lo[4], hi[4];
          pipe_static_for<0, 4>([&](auto T_) {
            constexpr int T = decltype(T_)::value;
            lo[T] = pipe_lds_read_tr<kt * 8192 + kk * 32 * 128>(a1[T] + sb);
            hi[T] = pipe_lds_read_tr<kt * 8192 + kk * 32 * 128 + 4 * 128>(a1[T] + sb);
          });
          bf16x8 w[4];
#pragma unroll
          for (int T = 0; T < 4; ++T) w[T] = __builtin_shufflevector(lo[T], hi[T], 0, 1, 2, 3, 4, 5, 6, 7);
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
#pragma unroll
          for (int T = 0; T < 4; ++T) v[T] = mma16(w[T], xa[2 * kt + kk], v[T]);
        });
      });
      // ---- du = v x h' (the arithmetic of EPI_MUL_AUX), 8 consecutive hidden units per lane and s; stored once, kept for product 2
      bf16x8 duf[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float x0 = v[2 * s][e] * p.g1.alpha, x1 = v[2 * s + 1][e] * p.g1.alpha;
          x0 *= (float)hgc[s][e];
          x1 *= (float)hgc[s][4 + e];
          duf[s][e] = (bf16_t)x0;
          duf[s][4 + e] = (bf16_t)x1;
        }
        if (mok) *reinterpret_cast<bf16x8*>(durow + 64 * st + 32 * s) = duf[s];
      }
      // ---- product 2: dacc[ct][e] += sum_{hidden of this step} W1[hidden][16 ct + 4 g + e] du[m][hidden]
      pipe_static_for<0, CT / 2>([&](auto jq_) {
        constexpr int jq = decltype(jq_)::value;
        bf16x4 lo[4], hi[4];
        pipe_static_for<0, 2>([&](auto jj_) {
          constexpr int jj = decltype(jj_)::value, ct = 2 * jq + jj;
          pipe_static_for<0, 2>([&](auto s_) {
            constexpr int s = decltype(s_)::value;
            const uint32_t a = a2 + sb + ((((ct * 16) >> 3) ^ tr_swz) << 4);
            lo[2 * jj + s] = pipe_lds_read_tr<s * 32 * C * 2>(a);
            hi[2 * jj + s] = pipe_lds_read_tr<s * 32 * C * 2 + 4 * C * 2>(a);
          });
        });
        bf16x8 w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) w[q] = __builtin_shufflevector(lo[q], hi[q], 0, 1, 2, 3, 4, 5, 6, 7);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          dacc[2 * jq + jj] = mma16(w[2 * jj], duf[0], dacc[2 * jq + jj]);
          dacc[2 * jq + jj] = mma16(w[2 * jj + 1], duf[1], dacc[2 * jq + jj]);
        }
      });
      slot ^= 1;
    }

    // ---- epilogue of the second product
    if constexpr (EPI1 == EPI_LN_BWD) {
      PipePre<float, EPI1, C, 1> pre;
      pipe_epilogue_prefetch<float, EPI1, BM, C, NW, 1>(p.g1, p.g1.resid, m0, 0, mbase, 0, lane, Gout, pre);
      pipe_epilogue_finish<float, EPI1, BM, C, NW, 1>(p.g1, p.g1.alpha, dacc, est, meE, m0, 0, mbase, 0, lane, Gout, zero_bias, pg, pb, lng, pre);
    } else {
      bf16_t* drow = p.dc + (long)m * C + 4 * g4;
#pragma unroll
      for (int j = 0; j < CT; ++j)
        if (mok) store4(drow + 16 * j, dacc[j] * p.g1.alpha);
    }
  }
  if constexpr (EPI1 == EPI_LN_BWD)
    pipe_ln_bwd_flush<C, NW, 1, 4>(p.g1, pg, pb, reinterpret_cast<float*>(wide_lds + R * SLOT_BYTES) + NW * 16 * L::WPITCH, 0, 0, wave, lane, tid);
}

}  // namespace focal_mlp_wide
using namespace focal_mlp_wide;

static MaskParams wide_mask(const focal_drop_desc& d, int ncols) {
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

// Launched back to back on cold operands the one-launch form only ties the two launches it replaces (profiles/r6_mlp_wide.txt: 0.88-1.07 x at
// 128 channels, 0.66-1.00 x at 256 -- both forms are bound by vector-instruction issue, not by the 8C bytes per token the fusion removes);
// INSIDE the replayed step, where the other modality's stream runs beside it, it wins: +1.5 % on the SW_Transformer step, three interleaved
// same-box repetitions (fewer launches, 75 MB less traffic per block, and at 256 channels a grid that leaves CUs to the other stream).
// FOCAL_MLP_WIDE=0 keeps the two launches, =128 / =256 fuses that width only (same-box A/B: tools/ab_wide.sh).
extern "C" int focal_mlp_wide_supported(int dtype, int C_, int hidden) {
  if (!(dtype == FOCAL_BF16 && (C_ == 128 || C_ == 256) && hidden == 4 * C_)) return 0;
  const char* on = getenv("FOCAL_MLP_WIDE");
  if (on == nullptr || strcmp(on, "1") == 0) return 1;
  return atoi(on) == C_;
}

template <int C, int EPI2, bool DROP, bool FULL, bool PROJ, int NWX = 0>
static int launch_wide_fwd_t(const WideFwdParams& p, hipStream_t st) {
  using L = WideLayout<C, NWX>;
  auto kern = mlp_wide_fwd_kernel<C, EPI2, DROP, FULL, PROJ, NWX>;
  static std::atomic<bool> attr_set{false};
  if (!attr_set.load(std::memory_order_acquire)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, L::LDS_BYTES) != hipSuccess) {
      focal_set_error("mlp_wide_fwd: cannot reserve %d bytes of LDS", L::LDS_BYTES);
      return FOCAL_EHIP;
    }
    attr_set.store(true, std::memory_order_release);
  }
  constexpr int BM = L::BM, NW = L::NW;
  const int ntiles = (p.M + BM - 1) / BM, cus = focal_cu_count();
  // persistent, one workgroup per CU; every workgroup the same number of tiles where the tile count allows
  const int rounds = (ntiles + cus - 1) / cus;
  const int grid = (ntiles + rounds - 1) / rounds;
  FOCAL_LAUNCH(kern, dim3(grid), dim3(64 * (NW + NLOAD)), L::LDS_BYTES, st, p);
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}

template <int C, int EPI2>
static int launch_wide_fwd(const WideFwdParams& p, hipStream_t st) {
  if constexpr (C == 256) {
    // 64-row tiles with four consumer waves while they fit ONE round of the chip (the step: the seismic encoder's 9 216 rows = 144 tiles; the
    // audio encoder's 18 432 rows would be 288 = two rounds and keep the 96-row tiles); only the step's own instantiation (dropout on, whole
    // tiles, proj folded).  Measured inside the step: 51 546 / 51 213 / 51 742 against 51 394 / 51 340 / 51 541 windows/s with six waves
    // everywhere (tools/ab_wide_nw4.sh) -- noise: the CUs the 96-tile launch leaves idle are the other encoder's anyway.  Opt-in
    // (FOCAL_LAB_WIDE_NW4=1).
    static const bool nw4 = [] { const char* e = getenv("FOCAL_LAB_WIDE_NW4"); return e != nullptr && e[0] == '1'; }();
    if (nw4 && p.o != nullptr && p.drop_h.p_elem > 0.f && p.M % 64 == 0 && p.M / 64 <= focal_cu_count() && p.M / 64 > focal_cu_count() / 3)
      return launch_wide_fwd_t<C, EPI2, true, true, true, 4>(p, st);
  }
  const bool drop = p.drop_h.p_elem > 0.f, full = p.M % WideLayout<C>::BM == 0;
  if (p.o != nullptr) {
    if (drop) return full ? launch_wide_fwd_t<C, EPI2, true, true, true>(p, st) : launch_wide_fwd_t<C, EPI2, true, false, true>(p, st);
    return full ? launch_wide_fwd_t<C, EPI2, false, true, true>(p, st) : launch_wide_fwd_t<C, EPI2, false, false, true>(p, st);
  }
  if (drop) return full ? launch_wide_fwd_t<C, EPI2, true, true, false>(p, st) : launch_wide_fwd_t<C, EPI2, true, false, false>(p, st);
  return full ? launch_wide_fwd_t<C, EPI2, false, true, false>(p, st) : launch_wide_fwd_t<C, EPI2, false, false, false>(p, st);
}

struct WideProjArgs { const void* o; const float* x; const void* wp; const float* bp; const focal_drop_desc* drop; const float* g2; const float* bt2; float* st2; };

static int mlp_wide_fwd_launch(const char* who, const focal_mlp_desc* d, const void* a, const float* resid, const void* w1, const float* b1, const void* w2,
                               const float* b2, float* y, void* h, void* hg, const float* ln_gamma, const float* ln_beta, void* y_ln,
                               float* ln_stats, const WideProjArgs* pj, void* stream) {
  FOCAL_CHECK_ARG(d != nullptr, "%s: null descriptor", who);
  FOCAL_CHECK_ARG(d->dtype == FOCAL_BF16 && (d->C == 128 || d->C == 256) && d->hidden == 4 * d->C,
                  "%s: bf16, C = 128 or 256, hidden = 4 C (got dtype %d, C %d, hidden %d)", who, d->dtype, d->C, d->hidden);
  FOCAL_CHECK_ARG(d->M > 0, "%s: M = %d", who, d->M);
  FOCAL_CHECK_ARG(a && resid && w1 && b1 && w2 && b2 && y && h && hg, "%s: null tensor", who);
  const bool ln = y_ln != nullptr;
  if (ln) FOCAL_CHECK_ARG(ln_gamma && ln_beta && ln_stats, "%s: the fused LayerNorm needs gamma, beta and a statistics buffer", who);
  WideFwdParams p;
  memset(&p, 0, sizeof(p));
  p.M = d->M;
  p.a = reinterpret_cast<const bf16_t*>(a);
  p.w1 = reinterpret_cast<const bf16_t*>(w1);
  p.b1 = b1;
  p.w2 = reinterpret_cast<const bf16_t*>(w2);
  p.h = reinterpret_cast<bf16_t*>(h);
  p.hg = reinterpret_cast<bf16_t*>(hg);
  p.drop_h = wide_mask(d->drop_hidden, d->hidden);
  GemmParams& g = p.g2;  // (as focal_linear_fwd / focal_linear_resid_ln_fwd fill it for fc2)
  g.M = d->M; g.N = d->C; g.K = d->hidden;
  g.C = y; g.ldc = d->C;
  g.batch = 1; g.splits = 1; g.alpha = 1.f;
  g.bias = b2;
  g.resid = resid; g.ldr = d->C;
  g.aux_out = y_ln;
  g.ln_gamma = ln_gamma; g.ln_beta = ln_beta; g.ln_stats = ln_stats; g.ln_eps = d->ln_eps;
  g.epi = wide_mask(d->drop_out, d->C);
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
    p.drop_p = wide_mask(dd, d->C);
    p.ng2 = pj->g2; p.nbt2 = pj->bt2; p.st2 = pj->st2; p.ln2_eps = d->ln_eps;
  }
#ifdef WIDE_STAMPS
  g.colsumA = reinterpret_cast<float*>(ln_stats && !ln ? ln_stats : nullptr);  // lab build: the stamp buffer rides in ln_stats when no LayerNorm is asked for
#endif
  hipStream_t st = (hipStream_t)stream;
  if (d->C == 128) return ln ? launch_wide_fwd<128, EPI_RESID_LN>(p, st) : launch_wide_fwd<128, EPI_RESID>(p, st);
  return ln ? launch_wide_fwd<256, EPI_RESID_LN>(p, st) : launch_wide_fwd<256, EPI_RESID>(p, st);
}

extern "C" int focal_mlp_wide_fwd(const focal_mlp_desc* d, const void* a, const float* resid, const void* w1, const float* b1, const void* w2,
                                  const float* b2, float* y, void* h, void* hg, const float* ln_gamma, const float* ln_beta, void* y_ln,
                                  float* ln_stats, void* stream) {
  return mlp_wide_fwd_launch("mlp_wide_fwd", d, a, resid, w1, b1, w2, b2, y, h, hg, ln_gamma, ln_beta, y_ln, ln_stats, nullptr, stream);
}

extern "C" int focal_mlp_wide_proj_supported(int dtype, int C_, int hidden) {
  return focal_mlp_wide_supported(dtype, C_, hidden) && focal_mlp_proj_width_enabled(C_);
}

extern "C" int focal_mlp_wide_proj_fwd(const focal_mlp_desc* d, const void* o, const float* x, const void* wp, const float* bp, const focal_drop_desc* drop_proj,
                                       const float* g2, const float* bt2, float* x_mid, void* a2, float* st2, const void* w1, const float* b1, const void* w2,
                                       const float* b2, float* y, void* h, void* hg, const float* ln_gamma, const float* ln_beta, void* y_ln,
                                       float* ln_stats, void* stream) {
  const WideProjArgs pj = {o, x, wp, bp, drop_proj, g2, bt2, st2};
  return mlp_wide_fwd_launch("mlp_wide_proj_fwd", d, a2, x_mid, w1, b1, w2, b2, y, h, hg, ln_gamma, ln_beta, y_ln, ln_stats, &pj, stream);
}

// NOT the default: inside the replayed step the one-launch backward data path is neutral at 128 channels and costs 1.2 % at 256 (three
// interleaved same-box repetitions, tools/ab_wide_bwd.sh; profiles/r6_mlp_wide.txt) -- the two launches it replaces are the ring GEMM's
// fastest shapes (26 us for both at M = 9 216 against 40 us for the forward pair), there is less to win and the transposed fragment reads
// (two ds_read_b64_tr_b16 per MFMA against one ds_read_b128) cost more.  FOCAL_MLP_WIDE_BWD=1 / 128 / 256 selects it.
extern "C" int focal_mlp_wide_bwd_supported(int dtype, int C_, int hidden) {
  const char* sel = getenv("FOCAL_MLP_WIDE_BWD");
  if (sel == nullptr || (strcmp(sel, "1") != 0 && atoi(sel) != C_)) return 0;
  return focal_mlp_wide_supported(dtype, C_, hidden);
}

template <int C, int EPI1>
static int launch_wide_bwd(const WideBwdParams& p, hipStream_t st) {
  using L = WideBwdLayout<C>;
  auto kern = mlp_wide_bwd_kernel<C, EPI1>;
  static std::atomic<bool> attr_set{false};
  if (!attr_set.load(std::memory_order_acquire)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, L::LDS_BYTES) != hipSuccess) {
      focal_set_error("mlp_wide_bwd: cannot reserve %d bytes of LDS", L::LDS_BYTES);
      return FOCAL_EHIP;
    }
    attr_set.store(true, std::memory_order_release);
  }
  constexpr int BM = L::BM, NW = L::NW;
  const int ntiles = (p.M + BM - 1) / BM, cus = focal_cu_count();
  const int rounds = (ntiles + cus - 1) / cus;
  const int grid = (ntiles + rounds - 1) / rounds;
  FOCAL_LAUNCH(kern, dim3(grid), dim3(64 * (NW + NLOAD)), L::LDS_BYTES, st, p);
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}

extern "C" int focal_mlp_wide_bwd_data(const focal_mlp_desc* d, const void* gm, const void* hg, const void* w1, const void* w2, void* du, void* dc,
                                       const float* ln_x, const float* ln_stats, const float* ln_gamma, float* g, void* g_masked,
                                       const focal_drop_desc* mask, float* dgamma, float* dbeta, void* stream) {
  FOCAL_CHECK_ARG(d != nullptr, "mlp_wide_bwd_data: null descriptor");
  FOCAL_CHECK_ARG(d->dtype == FOCAL_BF16 && (d->C == 128 || d->C == 256) && d->hidden == 4 * d->C,
                  "mlp_wide_bwd_data: bf16, C = 128 or 256, hidden = 4 C (got dtype %d, C %d, hidden %d)", d->dtype, d->C, d->hidden);
  FOCAL_CHECK_ARG(d->M > 0 && gm && hg && w1 && w2 && du, "mlp_wide_bwd_data: null tensor or M = %d", d->M);
  const bool ln = ln_x != nullptr;
  FOCAL_CHECK_ARG(ln || dc, "mlp_wide_bwd_data: neither dc nor the LayerNorm operands");
  if (ln) FOCAL_CHECK_ARG(d->C == 128 && ln_stats && ln_gamma && g && dgamma && dbeta && (g_masked || !mask),
                          "mlp_wide_bwd_data: the fused LayerNorm backward exists at 128 channels and needs x, statistics, gamma, g, dgamma, dbeta");
  WideBwdParams p;
  memset(&p, 0, sizeof(p));
  p.M = d->M;
  p.gm = reinterpret_cast<const bf16_t*>(gm);
  p.hg = reinterpret_cast<const bf16_t*>(hg);
  p.w1 = reinterpret_cast<const bf16_t*>(w1);
  p.w2 = reinterpret_cast<const bf16_t*>(w2);
  p.du = reinterpret_cast<bf16_t*>(du);
  p.dc = reinterpret_cast<bf16_t*>(dc);
  GemmParams& q = p.g1;  // (as focal_linear_bwd_data_ln fills it for the dX of fc1)
  q.M = d->M; q.N = d->C; q.K = d->hidden;
  q.C = g; q.ldc = d->C;
  q.batch = 1; q.splits = 1; q.alpha = 1.f;
  q.resid = ln_x; q.ldr = d->C;
  q.ln_stats = const_cast<float*>(ln_stats);
  q.ln_gamma = ln_gamma;
  q.ln_dgamma = dgamma; q.ln_dbeta = dbeta;
  q.aux_out = g_masked;
  focal_drop_desc dd;
  memset(&dd, 0, sizeof(dd));
  if (mask) dd = *mask;
  q.epi = wide_mask(dd, d->C);
  hipStream_t st = (hipStream_t)stream;
  if (ln) return launch_wide_bwd<128, EPI_LN_BWD>(p, st);
  return d->C == 128 ? launch_wide_bwd<128, EPI_STORE>(p, st) : launch_wide_bwd<256, EPI_STORE>(p, st);
}
