// Pipelined bf16 GEMM for the deep Swin stages (M = 9 k - 74 k tokens, K, N = 128 - 1024):
//
//   C[m][n] = epi( sum_k A[m][k] * W[n][k] + bias[n] )        A: [M][K] bf16, W: [N][K] bf16 (both k-contiguous)
//
// The 64 x 64 kernel of gemm.hpp hides load latency with occupancy (5-6 workgroups per CU, one k-step in flight
// each) and reads every fragment it multiplies from LDS 1.25 times per MFMA; at these shapes that leaves the matrix
// pipe 12 % busy.  This kernel trades occupancy for a pipeline inside the workgroup:
//   * 128-row tiles, 2 x 2 waves, 64 x (BN / 2) outputs per wave: 0.5 - 0.75 fragment reads per MFMA;
//   * operands go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass) into an
//     NST-deep ring; NST - 1 k-steps are in flight while one is multiplied; ONE raw s_barrier per k-step with a
//     counted s_waitcnt vmcnt (never a __syncthreads(): its fence would drain the ring);
//   * an LDS-DMA wave-instruction writes 1 KB contiguously (8 rows x 128 B), so rows cannot be padded: the 16-byte
//     chunk c of tile row r is stored at chunk position c ^ ((r >> 1) & 7) -- applied on the per-lane SOURCE address
//     when filling and on the ds_read_b128 address when reading (16 lanes of a fragment read then cover all 64 banks).
// The W operand is either [N][K] (forward) or [K][N] (TRB: the same weight read for a data gradient; fragments by
// ds_read_b64_tr_b16 from a [64 k][BN] image with its own swizzle) -- no transposed copy of a weight exists in HBM.
// Requires N % BN == 0, K % 64 == 0, 16-byte aligned rows (the dispatcher checks; everything else stays on gemm.hpp's
// kernel); M may be ragged.
#pragma once
#include <type_traits>
#include "gemm.hpp"

template <int N> __device__ __forceinline__ void pipe_wait_barrier() {
  // all of this wave's LDS-DMA pieces except the newest N have landed; then meet the other waves
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

typedef __attribute__((address_space(3))) void* pipe_lds_ptr;
typedef const __attribute__((address_space(1))) void* pipe_glb_ptr;

// Row addressing in the epilogues: a uniform pointer to the tile's first row + a 32-bit lane offset (24-bit multiply: the row inside the tile
// x the row pitch in bytes, both far below 2^24).  `ptr + (long)m * ld + n` per row and tensor was a 64-bit multiply-add chain at quarter rate --
// 190-240 of the 2 200-3 100 vector instructions of these kernels, a quarter of the epilogues' issue slots.
template <class T> __device__ __forceinline__ const T* pipe_row(const T* tile0, int lrow, long ld, int n) {
  return reinterpret_cast<const T*>(reinterpret_cast<const char*>(tile0) + (__umul24((uint32_t)lrow, (uint32_t)ld * (uint32_t)sizeof(T)) + (uint32_t)n * (uint32_t)sizeof(T)));
}
template <class T> __device__ __forceinline__ T* pipe_row(T* tile0, int lrow, long ld, int n) {
  return reinterpret_cast<T*>(reinterpret_cast<char*>(tile0) + (__umul24((uint32_t)lrow, (uint32_t)ld * (uint32_t)sizeof(T)) + (uint32_t)n * (uint32_t)sizeof(T)));
}
// LDS fragment reads as inline asm.  hipcc cannot tell which ring stage a visible LDS read touches, so it guards EVERY such read with
// s_waitcnt vmcnt(0) while an LDS-DMA is in flight -- i.e. it waits for the prefetch it has just issued, and a ring of any depth
// degenerates to load -> wait -> multiply (the r1 kernels ran like that; build/isa shows the wait in front of the first ds_read of
// every k-step).  Reads the compiler cannot see are not guarded; the kernel orders them itself: the counted vmcnt of
// pipe_wait_barrier before, pipe_lds_wait (lgkmcnt(0), tied to the fragment registers so no MFMA can move above it) after.
__device__ __forceinline__ uint32_t pipe_lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}
template <int OFF> __device__ __forceinline__ bf16x8 pipe_lds_read128(uint32_t a) {
  bf16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF));
  return v;
}
template <int OFF> __device__ __forceinline__ bf16x4 pipe_lds_read_tr(uint32_t a) {
  bf16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF));
  return v;
}
// The wait that orders the inline-asm fragment reads before the MFMAs: lgkmcnt(0) tied to the fragment registers (an MFMA is
// register-only code, so nothing else stops the scheduler from moving it above a bare s_waitcnt).  Volatile asm statements keep their
// program order, so fragments beyond the first statement's operand budget are tied by empty statements behind it.
template <int N> __device__ __forceinline__ void pipe_tie(bf16x8 (&v)[N], int from) {
#pragma unroll
  for (int i = from; i + 3 < N; i += 4) asm volatile("" : "+v"(v[i]), "+v"(v[i + 1]), "+v"(v[i + 2]), "+v"(v[i + 3]));
}
template <int NA, int NB> __device__ __forceinline__ void pipe_lds_wait(bf16x8 (&a)[NA], bf16x8 (&b)[NB]) {
  static_assert((NA == 4 && (NB == 2 || NB == 4)) || (NA == 2 && NB == 2) || ((NA == 1 || NA == 2) && (NB == 4 || NB == 8 || NB == 16)),
                "fragment counts of the wave tiles in use");
  if constexpr (NA == 4 && NB == 4)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
  else if constexpr (NA == 4)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]));
  else if constexpr (NA == 2 && NB == 2)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]));
  else {  // row-complete wave tiles (1 or 2 row fragments x 4 / 8 / 16 column fragments): every MFMA reads an `a` fragment
    if constexpr (NA == 2)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
    else
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
    pipe_tie(b, 4);
  }
}
template <int I, int N, typename F> __device__ __forceinline__ void pipe_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    pipe_static_for<I + 1, N>(f);
  }
}

// LayerNorm epilogues when TWO waves side by side cover a row (256 channels: 4 x 2 waves of 16 x 128 outputs on a 64 x 256 tile -- a
// single wave per row means 16 rows x 256 columns of serial epilogue per wave at 8 waves per CU, which lost to GEMM + LayerNorm as two
// launches).  Each wave reduces its half row, the halves meet through one LDS exchange per pass:
//   forward  (EPI_RESID_LN): half-row mean and centred sum of squares, merged as  mean = (ma + mb) / 2,  M2 = M2a + M2b + (n/2)(ma - mb)^2 / 2 * 2
//   backward (EPI_LN_BWD):   the two row sums of  g = dy gamma  and  g xhat  are linear: the halves' partial sums are added.
template <int EPI, int BM, int BN, int WGM, int TM, int TN>
__device__ __forceinline__ void pipe_ln_epilogue_two_waves(const GemmParams& p, f32x4 (&acc)[TM][TN], float* lds, int m0, int n0, int wm, int wn,
                                                           int wave, int lane, int tid, float* Cf) {
  constexpr int NW = WGM * 2, WC = BN / 2, WR = BM / WGM, WPITCH = WC + 4;
  constexpr int LPR = WC / 4, RPI = 64 / LPR, NIT = 16 / RPI;
  static_assert(LPR == 32, "two waves per row: 128 columns per wave");
  float* est = lds + wave * 16 * WPITCH;
  float2* xch = reinterpret_cast<float2*>(lds + NW * 16 * WPITCH);   // [BM rows][2 halves]
  float* red = lds + NW * 16 * WPITCH + BM * 4;                       // [NW][2][WC] (EPI_LN_BWD)
  MaskEval meE;
  meE.init(p.epi);
  const int c = (lane % LPR) * 4, n = n0 + wn * WC + c;
  float bias[4] = {0.f, 0.f, 0.f, 0.f}, lng[4], lnb[4] = {0.f, 0.f, 0.f, 0.f};
  if (EPI == EPI_RESID_LN && p.bias) loadN<4>(p.bias + n, bias);
  loadN<4>(p.ln_gamma + n, lng);
  if (EPI == EPI_RESID_LN) loadN<4>(p.ln_beta + n, lnb);
  float pg[4] = {0.f, 0.f, 0.f, 0.f}, pb[4] = {0.f, 0.f, 0.f, 0.f};
  // (uniform pointers to the tile's first row: pipe_row)
  const float* resid_t = p.resid + (long)m0 * p.ldr;
  float* Cf_t = Cf + (long)m0 * p.ldc;
  bf16_t* auxo_t = reinterpret_cast<bf16_t*>(p.aux_out) + (long)m0 * p.ldc;
  float* stats_t = p.ln_stats + 2 * (long)m0;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int lrow0 = wm * WR + i * 16, mbase = m0 + lrow0;
    float rpre[NIT][4], gpre[EPI == EPI_LN_BWD ? NIT : 1][4];
    float2 spre[EPI == EPI_LN_BWD ? NIT : 1];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {  // every global read of the pass is requested before the accumulators are staged
      const int mm = min(mbase + it * RPI + lane / LPR, p.M - 1);
      loadN<4>(pipe_row(resid_t, mm - m0, p.ldr, n), rpre[it]);
      if (EPI == EPI_LN_BWD) {
        if (Cf != nullptr) loadN<4>(pipe_row(Cf_t, mm - m0, p.ldc, n), gpre[it]);  // (NULL: dgamma / dbeta only, see the one-wave-per-row epilogue below)
        spre[it] = *reinterpret_cast<const float2*>(stats_t + 2 * (mm - m0));
      }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const f32x4 v = acc[i][j] * p.alpha;
      *reinterpret_cast<float4*>(est + (lane & 15) * WPITCH + j * 16 + (lane >> 4) * 4) = make_float4(v[0], v[1], v[2], v[3]);
    }
    float a[NIT][4], b[NIT][4];  // forward: the finished residual row (a); backward: g = dy gamma (a) and xhat (b)
    float h1[NIT], h2[NIT];      // this half row's two reduced quantities
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int row = it * RPI + lane / LPR, m = mbase + row;
      float v[4];
      loadN<4>(est + row * WPITCH + c, v);
      if (EPI == EPI_RESID_LN) {
        const float rowm = meE.row_mult(min(m, p.M - 1));
        float s1 = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          a[it][e] = rpre[it][e] + (v[e] + bias[e]) * rowm * meE.elem_mult(min(m, p.M - 1), n + e);
          s1 += a[it][e];
        }
        if (m < p.M) storeN<4>(pipe_row(Cf_t, m - m0, p.ldc, n), a[it]);
        s1 = row16_sum(s1);
        s1 = xadd16(s1);
        const float mh = s1 * (1.0f / WC);
        float s2 = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) s2 += (a[it][e] - mh) * (a[it][e] - mh);
        s2 = row16_sum(s2);
        s2 = xadd16(s2);
        h1[it] = mh; h2[it] = s2;
      } else {
        const float mean = spre[it].x, rstd = spre[it].y;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          b[it][e] = (rpre[it][e] - mean) * rstd;
          a[it][e] = v[e] * lng[e];
          s1 += a[it][e];
          s2 += a[it][e] * b[it][e];
          if (m < p.M) { pg[e] += v[e] * b[it][e]; pb[e] += v[e]; }
        }
        s1 = row16_sum(s1); s2 = row16_sum(s2);
        s1 = xadd16(s1); s2 = xadd16(s2);
        h1[it] = s1; h2[it] = s2;
      }
      if ((lane % LPR) == 0) xch[(lrow0 + row) * 2 + wn] = make_float2(h1[it], h2[it]);
    }
    __syncthreads();  // (every wave runs the same TM passes: the halves of a row meet here)
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int row = it * RPI + lane / LPR, m = mbase + row;
      const float2 o = xch[(lrow0 + row) * 2 + (wn ^ 1)];
      if (m >= p.M) continue;
      if (EPI == EPI_RESID_LN) {
        const float mean = 0.5f * (h1[it] + o.x), dm = h1[it] - o.x;
        const float rstd = rsqrtf((h2[it] + o.y + 0.5f * WC * dm * dm) * (1.0f / BN) + p.ln_eps);
        float y[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = (a[it][e] - mean) * rstd * lng[e] + lnb[e];
        storeN<4>(pipe_row(auxo_t, m - m0, p.ldc, n), y);
        if (wn == 0 && (lane % LPR) == 0) *reinterpret_cast<float2*>(stats_t + 2 * (m - m0)) = make_float2(mean, rstd);
      } else {
        if (Cf == nullptr) continue;
        const float rstd = spre[it].y, m1 = (h1[it] + o.x) * (1.0f / BN), m2 = (h2[it] + o.y) * (1.0f / BN);
        float g[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] = gpre[it][e] + rstd * (a[it][e] - m1 - b[it][e] * m2);
        storeN<4>(pipe_row(Cf_t, m - m0, p.ldc, n), g);
        if (p.aux_out) {
          const float rowm = meE.row_mult(m);
          float gq[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) gq[e] = g[e] * rowm * meE.elem_mult(m, n + e);
          storeN<4>(pipe_row(auxo_t, m - m0, p.ldc, n), gq);
        }
      }
    }
  }
  if (EPI == EPI_LN_BWD) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      pg[e] = xadd32(pg[e]);
      pb[e] = xadd32(pb[e]);
    }
    if (lane < LPR) {
      *reinterpret_cast<float4*>(red + wave * 2 * WC + c) = make_float4(pg[0], pg[1], pg[2], pg[3]);
      *reinterpret_cast<float4*>(red + wave * 2 * WC + WC + c) = make_float4(pb[0], pb[1], pb[2], pb[3]);
    }
    __syncthreads();
    for (int i = tid; i < 2 * BN; i += 64 * NW) {  // i -> (which sum, column of the full row): the WGM waves of that column half
      const int which = i / BN, col = i % BN, half = col / WC, cc = col % WC;
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < WGM; ++w) s += red[(w * 2 + half) * 2 * WC + which * WC + cc];
      atomicAdd((which ? p.ln_dbeta : p.ln_dgamma) + n0 + col, s);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------- epilogue of a tile
// (shared by focal_gemm_pipe_kernel, one tile per workgroup, and the persistent ring kernel of gemm_ring.hpp)
// One wave's WR x WC outputs: 16 rows at a time transposed through the wave-private LDS region `est`, then walked row-major (16 B per lane).
// pg / pb (EPI_LN_BWD): this lane's column partials of dy * xhat and dy -- accumulated over every tile the workgroup finishes, folded
// and added to dgamma / dbeta once by pipe_ln_bwd_flush; lng: the lane's four gamma values.
// pipe_epilogue_group: ONE 16-row group (rows mbase .. mbase + 15 of the tile at (m0, n0); accrow = the group's TN accumulator tiles);
// pipe_epilogue_rows: a wave's groups I0 .. I1 (all of them by default).  bias: the lane's CPL bias values (pipe_epilogue_bias).
template <int CPL> struct PipeBias { float v[CPL]; };
// (alpha, bias, resid travel as scalar arguments: read through the argument struct from inside the ring kernel's nested loops -- or
// packed into a small struct -- hipcc kept exactly these 20 bytes in a scratch copy)
template <typename TC, int EPI, int BN, int WGN>
__device__ __forceinline__ PipeBias<(sizeof(TC) == 2) ? 8 : 4> pipe_epilogue_bias(const float* bias_ptr, int n0, int wn, int lane) {
  constexpr int WC = BN / WGN, CPL = (sizeof(TC) == 2) ? 8 : 4, LPR = WC / CPL;
  const int n = n0 + wn * WC + (lane % LPR) * CPL;
  PipeBias<CPL> b;
#pragma unroll
  for (int e = 0; e < CPL; ++e) b.v[e] = 0.f;
  if (bias_ptr) {
    // (an explicit global-memory load: through the generic pointer this was a flat_load, whose result waits for vmcnt(0) AND lgkmcnt(0))
    typedef float f32x4g __attribute__((ext_vector_type(4)));
    const __attribute__((address_space(1))) f32x4g* g = (const __attribute__((address_space(1))) f32x4g*)(bias_ptr + n);
#pragma unroll
    for (int q = 0; q < CPL / 4; ++q) {
      const f32x4g t = g[q];
      b.v[4 * q] = t[0]; b.v[4 * q + 1] = t[1]; b.v[4 * q + 2] = t[2]; b.v[4 * q + 3] = t[3];
    }
  }
  return b;
}
// What a 16-row group's epilogue reads from memory, requested up front (pipe_epilogue_prefetch) and consumed by pipe_epilogue_finish:
//   r: EPI_RESID / EPI_RESID_LN / EPI_LN_BWD: the fp32 residual rows (the LayerNorm's input rows);  EPI_MUL_AUX: the aux rows (as floats)
//   g, s: EPI_LN_BWD: the residual-stream gradient rows the result is added to, the rows' {mean, rstd}
// The one-tile-per-workgroup kernel requests them at the top of the group (hipcc will not move a global load above the staging's LDS
// traffic: left inside the row loop each row exposed a full memory latency); the ring kernel requests a whole tile's rows BEFORE the tile's
// k loop, so they land while it multiplies (lab: the epilogue's exposed load latency was half of a MUL_AUX / RESID launch).
// Rows past M re-read the last row.
template <typename TC, int EPI, int BN, int WGN> struct PipePre {
  static constexpr int CPL = (sizeof(TC) == 2) ? 8 : 4, LPR = (BN / WGN) / CPL, RPI = 64 / LPR, NIT = 16 / RPI;
  static constexpr bool HAS_R = EPI == EPI_RESID || EPI == EPI_RESID_LN || EPI == EPI_LN_BWD || EPI == EPI_MUL_AUX;
  float r[HAS_R ? NIT : 1][CPL];
  float g[EPI == EPI_LN_BWD ? NIT : 1][CPL];
  float2 s[EPI == EPI_LN_BWD ? NIT : 1];
};
template <typename TC, int EPI, int BM, int BN, int WGM, int WGN>
__device__ __forceinline__ void pipe_epilogue_prefetch(const GemmParams& p, const float* ea_resid, int m0, int n0, int mbase, int wn, int lane, const TC* C,
                                                       PipePre<TC, EPI, BN, WGN>& pre) {
  using P = PipePre<TC, EPI, BN, WGN>;
  constexpr int WC = BN / WGN, CPL = P::CPL, LPR = P::LPR, RPI = P::RPI, NIT = P::NIT;
  const int c = (lane % LPR) * CPL, n = n0 + wn * WC + c;
  if constexpr (P::HAS_R) {
    const float* resid_t = ea_resid + (long)m0 * p.ldr;
    const float* Cc_t = reinterpret_cast<const float*>(C) + (long)m0 * p.ldc;
    const bf16_t* aux_t = reinterpret_cast<const bf16_t*>(p.aux) + (long)m0 * p.ldaux;
    const float* stats_t = p.ln_stats + 2 * (long)m0;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int mm = min(mbase + it * RPI + lane / LPR, p.M - 1);
      if constexpr (EPI == EPI_MUL_AUX) {
        loadN<CPL>(pipe_row(aux_t, mm - m0, p.ldaux, n), pre.r[it]);
      } else {
        loadN<CPL>(pipe_row(resid_t, mm - m0, p.ldr, n), pre.r[it]);
      }
      if constexpr (EPI == EPI_LN_BWD) {  // the LayerNorm's input row (resid), the residual-stream gradient it is added to, the row's statistics
        // (C == NULL: nobody needs the input gradient -- the first block behind a frozen patch embedding -- only dgamma / dbeta:
        // the 8 + 4 bytes per element of reading, updating and re-casting the residual-stream gradient are skipped)
        if (C != nullptr) loadN<CPL>(pipe_row(Cc_t, mm - m0, p.ldc, n), pre.g[it]);
        pre.s[it] = *reinterpret_cast<const float2*>(stats_t + 2 * (mm - m0));
      }
    }
  }
}
template <typename TC, int EPI, int BM, int BN, int WGM, int WGN>
__device__ __forceinline__ void pipe_epilogue_finish(const GemmParams& p, float ea_alpha, f32x4 (&accrow)[BN / WGN / 16], float* est, const MaskEval& meE, int m0, int n0,
                                                     int mbase, int wn, int lane, TC* C, const float (&bias)[(sizeof(TC) == 2) ? 8 : 4], float (&pg)[4],
                                                     float (&pb)[4], const float (&lng)[4], const PipePre<TC, EPI, BN, WGN>& pre) {
  constexpr int WR = BM / WGM, WC = BN / WGN, TM = WR / 16, TN = WC / 16, WPITCH = WC + 4;
  constexpr int CPL = (sizeof(TC) == 2) ? 8 : 4;
  constexpr int LPR = WC / CPL, RPI = 64 / LPR;
  const int c = (lane % LPR) * CPL, n = n0 + wn * WC + c;
  // (uniform pointers to the tile's first row: pipe_row)
  TC* C_t = C + (long)m0 * p.ldc;
  bf16_t* auxo_t = reinterpret_cast<bf16_t*>(p.aux_out) + (long)m0 * p.ldc;
  TC* auxoT_t = reinterpret_cast<TC*>(p.aux_out) + (long)m0 * p.ldc;
  float* stats_t = p.ln_stats + 2 * (long)m0;
  constexpr int NIT = 16 / RPI;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const f32x4 v = accrow[j] * ea_alpha;
    *reinterpret_cast<float4*>(est + (lane & 15) * WPITCH + j * 16 + (lane >> 4) * 4) = make_float4(v[0], v[1], v[2], v[3]);
  }
#pragma unroll
  for (int rr = 0; rr < 16; rr += RPI) {
    const int row = rr + lane / LPR, m = mbase + row;
    if (m >= p.M) continue;
    float v[CPL];
    loadN<CPL>(est + row * WPITCH + c, v);
#pragma unroll
    for (int e = 0; e < CPL; ++e) v[e] += bias[e];
    TC* dst = pipe_row(C_t, m - m0, p.ldc, n);
    if (EPI == EPI_STORE) {
      storeN<CPL>(dst, v);
    } else if (EPI == EPI_RESID) {
      const float rowm = meE.row_mult(m);
#pragma unroll
      for (int e = 0; e < CPL; ++e) v[e] = pre.r[EPI == EPI_RESID ? rr / RPI : 0][e] + v[e] * rowm * meE.elem_mult(m, n + e);
      storeN<CPL>(dst, v);
    } else if (EPI == EPI_RESID_LN) {
      // y = resid + drop(v); then the LayerNorm that follows in the block, finished while the row is in registers: the LPR lanes of a
      // row (16 / 32 / 64: a DPP row, two, or the whole wave) fold their sums -- statistics, normalised row and residual row leave
      // together, no second pass over the residual stream
      const float rowm = meE.row_mult(m);
#pragma unroll
      for (int e = 0; e < CPL; ++e) v[e] = pre.r[EPI == EPI_RESID_LN ? rr / RPI : 0][e] + v[e] * rowm * meE.elem_mult(m, n + e);
      storeN<CPL>(dst, v);
      float s1 = 0.f;
#pragma unroll
      for (int e = 0; e < CPL; ++e) s1 += v[e];
      s1 = row16_sum(s1);
      if (LPR >= 32) s1 = xadd16(s1);
      if (LPR >= 64) s1 = xadd32(s1);
      const float mean = s1 * (1.0f / WC);
      float s2 = 0.f;
#pragma unroll
      for (int e = 0; e < CPL; ++e) s2 += (v[e] - mean) * (v[e] - mean);
      s2 = row16_sum(s2);
      if (LPR >= 32) s2 = xadd16(s2);
      if (LPR >= 64) s2 = xadd32(s2);
      const float rstd = rsqrtf(s2 * (1.0f / WC) + p.ln_eps);
      float gq[CPL], bt[CPL], yq[CPL];
      loadN<CPL>(p.ln_gamma + n, gq);
      loadN<CPL>(p.ln_beta + n, bt);
#pragma unroll
      for (int e = 0; e < CPL; ++e) yq[e] = (v[e] - mean) * rstd * gq[e] + bt[e];
      storeN<CPL>(pipe_row(auxo_t, m - m0, p.ldc, n), yq);
      if ((lane % LPR) == 0) *reinterpret_cast<float2*>(stats_t + 2 * (m - m0)) = make_float2(mean, rstd);
    } else if (EPI == EPI_LN_BWD) {
      // v = dy of the LayerNorm (this GEMM's product, fp32 -- never written): dx = rstd (g - mean(g) - xhat mean(g xhat)), g = dy gamma;
      // the residual-stream gradient row gets += dx and leaves a second time as dtype(row * mask) for the next branch's GEMMs
      constexpr int IT = (EPI == EPI_LN_BWD) ? 1 : 0;
      const int it = IT * (rr / RPI);
      const float mean = pre.s[it].x, rstd = pre.s[it].y;
      float xh[CPL], gd[CPL];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int e = 0; e < CPL; ++e) {
        xh[e] = (pre.r[it][e] - mean) * rstd;
        gd[e] = v[e] * lng[e];
        s1 += gd[e];
        s2 += gd[e] * xh[e];
        pg[e] += v[e] * xh[e];
        pb[e] += v[e];
      }
      s1 = row16_sum(s1); s2 = row16_sum(s2);
      if (LPR >= 32) { s1 = xadd16(s1); s2 = xadd16(s2); }
      if (LPR >= 64) { s1 = xadd32(s1); s2 = xadd32(s2); }
      const float m1 = s1 * (1.0f / WC), m2 = s2 * (1.0f / WC);
      if (C == nullptr) continue;
      float o[CPL];
#pragma unroll
      for (int e = 0; e < CPL; ++e) o[e] = pre.g[it][e] + rstd * (gd[e] - m1 - xh[e] * m2);
      storeN<CPL>(dst, o);
      if (p.aux_out) {
        const float rowm = meE.row_mult(m);
        float om[CPL];
#pragma unroll
        for (int e = 0; e < CPL; ++e) om[e] = o[e] * rowm * meE.elem_mult(m, n + e);
        storeN<CPL>(pipe_row(auxo_t, m - m0, p.ldc, n), om);
      }
    } else if (EPI == EPI_MUL_AUX) {
#pragma unroll
      for (int e = 0; e < CPL; ++e) v[e] *= pre.r[EPI == EPI_MUL_AUX ? rr / RPI : 0][e];
      storeN<CPL>(dst, v);
    } else if (EPI == EPI_GELU_FWD) {
      float gq[CPL];
#pragma unroll
      for (int e = 0; e < CPL; e += 4) {  // one mask hash per four elements (MaskEval::elem_mult_quad)
        gelu_f2 mult[2];
        meE.elem_mult_quad(m, n + e, mult[0], mult[1]);
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          const gelu_f2 x = {v[e + 2 * h2], v[e + 2 * h2 + 1]};
          gelu_f2 cdf, pdf;
          gelu_parts2(x, cdf, pdf);
          const gelu_f2 gg = (x * pdf + cdf) * mult[h2], hh = x * cdf * mult[h2];
          gq[e + 2 * h2] = gg.x; gq[e + 2 * h2 + 1] = gg.y;
          v[e + 2 * h2] = hh.x; v[e + 2 * h2 + 1] = hh.y;
        }
      }
      storeN<CPL>(dst, v);
      storeN<CPL>(pipe_row(auxoT_t, m - m0, p.ldc, n), gq);
    }
  }
}
// one group, loads and all (the one-tile-per-workgroup kernel)
template <typename TC, int EPI, int BM, int BN, int WGM, int WGN>
__device__ __forceinline__ void pipe_epilogue_group(const GemmParams& p, float ea_alpha, const float* ea_resid, f32x4 (&accrow)[BN / WGN / 16], float* est, const MaskEval& meE, int m0, int n0,
                                                    int mbase, int wn, int lane, TC* C, const float (&bias)[(sizeof(TC) == 2) ? 8 : 4], float (&pg)[4],
                                                    float (&pb)[4], const float (&lng)[4]) {
  PipePre<TC, EPI, BN, WGN> pre;
  pipe_epilogue_prefetch<TC, EPI, BM, BN, WGM, WGN>(p, ea_resid, m0, n0, mbase, wn, lane, C, pre);
  pipe_epilogue_finish<TC, EPI, BM, BN, WGM, WGN>(p, ea_alpha, accrow, est, meE, m0, n0, mbase, wn, lane, C, bias, pg, pb, lng, pre);
}
template <typename TC, int EPI, int BM, int BN, int WGM, int WGN, int I0 = 0, int I1 = BM / WGM / 16>
__device__ __forceinline__ void pipe_epilogue_rows(const GemmParams& p, f32x4 (&acc)[BM / WGM / 16][BN / WGN / 16], float* est, const MaskEval& meE, int m0,
                                                   int n0, int wm, int wn, int lane, TC* C, float (&pg)[4], float (&pb)[4], const float (&lng)[4]) {
  constexpr int WR = BM / WGM;
  // (p.alpha / p.bias / p.resid passed as they are: copied into three locals first, hipcc turns the adjacent loads into a 20-byte copy
  // of the argument struct into scratch)
  const PipeBias<(sizeof(TC) == 2) ? 8 : 4> bias = pipe_epilogue_bias<TC, EPI, BN, WGN>(p.bias, n0, wn, lane);
#pragma unroll
  for (int i = I0; i < I1; ++i)
    pipe_epilogue_group<TC, EPI, BM, BN, WGM, WGN>(p, p.alpha, p.resid, acc[i], est, meE, m0, n0, m0 + wm * WR + i * 16, wn, lane, C, bias.v, pg, pb, lng);
}

// EPI_LN_BWD, after the workgroup's last tile: dgamma / dbeta from the lanes' column partials
template <int BN, int WGM, int WGN, int CPL>
__device__ __forceinline__ void pipe_ln_bwd_flush(const GemmParams& p, float (&pg)[4], float (&pb)[4], float* red, int n0, int wn, int wave, int lane, int tid) {
  constexpr int NW = WGM * WGN, WC = BN / WGN, LPR = WC / CPL;
  const int c = (lane % LPR) * CPL;
  {
    // dgamma / dbeta: the lanes of a wave that share columns fold by shuffles, one [2][WC] row per wave goes to LDS behind the staging
    // regions, the waves are summed and every column leaves as ONE atomic per workgroup
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      for (int o = LPR; o < 64; o <<= 1) {
        pg[e] = xadd(pg[e], o);
        pb[e] = xadd(pb[e], o);
      }
    }
    if (lane < LPR) {
      *reinterpret_cast<float4*>(red + wave * 2 * WC + c) = make_float4(pg[0], pg[1], pg[2], pg[3]);
      *reinterpret_cast<float4*>(red + wave * 2 * WC + WC + c) = make_float4(pb[0], pb[1], pb[2], pb[3]);
    }
    __syncthreads();
    for (int i = tid; i < 2 * WC; i += 64 * NW) {
      float a = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) a += red[w * 2 * WC + i];
      atomicAdd(i < WC ? p.ln_dgamma + n0 + i : p.ln_dbeta + n0 + (i - WC), a);
    }
  }
}

template <typename TC, int EPI, bool TRB, int BM, int BN, int NST, int WGM = 2, int WGN = 2>
__global__ __launch_bounds__(64 * WGM * WGN) void focal_gemm_pipe_kernel(const GemmParams p) {
  constexpr int NW = WGM * WGN;
  constexpr int BK = 64;                       // bf16 elements = 128 B = one LDS row
  constexpr int WR = BM / WGM, WC = BN / WGN;  // per-wave output
  constexpr int TM = WR / 16, TN = WC / 16;
  constexpr int ROWS = BM + BN;                // LDS rows per stage (A rows then W rows)
  constexpr int STAGE_BYTES = ROWS * 128;
  constexpr int LPW = ROWS / (8 * NW);         // LDS-DMA pieces (8 rows each) per wave per stage
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "a piece index must be an A piece or a W piece for all waves");
  constexpr int WPITCH = WC + 4;
  constexpr int EPI_BYTES = NW * 16 * WPITCH * 4 + (EPI == EPI_LN_BWD ? NW * 2 * WC * 4 : 0) +
                            (((EPI == EPI_RESID_LN || EPI == EPI_LN_BWD) && WGN == 2) ? BM * 16 : 0);
  constexpr int LDS_BYTES = NST * STAGE_BYTES > EPI_BYTES ? NST * STAGE_BYTES : EPI_BYTES;
  extern __shared__ __attribute__((aligned(1024))) char pipe_lds[];
  static_assert(LDS_BYTES <= 160 * 1024, "ring does not fit in LDS");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int tiles_n = p.N / BN, tiles_m = (p.M + BM - 1) / BM;
  const int ntiles = tiles_m * tiles_n;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int tile = logical % ntiles, bz = logical / ntiles;
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;

  const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A) + (long)bz * p.strideA;
  const bf16_t* W = reinterpret_cast<const bf16_t*>(p.B) + (long)bz * p.strideB;
  TC* C = reinterpret_cast<TC*>(p.C) + (long)bz * p.strideC;
  const int KT = p.K / BK;

  // ---- fill plan: piece q = wave + 4 t covers stage rows 8 q .. 8 q + 7; lane -> (row 8 q + lane / 8, position lane % 8)
  // and fetches chunk position ^ swizzle(row) of that row.  Per-lane byte offsets from the (uniform) operand bases.
  uint32_t goff[LPW];
  const char* gbase[LPW];
  long gstep[LPW];  // bytes per k-step
#pragma unroll
  for (int t = 0; t < LPW; ++t) {
    const int q = wave + NW * t;
    if (8 * NW * t < BM) {  // piece index t is an A piece for all waves or a W piece for all waves
      const int row = 8 * q + (lane >> 3), pos = lane & 7;
      const int chunk = pos ^ ((row >> 1) & 7);
      const int srow = min(row, p.M - 1 - m0);  // ragged last tile: re-read the last valid row (its products are never stored)
      goff[t] = (uint32_t)(((long)srow * p.lda + chunk * 8) * 2);
      gbase[t] = reinterpret_cast<const char*>(A + (long)m0 * p.lda);
      gstep[t] = 128;
    } else if (!TRB) {
      const int row = 8 * q + (lane >> 3) - BM, pos = lane & 7;
      const int chunk = pos ^ ((row >> 1) & 7);
      goff[t] = (uint32_t)(((long)row * p.ldb + chunk * 8) * 2);
      gbase[t] = reinterpret_cast<const char*>(W + (long)n0 * p.ldb);
      gstep[t] = 128;
    } else {
      // W stored [k][n] (the weight of a data-gradient product): the stage image is [64 k][BN], rows of BN * 2 bytes; a piece is
      // 4 (BN = 128) or 8 (BN = 64) k-rows.  16-byte chunk c of k-row r sits at position c ^ ((r & 3) << 1): the four k-rows
      // that one ds_read_b64_tr_b16 lane group touches then fall in four different 32-byte bank groups.
      constexpr int CPRW = BN / 8, RPP = 64 / CPRW;  // 16-byte chunks per k-row, k-rows per piece
      const int krow = (q - BM / 8) * RPP + lane / CPRW, pos = lane % CPRW;
      const int chunk = pos ^ ((krow & 3) << 1);
      goff[t] = (uint32_t)(((long)krow * p.ldb + chunk * 8) * 2);
      gbase[t] = reinterpret_cast<const char*>(W + n0);
      gstep[t] = (long)64 * p.ldb * 2;
    }
  }
  auto fill = [&](int kt, int stage) {
#pragma unroll
    for (int t = 0; t < LPW; ++t) {
      const int q = wave + NW * t;
      char* dst = pipe_lds + stage * STAGE_BYTES + q * 1024;
      __builtin_amdgcn_global_load_lds((pipe_glb_ptr)(gbase[t] + (long)kt * gstep[t] + goff[t]), (pipe_lds_ptr)dst, 16, 0, 0);
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment addresses inside a stage: row (lane & 15) of a 16-row tile, chunk (kk * 4 + lane / 16) ^ swizzle
  const int swz = (lane >> 1) & 7, g = lane >> 4;
  const uint32_t lds0 = pipe_lds_addr(pipe_lds);
  const uint32_t fo0 = lds0 + (lane & 15) * 128 + ((g ^ swz) << 4), fo1 = lds0 + (lane & 15) * 128 + (((4 + g) ^ swz) << 4);
  const int a_off = wm * WR * 128, b_off = (BM + wn * WC) * 128;
  // transposed W: this lane reads k-row 8 g + q (and + 4), columns 4 p .. 4 p + 3 of a 16-column tile (q = (lane & 15) >> 2, p = lane & 3)
  const int tr_krow = 8 * g + ((lane & 15) >> 2), tr_swz = ((lane & 15) >> 2) << 1;
  const int tr_in = ((lane & 3) >> 1) * 16 + (lane & 1) * 8;  // byte offset of columns 4 p .. 4 p + 3 inside their pair of 16-byte chunks
  uint32_t tr_a[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) tr_a[j] = lds0 + BM * 128 + tr_krow * (BN * 2) + ((((wn * WC + j * 16) >> 3) ^ tr_swz) << 4) + tr_in;

  auto compute = [&](int stage) {
    const uint32_t so = stage * STAGE_BYTES;
    pipe_static_for<0, 2>([&](auto kc) {
      constexpr int kk = decltype(kc)::value;
      const uint32_t fa = (kk ? fo1 : fo0) + so + a_off, fb = (kk ? fo1 : fo0) + so + b_off;
      bf16x8 xa[TM], wb[TN];
      pipe_static_for<0, TM>([&](auto ic) { xa[decltype(ic)::value] = pipe_lds_read128<decltype(ic)::value * 2048>(fa); });
      pipe_static_for<0, TN>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if constexpr (!TRB) {
          wb[j] = pipe_lds_read128<j * 2048>(fb);
        } else {
          const bf16x4 lo = pipe_lds_read_tr<kk * 32 * BN * 2>(tr_a[j] + so);
          const bf16x4 hi = pipe_lds_read_tr<kk * 32 * BN * 2 + 4 * BN * 2>(tr_a[j] + so);
          wb[j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
      });
      pipe_lds_wait(xa, wb);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = mma16(wb[j], xa[i], acc[i][j]);
    });
  };

#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < KT) fill(s, s);
  int stage = 0, fstage = NST - 1;
  for (int kt = 0; kt < KT; ++kt) {
    // k-steps issued so far: min(KT, kt + NST - 1); step kt must have landed -> leave min(NST - 2, KT - 1 - kt) steps in flight
    const int ahead = KT - 1 - kt;
    if (NST >= 4 && ahead >= 2) pipe_wait_barrier<2 * LPW>();
    else if (NST >= 3 && ahead >= 1) pipe_wait_barrier<(NST >= 3 ? LPW : 0)>();
    else pipe_wait_barrier<0>();
    if (kt + NST - 1 < KT) fill(kt + NST - 1, fstage);  // overwrites the stage multiplied in iteration kt - 1: every wave is past it
    compute(stage);
    stage = (stage + 1 == NST) ? 0 : stage + 1;
    fstage = (fstage + 1 == NST) ? 0 : fstage + 1;
  }
  asm volatile("s_barrier" ::: "memory");  // the ring is re-used as epilogue staging

  // ---- epilogue: transpose 16 rows at a time through a wave-private LDS region, then walk it row-major (16 B per lane)
  float* est = reinterpret_cast<float*>(pipe_lds) + wave * 16 * WPITCH;
  MaskEval meE;
  if (EPI == EPI_RESID || EPI == EPI_GELU_FWD || EPI == EPI_RESID_LN || EPI == EPI_LN_BWD) meE.init(p.epi);
  static_assert((EPI != EPI_RESID_LN && EPI != EPI_LN_BWD) || ((WGN == 1 || WGN == 2) && sizeof(TC) == 4 && BN == WGN * WC),
                "the LayerNorm epilogues need wave tiles that cover a row (one wave, or two side by side) and an fp32 residual stream");
  if constexpr ((EPI == EPI_RESID_LN || EPI == EPI_LN_BWD) && WGN == 2) {
    pipe_ln_epilogue_two_waves<EPI, BM, BN, WGM, TM, TN>(p, acc, reinterpret_cast<float*>(pipe_lds), m0, n0, wm, wn, wave, lane, tid,
                                                          reinterpret_cast<float*>(C));
    return;
  }
  float pg[4] = {0.f, 0.f, 0.f, 0.f}, pb[4] = {0.f, 0.f, 0.f, 0.f};  // EPI_LN_BWD: this lane's column partials of dy * xhat, dy
  float lng[4] = {0.f, 0.f, 0.f, 0.f};
  if (EPI == EPI_LN_BWD) loadN<4>(p.ln_gamma + (n0 + wn * WC + (lane % (WC / 4)) * 4), lng);
  pipe_epilogue_rows<TC, EPI, BM, BN, WGM, WGN>(p, acc, est, meE, m0, n0, wm, wn, lane, C, pg, pb, lng);
  if constexpr (EPI == EPI_LN_BWD)
    pipe_ln_bwd_flush<BN, WGM, WGN, 4>(p, pg, pb, reinterpret_cast<float*>(pipe_lds) + NW * 16 * WPITCH, n0, wn, wave, lane, tid);
}

template <typename TC, int EPI, bool TRB, int BM, int BN, int NST, int WGM = 2, int WGN = 2>
static inline hipError_t focal_launch_gemm_pipe(const GemmParams& p, hipStream_t stream) {
  constexpr int LDS_BYTES = NST * (BM + BN) * 128;
  auto kern = focal_gemm_pipe_kernel<TC, EPI, TRB, BM, BN, NST, WGM, WGN>;
  static std::atomic<bool> attr_set{false};  // (the grant is idempotent: two first callers may both issue it; the flag itself is race-free)
  if (!attr_set.load(std::memory_order_acquire)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_set.store(true, std::memory_order_release);
  }
  dim3 grid(((p.M + BM - 1) / BM) * (p.N / BN) * p.batch);
  FOCAL_LAUNCH(kern, grid, dim3(64 * WGM * WGN), LDS_BYTES, stream, p);
  return hipGetLastError();
}
