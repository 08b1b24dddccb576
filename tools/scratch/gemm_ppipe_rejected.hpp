// Persistent variant of focal_gemm_pipe_kernel measured in tools/scratch/gemm_lab.hip and REJECTED (1.2-1.4x slower than the
// non-persistent 2-stage form at every deep-stage shape: with 1-2 workgroups per CU the per-step LDS->MFMA latency is exposed).
// Kept as a record; not part of the library.
#pragma once
#include "gemm_pipe.hpp"

// ------------------------------------------------------------------------------------------------------------------
// Persistent form: a workgroup walks tiles  v = it * gridDim.x + blockIdx.x  (XCD-aware order) and treats (tile, k-step)
// as ONE sequence of steps, so the ring is filled NST - 1 steps ahead ACROSS tile boundaries: the next tile's operands
// are already landing while a wave writes the finished tile out, and no tile but a workgroup's first pays a pipeline
// fill.  Counted waits stay exact because vmcnt counts loads, stores and LDS-DMA together in issue order: a tile's
// epilogue stores are younger than the pieces issued before them, so the first NST - 1 steps of the next tile allow
// E more operations in flight.  The bias vector is staged in LDS once (an ordinary global load inside the loop would
// make hipcc drain the whole ring with vmcnt(0)).
// The epilogue's staging accesses are inline asm on purpose: hipcc orders every LDS access it can see against in-flight
// LDS-DMA with s_waitcnt vmcnt(0) (it cannot tell the staging region from the ring), which would drain the ring at every
// tile boundary.  LDS operations of one wave execute in order, so write -> read needs no wait; the read waits for itself.
__device__ __forceinline__ uint32_t pipe_lds_addr(const void* p) { return (uint32_t)(uintptr_t)(pipe_lds_ptr)(void*)p; }
__device__ __forceinline__ void pipe_lds_store4(uint32_t addr, f32x4 v) { asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
template <int N> __device__ __forceinline__ void pipe_lds_loadN(uint32_t addr, float* f) {
  if (N == 8) {
    f32x4 a, b;
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(addr) : "memory");
    f[0] = a[0]; f[1] = a[1]; f[2] = a[2]; f[3] = a[3]; f[4] = b[0]; f[5] = b[1]; f[6] = b[2]; f[7] = b[3];
  } else {
    f32x4 a;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(a) : "v"(addr) : "memory");
    f[0] = a[0]; f[1] = a[1]; f[2] = a[2]; f[3] = a[3];
  }
}

template <typename TC, int EPI, int BM, int BN, int NST, int WGM = 2, int WGN = 2>
__global__ __launch_bounds__(64 * WGM * WGN) void focal_gemm_ppipe_kernel(const GemmParams p) {
  constexpr int NW = WGM * WGN;
  constexpr int BK = 64;
  constexpr int WR = BM / WGM, WC = BN / WGN;
  constexpr int TM = WR / 16, TN = WC / 16;
  constexpr int ROWS = BM + BN;
  constexpr int STAGE_BYTES = ROWS * 128;
  constexpr int LPW = ROWS / (8 * NW);
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "a piece index must be an A piece or a W piece for all waves");
  constexpr int WPITCH = WC + 4;
  constexpr int RING_BYTES = NST * STAGE_BYTES, EPI_BYTES = NW * 16 * WPITCH * 4;
  static_assert(RING_BYTES + EPI_BYTES <= 160 * 1024, "ring does not fit in LDS");
  extern __shared__ __attribute__((aligned(1024))) char pipe_lds[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int tiles_n = p.N / BN, tiles_m = p.M / BM;
  // A workgroup keeps ONE column panel (n0 fixed: its bias slice lives in registers as the accumulators' initial value) and
  // walks row tiles mslot, mslot + MS, ...  Logical ids are contiguous per XCD (blocks b, b + 8, ... share an XCD), and the
  // tiles_n workgroups that read the same rows at the same time are neighbours in that order: they share an L2.
  const int G = gridDim.x;  // multiple of 8 * tiles_n (launcher)
  const int L = ((int)blockIdx.x & 7) * (G >> 3) + ((int)blockIdx.x >> 3);
  const int n0 = (L % tiles_n) * BN, mslot = L / tiles_n, MS = G / tiles_n;
  const int my_tiles = mslot < tiles_m ? (tiles_m - mslot + MS - 1) / MS : 0;
  const int KT = p.K / BK;
  const int S = my_tiles * KT;
  f32x4 bias4[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) bias4[j] = p.bias ? load4(p.bias + n0 + wn * WC + j * 16 + 4 * (lane >> 4)) : f32x4{0.f, 0.f, 0.f, 0.f};

  uint32_t goff[LPW];
#pragma unroll
  for (int t = 0; t < LPW; ++t) {
    const int q = wave + NW * t, row = 8 * q + (lane >> 3), pos = lane & 7;
    const int chunk = pos ^ ((row >> 1) & 7);
    if (8 * NW * t < BM) goff[t] = (uint32_t)(((long)row * p.lda + chunk * 8) * 2);
    else goff[t] = (uint32_t)(((long)(row - BM) * p.ldb + chunk * 8) * 2);
  }
  // fill-side cursor
  int f_it = 0, f_kt = 0, f_stage = 0;
  const char *fA = nullptr, *fW = nullptr;
  fW = reinterpret_cast<const char*>(reinterpret_cast<const bf16_t*>(p.B) + (long)n0 * p.ldb);
  auto fill_next = [&]() {
    if (f_kt == 0) fA = reinterpret_cast<const char*>(reinterpret_cast<const bf16_t*>(p.A) + (long)(mslot + f_it * MS) * BM * p.lda);
#pragma unroll
    for (int t = 0; t < LPW; ++t) {
      const int q = wave + NW * t;
      char* dst = pipe_lds + f_stage * STAGE_BYTES + q * 1024;
      const char* src = (8 * NW * t < BM ? fA : fW) + (long)f_kt * 128 + goff[t];
      __builtin_amdgcn_global_load_lds((pipe_glb_ptr)src, (pipe_lds_ptr)dst, 16, 0, 0);
    }
    f_stage = (f_stage + 1 == NST) ? 0 : f_stage + 1;
    if (++f_kt == KT) { f_kt = 0; ++f_it; }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = bias4[j];

  const int swz = (lane >> 1) & 7, g = lane >> 4;
  const int fo0 = (lane & 15) * 128 + ((g ^ swz) << 4), fo1 = (lane & 15) * 128 + (((4 + g) ^ swz) << 4);
  const int a_off = wm * WR * 128, b_off = (BM + wn * WC) * 128;
  auto compute = [&](int stage) {
    const char* s = pipe_lds + stage * STAGE_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int fo = kk ? fo1 : fo0;
      bf16x8 xa[TM], wb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) xa[i] = *reinterpret_cast<const bf16x8*>(s + a_off + i * 2048 + fo);
#pragma unroll
      for (int j = 0; j < TN; ++j) wb[j] = *reinterpret_cast<const bf16x8*>(s + b_off + j * 2048 + fo);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = mma16(wb[j], xa[i], acc[i][j]);
    }
  };

  constexpr int CPL = (sizeof(TC) == 2) ? 8 : 4;
  constexpr int LPR = WC / CPL, RPI = 64 / LPR;
  constexpr int NSTORE = TM * (16 / RPI) * (EPI == EPI_GELU_FWD ? 2 : 1);  // epilogue stores per wave per tile
  const uint32_t est = pipe_lds_addr(pipe_lds + RING_BYTES) + wave * 16 * WPITCH * 4;
  MaskEval meE;
  if (EPI == EPI_RESID || EPI == EPI_GELU_FWD) meE.init(p.epi);

  int issued = 0;
#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < S) { fill_next(); ++issued; }
  int stage = 0, kt = 0, it = 0;
  for (int s = 0; s < S; ++s) {
    // younger than step s's pieces: the pieces of steps s+1 .. min(S-1, s+NST-2), plus the previous tile's stores while
    // this tile is in its first NST - 1 steps
    const int ahead = min(NST - 2, S - 1 - s);
    const bool st = (it > 0) && (kt < NST - 1);
    if (st) {
      if (ahead >= 2) pipe_wait_barrier<(NST >= 4 ? 2 * LPW : 0) + NSTORE>();
      else if (ahead == 1) pipe_wait_barrier<(NST >= 3 ? LPW : 0) + NSTORE>();
      else pipe_wait_barrier<NSTORE>();
    } else {
      if (ahead >= 2) pipe_wait_barrier<(NST >= 4 ? 2 * LPW : 0)>();
      else if (ahead == 1) pipe_wait_barrier<(NST >= 3 ? LPW : 0)>();
      else pipe_wait_barrier<0>();
    }
    if (issued < S) { fill_next(); ++issued; }
    compute(stage);
    stage = (stage + 1 == NST) ? 0 : stage + 1;
    if (++kt == KT) {
      kt = 0;
      const int m0 = (mslot + it * MS) * BM;
      ++it;
      TC* C = reinterpret_cast<TC*>(p.C);
      const int c = (lane % LPR) * CPL, n = n0 + wn * WC + c;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const f32x4 v = acc[i][j];
          pipe_lds_store4(est + ((lane & 15) * WPITCH + j * 16 + (lane >> 4) * 4) * 4, v);
          acc[i][j] = bias4[j];
        }
        const int mbase = m0 + wm * WR + i * 16;
#pragma unroll
        for (int rr = 0; rr < 16; rr += RPI) {
          const int row = rr + lane / LPR, m = mbase + row;
          float v[CPL];
          pipe_lds_loadN<CPL>(est + (row * WPITCH + c) * 4, v);
          TC* dst = C + (long)m * p.ldc + n;
          if (EPI == EPI_STORE) {
            storeN<CPL>(dst, v);
          } else if (EPI == EPI_RESID) {
            float r[CPL];
            loadN<CPL>(p.resid + (long)m * p.ldr + n, r);
            const float rowm = meE.row_mult(m);
#pragma unroll
            for (int e = 0; e < CPL; ++e) v[e] = r[e] + v[e] * rowm * meE.elem_mult(m, n + e);
            storeN<CPL>(dst, v);
          } else if (EPI == EPI_MUL_AUX) {
            float a[CPL];
            loadN<CPL>(reinterpret_cast<const bf16_t*>(p.aux) + (long)m * p.ldaux + n, a);
#pragma unroll
            for (int e = 0; e < CPL; ++e) v[e] *= a[e];
            storeN<CPL>(dst, v);
          } else if (EPI == EPI_GELU_FWD) {
            float gq[CPL];
#pragma unroll
            for (int e = 0; e < CPL; e += 2) {
              const gelu_f2 x = {v[e], v[e + 1]};
              gelu_f2 cdf, pdf;
              gelu_parts2(x, cdf, pdf);
              const gelu_f2 mult = {meE.elem_mult(m, n + e), meE.elem_mult(m, n + e + 1)};
              const gelu_f2 gg = (x * pdf + cdf) * mult, hh = x * cdf * mult;
              gq[e] = gg.x; gq[e + 1] = gg.y;
              v[e] = hh.x; v[e + 1] = hh.y;
            }
            storeN<CPL>(dst, v);
            storeN<CPL>(reinterpret_cast<TC*>(p.aux_out) + (long)m * p.ldc + n, gq);
          }
        }
      }
    }
  }
}

template <typename TC, int EPI, int BM, int BN, int NST, int WGM = 2, int WGN = 2>
static inline hipError_t focal_launch_gemm_ppipe(const GemmParams& p, hipStream_t stream, int wg_per_cu = 1) {
  constexpr int NW = WGM * WGN;
  constexpr int LDS_BYTES = NST * (BM + BN) * 128 + NW * 16 * (BN / WGN + 4) * 4;
  auto kern = focal_gemm_ppipe_kernel<TC, EPI, BM, BN, NST, WGM, WGN>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const int tiles_n = p.N / BN, tiles_m = p.M / BM, unit = 8 * tiles_n;
  int grid = (256 * wg_per_cu) / unit * unit;  // whole groups of (8 XCDs x tiles_n column panels)
  if (grid < unit) grid = unit;
  const int need = (tiles_m * tiles_n + unit - 1) / unit * unit;
  if (grid > need) grid = need;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), LDS_BYTES, stream, p);
  return hipGetLastError();
}
