// Persistent LDS-DMA GEMM for the deep Swin stages: the tiles of gemm_pipe.hpp behind a ring that never drains.
//
//   C[m][n] = epi( sum_k A[m][k] * W[n][k] + bias[n] )        A: [M][K] bf16, W: [N][K] (or [K][N]: TRB) bf16
//
// focal_gemm_pipe_kernel runs ONE output tile per workgroup: K / 64 = 2 - 16 k-steps behind a 2-stage ring, i.e. fill, fill,
// multiply, multiply, then a fat epilogue (bias, GELU + derivative + dropout, LayerNorm ...) with nothing in flight -- every tile pays
// the whole ramp, and only the co-resident workgroups hide it (r4: 0.32 of the HBM peak over the family's 108 launches).  Here a
// workgroup is resident for the whole launch (at most one per CU-slot) and walks a list of tiles; the operand ring runs over the
// CONCATENATION of their k-steps:
//   * NLOAD loader waves do nothing but issue LDS-DMA fills (global_load_lds_dwordx4), up to R - 1 k-steps ahead of the consumers and
//     across tile boundaries: while the consumer waves run a tile's epilogue, the next tile's operands are landing.  A loader wave's
//     vmcnt holds only its own fills, so its counted s_waitcnt is exact whatever the epilogue loads and stores (the consumers' vmcnt is
//     the compiler's business; no consumer ever issues an LDS-DMA, so hipcc has no reason to guard their LDS accesses with vmcnt(0));
//   * synchronisation, two forms:
//       barrier (ASYNC = false): ONE s_barrier per k-step joins the roles -- the loaders arrive once step g has landed, the consumers
//         once they have multiplied step g - 1, whose slot the loaders refill right behind the barrier.  Every consumer wave is then in the
//         same phase at the same time: all multiply, all run their epilogue (vector work and store issue, matrix pipe idle);
//       flags (ASYNC = true): no barrier.  A loader PUBLISHES a landed step in an LDS word per slot; a consumer wave waits for that word
//         only, counts itself out of the slot once its fragments are in registers, and the loader refills a slot when all NW consumers
//         have left it.  The consumer waves drift apart by up to R - 1 steps: one wave's epilogue overlaps its SIMD partner's products;
//   * WS (weight-stationary, K <= 256 at 128 columns): the workgroup owns ONE column panel of W -- BN x K bf16 <= 64 KB, loaded once, resident
//     in LDS for the launch -- and walks row tiles of that panel only; the ring carries A alone (half the fills per k-step);
//   * the epilogue staging has its own LDS region (the ring is never idle), otherwise the epilogues are gemm_pipe.hpp's code:
//     same k order, same arithmetic.
// Swizzles, fragment reads and the inline-asm ordering are those of gemm_pipe.hpp (see there).
// Measured and dropped (profiles/r5_ring_lab.txt): the epilogue of tile t spread over the k-steps of tile t + 1 inside each wave (software
// pipelining with a second accumulator set) -- 5-20 % slower than the plain loop.
#pragma once
#include "gemm_pipe.hpp"

// Lab-only in-kernel stamps (tools/gemm_ring_lab.hip builds with -DRING_STAMPS and passes a u64 buffer in p.colsumA): where the two
// roles spend their cycles.  Never defined in the product build.
#ifdef RING_STAMPS
#define RS_DECL(n) unsigned long long n = 0
#define RS_T0() const unsigned long long rs_t0_ = __builtin_amdgcn_s_memtime()
#define RS_ADD(n) n += __builtin_amdgcn_s_memtime() - rs_t0_
#define RS_OUT(i, n) if (lane == 0 && p.colsumA) reinterpret_cast<unsigned long long*>(p.colsumA)[((long)blockIdx.x * 16 + wave) * 8 + (i)] = n
#else
#define RS_DECL(n)
#define RS_T0()
#define RS_ADD(n)
#define RS_OUT(i, n)
#endif

template <int N> __device__ __forceinline__ void ring_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N < 63 ? N : 63) : "memory"); }
__device__ __forceinline__ void ring_barrier() { asm volatile("s_barrier" ::: "memory"); }
// LDS control words by inline asm: a compiler-visible LDS access in a wave that issues LDS-DMA is guarded with s_waitcnt vmcnt(0) by hipcc
// (gemm_pipe.hpp) -- in the loader that would drain the whole ring before every publication
__device__ __forceinline__ uint32_t ring_lds_load32(uint32_t addr) {
  uint32_t v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ void ring_lds_store32(uint32_t addr, uint32_t v) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void ring_lds_add32(uint32_t addr, uint32_t v) { asm volatile("ds_add_u32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }

// LDS bytes of a configuration at K = 64 kt (host and device agree through this)
template <typename TC, int EPI, int BM, int BN, int R, bool WS, int WGM, int WGN>
struct RingLayout {
  static constexpr int NW = WGM * WGN, WC = BN / WGN, WPITCH = WC + 4;
  static constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128;
  static constexpr int SLOT_BYTES = WS ? A_BYTES : A_BYTES + W_BYTES;
  static constexpr int STG_BYTES = NW * 16 * WPITCH * 4 + (EPI == EPI_LN_BWD ? NW * 2 * WC * 4 : 0);
  static constexpr int CTRL_BYTES = 256;  // flags form: full[2][R] (step + 1 a loader wave has landed in slot s) | done[R] (uses of slot s finished)
  static_assert(3 * R * 4 <= CTRL_BYTES, "control words");
  static inline int panel_bytes(int kt) { return WS ? kt * W_BYTES : 0; }
  static inline int total(int kt) { return panel_bytes(kt) + R * SLOT_BYTES + STG_BYTES + CTRL_BYTES; }
};

template <typename TC, int EPI, bool TRB, int BM, int BN, int R, bool WS, int WGM, int WGN, int NLOAD, bool ASYNC = false, bool HOIST = true>
__global__ __launch_bounds__(64 * (WGM * WGN + NLOAD)) void focal_gemm_ring_kernel(const GemmParams p) {
  using L = RingLayout<TC, EPI, BM, BN, R, WS, WGM, WGN>;
  constexpr int NW = WGM * WGN;
  constexpr int WR = BM / WGM, WC = BN / WGN, TM = WR / 16, TN = WC / 16, WPITCH = WC + 4;
  constexpr int A_BYTES = L::A_BYTES, W_BYTES = L::W_BYTES, SLOT_BYTES = L::SLOT_BYTES;
  static_assert(R >= 2 && NLOAD >= 1 && NLOAD <= 2, "at least one k-step in flight; one or two loader waves");
  static_assert((EPI != EPI_RESID_LN && EPI != EPI_LN_BWD) || (WGN == 1 && sizeof(TC) == 4), "LayerNorm epilogues: one wave per row here");
  extern __shared__ __attribute__((aligned(1024))) char ring_lds[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int KT = p.K / 64;
  const int tiles_n = p.N / BN, tiles_m = (p.M + BM - 1) / BM, ntiles = tiles_m * tiles_n;
  const int G = gridDim.x;
  const int logical = xcd_remap(blockIdx.x, G);
  const int ntl = (ntiles - logical + G - 1) / G;  // tiles of this workgroup: logical, logical + G, ... (the host launches G <= ntiles; WS: G % tiles_n == 0,
                                                    // so the column panel logical % tiles_n is the same for all of them)
  const int total = ntl * KT;                       // k-steps of this workgroup
  const uint32_t lds0 = pipe_lds_addr(ring_lds);
  const int panel_bytes = WS ? KT * W_BYTES : 0;
  const uint32_t ring0 = panel_bytes;               // byte offsets inside ring_lds
  const uint32_t stg0 = ring0 + R * SLOT_BYTES;
  const uint32_t ctrl0 = lds0 + stg0 + L::STG_BYTES;  // LDS address of full[2][R] | done[R]
  if (ASYNC) {
    if (tid < L::CTRL_BYTES / 4) reinterpret_cast<uint32_t*>(ring_lds + stg0 + L::STG_BYTES)[tid] = 0u;
    __syncthreads();
  }

  const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
  const bf16_t* W = reinterpret_cast<const bf16_t*>(p.B);
  TC* C = reinterpret_cast<TC*>(p.C);

  if (wave >= NW) {
    // ================================================================================================ loader waves
    const int lw = wave - NW;
    constexpr int PA = BM / 8, PW = BN / 8;                 // 1-KB pieces (8 rows x 128 B) of an A image / a W image
    static_assert(PA % NLOAD == 0 && PW % NLOAD == 0, "pieces split evenly over the loader waves");
    constexpr int LA = PA / NLOAD, LW = PW / NLOAD;         // per loader wave
    constexpr int LSTEP = WS ? LA : LA + LW;                // LDS-DMA instructions per loader wave and k-step
    // per-lane byte offsets of this wave's pieces from the (uniform) image origins; piece q covers image rows 8 q .. 8 q + 7,
    // lane -> (row 8 q + lane / 8, position lane % 8) fetches chunk position ^ swizzle(row)   (gemm_pipe.hpp)
    uint32_t offW[LW];
#pragma unroll
    for (int t = 0; t < LW; ++t) {
      const int q = lw + NLOAD * t;
      if (!TRB) {
        const int row = 8 * q + (lane >> 3), pos = lane & 7, chunk = pos ^ ((row >> 1) & 7);
        offW[t] = (uint32_t)(((long)row * p.ldb + chunk * 8) * 2);
      } else {
        constexpr int CPRW = BN / 8, RPP = 64 / CPRW;
        const int krow = q * RPP + lane / CPRW, pos = lane % CPRW, chunk = pos ^ ((krow & 3) << 1);
        offW[t] = (uint32_t)(((long)krow * p.ldb + chunk * 8) * 2);
      }
    }
    const long wstep = TRB ? (long)64 * p.ldb * 2 : 128;
    auto w_origin = [&](int n0) __attribute__((always_inline)) { return reinterpret_cast<const char*>(TRB ? W + n0 : W + (long)n0 * p.ldb); };
    auto fill_w = [&](const char* wb, int kt, uint32_t dst0) __attribute__((always_inline)) {
#pragma unroll
      for (int t = 0; t < LW; ++t) {
        const int q = lw + NLOAD * t;
        __builtin_amdgcn_global_load_lds((pipe_glb_ptr)(wb + (long)kt * wstep + offW[t]), (pipe_lds_ptr)(ring_lds + dst0 + q * 1024), 16, 0, 0);
      }
    };
    // the issue cursor: (tile, k-step) of the next step to fill
    int f_kt = 0, f_tile = logical, issued = 0;
    const char* f_ab = nullptr;
    const char* f_wb = nullptr;
    uint32_t offA[LA];
    auto open_tile = [&]() __attribute__((always_inline)) {
      const int m0 = (f_tile / tiles_n) * BM, n0 = (f_tile % tiles_n) * BN;
      const int rmax = p.M - 1 - m0;  // ragged last row tile: rows beyond M re-read the last valid row (their products are never stored)
#pragma unroll
      for (int t = 0; t < LA; ++t) {
        const int q = lw + NLOAD * t;
        const int row = 8 * q + (lane >> 3), pos = lane & 7, chunk = pos ^ ((row >> 1) & 7);
        offA[t] = (uint32_t)(((long)min(row, rmax) * p.lda + chunk * 8) * 2);
      }
      f_ab = reinterpret_cast<const char*>(A + (long)m0 * p.lda);
      if (!WS) f_wb = w_origin(n0);
    };
    auto issue_next = [&]() __attribute__((always_inline)) {
      if (f_kt == 0) open_tile();
      const uint32_t slot = ring0 + (uint32_t)(issued % R) * SLOT_BYTES;
#pragma unroll
      for (int t = 0; t < LA; ++t) {
        const int q = lw + NLOAD * t;
        __builtin_amdgcn_global_load_lds((pipe_glb_ptr)(f_ab + (long)f_kt * 128 + offA[t]), (pipe_lds_ptr)(ring_lds + slot + q * 1024), 16, 0, 0);
      }
      if (!WS) fill_w(f_wb, f_kt, slot + A_BYTES);
      ++issued;
      if (++f_kt == KT) { f_kt = 0; f_tile += G; }
    };
    if (WS) {  // the panel: all KT images of this workgroup's columns, once
      const char* wb = w_origin((logical % tiles_n) * BN);
      for (int kt = 0; kt < KT; ++kt) fill_w(wb, kt, (uint32_t)kt * W_BYTES);
    }
    RS_DECL(rs_wait); RS_DECL(rs_bar); RS_DECL(rs_issue); RS_DECL(rs_all);
#ifdef RING_STAMPS
    const unsigned long long rs_begin = __builtin_amdgcn_s_memtime();
#endif
    if (!ASYNC) {
      for (int s = 0; s < R - 1 && s < total; ++s) issue_next();
      for (int g = 0; g < total; ++g) {
        // step g has landed once all but the fills issued behind it are done (the panel's fills are older than every step)
        const int younger = issued - (g + 1);
        {
          RS_T0();
          pipe_static_for<0, R - 1>([&](auto yc) {
            constexpr int y = decltype(yc)::value;
            if (younger == y) ring_vmcnt<y * LSTEP>();
          });
          RS_ADD(rs_wait);
        }
        {
          RS_T0();
          ring_barrier();              // B_g: the consumers are past step g - 1 -> its slot is free
          RS_ADD(rs_bar);
        }
        {
          RS_T0();
          if (issued < total) issue_next();
          RS_ADD(rs_issue);
        }
      }
    } else {
      // Issue while a slot is free (its previous content consumed by all NW waves), publish the oldest outstanding step when nothing can
      // be issued.  (Publishing is a counted vmcnt wait on this wave's own fills, then one LDS word.)
      int published = 0;
      const uint32_t full_w = ctrl0 + (uint32_t)lw * R * 4, done_w = ctrl0 + 2 * R * 4;
      while (published < total) {
        bool did = false;
        if (issued < total && issued - published < R) {
          const uint32_t uses = (uint32_t)(issued / R) * NW;
          RS_T0();
          if (uses == 0 || ring_lds_load32(done_w + (uint32_t)(issued % R) * 4) >= uses) {
            issue_next();
            did = true;
          }
          RS_ADD(rs_issue);
        }
        if (!did) {
          if (published < issued) {
            const int younger = issued - published - 1;
            RS_T0();
            pipe_static_for<0, R>([&](auto yc) {
              constexpr int y = decltype(yc)::value;
              if (younger == y) ring_vmcnt<y * LSTEP>();
            });
            RS_ADD(rs_wait);
            ring_lds_store32(full_w + (uint32_t)(published % R) * 4, (uint32_t)published + 1u);
            ++published;
          } else {
            RS_T0();
            __builtin_amdgcn_s_sleep(2);
            RS_ADD(rs_bar);
          }
        }
      }
    }
    if (EPI == EPI_LN_BWD) ring_barrier();  // the consumers' __syncthreads() in pipe_ln_bwd_flush
#ifdef RING_STAMPS
    rs_all = __builtin_amdgcn_s_memtime() - rs_begin;
    RS_OUT(0, rs_wait); RS_OUT(1, rs_bar); RS_OUT(2, rs_issue); RS_OUT(3, rs_all); RS_OUT(4, 1ull);  // [4] = 1: a loader wave's record
#endif
    return;
  }

  // ==================================================================================================== consumer waves
  const int wm = wave / WGN, wn = wave % WGN;
  // fragment addresses inside an image: row (lane & 15) of a 16-row tile, chunk (kk * 4 + lane / 16) ^ swizzle
  const int swz = (lane >> 1) & 7, g4 = lane >> 4;
  const uint32_t fo0 = lds0 + (lane & 15) * 128 + ((g4 ^ swz) << 4), fo1 = lds0 + (lane & 15) * 128 + (((4 + g4) ^ swz) << 4);
  const int a_off = wm * WR * 128, b_off = wn * WC * 128;
  // transposed W: this lane reads k-row 8 g + q (and + 4), columns 4 p .. 4 p + 3 of a 16-column tile (q = (lane & 15) >> 2, p = lane & 3)
  const int tr_krow = 8 * g4 + ((lane & 15) >> 2), tr_swz = ((lane & 15) >> 2) << 1;
  const int tr_in = ((lane & 3) >> 1) * 16 + (lane & 1) * 8;
  uint32_t tr_a[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) tr_a[j] = lds0 + tr_krow * (BN * 2) + ((((wn * WC + j * 16) >> 3) ^ tr_swz) << 4) + tr_in;

  f32x4 acc[TM][TN];
  // reads_done(): called once the step's last fragments are in registers (before its last MFMAs)
  auto compute = [&](uint32_t sa, uint32_t sw, auto&& reads_done) __attribute__((always_inline)) {  // byte offsets of the step's A image and W image
    pipe_static_for<0, 2>([&](auto kc) {
      constexpr int kk = decltype(kc)::value;
      const uint32_t fa = (kk ? fo1 : fo0) + sa + a_off, fb = (kk ? fo1 : fo0) + sw + b_off;
      bf16x8 xa[TM], wb[TN];
      pipe_static_for<0, TM>([&](auto ic) { xa[decltype(ic)::value] = pipe_lds_read128<decltype(ic)::value * 2048>(fa); });
      pipe_static_for<0, TN>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if constexpr (!TRB) {
          wb[j] = pipe_lds_read128<j * 2048>(fb);
        } else {
          const bf16x4 lo = pipe_lds_read_tr<kk * 32 * BN * 2>(tr_a[j] + sw);
          const bf16x4 hi = pipe_lds_read_tr<kk * 32 * BN * 2 + 4 * BN * 2>(tr_a[j] + sw);
          wb[j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
      });
      pipe_lds_wait(xa, wb);
      if constexpr (kk == 1) reads_done();
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = mma16(wb[j], xa[i], acc[i][j]);
    });
  };

  float* est = reinterpret_cast<float*>(ring_lds + stg0) + wave * 16 * WPITCH;
  MaskEval meE;
  if (EPI == EPI_RESID || EPI == EPI_GELU_FWD || EPI == EPI_RESID_LN || EPI == EPI_LN_BWD) meE.init(p.epi);
  float pg[4] = {0.f, 0.f, 0.f, 0.f}, pb[4] = {0.f, 0.f, 0.f, 0.f};
  float lng[4] = {0.f, 0.f, 0.f, 0.f};
  if (EPI == EPI_LN_BWD) loadN<4>(p.ln_gamma + (wn * WC + (lane % (WC / 4)) * 4), lng);  // (BN == N: n0 = 0)

  int slot = 0, g = 0;
  RS_DECL(rs_cbar); RS_DECL(rs_cmma); RS_DECL(rs_cepi); RS_DECL(rs_call);
#ifdef RING_STAMPS
  const unsigned long long rs_cbegin = __builtin_amdgcn_s_memtime();
#endif
  const uint32_t done_w = ctrl0 + 2 * R * 4;
#pragma unroll 1
  for (int i = 0, tile = logical; i < ntl; ++i, tile += G) {
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    // everything the tile's epilogue reads from memory (residual / aux rows, LayerNorm rows and statistics, bias) is requested HERE and
    // lands while the k loop multiplies; the loop's asm barriers keep hipcc from sinking the loads to their uses
    // (HOIST = false: the LayerNorm backward at 128 columns -- 80 more registers per lane across the k loop made it slower, 0.97 against 0.88)
    PipePre<TC, EPI, BN, WGN> pre[TM];
    if constexpr (HOIST) {
#pragma unroll
      for (int a = 0; a < TM; ++a) pipe_epilogue_prefetch<TC, EPI, BM, BN, WGM, WGN>(p, p.resid, m0, n0, m0 + wm * WR + a * 16, wn, lane, C, pre[a]);
    }
    const PipeBias<(sizeof(TC) == 2) ? 8 : 4> bias = pipe_epilogue_bias<TC, EPI, BN, WGN>(p.bias, n0, wn, lane);
    {
      RS_T0();
#pragma unroll 1
      for (int kt = 0; kt < KT; ++kt, ++g) {
        if (!ASYNC) {
          ring_barrier();  // B_g: step g is in LDS (the loaders waited for it)
        } else {           // has every loader wave published step g?
          const uint32_t fw = ctrl0 + (uint32_t)slot * 4;
          while (ring_lds_load32(fw) <= (uint32_t)g || (NLOAD == 2 && ring_lds_load32(fw + R * 4) <= (uint32_t)g)) __builtin_amdgcn_s_sleep(1);
        }
        const uint32_t sa = ring0 + (uint32_t)slot * SLOT_BYTES;
        compute(sa, WS ? (uint32_t)kt * W_BYTES : sa + A_BYTES, [&]() __attribute__((always_inline)) {
          if (ASYNC && lane == 0) ring_lds_add32(done_w + (uint32_t)slot * 4, 1u);  // this wave is out of the slot
        });
        slot = (slot + 1 == R) ? 0 : slot + 1;
      }
      RS_ADD(rs_cmma);
    }
    {
      RS_T0();
#pragma unroll
      for (int a = 0; a < TM; ++a) {
        if constexpr (!HOIST) pipe_epilogue_prefetch<TC, EPI, BM, BN, WGM, WGN>(p, p.resid, m0, n0, m0 + wm * WR + a * 16, wn, lane, C, pre[a]);
        pipe_epilogue_finish<TC, EPI, BM, BN, WGM, WGN>(p, p.alpha, acc[a], est, meE, m0, n0, m0 + wm * WR + a * 16, wn, lane, C, bias.v, pg, pb, lng, pre[a]);
      }
      RS_ADD(rs_cepi);
    }
  }
#ifdef RING_STAMPS
  rs_call = __builtin_amdgcn_s_memtime() - rs_cbegin;
  RS_OUT(0, rs_cbar); RS_OUT(1, rs_cmma); RS_OUT(2, rs_cepi); RS_OUT(3, rs_call); RS_OUT(4, 2ull);  // [4] = 2: a consumer wave's record
#endif
  if constexpr (EPI == EPI_LN_BWD)
    pipe_ln_bwd_flush<BN, WGM, WGN, 4>(p, pg, pb, reinterpret_cast<float*>(ring_lds + stg0) + NW * 16 * WPITCH, 0, wn, wave, lane, tid);
}

// Workgroups the launch uses: at most `slots` (CUs x resident workgroups per CU), every workgroup the same number of tiles where the
// tile count allows; WS: a multiple of the panel count (a workgroup stays on one panel).
static inline int focal_ring_grid(int tiles_m, int tiles_n, bool ws, int slots) {
  if (ws) {
    int streams = slots / tiles_n;
    if (streams < 1) streams = 1;
    if (streams > tiles_m) streams = tiles_m;
    const int rounds = (tiles_m + streams - 1) / streams;
    streams = (tiles_m + rounds - 1) / rounds;
    return streams * tiles_n;
  }
  const long ntiles = (long)tiles_m * tiles_n;
  if (ntiles <= slots) return (int)ntiles;
  const long rounds = (ntiles + slots - 1) / slots;
  return (int)((ntiles + rounds - 1) / rounds);
}

// FOCAL_NO_RING=1: every product stays on focal_gemm_pipe_kernel (same-box A/B of the step: tools/ab_env.sh; tests/test_kernels_gpu.py runs both)
static inline bool focal_ring_disabled() {
  static const bool off = getenv("FOCAL_NO_RING") != nullptr;
  return off;
}

static inline int focal_cu_count() {
  // (initialised once, thread-safe: C++11 magic static; `static const bool off` above is one too)
  static const int n = [] {
    if (const char* lab = getenv("FOCAL_LAB_CUS")) { if (atoi(lab) > 0) return atoi(lab); }  // lab: size the persistent grids for fewer CUs
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) return v;
    return 256;
  }();
  return n;
}

// false: the configuration does not take this problem (LDS), the caller uses another kernel
template <typename TC, int EPI, bool TRB, int BM, int BN, int R, bool WS, int WGM, int WGN, int NLOAD, bool ASYNC = false, bool HOIST = true>
static inline bool focal_ring_fits(const GemmParams& p) {
  using L = RingLayout<TC, EPI, BM, BN, R, WS, WGM, WGN>;
  return p.batch == 1 && p.splits == 1 && p.K % 64 == 0 && p.N % BN == 0 && L::total(p.K / 64) <= 160 * 1024;
}

template <typename TC, int EPI, bool TRB, int BM, int BN, int R, bool WS, int WGM, int WGN, int NLOAD, bool ASYNC = false, bool HOIST = true>
static inline hipError_t focal_launch_gemm_ring(const GemmParams& p, hipStream_t stream) {
  using L = RingLayout<TC, EPI, BM, BN, R, WS, WGM, WGN>;
  const int lds_bytes = L::total(p.K / 64);
  auto kern = focal_gemm_ring_kernel<TC, EPI, TRB, BM, BN, R, WS, WGM, WGN, NLOAD, ASYNC, HOIST>;
  static std::atomic<bool> attr_set{false};  // (the grant is idempotent: two first callers may both issue it; the flag itself is race-free)
  if (!attr_set.load(std::memory_order_acquire)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set.store(true, std::memory_order_release);
  }
  int per_cu = (160 * 1024) / lds_bytes;
  const int by_waves = 32 / (WGM * WGN + NLOAD);
  if (per_cu > by_waves) per_cu = by_waves;
  if (per_cu < 1) per_cu = 1;
  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = p.N / BN;
  const int grid = focal_ring_grid(tiles_m, tiles_n, WS, focal_cu_count() * per_cu);
  FOCAL_LAUNCH(kern, dim3(grid), dim3(64 * (WGM * WGN + NLOAD)), lds_bytes, stream, p);
  return hipGetLastError();
}
