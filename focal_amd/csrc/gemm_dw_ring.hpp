// Weight-gradient GEMM on an LDS-DMA ring (bf16 operands, 64 x 64 output tiles):
//
//   C[m][n] += sum_r A[r][m] * B[r][n]        A = dy [rows][M] bf16, B = x [rows][N] bf16 (both token-major: the contraction index
//                                             is the memory row), C fp32 [M][N] accumulated with atomics, rows split over workgroups
//
// Same tiles and launch plan as the register-staged kernel of gemm.hpp (launch_dw: 64 x 64 outputs, ~512 workgroups -- more
// workgroups for the same atomic volume than any wider tile), different way for the bytes to arrive:
//   * both operands go L2 -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass) into a 4-deep ring of
//     16 KB stages (64 rows of A | 64 rows of B), 3 stages in flight, ONE raw s_barrier + counted vmcnt per stage;
//   * rows are stored as they are ([r][64] = 128 B) and read as MFMA fragments with ds_read_b64_tr_b16; the 16-byte chunk c of row r
//     sits at c ^ sw(r), sw(r) = bit1(r) << 1 | bit3(r) << 2: the eight rows {r0 .. r0 + 3, r0 + 8 .. r0 + 11} that the 32 lanes of
//     one LDS pass touch (32 bytes each) fall in eight different 32-byte groups of the 256-byte bank period;
//   * the fragment reads are inline asm (pipe_lds_read_tr): a visible LDS read makes hipcc wait vmcnt(0), i.e. for every stage in
//     flight (gemm_pipe.hpp) -- the round-2 ring that lost to the register-staged kernel (profiles/r2_dw_variants.txt (2)) had
//     exactly that wait in front of each stage;
//   * 2 x 2 waves of 32 x 32 outputs: 2 + 2 transposed fragments for 4 MFMAs per 32 rows;
//   * the bias gradient (column sums of A) is one extra MFMA per A fragment against a ones operand, issued branch-free by every wave
//     (a zero operand where the sums are not wanted: a branch would cut the k-step's scheduling region);
//   * accumulators leave through a wave-private LDS transpose as contiguous fp32 atomic rows (two 128-byte rows per instruction).
// 63 registers, 64 KB of LDS: two workgroups per CU at the plan's 512 workgroups.  Measured (tools/mb_dw.py, cold operands, sum over
// the step's 12 shapes): 358 -> 322 us; ring depths 2 / 3 / 4: 348 / 337 / 331 us; same-box step A/B +1.5 % (3 stages), +2.7 %
// (4 stages) (profiles/r2_k_dw_ring.txt).  A 128 x 128 variant (half the L2 -> LDS bytes, 64 KB of atomics per workgroup) won only
// at 1024 x 256 outputs in isolation and nothing in the step; it is not built.
// Requirements (the dispatcher checks; everything else stays on launch_dw): bf16 operands without a loader prologue, M % 64 == 0,
// N % 64 == 0, rows % 64 == 0, 16-byte aligned rows, batch 1.
#pragma once
#include "gemm_pipe.hpp"

// (the kernel body is a device function of (problem, workgroup index, workgroups of the problem): focal_dw_ring_kernel runs it for its one
// problem, focal_dw_ring_group_kernel for the problem its blockIdx falls into)
template <int RPS, int NST, bool WITH_BIAS, int HALVES>
__device__ __forceinline__ void focal_dw_ring_body(const GemmParams& p, const int block, const int nblocks) {
  // HALVES = 2: two groups of four waves, each with its own ring and its own slice of the token range, work on the SAME output tile and
  // add their accumulators through LDS before the one atomic pass -- the 8 waves a CU held as two workgroups, with half the atomic
  // volume (the epilogue's fp32 atomics retire at ~1.3 TB/s at the memory side: 2-5 us per launch, profiles/r2_dw_fixed_cost.txt).
  constexpr int TILE = 64;
  constexpr int ROWB = TILE * 2;                 // bytes of a stage row (one operand)
  constexpr int CPR = ROWB / 16;                 // 16-byte chunks per row: 8
  constexpr int RPP = 1024 / ROWB;               // rows per 1 KB LDS-DMA piece: 8
  constexpr int A_BYTES = RPS * ROWB, STAGE_BYTES = 2 * A_BYTES;
  constexpr int APIECES = RPS / RPP;             // pieces of the A part; as many for B
  constexpr int LPW = 2 * APIECES / 4;           // pieces per wave per stage
  constexpr int WT = TILE / 2, NF = WT / 16;     // wave tile edge (2 x 2 waves), fragments per side: 2
  static_assert(APIECES % 4 == 0, "a piece index must be an A piece or a B piece for all four waves");
  constexpr int WPITCH = WT + 4;
  static_assert(NST * STAGE_BYTES >= 4 * (NF * NF + NF) * 1024 && NST * STAGE_BYTES >= 4 * 16 * WPITCH * 4, "epilogue staging must fit in the ring");
  extern __shared__ __attribute__((aligned(1024))) char dww_lds_all[];
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, l15 = lane & 15, tq = l15 >> 2, tp = l15 & 3;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = wave8 >> 2, wave = wave8 & 3;
  char* dww_lds = dww_lds_all + half * (NST * STAGE_BYTES);  // this half's ring
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = p.N / TILE, ntiles = (p.M / TILE) * tiles_n;
  const int logical = xcd_remap(block, nblocks);
  const int tile = logical % ntiles, sp = logical / ntiles;
  const int m0 = (tile / tiles_n) * TILE, n0 = (tile % tiles_n) * TILE;
  const int KT = p.K / RPS;
  const int kt_per = (KT + p.splits * HALVES - 1) / (p.splits * HALVES);  // stages per token slice; slice = sp * HALVES + half
  const int kt0 = min(KT, (sp * HALVES + half) * kt_per), kt1 = min(KT, kt0 + kt_per);
  const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
  const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B);
  float* C = reinterpret_cast<float*>(p.C);

  // swizzle of the 16-byte chunk index by the stage row: the eight rows {r0 .. r0 + 3, r0 + 8 .. r0 + 11} that the 32 lanes of one
  // transposed-read pass touch (32 bytes each) must fall in eight different 32-byte groups of the 256-byte bank period
  // (128-byte rows: r and r + 1 never share banks, so bits 1 and 3 of the row separate the two classes of four rows)
  auto swz = [](int r) { return (((r >> 1) & 1) << 1) | (((r >> 3) & 1) << 2); };

  // ---- fill plan: piece q = wave + 4 t covers RPP stage rows of A (q < APIECES) or B; lane -> row + lane / CPR, position lane % CPR
  uint32_t goff[LPW];
  const char* gbase[LPW];
  long gstep[LPW];
#pragma unroll
  for (int t = 0; t < LPW; ++t) {
    const int q = wave + 4 * t;
    const bool isA = 4 * t < APIECES;
    const int row = RPP * (isA ? q : q - APIECES) + lane / CPR, pos = lane % CPR;
    const int chunk = pos ^ swz(row);
    if (isA) {
      goff[t] = (uint32_t)(((long)row * p.lda + m0 + chunk * 8) * 2);
      gbase[t] = reinterpret_cast<const char*>(A);
      gstep[t] = (long)RPS * p.lda * 2;
    } else {
      goff[t] = (uint32_t)(((long)row * p.ldb + n0 + chunk * 8) * 2);
      gbase[t] = reinterpret_cast<const char*>(B);
      gstep[t] = (long)RPS * p.ldb * 2;
    }
  }
  auto fill = [&](int kt, int stage) {
#pragma unroll
    for (int t = 0; t < LPW; ++t) {
      const int q = wave + 4 * t;
      char* dst = dww_lds + stage * STAGE_BYTES + q * 1024;
      __builtin_amdgcn_global_load_lds((pipe_glb_ptr)(gbase[t] + (long)kt * gstep[t] + goff[t]), (pipe_lds_ptr)dst, 16, 0, 0);
    }
  };

  f32x4 acc[NF][NF], accb[NF];
#pragma unroll
  for (int i = 0; i < NF; ++i) {
    accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const bool do_bias = WITH_BIAS && p.colsumA != nullptr && (n0 == 0) && (wn == 0);
  // the column-sum MFMAs are issued by every wave (against a zero operand where the sums are not wanted): a branch here would split
  // the k-step into basic blocks and stop the scheduler from moving the next fragment reads above this step's MFMAs
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (bf16_t)(do_bias ? 1.0f : 0.0f);

  // transposed-fragment addresses inside a stage: row 8 g + tq (+ 4; + 32 kk), columns 16 tile + 4 tp .. + 3
  const int row_lo = 8 * g + tq;
  const int sw = swz(row_lo);  // unchanged by + 4 and + 32
  int fr_a[NF], fr_b[NF];
#pragma unroll
  for (int i = 0; i < NF; ++i) {
    fr_a[i] = row_lo * ROWB + (((2 * NF * wm + 2 * i + (tp >> 1)) ^ sw) << 4) + (tp & 1) * 8;
    fr_b[i] = A_BYTES + row_lo * ROWB + (((2 * NF * wn + 2 * i + (tp >> 1)) ^ sw) << 4) + (tp & 1) * 8;
  }
  const uint32_t lds0 = pipe_lds_addr(dww_lds);
  auto compute = [&](int stage) {
    const uint32_t sb = lds0 + stage * STAGE_BYTES;
    pipe_static_for<0, RPS / 32>([&](auto kc) {
      constexpr int KO = decltype(kc)::value * 32 * ROWB;
      bf16x4 al[NF], ah[NF], bl[NF], bh[NF];
#pragma unroll
      for (int i = 0; i < NF; ++i) {
        al[i] = pipe_lds_read_tr<KO>(sb + fr_a[i]);
        ah[i] = pipe_lds_read_tr<KO + 4 * ROWB>(sb + fr_a[i]);
      }
#pragma unroll
      for (int j = 0; j < NF; ++j) {
        bl[j] = pipe_lds_read_tr<KO>(sb + fr_b[j]);
        bh[j] = pipe_lds_read_tr<KO + 4 * ROWB>(sb + fr_b[j]);
      }
      bf16x8 xa[NF], wb[NF];
#pragma unroll
      for (int i = 0; i < NF; ++i) {
        xa[i] = __builtin_shufflevector(al[i], ah[i], 0, 1, 2, 3, 4, 5, 6, 7);
        wb[i] = __builtin_shufflevector(bl[i], bh[i], 0, 1, 2, 3, 4, 5, 6, 7);
      }
      pipe_lds_wait(xa, wb);
#pragma unroll
      for (int i = 0; i < NF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = mma16(wb[j], xa[i], acc[i][j]);
      if (WITH_BIAS) {
#pragma unroll
        for (int i = 0; i < NF; ++i) accb[i] = mma16(ones, xa[i], accb[i]);
      }
    });
  };

  const int nk = kt1 - kt0;
#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < nk) fill(kt0 + s, s);
  int stage = 0, fstage = NST - 1;
  // every wave of the workgroup meets the same number of barriers: the trip count is the slice length of a FULL slice (the last
  // slices of the token range may be shorter or empty)
  for (int i = 0; i < (HALVES == 1 ? nk : kt_per); ++i) {
    const int ahead = nk - 1 - i;  // stages issued beyond this one: min(NST - 2, ahead) stay in flight
    if (NST >= 4 && ahead >= 2) pipe_wait_barrier<2 * LPW>();
    else if (NST >= 3 && ahead >= 1) pipe_wait_barrier<LPW>();
    else pipe_wait_barrier<0>();
    if (i + NST - 1 < nk) fill(kt0 + i + NST - 1, fstage);
    if (HALVES == 1 || i < nk) compute(stage);
    stage = (stage + 1 == NST) ? 0 : stage + 1;
    fstage = (fstage + 1 == NST) ? 0 : fstage + 1;
  }
  asm volatile("s_barrier" ::: "memory");  // the rings are re-used as epilogue staging
  if (HALVES >= 2) {
    // the other groups hand their accumulators over (16 bytes per lane and tile, each in its own ring), the first group adds them
    if (half >= 1) {
      float4* xch = reinterpret_cast<float4*>(dww_lds) + wave * (NF * NF + NF) * 64 + lane;
#pragma unroll
      for (int i = 0; i < NF; ++i) {
#pragma unroll
        for (int j = 0; j < NF; ++j) xch[(i * NF + j) * 64] = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        xch[(NF * NF + i) * 64] = make_float4(accb[i][0], accb[i][1], accb[i][2], accb[i][3]);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (half >= 1) return;
#pragma unroll
    for (int h = 1; h < HALVES; ++h) {
      const float4* xch = reinterpret_cast<const float4*>(dww_lds_all + h * (NST * STAGE_BYTES)) + wave * (NF * NF + NF) * 64 + lane;
#pragma unroll
      for (int i = 0; i < NF; ++i) {
#pragma unroll
        for (int j = 0; j < NF; ++j) {
          const float4 v = xch[(i * NF + j) * 64];
          acc[i][j] += f32x4{v.x, v.y, v.z, v.w};
        }
        const float4 vb = xch[(NF * NF + i) * 64];
        accb[i] += f32x4{vb.x, vb.y, vb.z, vb.w};
      }
    }
  }

  // ---- epilogue: D[n][m] accumulators -> this wave's [16 m][32 n] fp32 staging -> contiguous atomics, two 128-byte output rows per
  // wave instruction
  float* est = reinterpret_cast<float*>(dww_lds) + wave * 16 * WPITCH;
  constexpr int RPI = 64 / WT;  // output rows per wave instruction
  const int ecol = lane % WT, erow = lane / WT;
  const int nc = n0 + WT * wn + ecol;
  const float badd = (p.bias && sp == 0) ? p.bias[nc] : 0.f;
#pragma unroll
  for (int i = 0; i < NF; ++i) {
#pragma unroll
    for (int j = 0; j < NF; ++j) {
      const f32x4 v = acc[i][j] * p.alpha;
      *reinterpret_cast<float4*>(est + l15 * WPITCH + j * 16 + g * 4) = make_float4(v[0], v[1], v[2], v[3]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private region: no barrier needed
    float* crow = C + (long)(m0 + WT * wm + 16 * i + erow) * p.ldc + nc;
#pragma unroll 4
    for (int r = 0; r < 16; r += RPI) atomicAdd(crow + (long)r * p.ldc, est[(r + erow) * WPITCH + ecol] + badd);
    if (do_bias && g == 0) atomicAdd(p.colsumA + m0 + WT * wm + 16 * i + l15, accb[i][0]);
  }
}

template <int RPS, int NST, bool WITH_BIAS, int HALVES>
__global__ __launch_bounds__(256 * HALVES) void focal_dw_ring_kernel(const GemmParams p) {
  focal_dw_ring_body<RPS, NST, WITH_BIAS, HALVES>(p, blockIdx.x, gridDim.x);
}

// Several weight gradients of 64-tile shapes as ONE launch (the qkv and proj gradients of a 64-channel Swin block: [192, 64] and
// [64, 64] outputs -- too narrow for the 128 x 128 tiles of gemm_dw_group.hpp): every problem keeps the tiles, ring and atomic epilogue
// of focal_dw_ring_kernel; a prefix table in the kernel arguments maps blockIdx to (problem, workgroup of the problem).
constexpr int DWR_MAX_PROBLEMS = 4;
struct DwRingGroupParams {
  int nprob;
  int wg_end[DWR_MAX_PROBLEMS];  // exclusive prefix ends of the problems' workgroup ranges
  GemmParams prob[DWR_MAX_PROBLEMS];
};
template <int RPS, int NST, int HALVES>
__global__ __launch_bounds__(256 * HALVES) void focal_dw_ring_group_kernel(const DwRingGroupParams gp) {
  const int b = blockIdx.x;
  int pi = 0;
#pragma unroll
  for (int q = 0; q < DWR_MAX_PROBLEMS - 1; ++q) pi += (q < gp.nprob - 1 && b >= gp.wg_end[q]) ? 1 : 0;
  const int start = pi > 0 ? gp.wg_end[pi - 1] : 0;
  focal_dw_ring_body<RPS, NST, true, HALVES>(gp.prob[pi], b - start, gp.wg_end[pi] - start);
}

// splits (workgroups per tile, each taking HALVES token slices) such that the whole group is one round of 256 one-per-CU workgroups
template <int HALVES> static inline int focal_dw_ring_group_plan(DwRingGroupParams& gp) {
  int tiles = 0;
  long rows_min = 1L << 40;
  for (int i = 0; i < gp.nprob; ++i) {
    tiles += (gp.prob[i].M / 64) * (gp.prob[i].N / 64);
    if (gp.prob[i].K < rows_min) rows_min = gp.prob[i].K;
  }
  int per = 256 / (tiles > 0 ? tiles : 1);
  const long max_per = (rows_min + 256L * HALVES - 1) / (256L * HALVES);  // at least 256 reduction rows per token slice
  if (per > max_per) per = (int)max_per;
  if (per < 1) per = 1;
  int end = 0;
  for (int i = 0; i < gp.nprob; ++i) {
    gp.prob[i].splits = per;
    end += (gp.prob[i].M / 64) * (gp.prob[i].N / 64) * per;
    gp.wg_end[i] = end;
  }
  return end;
}

template <int RPS, int NST, int HALVES>
static inline hipError_t focal_launch_dw_ring_group(const DwRingGroupParams& gp, int wgs, hipStream_t stream) {
  constexpr int LDS_BYTES = HALVES * NST * RPS * 64 * 4;
  auto kern = focal_dw_ring_group_kernel<RPS, NST, HALVES>;
  static std::atomic<bool> attr_set{false};  // (the grant is idempotent: two first callers may both issue it; the flag itself is race-free)
  if (!attr_set.load(std::memory_order_acquire)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_set.store(true, std::memory_order_release);
  }
  FOCAL_LAUNCH(kern, dim3(wgs), dim3(256 * HALVES), LDS_BYTES, stream, gp);
  return hipGetLastError();
}

template <int RPS, int NST, int HALVES>
static inline hipError_t focal_launch_dw_ring(const GemmParams& p, hipStream_t stream) {
  constexpr int TILE = 64;
  constexpr int LDS_BYTES = HALVES * NST * RPS * TILE * 4;
  auto kern = focal_dw_ring_kernel<RPS, NST, true, HALVES>;
  auto kern0 = focal_dw_ring_kernel<RPS, NST, false, HALVES>;
  static std::atomic<bool> attr_set{false};  // (the grant is idempotent: two first callers may both issue it; the flag itself is race-free)
  if (!attr_set.load(std::memory_order_acquire)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern0), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_set.store(true, std::memory_order_release);
  }
  dim3 grid((p.M / TILE) * (p.N / TILE) * p.splits);
  if (p.colsumA) FOCAL_LAUNCH(kern, grid, dim3(256 * HALVES), LDS_BYTES, stream, p);
  else FOCAL_LAUNCH(kern0, grid, dim3(256 * HALVES), LDS_BYTES, stream, p);
  return hipGetLastError();
}
