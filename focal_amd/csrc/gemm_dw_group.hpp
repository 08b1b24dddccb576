// Weight gradients of SEVERAL linear layers in one launch, 128 x 128 output tiles on an LDS-DMA ring (bf16 operands):
//
//   for each problem p:   C_p[m][n] += sum_r A_p[r][m] * B_p[r][n]      A_p = dy [rows][M] bf16, B_p = x [rows][N] bf16 (token-major),
//                                                                     C_p fp32 [M][N]; colsum_p[m] += sum_r A_p[r][m] (bias gradient)
//
// Why (round 3; profiles/r2_dw_fixed_cost.txt): one weight gradient of a deep Swin stage is a tiny output (0.25 - 1 MB)
// reduced over 10^4 - 10^5 tokens.  The 64 x 64 ring kernel (gemm_dw_ring.hpp) runs it as ~256 workgroups and is bound by (1) the
// L2 -> LDS fill rate -- a 64-column tile moves 16 KB into LDS per 64 x 64 x 64 MACs -- and (2) ~6 us of launch ramp + atomic
// epilogue per launch, four launches per Swin block.  A 128 x 128 tile halves the fill bytes per MAC, but one layer has too few
// such tiles: filling the chip needs 8 - 20 token slices per tile, i.e. 16 MB of fp32 atomics per launch (the r2 128 x 128 variant
// lost exactly there).  The four linears of a block (qkv, proj, fc1, fc2) as ONE launch have 4x the tiles: the same atomic volume
// serves four layers, one ramp instead of four, and every workgroup runs the wide tile.
//
// Workgroup = 8 waves (2 x 4, 64 x 32 outputs per wave: 12 transposed fragment reads per 8 MFMAs; the 64 x 64 kernel needs 8 per 4).
// A stage of the ring is RPS token rows of the A panel (128 columns) and of the B panel, stored as four [RPS][64] sub-tiles
// (A0 | A1 | B0 | B1) in exactly the layout of gemm_dw_ring.hpp -- 128-byte rows, 16-byte chunk c of row r at c ^ (bit1(r) << 1 |
// bit3(r) << 2), ds_read_b64_tr_b16 fragments, inline-asm reads under the kernel's own counted vmcnt -- so the addressing is that
// kernel's, per sub-tile.  A problem's token range is cut into `splits` slices; a tile owned by ONE workgroup (splits == 1) leaves
// with plain read-add-write (problem flag `exclusive`: nobody else adds to C_p during the launch), otherwise with fp32 atomics.
// blockIdx -> (problem, tile, slice) through a small prefix table in the kernel arguments; inside a problem the tile index runs
// fastest, so neighbouring workgroups (one XCD's run after xcd_remap) read the same token slice and share it in L2.
#pragma once
#include "gemm_pipe.hpp"

constexpr int DWG_MAX_PROBLEMS = FOCAL_DW_GROUP_MAX_PROBLEMS;  // (include/focal_hip.h)

struct DwGroupProblem {
  const bf16_t* A; const bf16_t* B; float* C; float* colsum;  // colsum may be null
  int lda, ldb, ldc;
  int M, N, rows;      // output [M][N], reduction over `rows` tokens
  int splits;          // token slices per tile
  int exclusive;       // 1: plain read-add-write is allowed when splits == 1
};
struct DwGroupParams {
  int nprob;
  int wg_end[DWG_MAX_PROBLEMS];  // exclusive prefix ends of the problems' workgroup ranges
  DwGroupProblem prob[DWG_MAX_PROBLEMS];
};

template <int RPS, int NST>
__global__ __launch_bounds__(512) void focal_dw_group_kernel(const DwGroupParams gp) {
  constexpr int SUB_BYTES = RPS * 128;           // one [RPS][64] bf16 sub-tile
  constexpr int STAGE_BYTES = 4 * SUB_BYTES;     // A0 | A1 | B0 | B1
  constexpr int PPS = RPS / 8;                   // 1 KB LDS-DMA pieces (8 rows) per sub-tile
  constexpr int LPW = 4 * PPS / 8;               // pieces per wave per stage
  static_assert(RPS % 32 == 0 && (4 * PPS) % 8 == 0, "stage geometry");
  constexpr int NFA = 4, NFB = 2;                // 16-wide fragments per wave: 64 A columns, 32 B columns
  constexpr int WPITCH = 32 + 4;
  static_assert(NST * STAGE_BYTES >= 8 * 16 * WPITCH * 4, "epilogue staging must fit in the ring");
  extern __shared__ __attribute__((aligned(1024))) char dwg_lds[];
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, l15 = lane & 15, tq = l15 >> 2, tp = l15 & 3;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  // ---- which problem, tile, slice
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  int pi = 0;
#pragma unroll
  for (int q = 0; q < DWG_MAX_PROBLEMS - 1; ++q) pi += (q < gp.nprob - 1 && logical >= gp.wg_end[q]) ? 1 : 0;
  const DwGroupProblem& p = gp.prob[pi];
  const int local = logical - (pi > 0 ? gp.wg_end[pi - 1] : 0);
  const int tiles_n = p.N >> 7, ntiles = (p.M >> 7) * tiles_n;
  const int tile = local % ntiles, sp = local / ntiles;
  const int m0 = (tile / tiles_n) << 7, n0 = (tile % tiles_n) << 7;
  const int KT = p.rows / RPS;
  const int kt_per = (KT + p.splits - 1) / p.splits;
  const int kt0 = min(KT, sp * kt_per), kt1 = min(KT, kt0 + kt_per);
  const int nk = kt1 - kt0;

  auto swz = [](int r) { return (((r >> 1) & 1) << 1) | (((r >> 3) & 1) << 2); };

  // ---- fill plan: piece q = wave + 8 t of a stage: sub-tile q / PPS (A0, A1, B0, B1), rows 8 (q % PPS) .. + 7; lane -> row + lane / 8,
  // chunk position lane % 8 (holding chunk position ^ swizzle(row) of the memory row)
  uint32_t goff[LPW];
  const char* gbase[LPW];
  long gstep[LPW];
#pragma unroll
  for (int t = 0; t < LPW; ++t) {
    const int q = wave + 8 * t;
    const int sub = q / PPS, row = 8 * (q % PPS) + (lane >> 3), pos = lane & 7;
    const int chunk = pos ^ swz(row);
    if (sub < 2) {
      goff[t] = (uint32_t)(((long)row * p.lda + m0 + 64 * sub + chunk * 8) * 2);
      gbase[t] = reinterpret_cast<const char*>(p.A);
      gstep[t] = (long)RPS * p.lda * 2;
    } else {
      goff[t] = (uint32_t)(((long)row * p.ldb + n0 + 64 * (sub - 2) + chunk * 8) * 2);
      gbase[t] = reinterpret_cast<const char*>(p.B);
      gstep[t] = (long)RPS * p.ldb * 2;
    }
  }
  auto fill = [&](int kt, int stage) {
#pragma unroll
    for (int t = 0; t < LPW; ++t) {
      const int q = wave + 8 * t;
      char* dst = dwg_lds + stage * STAGE_BYTES + q * 1024;
      __builtin_amdgcn_global_load_lds((pipe_glb_ptr)(gbase[t] + (long)kt * gstep[t] + goff[t]), (pipe_lds_ptr)dst, 16, 0, 0);
    }
  };

  f32x4 acc[NFA][NFB], accb[NFA];
#pragma unroll
  for (int i = 0; i < NFA; ++i) {
    accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NFB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const bool do_bias = (p.colsum != nullptr) && (n0 == 0) && (wn == 0);  // wave-uniform
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (bf16_t)1.0f;

  // transposed-fragment addresses inside a stage (gemm_dw_ring.hpp): row 8 g + tq (+ 4; + 32 kc), fragment f of a sub-tile = chunks 2 f, 2 f + 1
  const int row_lo = 8 * g + tq;
  const int sw = swz(row_lo);
  int fr_a[NFA], fr_b[NFB];
#pragma unroll
  for (int i = 0; i < NFA; ++i) fr_a[i] = wm * SUB_BYTES + row_lo * 128 + (((2 * i + (tp >> 1)) ^ sw) << 4) + (tp & 1) * 8;
#pragma unroll
  for (int j = 0; j < NFB; ++j)
    fr_b[j] = (2 + (wn >> 1)) * SUB_BYTES + row_lo * 128 + (((2 * (2 * (wn & 1) + j) + (tp >> 1)) ^ sw) << 4) + (tp & 1) * 8;
  const uint32_t lds0 = pipe_lds_addr(dwg_lds);
  auto compute = [&](int stage) {
    const uint32_t sb = lds0 + stage * STAGE_BYTES;
    pipe_static_for<0, RPS / 32>([&](auto kc) {
      constexpr int KO = decltype(kc)::value * 32 * 128;
      bf16x4 al[NFA], ah[NFA], bl[NFB], bh[NFB];
#pragma unroll
      for (int i = 0; i < NFA; ++i) {
        al[i] = pipe_lds_read_tr<KO>(sb + fr_a[i]);
        ah[i] = pipe_lds_read_tr<KO + 4 * 128>(sb + fr_a[i]);
      }
#pragma unroll
      for (int j = 0; j < NFB; ++j) {
        bl[j] = pipe_lds_read_tr<KO>(sb + fr_b[j]);
        bh[j] = pipe_lds_read_tr<KO + 4 * 128>(sb + fr_b[j]);
      }
      bf16x8 xa[NFA], wb[NFB];
#pragma unroll
      for (int i = 0; i < NFA; ++i) xa[i] = __builtin_shufflevector(al[i], ah[i], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
      for (int j = 0; j < NFB; ++j) wb[j] = __builtin_shufflevector(bl[j], bh[j], 0, 1, 2, 3, 4, 5, 6, 7);
      pipe_lds_wait(xa, wb);
#pragma unroll
      for (int i = 0; i < NFA; ++i)
#pragma unroll
        for (int j = 0; j < NFB; ++j) acc[i][j] = mma16(wb[j], xa[i], acc[i][j]);
      if (do_bias) {  // column sums of A against a ones operand: two of the eight waves, a scalar branch
#pragma unroll
        for (int i = 0; i < NFA; ++i) accb[i] = mma16(ones, xa[i], accb[i]);
      }
    });
  };

#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < nk) fill(kt0 + s, s);
  int stage = 0, fstage = NST - 1;
  for (int i = 0; i < nk; ++i) {
    const int ahead = nk - 1 - i;  // stages issued beyond this one: min(NST - 2, ahead) stay in flight
    if (NST >= 4 && ahead >= 2) pipe_wait_barrier<2 * LPW>();
    else if (NST >= 3 && ahead >= 1) pipe_wait_barrier<LPW>();
    else pipe_wait_barrier<0>();
    if (i + NST - 1 < nk) fill(kt0 + i + NST - 1, fstage);
    compute(stage);
    stage = (stage + 1 == NST) ? 0 : stage + 1;
    fstage = (fstage + 1 == NST) ? 0 : fstage + 1;
  }
  if (nk == 0) return;  // an empty slice (the last slices of a short token range) adds nothing
#if defined(DWG_LAB_NOEPI)  // lab: the launch without its epilogue (tools/mb_dw.py; every accumulator stays live through a never-taken store)
  {
    float lab = 0.f;
#pragma unroll
    for (int i = 0; i < NFA; ++i) {
      lab += accb[i][0];
#pragma unroll
      for (int j = 0; j < NFB; ++j) lab += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    }
    if (lab == 12345.678f) p.C[0] = 1.f;
    return;
  }
#endif

  // ---- epilogue.  acc[i][j][e] = C[m0 + 64 wm + 16 i + l15][n0 + 32 wn + 16 j + 4 g + e]
  const int mrow = m0 + 64 * wm, ncol = n0 + 32 * wn;
  if (p.splits == 1 && p.exclusive) {
    // the tile is this workgroup's alone: read-add-write, 16 bytes per lane (the gradient buffer may already hold another pass's sum)
#pragma unroll
    for (int i = 0; i < NFA; ++i)
#pragma unroll
      for (int j = 0; j < NFB; ++j) {
        float4* dst = reinterpret_cast<float4*>(p.C + (long)(mrow + 16 * i + l15) * p.ldc + ncol + 16 * j + 4 * g);
        float4 v = *dst;
        v.x += acc[i][j][0]; v.y += acc[i][j][1]; v.z += acc[i][j][2]; v.w += acc[i][j][3];
        *dst = v;
      }
  } else {
    asm volatile("s_barrier" ::: "memory");  // the ring is re-used as epilogue staging
    float* est = reinterpret_cast<float*>(dwg_lds) + wave * 16 * WPITCH;
    const int ecol = lane & 31, erow = lane >> 5;  // two 128-byte output rows per wave instruction
#pragma unroll
    for (int i = 0; i < NFA; ++i) {
#pragma unroll
      for (int j = 0; j < NFB; ++j)
        *reinterpret_cast<float4*>(est + l15 * WPITCH + j * 16 + g * 4) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private region: no barrier needed
      float* crow = p.C + (long)(mrow + 16 * i + erow) * p.ldc + ncol + ecol;
#pragma unroll 4
#if defined(DWG_LAB_STORE)  // lab: plain stores where the atomics are (same bytes, wrong sums)
      for (int r = 0; r < 16; r += 2) crow[(long)r * p.ldc] = est[(r + erow) * WPITCH + ecol];
#else
      for (int r = 0; r < 16; r += 2) atomicAdd(crow + (long)r * p.ldc, est[(r + erow) * WPITCH + ecol]);
#endif
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the reads above are done before the next fragment row overwrites the region
    }
  }
  if (do_bias && g == 0) {
#pragma unroll
    for (int i = 0; i < NFA; ++i) atomicAdd(p.colsum + mrow + 16 * i + l15, accb[i][0]);
  }
}

// Token slices per problem: ~one workgroup per CU over the whole launch (the ring takes most of a CU's LDS), every workgroup about the
// same number of ring stages; a slice is never shorter than `min_steps` stages.  Returns the total workgroup count.
template <int RPS>
static inline int focal_dw_group_plan(DwGroupParams& gp, int target_wgs, int min_steps) {
  long work = 0;
  for (int q = 0; q < gp.nprob; ++q) work += (long)(gp.prob[q].M >> 7) * (gp.prob[q].N >> 7) * (gp.prob[q].rows / RPS);
  long per = (work + target_wgs - 1) / target_wgs;
  if (per < min_steps) per = min_steps;
  int total = 0;
  for (int it = 0; it < 64; ++it) {
    total = 0;
    for (int q = 0; q < gp.nprob; ++q) {
      DwGroupProblem& p = gp.prob[q];
      const int KT = p.rows / RPS;
      int s = (int)((KT + per - 1) / per);
      if (s < 1) s = 1;
      p.splits = s;
      total += (p.M >> 7) * (p.N >> 7) * s;
    }
    bool can_shrink = false;
    for (int q = 0; q < gp.nprob; ++q) can_shrink = can_shrink || gp.prob[q].splits > 1;
    if (total <= target_wgs || !can_shrink) break;
    per += (per + 15) / 16;
  }
  int end = 0;
  for (int q = 0; q < gp.nprob; ++q) {
    end += (gp.prob[q].M >> 7) * (gp.prob[q].N >> 7) * gp.prob[q].splits;
    gp.wg_end[q] = end;
  }
  for (int q = gp.nprob; q < DWG_MAX_PROBLEMS; ++q) gp.wg_end[q] = end;
  return end;
}

template <int RPS, int NST>
static inline hipError_t focal_launch_dw_group(const DwGroupParams& gp, int wgs, hipStream_t stream) {
  constexpr int LDS_BYTES = NST * 4 * RPS * 128;
  auto kern = focal_dw_group_kernel<RPS, NST>;
  static std::atomic<bool> attr_set{false};  // (the grant is idempotent: two first callers may both issue it; the flag itself is race-free)
  if (!attr_set.load(std::memory_order_acquire)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_set.store(true, std::memory_order_release);
  }
  FOCAL_LAUNCH(kern, dim3(wgs), dim3(512), LDS_BYTES, stream, gp);
  return hipGetLastError();
}
