// MFMA GEMM family for the FOCAL hot path (gfx950).
//
//   C[m][n] (+)= epi( alpha * sum_r  proA(A)[m][r] * proB(B)[n][r] )
//
// Each operand is either "direct" (memory [i][r], r contiguous -> forward / dX's activation side) or
// "transposed" (memory [r][i], i contiguous -> weights in dX, both operands in dW).  Transposed tiles are staged
// as-is into LDS and read back as MFMA fragments with ds_read_b64_tr_b16 (bf16) or strided ds_read_b32 (f32), so
// no transposed copy of any tensor ever exists in HBM.
// Compute type CT is bf16 (v_mfma_f32_16x16x32_bf16) or f32 (v_mfma_f32_16x16x4_f32, exact fp32 -- the parity mode).
// The MFMA is issued as D = Bfrag x Afrag so that a lane ends up with 4 consecutive n for one m: 8/16-byte stores.
#pragma once
#include <stdlib.h>
#include <type_traits>
#include "common.hpp"

enum GemmPro { PRO_NONE = 0, PRO_GELU = 1, PRO_MASK = 2, PRO_CONV = 3 };
enum GemmEpi { EPI_STORE = 0, EPI_RESID = 1, EPI_MUL_AUX = 2, EPI_RELU = 3, EPI_RELU_BWD = 4, EPI_ATOMIC = 5, EPI_GELU_FWD = 6,
               EPI_RESID_LN = 7,    // EPI_RESID, then LayerNorm of the finished row (N == 64 == one wave's tile width): y_ln, statistics
               EPI_LN_BWD = 8,      // the product IS the gradient w.r.t. a LayerNorm's output: finish that LayerNorm's backward on the row
                                    // (gemm_pipe.hpp, row-complete wave tiles): C (fp32) += dx, aux_out = dtype(C * mask), dgamma / dbeta
               EPI_STORE_STATS = 9 };  // EPI_STORE (fp32 C) + the BatchNorm statistics of C's columns: per-column sum / sum of squares into
                                    // bn_sums (16 slots x 2N, zeroed by the caller), the last workgroup to arrive finalises mean / rstd

struct MaskParams {
  const uint32_t* seed;  // device word (null -> seed 0)
  uint32_t stream_elem;  // element-wise dropout stream id
  float p_elem;          // element-wise dropout probability (0 = off)
  uint32_t stream_path;  // per-sample stochastic-depth stream id
  float p_path;          // DropPath probability (0 = off)
  int rows_per_sample;   // memory rows per sample (for DropPath)
  int ncols;             // memory columns (linear element index = row * ncols + col)
};

struct GemmParams {
  int M, N, K;
  const void* A; long lda; long strideA;
  const void* B; long ldb; long strideB;
  void* C; long ldc; long strideC;
  int batch, splits;
  int dw_target;                  // weight-gradient launches: workgroups the token-split plan aims at (0 = default; focal_dw_plan)
  float alpha;
  const float* bias;              // [N] f32 or null
  const float* resid; long ldr;   // f32 [M][N] (EPI_RESID)
  const void* aux; long ldaux;    // [M][N]: EPI_MUL_AUX: CT multiplier (saved activation derivative); EPI_RELU_BWD: relu output (TC)
  void* aux_out;                  // TC [M][N], ldc: EPI_GELU_FWD writes d gelu/dx * dropout mask here
  MaskParams proA, proB, epi;
  float* colsumA;                 // f32 [M] (+=): sum_r proA(A)[m][r]; only with transposed A (bias gradient)
  // EPI_RESID_LN: the next LayerNorm, applied to the finished residual row (aux_out = its CT output [M][N], ldc)
  const float* ln_gamma; const float* ln_beta; float* ln_stats; float ln_eps;  // stats f32 [M][2] = {mean, rstd}
  // EPI_LN_BWD: resid = the LayerNorm's input x (fp32 [M][N], ldr), ln_stats read, ln_gamma; column sums of dy * xhat / dy are added here
  float* ln_dgamma; float* ln_dbeta;
  // EPI_STORE_STATS (training-mode BatchNorm behind a [1,k] convolution, DeepSense): bn_sums = 16 slots of {sum[N], sum of squares[N]} followed
  // by the arrival counter, all zero on entry; statistics over bn_rows rows; outputs as focal_bn_stats in FOCAL_BN_TRAIN mode
  float* bn_sums; float* bn_mean_rstd; float* bn_run_mean; float* bn_run_var; long bn_rows; float bn_eps, bn_momentum;
  // bn_groups > 1: rows [g bn_rows, (g + 1) bn_rows) have statistics of their own (bn_rows a multiple of the tile height); sums / counter,
  // mean_rstd, run_* are bn_groups copies of the one-group layout
  int bn_groups;
};
constexpr int BN_STAT_SLOTS = 16;

struct MaskEval {
  DropCtx e, p;
  bool on_e, on_p;
  int rps, ncols;
  uint32_t pad, rps_magic;  // ceil(2^32 / rps): row / rps == mulhi(row, magic) while row * rps < 2^32
  __device__ __forceinline__ void init(const MaskParams& m) {
    pad = m.stream_elem;
    on_e = m.p_elem > 0.f;
    on_p = m.p_path > 0.f;
    e = make_drop(m.seed, m.stream_elem, m.p_elem);
    p = make_drop(m.seed, m.stream_path, m.p_path);
    rps = m.rows_per_sample > 0 ? m.rows_per_sample : 1;
    ncols = m.ncols;
    rps_magic = 0xFFFFFFFFu / (uint32_t)rps + 1u;
  }
  // PRO_CONV: the operand is a [1, k] "same" convolution window over channel-last tokens: memory row m, column
  // kk = tap * Cin + ci reads token m + tap - pad, valid only while it stays inside the same interval of S tokens
  // (rows_per_sample = S, ncols = Cin, stream_elem = pad).
  __device__ __forceinline__ bool conv_valid(int row, int col, uint32_t pad) const {
    const int s = row % rps, t = col / ncols;
    const int q = s + t - (int)pad;
    return q >= 0 && q < rps;
  }
  // (the sample index by multiply-high -- a runtime division is ~20 vector instructions per row, two of them at quarter rate; exact for
  //  row < 2^20, rps < 2^12, else the division -- and the element index by a 24-bit multiply: rows and columns are far below 2^24, the
  //  launchers check)
  __device__ __forceinline__ uint32_t sample_of(int row) const {
    if (rps == 1) return (uint32_t)row;
    return ((uint32_t)row < (1u << 20) && rps < (1 << 12)) ? __umulhi((uint32_t)row, rps_magic) : (uint32_t)(row / rps);
  }
  __device__ __forceinline__ float row_mult(int row) const { return on_p ? drop_mult(p, sample_of(row)) : 1.0f; }
  __device__ __forceinline__ float elem_mult(int row, int col) const {
    return on_e ? drop_mult(e, __umul24((uint32_t)row, (uint32_t)ncols) + (uint32_t)col) : 1.0f;
  }
  // FOUR neighbouring elements (col % 4 == 0) from ONE hash and a three-instruction second word: 16-bit fields against a 16-bit
  // threshold.  Only for sites whose mask is consumed where it is drawn and never regenerated elsewhere (the fc1 GELU epilogue: the
  // saved derivative carries it).  Round 5: the epilogue is vector-issue-bound (profiles/r5_ring_lab.txt) and the per-pair hash was
  // a third of its instructions; keep rates, pairwise and lag correlations of the four decisions: tools/hash_eval.py (all < 1e-3).
  __device__ __forceinline__ void elem_mult_quad(int row, int col, gelu_f2& m01, gelu_f2& m23) const {
    if (!on_e) { m01 = gelu_f2{1.0f, 1.0f}; m23 = gelu_f2{1.0f, 1.0f}; return; }
    const uint32_t idx = (__umul24((uint32_t)row, (uint32_t)ncols) + (uint32_t)col) >> 2;
    const uint32_t h = focal_hash24(idx ^ e.key), t16 = e.thresh >> 8;
    uint32_t g = h ^ (h >> 13);
    g = __umul24(g, 0xC2B2AFu) + 0x165667B1u;
    g ^= g >> 15;
    m01 = gelu_f2{(h & 0xffffu) < t16 ? 0.0f : e.scale, (h >> 16) < t16 ? 0.0f : e.scale};
    m23 = gelu_f2{(g & 0xffffu) < t16 ? 0.0f : e.scale, (g >> 16) < t16 ? 0.0f : e.scale};
  }
};

template <typename CT> struct GemmCfg;
template <> struct GemmCfg<bf16_t> {
  static constexpr int BK = 64, EC = 8, KI = 32, PADK = 8, PADI = 8;
};
template <> struct GemmCfg<float> {
  static constexpr int BK = 32, EC = 4, KI = 4, PADK = 4, PADI = 4;
};

// ---------------------------------------------------------------------------------------------- staging helpers
template <typename T, int N> struct RawChunk {
  static constexpr int R = (N * (int)sizeof(T)) / 16;
  uint4 v[R];
};
template <typename T, int N>
__device__ __forceinline__ void raw_load(RawChunk<T, N>& r, const T* p, bool pred) {
#pragma unroll
  for (int i = 0; i < RawChunk<T, N>::R; ++i) r.v[i] = pred ? reinterpret_cast<const uint4*>(p)[i] : make_uint4(0, 0, 0, 0);
}
template <int N> __device__ __forceinline__ void raw_to_f32(const RawChunk<float, N>& r, float* f) {
#pragma unroll
  for (int i = 0; i < N / 4; ++i) {
    f[4 * i + 0] = __uint_as_float(r.v[i].x);
    f[4 * i + 1] = __uint_as_float(r.v[i].y);
    f[4 * i + 2] = __uint_as_float(r.v[i].z);
    f[4 * i + 3] = __uint_as_float(r.v[i].w);
  }
}
__device__ __forceinline__ void raw_to_f32(const RawChunk<bf16_t, 8>& r, float* f) {
  const uint32_t w[4] = {r.v[0].x, r.v[0].y, r.v[0].z, r.v[0].w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __uint_as_float(w[i] << 16);
    f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
  }
}
__device__ __forceinline__ void store_chunk(float* dst, const float* f) {
  *reinterpret_cast<float4*>(dst) = make_float4(f[0], f[1], f[2], f[3]);
}
__device__ __forceinline__ void store_chunk(bf16_t* dst, const float* f) {
  bf16x8 v;
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (bf16_t)f[i];
  *reinterpret_cast<bf16x8*>(dst) = v;
}

template <int PRO>
__device__ __forceinline__ void apply_prologue(float* f, int n, int row, int col0, const MaskEval& me) {
  if (PRO == PRO_NONE) return;
  if (PRO == PRO_GELU) {
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (e < n) f[e] = gelu_f(f[e]) * me.elem_mult(row, col0 + e);
  } else if (PRO == PRO_CONV) {
    // nothing to do: invalid taps were predicated off at load time and arrive as zeros
  } else {  // PRO_MASK
    const float rm = me.row_mult(row);
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (e < n) f[e] = f[e] * rm * me.elem_mult(row, col0 + e);
  }
}

// One operand's staging state for one workgroup (256 threads).
template <typename CT, typename TG, bool TRANS, int PRO, int BI, int BKM> struct OperandStage {
  using Cfg = GemmCfg<CT>;
  static constexpr int BK = Cfg::BK * BKM, EC = Cfg::EC;
  static constexpr int PITCH = TRANS ? (BI + Cfg::PADI) : (BK + Cfg::PADK);
  static constexpr int LDS_ELEMS = TRANS ? BK * PITCH : BI * PITCH;
  static constexpr int NCH = (BK / EC) * BI / 256;  // 16-byte chunks per thread
  static constexpr int CPR = TRANS ? BI / EC : BK / EC;
  RawChunk<TG, EC> raw[NCH];
  long off[NCH];  // element offset of each of this thread's chunks at r0 = 0: the 64-bit row * ld products are formed once per
                  // workgroup, not once per k-step (they were ~6 vector instructions per load, three of them quarter-rate)

  __device__ __forceinline__ void init(long ld, int i0, int tid) {
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = tid + 256 * j;
      const int a = c / CPR, b = (c % CPR) * EC;
      off[j] = TRANS ? (long)a * ld + (i0 + b) : (long)(i0 + a) * ld + b;
    }
  }

  // i_ext: extent of the non-reduced index, r_ext: extent of the reduced index (both in elements)
  __device__ __forceinline__ void load(const TG* base, long ld, int i0, int r0, int i_ext, int r_end, int tid,
                                       const MaskEval& me) {
    const TG* step_base = base + (TRANS ? (long)r0 * ld : (long)r0);  // uniform: scalar arithmetic
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = tid + 256 * j;
      const int a = c / CPR, b = (c % CPR) * EC;
      if (TRANS) {  // memory [r][i]
        const int r = r0 + a, i = i0 + b;
        bool ok = (r < r_end) && (i < i_ext);
        if (PRO == PRO_CONV) ok = ok && me.conv_valid(r, i, me.pad);  // out-of-window taps are never dereferenced
        raw_load(raw[j], step_base + off[j], ok);
      } else {  // memory [i][r]
        const int i = i0 + a, r = r0 + b;
        bool ok = (i < i_ext) && (r < r_end);
        if (PRO == PRO_CONV) ok = ok && me.conv_valid(i, r, me.pad);
        raw_load(raw[j], step_base + off[j], ok);
      }
    }
  }
  // csum (transposed operands only): running column sums of the prologue'd values.  A thread's chunks always cover
  // the same EC columns ((tid % CPR) * EC, since 256 % CPR == 0), so the bias gradient accumulates in registers
  // during staging -- no extra LDS or HBM traffic.
  template <bool CSUM>
  __device__ __forceinline__ void store(CT* lds, int i0, int r0, int tid, const MaskEval& me, float* csum) {
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = tid + 256 * j;
      const int a = c / CPR, b = (c % CPR) * EC;
      if (std::is_same<TG, CT>::value && PRO == PRO_NONE && !CSUM) {
        // already in the operand dtype and nothing to apply: a 16-byte copy (going through fp32 and back costs 12 vector
        // instructions per chunk that the compiler cannot fold away -- NaN payloads -- 48 per k-step beside 8 MFMAs)
#pragma unroll
        for (int q = 0; q < RawChunk<TG, EC>::R; ++q) reinterpret_cast<uint4*>(lds + a * PITCH + b)[q] = raw[j].v[q];
        continue;
      }
      float f[8];
      raw_to_f32(raw[j], f);
      if (TRANS) apply_prologue<PRO>(f, EC, r0 + a, i0 + b, me);
      else apply_prologue<PRO>(f, EC, i0 + a, r0 + b, me);
      if (CSUM) {
#pragma unroll
        for (int e = 0; e < EC; ++e) csum[e] += f[e];
      }
      store_chunk(lds + a * PITCH + b, f);
    }
  }
};

// ---------------------------------------------------------------------------------------------- fragments
template <typename CT, bool TRANS, int PITCH> struct FragLoad;
template <int PITCH> struct FragLoad<bf16_t, false, PITCH> {
  using Frag = bf16x8;
  // tile [i][k]; 16 rows starting at ibase, k-step kk (32 wide)
  static __device__ __forceinline__ Frag load(const bf16_t* lds, int ibase, int kk, int lane) {
    return *reinterpret_cast<const bf16x8*>(lds + (ibase + (lane & 15)) * PITCH + kk * 32 + 8 * (lane >> 4));
  }
};
template <int PITCH> struct FragLoad<bf16_t, true, PITCH> {
  using Frag = bf16x8;
  // tile [k][i]; hardware transpose read: two 4x16 blocks give this lane's 8 consecutive k for column ibase+(lane&15)
  static __device__ __forceinline__ Frag load(const bf16_t* lds, int ibase, int kk, int lane) {
    const int q = (lane & 15) >> 2, p = lane & 3;
    const bf16_t* a0 = lds + (kk * 32 + 8 * (lane >> 4) + q) * PITCH + ibase + 4 * p;
    typedef __attribute__((address_space(3))) bf16x4* lds_ptr;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(a0));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(a0 + 4 * PITCH));
    Frag f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
  }
};
template <int PITCH> struct FragLoad<float, false, PITCH> {
  using Frag = float;
  static __device__ __forceinline__ Frag load(const float* lds, int ibase, int kk, int lane) {
    return lds[(ibase + (lane & 15)) * PITCH + kk * 4 + (lane >> 4)];
  }
};
template <int PITCH> struct FragLoad<float, true, PITCH> {
  using Frag = float;
  static __device__ __forceinline__ Frag load(const float* lds, int ibase, int kk, int lane) {
    return lds[(kk * 4 + (lane >> 4)) * PITCH + ibase + (lane & 15)];
  }
};
__device__ __forceinline__ f32x4 mma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// ---------------------------------------------------------------------------------------------- output helpers
__device__ __forceinline__ void store4(float* p, f32x4 v) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ void store4(bf16_t* p, f32x4 v) {
  bf16x4 o;
  o[0] = (bf16_t)v[0]; o[1] = (bf16_t)v[1]; o[2] = (bf16_t)v[2]; o[3] = (bf16_t)v[3];
  *reinterpret_cast<bf16x4*>(p) = o;
}
__device__ __forceinline__ f32x4 load4(const float* p) { float4 t = *reinterpret_cast<const float4*>(p); return f32x4{t.x, t.y, t.z, t.w}; }
__device__ __forceinline__ f32x4 load4(const bf16_t* p) {
  bf16x4 t = *reinterpret_cast<const bf16x4*>(p);
  return f32x4{(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
}

// N-element (16-byte for bf16 x 8 / fp32 x 4, 32-byte for fp32 x 8) vector load / store helpers, fp32 in registers
template <int N> __device__ __forceinline__ void loadN(const float* p, float* f) {
#pragma unroll
  for (int i = 0; i < N / 4; ++i) {
    const float4 t = reinterpret_cast<const float4*>(p)[i];
    f[4 * i] = t.x; f[4 * i + 1] = t.y; f[4 * i + 2] = t.z; f[4 * i + 3] = t.w;
  }
}
template <int N> __device__ __forceinline__ void loadN(const bf16_t* p, float* f) {
  if (N == 8) {
    const bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (float)t[i];
  } else {
    const bf16x4 t = *reinterpret_cast<const bf16x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = (float)t[i];
  }
}
template <int N> __device__ __forceinline__ void storeN(float* p, const float* f) {
#pragma unroll
  for (int i = 0; i < N / 4; ++i) reinterpret_cast<float4*>(p)[i] = make_float4(f[4 * i], f[4 * i + 1], f[4 * i + 2], f[4 * i + 3]);
}
template <int N> __device__ __forceinline__ void storeN(bf16_t* p, const float* f) {
  if (N == 8) {
    bf16x8 t;
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = (bf16_t)f[i];
    *reinterpret_cast<bf16x8*>(p) = t;
  } else {
    bf16x4 t;
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = (bf16_t)f[i];
    *reinterpret_cast<bf16x4*>(p) = t;
  }
}

// XCD-aware tile order: blocks b and b+8 share an XCD (and its L2), so give each XCD a contiguous run of logical
// tiles; with n fastest, the column tiles that re-read the same activation rows hit that XCD's L2.  Bijective for
// any block count.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, x = bid & 7, i = bid >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// BKM: reduction-tile multiplier.  The weight-gradient GEMMs (reduction over 1e4-1e5 tokens, tiny output) use 2: twice
// the bytes in flight per workgroup and half the barriers in a loop that is bound by memory latency.
// (the kernel is a device function of (problem, workgroup index, workgroups of the problem): focal_gemm_kernel runs it for its one problem,
// the loss head's grouped launch -- loss.hip -- for the problem its blockIdx falls into)
template <typename CT, typename TA, typename TB, typename TC, bool TRA, bool TRB, int PROA, int PROB, int EPI, int BM, int BN, int BKM = 1>
__device__ __forceinline__ void focal_gemm_body(const GemmParams& p, const int block, const int nblocks) {
#include "gemm_body.inc"
}

template <typename CT, typename TA, typename TB, typename TC, bool TRA, bool TRB, int PROA, int PROB, int EPI, int BM, int BN, int BKM = 1>
__global__ __launch_bounds__(256) void focal_gemm_kernel(const GemmParams p) {
  const int block = blockIdx.x, nblocks = gridDim.x;
#include "gemm_body.inc"
}

// Several weight gradients with fp32 operands (dW = dy^T x; DeepSense's GRU: eight per layer pair, 16 us each at M = 5 120 rows) as ONE
// launch of the same 64 x 64 tiles behind a problem table (round 5).  The body is included textually, as in focal_gemm_kernel: as a called
// device function the BKM = 2 instances ran 1.7x slower.
constexpr int DW_TAIL_MAX = 8;
struct DwTailGroup { int n; int wg_end[DW_TAIL_MAX]; GemmParams p[DW_TAIL_MAX]; };
static_assert(sizeof(DwTailGroup) <= 4096, "kernel arguments");
template <typename CT>
__global__ __launch_bounds__(256) void focal_dw_tail_group_kernel(const DwTailGroup g) {
  using TA = float;
  using TB = float;
  using TC = float;
  constexpr bool TRA = true, TRB = true;
  constexpr int PROA = PRO_NONE, PROB = PRO_NONE, EPI = EPI_ATOMIC, BM = 64, BN = 64, BKM = 2;
  int pi = 0;
#pragma unroll
  for (int q = 0; q < DW_TAIL_MAX - 1; ++q) pi += (q < g.n - 1 && (int)blockIdx.x >= g.wg_end[q]) ? 1 : 0;
  const int start = pi > 0 ? g.wg_end[pi - 1] : 0;
  const GemmParams& p = g.p[pi];
  const int block = (int)blockIdx.x - start, nblocks = g.wg_end[pi] - start;
#include "gemm_body.inc"
}

// Launch plan of a weight-gradient GEMM (output [M][N], reduction over `rows` tokens split across workgroups): tile shape and split
// count.  Shared by the dispatcher (gemm_dispatch.inc: launch_dw) and by focal_linear_bwd_weight_workgroups, which tells a caller
// how a launch will show up in a profiler trace.
static inline void focal_dw_plan(int M, int N, long rows, int target, int* bm_out, int* bn_out, int* splits_out) {
  // 64 x 64 tiles, at least 256 reduction rows each: the wide register-staged shapes (256 x 64 ... 128 x 128) were swept in rounds 1-2 and
  // lost everywhere (profiles/r1_i_dw_tile_sweep.txt, r2_dw_variants.txt); round 4 removed them
  const int bm = 64, bn = 64;
  const long tiles = (long)((M + bm - 1) / bm) * ((N + bn - 1) / bn);
  // ~512 workgroups by default; a caller that runs several passes side by side lowers the target (the descriptor's dw_workgroups): what is left
  // on this path after the grouped launches took the Swin blocks are DeepSense's weight gradients -- tiny outputs (64 x 192 ... 768 x 512)
  // reduced over 2 560 - 51 200 rows, four passes of them on four streams -- and there the fp32 atomic epilogue is the cost: 513 workgroups of a
  // [64, 192] convolution gradient are 2.1 M atomics onto 12 k addresses.  Same-box sweep, DeepSense windows/s: 512: 113 500 / 114 200,
  // 256: 115 500 / 116 300, 192: 119 000 / 118 700, 128: 117 500 / 118 000 / 115 400 / 119 000, 96: 114 900 / 115 100, 64: 111 900 / 112 600.
  const long target_wg = target > 0 ? target : 512;
  long splits = (target_wg + tiles - 1) / tiles;
  const long min_rows = 256;  // reduction rows per workgroup, at least
  const long max_splits = (rows + min_rows - 1) / min_rows;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  *bm_out = bm; *bn_out = bn; *splits_out = (int)splits;
}

// Shapes the LDS-DMA ring weight-gradient kernel (gemm_dw_ring.hpp) takes, given bf16 operands without a loader prologue.
static inline bool focal_dw_ring_shape(int M, int N, long rows) {
  return M % 64 == 0 && N % 64 == 0 && rows % 64 == 0 && rows >= 64;
}

// Workgroups per output tile of the ring kernel when every workgroup takes two token slices (8 waves, two rings = 128 KB of LDS: ONE
// workgroup per CU, so the launch must not exceed the 256 CUs by a few workgroups -- 288 would run as two rounds).
static inline int focal_dw_ring_pairs(int M, int N, int splits) {
  const int tiles = (M / 64) * (N / 64);
  int pairs = (splits + 1) / 2;
  if (tiles <= 256 && tiles * pairs > 256) pairs = 256 / tiles;
  return pairs < 1 ? 1 : pairs;
}

// Host-side dispatch (gemm_dispatch.inc, instantiated per compute type in gemm_bf16.hip / gemm_f32.hip).
// dtype codes: FOCAL_F32 / FOCAL_BF16.  Returns a focal error code.
struct GemmSpec {
  int compute;  // compute type CT
  int a_dtype, b_dtype, c_dtype;
  bool tra, trb;
  int proA, proB, epi;
};
int focal_launch_gemm_bf16(const GemmSpec& s, const GemmParams& p, hipStream_t stream);
int focal_launch_gemm_f32(const GemmSpec& s, const GemmParams& p, hipStream_t stream);
int focal_launch_dw_tail_bf16(const DwTailGroup& g, hipStream_t stream);
int focal_launch_dw_tail_f32(const DwTailGroup& g, hipStream_t stream);
static inline int focal_launch_gemm(const GemmSpec& s, const GemmParams& p, hipStream_t stream) {
  if ((p.epi.p_elem > 0.f || p.proA.p_elem > 0.f || p.proB.p_elem > 0.f) && ((long)p.M >= (1L << 24) || (long)p.K >= (1L << 24) || (long)p.N >= (1L << 24))) {
    focal_set_error("gemm: a dimension beyond 2^24 with an element mask (MaskEval indexes elements with 24-bit multiplies)");
    return FOCAL_EUNSUPPORTED;
  }
  return s.compute == FOCAL_F32 ? focal_launch_gemm_f32(s, p, stream) : focal_launch_gemm_bf16(s, p, stream);
}
