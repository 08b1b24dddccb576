// Backward of the fused Swin MLP branch at 64 channels (see mlp.hip for the forward and the reference lines).
//
// One pass over gm = dL/dx_out (already x the branch's dropout / drop-path mask, bf16) and a2 (the saved LayerNorm output)
// produces dL/da2 and all four parameter gradients; the hidden activation h and its derivative are RECOMPUTED per tile instead of
// being read back from two [M, 256] tensors by four GEMM launches:
//
//   per 128-token tile, 32 hidden units at a time, two roles of 8 waves each:
//     A waves   u = a2 W1c^T + b1c -> h = drop(gelu(u)), h' = drop(gelu'(u))        (recompute; mask regenerated from the hash)
//     (16       dh = gm W2c -> du = dh . h'                                         (accumulator layouts match: no shuffles)
//     tokens    da2 += du W1c                                                       (du re-enters the matrix core from registers)
//     each)     h, du -> LDS (bf16, [token][32])
//     B waves   dW2c += gm^T h, dW1c += du^T a2 over all 128 tokens of the PREVIOUS chunk: every B wave owns 2 of a chunk's 16 output
//               tiles and keeps its 16 accumulators (64 VGPRs) for the WHOLE kernel -- the weight gradient leaves the CU once, as
//               256-byte contiguous fp32 atomics staged through LDS.
//   (double-buffered h / du tiles, one barrier per chunk).
//
// Operands that are contracted over their row index (W2 for dh, W1 for da2, gm / a2 / h / du for the weight gradients) are read
// with ds_read_b64_tr_b16; all LDS images are XOR-swizzled so that both their direct and their transposed fragment reads are
// bank-conflict free (derivations next to each swizzle function).
#include "gemm.hpp"
#include "mlp.hpp"

namespace focal_mlp_kernels {

constexpr int C = MLP_C, H = MLP_H, BM = 128;
// LDS map (bytes).  Every image is laid out so that (a) its fragment reads -- direct ds_read_b128 and / or transposed
// ds_read_b64_tr_b16 -- are bank-conflict free and (b) an address is (per-lane base) + (compile-time constant): the swizzles only mix
// LANE-dependent bits, loop indices select sub-images.  (The first version XOR-ed loop constants into the chunk index: half of the
// kernel's 4 700 vector instructions per tile were address arithmetic, and the kernel ran VALU-bound at 1/13 of the MFMA rate.)
constexpr int L_W1 = 0;                  // 2 half images (c < 32 | c >= 32) of [256 h][32 c] bf16: 64-B rows, chunk ^ P[(h >> 2) & 3]
constexpr int L_W2 = 32768;              // 16 sub-images (16 hidden units each) of [64 c][16 h] bf16: 32-B rows at row' = c ^ ((c >> 3 & 1) << 2)
constexpr int L_GM = 65536;              // [128 m][64 c] bf16: 128-B rows, chunk ^ sw_tile(m)
constexpr int L_A2 = L_GM + 16384;
constexpr int L_HB = L_A2 + 16384;       // 2 buffers x { h, du } x 2 sub-tiles (16 hidden units) of [128 m][16] bf16: 32-B rows at row'(m)
constexpr int L_B1 = L_HB + 32768;       // [256] f32
constexpr int L_END = L_B1 + 1024;
constexpr int LDS_BWD_BYTES = L_END;     // 129 KB: one workgroup per CU

typedef __attribute__((address_space(3))) bf16x4* lds_tr_ptr;

// chunk swizzle of the W1 half images: conflict-free for the direct reads of 16 consecutive rows (ds_read_b128 serves lanes
// {0-3, 12-15, 20-27} etc. together: the four 4-row blocks of a tile must land on different 16-byte columns for both chunk parities)
// and for the transposed reads over rows 32 q + 4 g + {0..3}
__device__ __forceinline__ int sw_p(int blk) { return (0x1230 >> (4 * (blk & 3))) & 3; }  // {0, 3, 2, 1}
// token tiles (gm, a2): direct reads of 16 consecutive rows, transposed reads over rows 32 ks + 8 g + {0..3} (+4): bits 1 and 3 of the row
__device__ __forceinline__ int sw_tile(int row) { return (((row >> 1) & 1) << 1) | (((row >> 3) & 1) << 2); }
// 32-byte-row images: rows r and r + 8 would share banks; swapping the two 4-row halves of every second 8-row block separates them
__device__ __forceinline__ int rowp(int r) { return r ^ (((r >> 3) & 1) << 2); }

__device__ __forceinline__ int a_w1(int h, int c) { return L_W1 + (c >> 5) * 16384 + h * 64 + (((((c & 31) >> 3)) ^ sw_p(h >> 2)) << 4) + (c & 7) * 2; }
__device__ __forceinline__ int a_w2(int c, int h) { return L_W2 + (h >> 4) * 2048 + rowp(c) * 32 + (h & 15) * 2; }

// (an element-wise bf16x8{lo[0], .., hi[3]} makes hipcc unpack and re-pack every 16-bit element: ~700 vector instructions per tile)
__device__ __forceinline__ bf16x8 join(bf16x4 lo, bf16x4 hi) { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7); }
__device__ __forceinline__ bf16x4 tr_read(const char* lds, int addr) { return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_tr_ptr)(lds + addr)); }
// workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt, i.e. it would expose the full HBM latency of the
// next tile's prefetch at the first barrier behind it
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ float sum8(bf16x8 v) {
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) s += (float)v[e];
  return s;
}

// The two halves of a barrier interval run on DIFFERENT waves: waves 0-7 ("A") recompute h / h' and produce du and dL/da2 for their
// 16 tokens, waves 8-15 ("B") contract the previous chunk's h / du sub-tiles into the weight gradients.  The first version ran A then
// B in every one of 8 waves and was latency-bound (39 % of its wave-cycles waiting, vector ALU 31 % / matrix pipe 9.5 % busy:
// profiles/r2_mlp_pmc.txt); with the roles on separate waves an interval lasts max(A, B) instead of A + B and every SIMD holds four
// waves instead of two: 186 -> 165 us at the stage-0 audio shape, +0.9 % on the step (same-box A/B, profiles/r2_mlp_ab.txt (3)).  Each
// role is its own loop -- its own register budget: the B waves' 64 accumulator registers and the A waves' GELU temporaries never live
// in the same wave (128 VGPRs, no spills) -- and both execute the same ten workgroup barriers per tile.
template <bool DROP>
__global__ __launch_bounds__(1024, 4) void mlp_bwd_kernel(const MlpBwdParams p) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, l15 = lane & 15, tq = l15 >> 2, tp = l15 & 3;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool role_a = wave < 8;
  const int wa = wave & 7;
  const bool c_side_first = wa < 4;   // B waves 8-11 own dW2 tiles (c-tile cw, all hidden tiles), 12-15 dW1 tiles (all hidden tiles, c-tile cw)
  const int cw = wa & 3;

  // ---- weights -> LDS
  {
    uint4 v1[2], v2[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      v1[i] = reinterpret_cast<const uint4*>(p.w1)[tid + 1024 * i];
      v2[i] = reinterpret_cast<const uint4*>(p.w2)[tid + 1024 * i];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q = tid + 1024 * i;
      *reinterpret_cast<uint4*>(lds + a_w1(q >> 3, (q & 7) * 8)) = v1[i];
      *reinterpret_cast<uint4*>(lds + a_w2(q >> 5, (q & 31) * 8)) = v2[i];
    }
    if (tid < H) reinterpret_cast<float*>(lds + L_B1)[tid] = p.b1[tid];
  }
  const int ntiles = (p.M + BM - 1) / BM;
  // every thread stages one 16-byte chunk of gm and one of a2 per tile: chunk id = tid -> row tid >> 3, column chunk tid & 7
  const int b_st = L_GM + (tid >> 3) * 128 + (((tid & 7) ^ sw_tile(tid >> 3)) << 4);
  uint4 pg, pa;
  auto prefetch = [&](int tile) {
    const long m = (long)tile * BM + (tid >> 3);
    const bool ok = m < p.M;
    pg = ok ? *reinterpret_cast<const uint4*>(p.gm + m * C + (tid & 7) * 8) : make_uint4(0, 0, 0, 0);
    pa = ok ? *reinterpret_cast<const uint4*>(p.a + m * C + (tid & 7) * 8) : make_uint4(0, 0, 0, 0);
  };
  auto stage = [&]() {
    *reinterpret_cast<uint4*>(lds + b_st) = pg;
    *reinterpret_cast<uint4*>(lds + b_st + 16384) = pa;
  };
  if ((int)blockIdx.x < ntiles) prefetch(blockIdx.x);
  const int rp_lo = rowp(8 * g + tq) * 32 + tp * 8, rp_hi = rowp(8 * g + tq + 4) * 32 + tp * 8;

  if (role_a) {
    // ================================================================================================= A: recompute, du, dL/da2
    MlpDropStream ds;
    ds.init(p.drop_h);
    const uint32_t ds_key = ds.s;
    const int mloc = wa * 16 + l15;
    const int b_w1d = L_W1 + l15 * 64 + ((g ^ sw_p(l15 >> 2)) << 4);
    int b_w1t[2];
#pragma unroll
    for (int jb = 0; jb < 2; ++jb) b_w1t[jb] = L_W1 + (4 * g + tq) * 64 + ((((2 * jb + (tp >> 1))) ^ sw_p(g)) << 4) + (tp & 1) * 8;
    const int b_w2_lo = L_W2 + rp_lo, b_w2_hi = L_W2 + rp_hi;
    const int b_hw = L_HB + rowp(mloc) * 32 + 8 * g;
    int b_td[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) b_td[kk] = L_GM + mloc * 128 + (((4 * kk + g) ^ sw_tile(mloc)) << 4);
    const int b_b1 = L_B1 + 16 * g;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
      stage();
      lds_barrier();
      if (tile + (int)gridDim.x < ntiles) prefetch(tile + gridDim.x);
      const int m = tile * BM + mloc;
      uint32_t dst = DROP ? ds.start(ds_key, m, g) : 0u;
      bf16x8 xa[2], xg[2];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        xg[kk] = *reinterpret_cast<const bf16x8*>(lds + b_td[kk]);
        xa[kk] = *reinterpret_cast<const bf16x8*>(lds + b_td[kk] + 16384);
      }
      f32x4 dc[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) dc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      auto phase_a = [&](int q) {  // hidden units 32 q .. 32 q + 31: u, dh -> h, h', du -> LDS; da2 += du W1c
        const int buf = (q & 1) * 16384;
        bf16x4 dqs[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int ht = 2 * q + t;
          f32x4 u = *reinterpret_cast<const f32x4*>(lds + b_b1 + ht * 64), dh = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
            const bf16x8 w1f = *reinterpret_cast<const bf16x8*>(lds + b_w1d + kk * 16384 + ht * 1024);
            u = mma16(w1f, xa[kk], u);
            const bf16x8 w2f = join(tr_read(lds, b_w2_lo + ht * 2048 + kk * 1024), tr_read(lds, b_w2_hi + ht * 2048 + kk * 1024));
            dh = mma16(w2f, xg[kk], dh);
          }
          bf16x4 hq, dq;
#pragma unroll
          for (int e = 0; e < 4; e += 2) {
            gelu_f2 hh, gg;
            mlp_gelu_bwd(gelu_f2{u[e], u[e + 1]}, hh, gg);
            if (DROP) {
              const gelu_f2 mult = ds.next(dst);
              hh = mlp_mul2(hh, mult);
              gg = mlp_mul2(gg, mult);
            }
            const gelu_f2 dd = mlp_mul2(gelu_f2{dh[e], dh[e + 1]}, gg);
            hq[e] = (bf16_t)hh.x; hq[e + 1] = (bf16_t)hh.y;
            dq[e] = (bf16_t)dd.x; dq[e + 1] = (bf16_t)dd.y;
          }
          *reinterpret_cast<bf16x4*>(lds + b_hw + buf + t * 4096) = hq;
          *reinterpret_cast<bf16x4*>(lds + b_hw + buf + 8192 + t * 4096) = dq;
          dqs[t] = dq;
        }
        const bf16x8 duf = join(dqs[0], dqs[1]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int a0 = b_w1t[j & 1] + (j >> 1) * 16384 + q * 2048;
          const bf16x8 w1t = join(tr_read(lds, a0), tr_read(lds, a0 + 1024));
          dc[j] = mma16(w1t, duf, dc[j]);
        }
      };
      phase_a(0);
      lds_barrier();
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (q < 7) phase_a(q + 1);
        lds_barrier();
      }
      if (m < p.M) {
#pragma unroll
        for (int j = 0; j < 4; ++j) store4(p.da + (long)m * C + 16 * j + 4 * g, dc[j]);
      }
    }
    __syncthreads();  // B waves have put their tiles into LDS
  } else {
    // ================================================================================================= B: weight / bias gradients
    const int hsel = c_side_first ? 0 : 8192;
    const int b_hb_lo = L_HB + hsel + rp_lo, b_hb_hi = L_HB + hsel + rp_hi;
    const int b_tr = (c_side_first ? L_GM : L_A2) + (8 * g + tq) * 128 + ((((2 * cw + (tp >> 1))) ^ sw_tile(8 * g + tq)) << 4) + (tp & 1) * 8;
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dbh[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) dbh[i] = 0.f;
    float dbc = 0.f;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
      stage();
      lds_barrier();
      if (tile + (int)gridDim.x < ntiles) prefetch(tile + gridDim.x);
      lds_barrier();  // chunk 0 of this tile is in LDS
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int buf = (q & 1) * 16384;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const bf16x8 fc = join(tr_read(lds, b_tr + ks * 4096), tr_read(lds, b_tr + ks * 4096 + 512));
          if (c_side_first && q == 0) dbc += sum8(fc);  // db2: column sums of gm, once per tile
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const bf16x8 fh = join(tr_read(lds, b_hb_lo + buf + t * 4096 + ks * 1024), tr_read(lds, b_hb_hi + buf + t * 4096 + ks * 1024));
            acc[2 * q + t] = mma16(fc, fh, acc[2 * q + t]);
            if (!c_side_first && ks == cw) dbh[2 * q + t] += sum8(fh);  // db1: wave 12 + cw takes the 32 tokens of k-step cw
          }
        }
        lds_barrier();
      }
    }
    // ---- tiles -> LDS ([64][256] dW2 then [256][64] dW1, fp32)
    float* F2 = reinterpret_cast<float*>(lds);
    float* F1 = reinterpret_cast<float*>(lds + 65536);
#pragma unroll
    for (int ht = 0; ht < 16; ++ht) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (c_side_first) F2[(16 * cw + 4 * g + r) * H + 16 * ht + l15] = acc[ht][r];
        else F1[(16 * ht + l15) * C + 16 * cw + 4 * g + r] = acc[ht][r];
      }
    }
    if (c_side_first) {
      dbc += __shfl_xor(dbc, 16, 64);
      dbc += __shfl_xor(dbc, 32, 64);
      if (g == 0 && p.db2) atomicAdd(p.db2 + 16 * cw + l15, dbc);
    } else if (p.db1) {
#pragma unroll
      for (int ht = 0; ht < 16; ++ht) {
        float v = dbh[ht];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (g == 0) atomicAdd(p.db1 + 16 * ht + l15, v);
      }
    }
    __syncthreads();
  }
  // ---- 256-byte contiguous atomics of the two weight gradients, all 1024 threads
  const float* F2 = reinterpret_cast<const float*>(lds);
  const float* F1 = reinterpret_cast<const float*>(lds + 65536);
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const int idx = tid + 1024 * i;
    atomicAdd(p.dw2 + idx, F2[idx]);
    atomicAdd(p.dw1 + idx, F1[idx]);
  }
}

}  // namespace focal_mlp_kernels
using namespace focal_mlp_kernels;

static MaskParams mlp_bwd_mask(const focal_drop_desc& d, int ncols) {
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

extern "C" int focal_mlp_bwd(const focal_mlp_desc* d, const void* gm, const void* a, const void* w1, const float* b1, const void* w2,
                             void* da, float* dw1, float* db1, float* dw2, float* db2, const float* ln_x, const float* ln_stats,
                             const float* ln_gamma, float* g, void* gm_next, const focal_drop_desc* next_mask, float* dgamma,
                             float* dbeta, void* stream) {
  if (int rc = mlp_check_desc(d, "mlp_bwd")) return rc;
  FOCAL_CHECK_ARG(gm && a && w1 && b1 && w2 && dw1 && dw2, "mlp_bwd: null tensor");
  if (ln_x != nullptr) {
    focal_set_error("mlp_bwd: the fused LayerNorm backward is not built in this version (pass ln_x = NULL and run focal_layernorm_bwd)");
    return FOCAL_EUNSUPPORTED;
  }
  FOCAL_CHECK_ARG(da != nullptr, "mlp_bwd: da is required without the fused LayerNorm backward");
  (void)ln_stats; (void)ln_gamma; (void)g; (void)gm_next; (void)next_mask; (void)dgamma; (void)dbeta;
  MlpBwdParams p;
  memset(&p, 0, sizeof(p));
  p.M = d->M;
  p.gm = reinterpret_cast<const bf16_t*>(gm);
  p.a = reinterpret_cast<const bf16_t*>(a);
  p.w1 = reinterpret_cast<const bf16_t*>(w1);
  p.b1 = b1;
  p.w2 = reinterpret_cast<const bf16_t*>(w2);
  p.da = reinterpret_cast<bf16_t*>(da);
  p.dw1 = dw1; p.db1 = db1; p.dw2 = dw2; p.db2 = db2;
  p.drop_h = mlp_bwd_mask(d->drop_hidden, MLP_H);
  const bool drop = d->drop_hidden.p_elem > 0.f;
  void (*kern)(const MlpBwdParams) = drop ? mlp_bwd_kernel<true> : mlp_bwd_kernel<false>;
  static bool attr_set[2] = {false, false};
  if (!attr_set[drop]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BWD_BYTES) != hipSuccess) {
      focal_set_error("mlp_bwd: cannot reserve %d bytes of LDS", LDS_BWD_BYTES);
      return FOCAL_EHIP;
    }
    attr_set[drop] = true;
  }
  const int ntiles = (d->M + BM - 1) / BM;
  const int grid = ntiles < 256 ? ntiles : 256;  // one persistent 16-wave workgroup per CU
  FOCAL_LAUNCH(kern, dim3(grid), dim3(1024), LDS_BWD_BYTES, (hipStream_t)stream, p);
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}
