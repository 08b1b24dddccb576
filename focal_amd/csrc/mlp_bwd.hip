// Backward of the fused Swin MLP branch at 64 channels (see mlp.hip for the forward and the reference lines).
//
// One pass over gm = dL/dx_out (already x the branch's dropout / drop-path mask, bf16) and a2 (the saved LayerNorm output)
// produces dL/da2 and all four parameter gradients; the hidden activation h and its derivative are RECOMPUTED per tile instead of
// being read back from two [M, 256] tensors by four GEMM launches.
//
// Round 4 layout -- every wave owns a HIDDEN slice, not a token slice.  Wave w of the 16 keeps hidden units 16 w .. 16 w + 15 for the
// whole kernel: its W1 rows and W2 columns are ready-made B operands in registers, and so are the accumulators of dW1^T[:, slice] and
// dW2[:, slice] (8 tiles).  Per 128-token tile and 16-token block:
//     u  = a2 W1s^T + b1s,  dh = gm W2s          D[token][hidden]: a lane holds 4 consecutive TOKENS of one hidden unit
//     h  = drop(gelu(u)),   du = dh gelu'(u) drop
// and two blocks' h / du (bf16) ARE the B operands of the weight-gradient products -- the token index moves from the accumulator's
// register slots to the contraction slots of the next MFMA (k-slot (g, e) <-> token 16 (2p + e / 4) + 4 g + e % 4, the order in which
// the transposed reads of gm^T / a2^T fetch their rows):
//     dW2[:, slice] += gm^T h,   dW1^T[:, slice] += a2^T du
// so the hidden activation makes NO LDS round trip for the weight gradients and nothing is exchanged between waves for them.  Only
// dL/da2 = du W1 contracts over all 256 hidden units: du goes to LDS as a [hidden][token] image (one 8-byte store per lane and block)
// and, behind the tile's single exchange barrier, every wave contracts two 16 x 16 output tiles over the full hidden axis.
// Two workgroup barriers per 128 tokens (the first version had ten, and two wave roles that waited for each other 48 % of the
// time: profiles/r2_mlp_pmc_split.txt); every wave runs the same instruction mix, so four of them per SIMD overlap their phases.
// The dropout mask of the hidden activation comes from the forward kernel as one bit per element (mlp.hip: [M][8] words, 32 B per
// token) -- 2 vector instructions per element instead of regenerating the xorshift stream (9, in a layout where a lane walks tokens).
//
// LDS images are XOR-swizzled so that every access pattern below is bank-conflict free (MI355X_MICROARCH.md, LDS table: ds_read_b128
// in four 16-lane groups and ds_read_b64_tr_b16 in two halves on 64 banks; stores on 32 banks); derivations next to each swizzle.
#include "gemm.hpp"
#include "mlp.hpp"

namespace focal_mlp_kernels {

constexpr int C = MLP_C, H = MLP_H, BM = 128;
constexpr int L_GM = 0;                  // [128 tok][64 c] bf16: 128-B rows, 16-B chunk ^ sw_tok(row)
constexpr int L_A2 = 16384;
constexpr int L_DU = 32768;              // du^T [256 hid][128 tok] bf16: 256-B rows, 8-B slot (4 tokens) ^ du_sw(hid)
constexpr int L_W1T = L_DU + 65536;      // W1^T [64 c][256 hid] bf16: 512-B rows, 16-B chunk ^ (c & 15)
constexpr int MB_LD = 132;               // mask words [8 columns][128 tok (+ 4 pad)]: the pad spreads the 8 columns over the banks
constexpr int L_MB = L_W1T + 32768;
constexpr int L_DB2 = L_MB + 8 * MB_LD * 4;  // db2 [64] f32: per-tile column sums of gm, added up in LDS
constexpr int L_LNX = L_DB2 + 256;       // LN: half-row sums {sum dxhat, sum dxhat xhat} [2 halves][128 tok] float2, exchanged between the two waves of a row
constexpr int L_DGB = L_LNX + 2048;      // LN: dgamma [64] | dbeta [64] f32, summed in LDS over the launch
constexpr int L_GAM = L_DGB + 512;       // LN: gamma [64] f32
constexpr int L_END = L_GAM + 256;
constexpr int LDS_BWD_BYTES = L_END;     // 132 KB: one workgroup per CU; the final flush reuses [0, 128 KB) as two fp32 images

typedef __attribute__((address_space(3))) bf16x4* lds_tr_ptr;

// Token tiles.  Direct fragment reads (ds_read_b128, lane = token l15, chunk 4 kk + g): a 16-lane group holds rows {0-3, 12-15} with one
// g and rows {4-11} with the next: sw = 2 * bits(1..2) of the row gives the first set even and the second odd chunk columns in both
// 128-byte halves of the bank row -> 16 distinct 16-byte slots.  Transposed reads (lanes (g, tq) address rows 4 g + tq, chunk
// 2 ct + tp / 2): a half-wave holds rows 0..7 (8..15), two chunks each -> rows of equal parity need four distinct even offsets: the same sw.
__device__ __forceinline__ int sw_tok(int row) { return ((row >> 1) & 3) << 1; }
// du^T image.  Stores (ds_write_b64, 16 consecutive lanes = 16 hidden rows, one slot): the slot offsets must differ modulo 16 (32 banks
// = 128 bytes) -> bits 0-3 of du_sw are a permutation of the row's low 4 bits.  Transposed reads (lanes (g, tq) address rows 8 g + tq
// (+ 4), slots 4 tb + tp): the 8 rows of a half-wave must differ in bits 2-4 of the offset (64 banks = all 32 slots of a row) -> bits
// 2-3 = tq, bit 4 = g.
__device__ __forceinline__ int du_sw(int h) { return ((h & 3) << 2) | ((h >> 2) & 3) | (((h >> 3) & 1) << 4); }

__device__ __forceinline__ bf16x8 join(bf16x4 lo, bf16x4 hi) { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7); }
__device__ __forceinline__ bf16x4 tr_read(const char* lds, int addr) { return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_tr_ptr)(lds + addr)); }
// workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt, i.e. it would expose the full HBM latency of the
// next tile's prefetch at the first barrier behind it
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// LN: the backward of the LayerNorm that produced a2 (norm2) runs in the exchange phase on the fp32 accumulators of dL/da2 -- dL/da2 is
// never stored, the separate LayerNorm-backward launch (one pass over 1 KB per token) is gone: g (the fp32 residual-stream gradient)
// += dLN, gm_next = bf16(g x next_mask) for the attention branch, dgamma / dbeta accumulated.  A token's row is split over the two
// waves that share its block (32 channels each): they exchange their two half-row sums through LDS -- one more barrier per tile.
template <bool DROP, bool LN>
__global__ __launch_bounds__(1024) void mlp_bwd_kernel(const MlpBwdParams p) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, l15 = lane & 15, tq = l15 >> 2, tp = l15 & 3;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- W1^T image (once): thread -> 8 channels of one hidden row, scattered as 2-byte stores
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int q = tid + 1024 * i, h = q >> 3, c0 = (q & 7) * 8;
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(p.w1 + h * C + c0);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = c0 + e;
      *reinterpret_cast<bf16_t*>(lds + L_W1T + c * 512 + ((((h >> 3)) ^ (c & 15)) << 4) + (h & 7) * 2) = v[e];
    }
  }
  // ---- this wave's hidden slice as B operands (k = channel 32 kk + 8 g + e, n = hidden 16 w + l15)
  const int hid = 16 * w + l15;
  bf16x8 w1f[2], w2f[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    w1f[kk] = *reinterpret_cast<const bf16x8*>(p.w1 + hid * C + 32 * kk + 8 * g);
#pragma unroll
    for (int e = 0; e < 8; ++e) w2f[kk][e] = p.w2[(32 * kk + 8 * g + e) * H + hid];
  }
  const float b1v = p.b1[hid];
  uint32_t scale_bits = 0;
  int mask_shift = 0;
  if (tid < C) reinterpret_cast<float*>(lds + L_DB2)[tid] = 0.f;
  MaskEval mk;
  if (LN) {
    mk.init(p.next_mask);  // (uniform values: into scalar registers, the vector file is full)
    mk.e.key = __builtin_amdgcn_readfirstlane(mk.e.key); mk.e.thresh = __builtin_amdgcn_readfirstlane(mk.e.thresh);
    mk.e.scale = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(mk.e.scale)));
    mk.p.key = __builtin_amdgcn_readfirstlane(mk.p.key); mk.p.thresh = __builtin_amdgcn_readfirstlane(mk.p.thresh);
    mk.p.scale = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(mk.p.scale)));
    if (tid < 2 * C) reinterpret_cast<float*>(lds + L_DGB)[tid] = 0.f;
    if (tid < C) reinterpret_cast<float*>(lds + L_GAM)[tid] = p.ln_gamma[tid];
  }
  if (DROP) {
    scale_bits = __builtin_amdgcn_readfirstlane(__float_as_uint(1.0f / (1.0f - p.drop_h.p_elem)));
    mask_shift = (w & 7) * 4 + (l15 & 3);  // the forward kernel's bit of (token, hidden 16 T + 4 G + e): word 2 G + T / 8, bit 4 (T % 8) + e
  }

  const int ntiles = (p.M + BM - 1) / BM;
  // staging: thread -> one 16-byte chunk of gm and of a2 (row tid / 8, chunk tid % 8) and one mask word (token tid / 8, column tid % 8)
  const int b_st = L_GM + (tid >> 3) * 128 + (((tid & 7) ^ sw_tok(tid >> 3)) << 4);
  const int b_stm = L_MB + ((tid & 7) * MB_LD + (tid >> 3)) * 4;
  uint4 pg, pa;
  uint32_t pm = 0;
  // (32-bit offsets from the uniform base pointers: the addresses stay in scalar registers + one vector offset; 64-bit per-thread
  // pointers cost this kernel eight vector registers it does not have -- the first build spilled them)
  const char* gm_base = reinterpret_cast<const char*>(p.gm);
  const char* a_base = reinterpret_cast<const char*>(p.a);
  const char* mb_base = reinterpret_cast<const char*>(p.mask_bits);
  const uint32_t pf_off = (uint32_t)tid * 16u;  // the tile is 1024 contiguous 16-byte chunks of gm / a2 (and 1024 mask words)
  auto prefetch = [&](int tile) {
    const bool ok = tile * BM + (tid >> 3) < p.M;
    const uint32_t off = (uint32_t)tile * (BM * C * 2) + pf_off;
    pg = ok ? *reinterpret_cast<const uint4*>(gm_base + off) : make_uint4(0, 0, 0, 0);
    pa = ok ? *reinterpret_cast<const uint4*>(a_base + off) : make_uint4(0, 0, 0, 0);
    if (DROP) pm = ok ? *reinterpret_cast<const uint32_t*>(mb_base + ((uint32_t)tile * (BM * 32) + (uint32_t)tid * 4u)) : 0u;
  };
  auto stage = [&]() {
    *reinterpret_cast<uint4*>(lds + b_st) = pg;
    *reinterpret_cast<uint4*>(lds + b_st + 16384) = pa;
    if (DROP) *reinterpret_cast<uint32_t*>(lds + b_stm) = pm;
  };
  if ((int)blockIdx.x < ntiles) prefetch(blockIdx.x);

  // (per-lane LDS addresses are derived inside the tile loop, per phase, from a laundered copy of the thread index: as loop invariants
  // they would occupy eleven registers through both phases)
  const int tb2 = w >> 1, ctb = 2 * (w & 1);                                  // exchange phase: token block and the first of two channel tiles
  f32x4 acc1[4], acc2[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc1[i] = acc2[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float db1acc = 0.f;
  __syncthreads();  // W1^T image complete

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    stage();
    lds_barrier();  // the tile is in LDS; every wave has left the previous tile's exchange phase (the du image is free)
    int t1 = tid;
    asm volatile("" : "+v"(t1));
    const int l1 = t1 & 15, g1 = (t1 >> 4) & 3, tq1 = l1 >> 2, tp1 = l1 & 3;
    // one register per image; what a loop index adds to a swizzled chunk index is an XOR constant (the swizzles only mix bits the
    // address's other terms leave free)
    const int b_dir = L_A2 + l1 * 128 + ((g1 ^ sw_tok(l1)) << 4);              // a2 fragment kk of token block tb: (b_dir ^ 64 kk) + 2048 tb; gm: - 16384
    const int b_tr = L_GM + (4 * g1 + tq1) * 128 + (((tp1 >> 1) ^ sw_tok(4 * g1 + tq1)) << 4) + (tp1 & 1) * 8;  // gm^T of channel tile ct: (b_tr ^ 32 ct) + 4096 p (+ 2048)
    const int b_duw = L_DU + (16 * w + l1) * 256, du_xs = g1 ^ du_sw(l1);      // store slot (4 tb + g1) ^ du_sw = du_xs ^ 4 tb
    const int b_mb = L_MB + ((2 * (l1 >> 2) + (w >> 3)) * MB_LD + 4 * g1) * 4;  // + 64 tb
    // ================================================================================ recompute + weight gradients, all waves alike
#pragma unroll 1
    for (int pr = 0; pr < 4; ++pr) {
      bf16x4 h4[2], d4[2];
      int dir0 = b_dir, tr0 = b_tr;
      asm volatile("" : "+v"(dir0), "+v"(tr0));  // (the XOR variants of these addresses are recomputed per pair, not kept in 5 registers across the loop)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int tb = 2 * pr + hf;
        bf16x8 xa[2], xg[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          xa[kk] = *reinterpret_cast<const bf16x8*>(lds + (dir0 ^ (64 * kk)) + tb * 2048);
          xg[kk] = *reinterpret_cast<const bf16x8*>(lds + (dir0 ^ (64 * kk)) + tb * 2048 - 16384);
        }
        f32x4 u = mma16(xa[0], w1f[0], f32x4{0.f, 0.f, 0.f, 0.f});
        u = mma16(xa[1], w1f[1], u);
        u += b1v;
        f32x4 dh = mma16(xg[0], w2f[0], f32x4{0.f, 0.f, 0.f, 0.f});
        dh = mma16(xg[1], w2f[1], dh);
        uint4 mq = make_uint4(0, 0, 0, 0);
        if (DROP) mq = *reinterpret_cast<const uint4*>(lds + b_mb + tb * 64);
        const uint32_t mw[4] = {mq.x, mq.y, mq.z, mq.w};
        bf16x4 hq, dq;
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
          gelu_f2 hh, gg;
          mlp_gelu_bwd(gelu_f2{u[e], u[e + 1]}, hh, gg);
          if (DROP) {
            // keep bit -> 0 / 1/(1-p): sign-extended one-bit field AND the multiplier's bits
            const gelu_f2 mult = gelu_f2{__uint_as_float((uint32_t)__builtin_amdgcn_sbfe(mw[e], mask_shift, 1) & scale_bits),
                                         __uint_as_float((uint32_t)__builtin_amdgcn_sbfe(mw[e + 1], mask_shift, 1) & scale_bits)};
            hh = mlp_mul2(hh, mult);
            gg = mlp_mul2(gg, mult);
          }
          const gelu_f2 dd = mlp_mul2(gelu_f2{dh[e], dh[e + 1]}, gg);
          db1acc += dd.x + dd.y;
          hq[e] = (bf16_t)hh.x; hq[e + 1] = (bf16_t)hh.y;
          dq[e] = (bf16_t)dd.x; dq[e + 1] = (bf16_t)dd.y;
        }
        *reinterpret_cast<bf16x4*>(lds + b_duw + ((du_xs ^ (4 * tb)) << 3)) = dq;
        h4[hf] = hq;
        d4[hf] = dq;
      }
      const bf16x8 h8 = join(h4[0], h4[1]), du8 = join(d4[0], d4[1]);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const int a0 = (tr0 ^ (32 * ct)) + pr * 4096;
        const bf16x8 gt = join(tr_read(lds, a0), tr_read(lds, a0 + 2048));
        const bf16x8 at = join(tr_read(lds, a0 + 16384), tr_read(lds, a0 + 16384 + 2048));
        acc2[ct] = mma16(gt, h8, acc2[ct]);
        acc1[ct] = mma16(at, du8, acc1[ct]);
      }
    }
    if (tile + (int)gridDim.x < ntiles) prefetch(tile + gridDim.x);  // lands during the exchange phase; staged at the top of the loop
    // LN: everything the exchange phase needs from memory (x_mid, the old g, the statistics of this wave's 16 tokens x 32 channels) is
    // requested here, in front of the barrier, so that it arrives while the waves wait and multiply
    float4 xv[2], go[2];
    float mean = 0.f, rstd = 0.f;
    if (LN) {
      const int m = tile * BM + 16 * tb2 + l1;
      const bool ok = m < p.M;
      const float2 stt = ok ? *reinterpret_cast<const float2*>(p.stats + 2 * (long)m) : make_float2(0.f, 0.f);
      mean = stt.x; rstd = stt.y;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const long off = (long)m * C + 16 * (ctb + j) + 4 * g1;
        xv[j] = ok ? *reinterpret_cast<const float4*>(p.x + off) : make_float4(0.f, 0.f, 0.f, 0.f);
        go[j] = ok ? *reinterpret_cast<const float4*>(p.g + off) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    lds_barrier();  // the du image of all 256 hidden units is complete
    int t2 = tid;
    asm volatile("" : "+v"(t2));
    const int l2 = t2 & 15, g2 = (t2 >> 4) & 3, tq2 = l2 >> 2, tp2 = l2 & 3;
    const int h_lo = 8 * (g2 & 1) + tq2;                                         // (row & 15) of this lane's transposed du reads (rows 32 ks + 8 g2 + tq, + 4)
    const int a_dur_lo = L_DU + (8 * g2 + tq2) * 256 + (((4 * tb2 + tp2) ^ du_sw(h_lo)) << 3);       // + 8192 ks
    const int a_dur_hi = L_DU + (8 * g2 + tq2 + 4) * 256 + (((4 * tb2 + tp2) ^ du_sw(h_lo + 4)) << 3);
    const int a_w1t = L_W1T + (16 * ctb + l2) * 512 + ((g2 ^ l2) << 4);        // chunk (4 ks + g2) ^ l2: a_w1t ^ 64 ks; second tile + 8192
    const int b_tr2 = L_GM + (4 * g2 + tq2) * 128 + (((tp2 >> 1) ^ sw_tok(4 * g2 + tq2)) << 4) + (tp2 & 1) * 8;
    // ================================================================================ dL/da2[tb2][ctb, ctb + 1] = W1^T du^T over all hidden units
    {
      f32x4 dc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      const int m = tile * BM + 16 * tb2 + l2;
      int w1t0 = a_w1t, tr0 = b_tr2;
      {  // db2[c] += sum over tokens of gm[.][c]: wave w takes block pair w / 4, channel tile w % 4 (gm^T fragment x ones), summed in LDS
        const int a0 = (tr0 ^ (32 * (w & 3))) + (w >> 2) * 4096;
        const bf16x8 gt = join(tr_read(lds, a0), tr_read(lds, a0 + 2048));
        const bf16_t one_b = (bf16_t)1.0f;
        const f32x4 cs = mma16(gt, bf16x8{one_b, one_b, one_b, one_b, one_b, one_b, one_b, one_b}, f32x4{0.f, 0.f, 0.f, 0.f});
        if (l2 == 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) atomicAdd(reinterpret_cast<float*>(lds + L_DB2) + 16 * (w & 3) + 4 * g2 + r, cs[r]);
        }
      }
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const bf16x8 duf = join(tr_read(lds, a_dur_lo + ks * 8192), tr_read(lds, a_dur_hi + ks * 8192));
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const bf16x8 wt = *reinterpret_cast<const bf16x8*>(lds + (w1t0 ^ (ks << 6)) + j * 8192);
          dc[j] = mma16(wt, duf, dc[j]);
        }
      }
      if (LN) {
        // D[c = 16 ct + 4 g2 + r][token l2]: this lane holds 8 of its token's 64 channels
        float xh[2][4], dxh[2][4], s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const f32x4 gam = *reinterpret_cast<const f32x4*>(lds + L_GAM + (16 * (ctb + j) + 4 * g2) * 4);
          const float xr[4] = {xv[j].x, xv[j].y, xv[j].z, xv[j].w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            xh[j][r] = (xr[r] - mean) * rstd;
            dxh[j][r] = dc[j][r] * gam[r];
            s1 += dxh[j][r];
            s2 += dxh[j][r] * xh[j][r];
            // dgamma / dbeta: sums over the 16 tokens of the block (the lanes of a row of 16), then one LDS add per channel and wave
            const float pg = row16_sum(dc[j][r] * xh[j][r]), pb = row16_sum(dc[j][r]);
            if (l2 == 0) {
              atomicAdd(reinterpret_cast<float*>(lds + L_DGB) + 16 * (ctb + j) + 4 * g2 + r, pg);
              atomicAdd(reinterpret_cast<float*>(lds + L_DGB) + C + 16 * (ctb + j) + 4 * g2 + r, pb);
            }
          }
        }
        s1 = xadd16(s1); s1 = xadd32(s1);
        s2 = xadd16(s2); s2 = xadd32(s2);
        if (g2 == 0) *reinterpret_cast<float2*>(lds + L_LNX + ((w & 1) * BM + 16 * tb2 + l2) * 8) = make_float2(s1, s2);
        lds_barrier();  // both halves of every row are in LDS
        const float2 oth = *reinterpret_cast<const float2*>(lds + L_LNX + (((w & 1) ^ 1) * BM + 16 * tb2 + l2) * 8);
        const float m1 = (s1 + oth.x) * (1.0f / C), m2 = (s2 + oth.y) * (1.0f / C);
        if (m < p.M) {
          const float rowm = mk.row_mult(m);
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int c0 = 16 * (ctb + j) + 4 * g2;
            const float gr[4] = {go[j].x, go[j].y, go[j].z, go[j].w};
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = gr[r] + rstd * (dxh[j][r] - m1 - xh[j][r] * m2);
            *reinterpret_cast<float4*>(p.g + (long)m * C + c0) = make_float4(o[0], o[1], o[2], o[3]);
            if (p.gm_next) {
              bf16x4 t;
#pragma unroll
              for (int r = 0; r < 4; ++r) t[r] = (bf16_t)(o[r] * rowm * mk.elem_mult(m, c0 + r));
              *reinterpret_cast<bf16x4*>(p.gm_next + (long)m * C + c0) = t;
            }
          }
        }
      } else if (m < p.M) {
        char* da_base = reinterpret_cast<char*>(p.da);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          bf16x4 o;
          o[0] = (bf16_t)dc[j][0]; o[1] = (bf16_t)dc[j][1]; o[2] = (bf16_t)dc[j][2]; o[3] = (bf16_t)dc[j][3];
          *reinterpret_cast<bf16x4*>(da_base + ((uint32_t)m * (C * 2) + (uint32_t)(16 * (ctb + j) + 4 * g2) * 2u)) = o;
        }
      }
    }
    // Without LN nothing separates this phase's gm^T reads (the db2 column sums) from the next iteration's stage(), which overwrites
    // the gm / a2 / mask images in front of the loop's first barrier: a wave that runs ahead would corrupt a slower wave's db2
    // (ADVICE r4).  With LN the row-exchange barrier above already stands between the two.
    if (!LN) lds_barrier();
  }
  __syncthreads();  // every wave is out of the loop: the images are free
  // ---- weight gradients -> two fp32 images ([64][256] dW2, [256][64] dW1), bias gradients
  float* F2 = reinterpret_cast<float*>(lds);
  float* F1 = reinterpret_cast<float*>(lds + 65536);
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
#pragma unroll
    for (int r = 0; r < 4; ++r) F2[(16 * ct + 4 * g + r) * H + hid] = acc2[ct][r];
    *reinterpret_cast<f32x4*>(F1 + hid * C + 16 * ct + 4 * g) = acc1[ct];
  }
  if (p.db1) {
    float v = db1acc;
    v = xadd16(v);
    v = xadd32(v);
    if (g == 0) atomicAdd(p.db1 + hid, v);
  }
  if (p.db2 && tid < C) atomicAdd(p.db2 + tid, reinterpret_cast<const float*>(lds + L_DB2)[tid]);
  if (LN && tid < 2 * C) atomicAdd((tid < C ? p.dgamma : p.dbeta - C) + tid, reinterpret_cast<const float*>(lds + L_DGB)[tid]);
  __syncthreads();
  // ---- 256-byte contiguous atomics of the two weight gradients, all 1024 threads
  // (33.5 MB of fp32 atomics per launch -- 256 workgroups x 128 KB -- drain at the memory side in ~30 us (1.3 TB/s); the waves end as
  // soon as they are issued, so the CUs are free for the other encoder stream's kernels meanwhile: profiles/r4_mlp_bwd.txt)
  if (p.partials != nullptr) {
    // Round 5: plain 16-byte stores of the workgroup's 128 KB into its own slot of a workspace (6 TB/s class), summed over the
    // workgroups by mlp_bwd_reduce_kernel behind this launch -- 33.5 MB of fp32 atomics drain at 1.3 TB/s at the memory side
    float4* dst = reinterpret_cast<float4*>(p.partials + (long)blockIdx.x * (2 * C * H));
    const float4* src = reinterpret_cast<const float4*>(lds);
#pragma unroll
    for (int i = 0; i < 8; ++i) dst[tid + 1024 * i] = src[tid + 1024 * i];
    return;
  }
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const int idx = tid + 1024 * i;
    atomicAdd(p.dw2 + idx, F2[idx]);
    atomicAdd(p.dw1 + idx, F1[idx]);
  }

}

// dw2 / dw1 += the sum over G workgroup images [dW2 (C x H) | dW1 (H x C)] of mlp_bwd_kernel: grid (32 column chunks of 1024 floats, 8
// slices of the images); a thread sums one float4 column over its slice, the chunk's 1024 sums cross LDS so that the eight adds per
// address leave as contiguous 256-byte atomic rows.
__global__ __launch_bounds__(256) void mlp_bwd_reduce_kernel(const float* __restrict__ partials, int G, float* __restrict__ dw2, float* __restrict__ dw1) {
  __shared__ float4 sums[256];
  const int tid = threadIdx.x, chunk = blockIdx.x, per = (G + (int)gridDim.y - 1) / (int)gridDim.y;
  const int g0 = blockIdx.y * per, g1 = min(G, g0 + per);
  const float4* src = reinterpret_cast<const float4*>(partials) + chunk * 256 + tid;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
  for (int g = g0; g < g1; ++g) {
    const float4 v = src[(long)g * (2 * C * H / 4)];
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  sums[tid] = acc;
  __syncthreads();
  if (g1 <= g0) return;
  const float* sf = reinterpret_cast<const float*>(sums);
  float* out = chunk < 16 ? dw2 + chunk * 1024 : dw1 + (chunk - 16) * 1024;
#pragma unroll
  for (int j = 0; j < 4; ++j) atomicAdd(out + j * 256 + tid, sf[j * 256 + tid]);
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
                             float* dbeta, const uint32_t* mask_bits, float* dw_partials, void* stream) {
  if (int rc = mlp_check_desc(d, "mlp_bwd")) return rc;
  FOCAL_CHECK_ARG(gm && a && w1 && b1 && w2 && dw1 && dw2, "mlp_bwd: null tensor");
  const bool ln = ln_x != nullptr;
  if (ln) FOCAL_CHECK_ARG(ln_stats && ln_gamma && g && dgamma && dbeta, "mlp_bwd: the fused LayerNorm backward needs ln_stats, ln_gamma, g, dgamma and dbeta");
  else FOCAL_CHECK_ARG(da != nullptr, "mlp_bwd: da is required without the fused LayerNorm backward");
  MlpBwdParams p;
  memset(&p, 0, sizeof(p));
  if (ln) {
    p.x = ln_x; p.stats = ln_stats; p.ln_gamma = ln_gamma; p.g = g; p.gm_next = reinterpret_cast<bf16_t*>(gm_next); p.dgamma = dgamma; p.dbeta = dbeta;
    if (next_mask) p.next_mask = mlp_bwd_mask(*next_mask, MLP_C);
  }
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
  FOCAL_CHECK_ARG(!drop || mask_bits != nullptr, "mlp_bwd: hidden dropout is on (p = %g) but mask_bits is NULL: pass the [M][8] words focal_mlp_fwd wrote", (double)d->drop_hidden.p_elem);
  p.mask_bits = mask_bits;
  p.partials = dw_partials;
  FOCAL_CHECK_ARG(dw_partials == nullptr || ((uintptr_t)dw_partials % 16) == 0, "mlp_bwd: dw_partials must be 16-byte aligned");
  void (*kern)(const MlpBwdParams) = ln ? (drop ? mlp_bwd_kernel<true, true> : mlp_bwd_kernel<false, true>)
                                        : (drop ? mlp_bwd_kernel<true, false> : mlp_bwd_kernel<false, false>);
  static std::atomic<bool> attr_set[4] = {{false}, {false}, {false}, {false}};
  const int ki = (ln ? 2 : 0) + (drop ? 1 : 0);
  if (!attr_set[ki].load(std::memory_order_acquire)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BWD_BYTES) != hipSuccess) {
      focal_set_error("mlp_bwd: cannot reserve %d bytes of LDS", LDS_BWD_BYTES);
      return FOCAL_EHIP;
    }
    attr_set[ki].store(true, std::memory_order_release);
  }
  const int ntiles = (d->M + BM - 1) / BM;
  const int grid = ntiles < 256 ? ntiles : 256;  // one persistent 16-wave workgroup per CU
  FOCAL_LAUNCH(kern, dim3(grid), dim3(1024), LDS_BWD_BYTES, (hipStream_t)stream, p);
  if (dw_partials != nullptr) FOCAL_LAUNCH(mlp_bwd_reduce_kernel, dim3(32, 8), dim3(256), 0, (hipStream_t)stream, dw_partials, grid, dw2, dw1);
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}

extern "C" long focal_mlp_bwd_partials_floats(const focal_mlp_desc* d) {
  if (mlp_check_desc(d, "mlp_bwd_partials_floats")) return 0;
  const int ntiles = (d->M + BM - 1) / BM;
  return (long)(ntiles < 256 ? ntiles : 256) * (2 * C * H);
}
