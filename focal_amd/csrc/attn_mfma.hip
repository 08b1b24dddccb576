// bf16 window attention on the matrix cores: one wave per (window, head), 16x16x16 MFMA tiles (9 tokens padded to 16).
// Every contraction is arranged so that a product's accumulator (lane = column, 4 consecutive rows per lane) is
// DIRECTLY the B operand of the next product (v_mfma_f32_16x16x16_bf16 takes k = 4*(lane>>4)+e, the same map), and the
// operands contracted over the token index come out of the [token][d] LDS tiles with the hardware transpose read
// (ds_read_b64_tr_b16).  Both orientations of the 16x16 score tile are computed (2 tiny MFMAs) so that no tile is ever
// transposed through LDS:
//   S   = Q K^T  (lane col j, rows i)  -> feeds dK^T = Q^T dS and dV^T = dO^T P
//   S^T = K Q^T  (lane col i, rows j)  -> feeds O^T = V^T P^T and dQ^T = K^T dS^T
// The gather (roll + window partition) and scatter are index arithmetic on the original token order, as in attn.hip.
#include "attn_geom.hpp"

typedef __attribute__((address_space(3))) bf16x4* lds_b4;

// All-reduce over a lane's four row groups (lanes l, l ^ 16, l ^ 32, l ^ 48) without the LDS pipeline (common.hpp: xadd16 / xadd32, the gfx950
// row / half swaps): six ds_bpermute round trips sat on the backward item's dependency chain.  Same pairing and operand order as
// `x op= shfl_xor(x, 16); x op= shfl_xor(x, 32)`: bit-identical.
__device__ __forceinline__ float rows4_sum(float v) { return xadd32(xadd16(v)); }
__device__ __forceinline__ float rows4_max(float v) { return xmax32(xmax16(v)); }
// (Scores in the base-2 domain -- exp2 without the multiply per element -- were tried: 3 instructions less, and the bf16 rank-loss term of
// the reference fixture moved from 0.98e-2 to 1.12e-2 of its bound-defining value; the natural-base form is kept bit for bit.)

// (32-bit byte offsets off the scalar base pointers: global_load v, v_off, s[base] -- a hoisted 64-bit per-lane pointer costs two registers each)
template <class T> __device__ __forceinline__ const T& br_ld(const void* base, uint32_t byte_off) { return *reinterpret_cast<const T*>(static_cast<const char*>(base) + byte_off); }
template <class T> __device__ __forceinline__ T& br_st(void* base, uint32_t byte_off) { return *reinterpret_cast<T*>(static_cast<char*>(base) + byte_off); }
// byte offset of a token's row: a 24-bit multiply (full rate; v_mul_lo_u32 and the 64-bit forms issue at quarter rate).  The launchers check
// tokens < 2^24 and tensor bytes < 2^32.
__device__ __forceinline__ uint32_t att_row(int tok, int row_bytes) { return __umul24((uint32_t)tok, (uint32_t)row_bytes); }


__device__ __forceinline__ f32x4 mma16x16(bf16x4 a, bf16x4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0); }
// Four fp32 -> bf16x4 as TWO v_cvt_pk_bf16_f32 (element-wise casts compile to four single conversions + two byte permutes: with ~9 such packs
// per item that was a fifth of the backward kernel's vector instructions).  pack4z: the same, zero unless `keep` -- the select runs on the two
// packed words, not on the four floats.
typedef float att_f2 __attribute__((ext_vector_type(2)));
typedef __bf16 att_b2 __attribute__((ext_vector_type(2)));
typedef uint32_t att_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16x4 pack4(const float* v) {
  const att_b2 lo = __builtin_convertvector(att_f2{v[0], v[1]}, att_b2), hi = __builtin_convertvector(att_f2{v[2], v[3]}, att_b2);
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}
__device__ __forceinline__ bf16x4 pack4z(const float* v, bool keep) {
  att_u2 w = __builtin_bit_cast(att_u2, pack4(v));
  w[0] = keep ? w[0] : 0u;
  w[1] = keep ? w[1] : 0u;
  return __builtin_bit_cast(bf16x4, w);
}
// direct fragment: rows = tokens; lane (row = l&15) holds d = kk*16 + 4*(l>>4) .. +3
__device__ __forceinline__ bf16x4 frag_rows(const bf16_t* tile, int P, int kk, int lane) {
  return *reinterpret_cast<const bf16x4*>(tile + (lane & 15) * P + kk * 16 + 4 * (lane >> 4));
}
// transposed fragment: lane (row = d = db*16 + (l&15)) holds tokens 4*(l>>4) .. +3
__device__ __forceinline__ bf16x4 frag_cols(const bf16_t* tile, int P, int db, int lane) {
  const int q = (lane & 15) >> 2, p = lane & 3;
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4)(tile + (4 * (lane >> 4) + q) * P + db * 16 + 4 * p));
}

// tok_own: this lane's token index for window slot (lane & 15); other slots come by wave shuffle
template <int HD>
__device__ __forceinline__ void load_tile(bf16_t* tile, const bf16_t* base, long row_stride, int tok_own, int N, int lane) {
  constexpr int P = HD + 4, CPR = HD / 4;
#pragma unroll
  for (int c = lane; c < 16 * CPR; c += 64) {
    const int t = c / CPR, dc = c % CPR;
    const int tok = __shfl(tok_own, t, 64);
    bf16x4 v = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
    if (t < N) v = br_ld<bf16x4>(base, att_row(tok, (int)row_stride * 2) + dc * 8);
    *reinterpret_cast<bf16x4*>(tile + t * P + dc * 4) = v;
  }
}

// Two-phase form: all of an item's tile loads are issued first (into registers), LDS is written later.  `load_tile` above
// compiles to load -> s_waitcnt vmcnt(0) -> ds_write per tile: four exposed memory latencies per (window, head) in the
// backward kernel, which was its whole duration (32 items per wave x ~5 us at stage 0).  With the register phase the NEXT
// item's tiles are requested before the current item is multiplied, so the latency is hidden behind ~500 instructions.
template <int HD> struct TileRegs { bf16x4 v[HD / 16]; };
template <int HD>
__device__ __forceinline__ void tile_fetch(TileRegs<HD>& r, const bf16_t* base, long row_stride, int tok_own, int N, int lane) {
  constexpr int CPR = HD / 4;
#pragma unroll
  for (int q = 0; q < HD / 16; ++q) {
    const int c = lane + 64 * q, t = c / CPR, dc = c % CPR;
    const int tok = __shfl(tok_own, t, 64);
    r.v[q] = bf16x4{(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
    if (t < N) r.v[q] = br_ld<bf16x4>(base, att_row(tok, (int)row_stride * 2) + dc * 8);
  }
}
template <int HD> __device__ __forceinline__ void tile_commit(bf16_t* tile, const TileRegs<HD>& r, int lane) {
  constexpr int P = HD + 4, CPR = HD / 4;
#pragma unroll
  for (int q = 0; q < HD / 16; ++q) {
    const int c = lane + 64 * q, t = c / CPR, dc = c % CPR;
    *reinterpret_cast<bf16x4*>(tile + t * P + dc * 4) = r.v[q];
  }
}

// The LDS tiles are private to one wave and DS operations of a wave execute in issue order, so no workgroup barrier
// is needed between filling a tile and reading fragments from it; this only stops the compiler from reordering.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Per-lane constants of a 16x16 score tile: for each of the lane's 4 accumulator rows, the relative-position index
// and validity of its (query i, key j).  They depend only on the lane, so they are computed ONCE per kernel: the
// item loop then contains no integer division (a runtime div/mod costs ~40 VALU instructions on CDNA).
struct TileIdx {
  int rel[4];   // relative_position_index(i, j) * heads
  int qi[4], kj[4];
  bool ok[4];   // i < N && j < N
  bool qpad[4]; // i >= N && j < N (padded query row: keep finite)
};
template <bool ROWS_ARE_KEYS> __device__ __forceinline__ TileIdx make_tile_idx(const AttnGeom& g, int lane) {
  TileIdx t;
  const int grp = lane >> 4, col = lane & 15;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * grp + r;
    const int i = ROWS_ARE_KEYS ? col : row, j = ROWS_ARE_KEYS ? row : col;
    t.qi[r] = i; t.kj[r] = j;
    t.ok[r] = i < g.N && j < g.N;
    t.qpad[r] = i >= g.N && j < g.N;
    t.rel[r] = ((i / g.ww - j / g.ww + g.wh - 1) * (2 * g.ww - 1) + (i % g.ww - j % g.ww + g.ww - 1)) * g.heads;
  }
  return t;
}

// softmax of a score tile.  ROWS_ARE_KEYS: lane holds keys j = 4g + r of query i = l & 15 (S^T layout), otherwise
// lane holds queries i = 4g + r and key j = l & 15 (S layout).  Returns probabilities in s[].
template <bool ROWS_ARE_KEYS>
__device__ __forceinline__ void tile_softmax(const AttnGeom& g, const TileIdx& t, float* s, const float* bias, int reg_own, bool edge) {
  // bias[r] = relative-position bias of this lane's (i, j) for the current head: it depends on the lane and the head only, so the
  // callers reload it when the head changes instead of fetching it from the table for every window
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    bool cross = false;
    if (edge) cross = __shfl(reg_own, t.qi[r], 64) != __shfl(reg_own, t.kj[r], 64);  // (wave-uniform branch: shuffles stay convergent)
    float v = -1.0e30f;
    if (t.ok[r]) {
      v = s[r] * g.scale + bias[r];
      if (cross) v += -100.0f;  // SwinModules.py:287
    } else if (t.qpad[r]) {
      v = 0.f;
    }
    s[r] = v;
  }
  if (ROWS_ARE_KEYS) {
    float m = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
    m = rows4_max(m);
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) { s[r] = __expf(s[r] - m); sum += s[r]; }
    sum = rows4_sum(sum);
    const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
    for (int r = 0; r < 4; ++r) s[r] *= inv;
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float m = row16_max(s[r]);
      const float e = __expf(s[r] - m);
      s[r] = e * __builtin_amdgcn_rcpf(row16_sum(e));
    }
  }
}


// ---------------------------------------------------------------------------------------------- q / k / v on the fly (round 4)
// At 64 channels (Swin stage 0: head_dim 16, 4 heads) the attention kernels can make their own q, k, v: a wave's head is fixed
// (item = workgroup slot x waves + wave, waves % heads == 0), so the 48 rows of Wqkv it needs are six loop-invariant A fragments, and an
// item's operands are the window's 9 LayerNorm-output rows -- two 16-byte loads per lane instead of three tile fetches.
//     T^T[d][slot] = W_T[h 16 + d][:] . a1[token(slot)][:] + b_T[h 16 + d]          (T = q, k, v; 2 MFMAs of 16 x 16 x 32 each)
// lands as 4 consecutive d of one token per lane: exactly the [token][d] tile the item loop reads.  The [M, 3C] qkv tensor then never
// exists in HBM: the forward pass saves a1 (the weight gradient needs it anyway) and the backward pass recomputes -- 1 KB of traffic per
// token and step, and one GEMM launch per block, less.  Padded slots are written as zeros, as the tile fetch did.
struct QkvFrags {
  bf16x8 w[3][2];     // A fragments: row = this lane's d (l & 15) of q / k / v for the wave's head, k = channels 32 kk + 8 (l >> 4) ..
  const float* bias;  // LDS copy of bqkv [3C] + this lane's offset h 16 + 4 (l >> 4): 12 registers less than keeping the three bias quads
};
__device__ __forceinline__ QkvFrags qkv_frags(const bf16_t* wqkv, const float* bias_lds, int C, int h, int lane) {
  QkvFrags f;
#pragma unroll
  for (int t = 0; t < 3; ++t) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) f.w[t][kk] = *reinterpret_cast<const bf16x8*>(wqkv + (long)(t * C + h * 16 + (lane & 15)) * C + 32 * kk + 8 * (lane >> 4));
  }
  f.bias = bias_lds + h * 16 + 4 * (lane >> 4);
  return f;
}
struct RowRegs { bf16x8 v[2]; };  // this lane's slot: channels 8 (l >> 4) .. + 7 and 32 + 8 (l >> 4) .. of its token's a1 row
// (a padded slot reads the row of token 0 -- `tok_own` is 0 there -- instead of being zero-filled under a branch: what is projected from it is
// either zeroed behind the product (backward: ZERO_PAD, the dO tile) or never used (forward); 16 moves and a divergent region less per item)
__device__ __forceinline__ void row_fetch(RowRegs& r, const bf16_t* a1, int C, int tok_own, bool /*valid*/, int lane) {
  const uint32_t off = att_row(tok_own, C * 2) + 16u * (lane >> 4);
  r.v[0] = br_ld<bf16x8>(a1, off);
  r.v[1] = br_ld<bf16x8>(a1, off + 64u);
}
// ZERO_PAD: rows of padded window slots are written as zeros (the backward kernel needs them: a padded query's softmax row meets Q / dO rows).
// The forward kernel does not: padded keys are masked by the additive -1e30, padded queries are never stored, and what stands in
// those rows -- the projection's bias, the input row is zero -- is finite.
template <int P, bool ZERO_PAD = true> __device__ __forceinline__ void qkv_project(bf16_t* Qt, bf16_t* Kt, bf16_t* Vt, const QkvFrags& f, const RowRegs& x, bool valid, int lane) {
  bf16_t* dst[3] = {Qt, Kt, Vt};
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    // (products first, bias last: the order of the GEMM epilogue this replaces, so that q / k / v round to the same bf16 values)
    f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.w[t][0], x.v[0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.w[t][1], x.v[1], acc, 0, 0, 0);
    acc += *reinterpret_cast<const f32x4*>(f.bias + t * 64);
    const float o[4] = {acc[0], acc[1], acc[2], acc[3]};
    *reinterpret_cast<bf16x4*>(dst[t] + (lane & 15) * P + 4 * (lane >> 4)) = ZERO_PAD ? pack4z(o, valid) : pack4(o);
  }
}

// Token index (and, in a shifted block, mask region) of this lane's window slot.  The window decomposes on the scalar unit (`win` is wave-
// uniform); what is left per lane is one add -- plus, shifted, four compares for the wrap-around of torch.roll and four for the region --
// against scalars.  No per-lane multiply (a v_mul_lo_u32 issues at quarter rate) and no divergent region: a padded slot (slot >= N) yields
// token 0 through a select.
struct SlotLane { int sy, sx, lin; bool valid; };  // slot coordinates inside the window, sy * W + sx
__device__ __forceinline__ SlotLane slot_lane(const AttnGeom& g, int slot) {
  SlotLane L;
  L.sy = slot / g.ww; L.sx = slot - L.sy * g.ww;
  L.lin = L.sy * g.W + L.sx;
  L.valid = slot < g.N;
  return L;
}
__device__ __forceinline__ int slot_token(const AttnGeom& g, int win, const SlotLane& L, int* region) {
  const int b = (int)att_div((uint32_t)win, g.m_nW, (uint32_t)g.nW), wl = win - b * g.nW;
  const int wy = (int)att_div((uint32_t)wl, g.m_nWx, (uint32_t)g.nWx), wx = wl - wy * g.nWx;
  const int Y0 = wy * g.wh, X0 = wx * g.ww;  // (scalar) the window's corner in the rolled frame
  int tok = (b * g.H + Y0) * g.W + X0 + L.lin, reg = 0;
  if (g.shifted) {  // rolled[Y] = original[(Y + sh) mod H]   (torch.roll by -shift, SwinModules.py:307); regions: SwinModules.py:276-287
    reg = 3 * ((L.sy >= g.H - g.wh - Y0) + (L.sy >= g.H - g.sh - Y0)) + (L.sx >= g.W - g.ww - X0) + (L.sx >= g.W - g.sw - X0);
    tok += g.sh * g.W + g.sw;
    tok -= (L.sy >= g.H - g.sh - Y0) ? g.H * g.W : 0;
    tok -= (L.sx >= g.W - g.sw - X0) ? g.W : 0;
  }
  *region = reg;
  return L.valid ? tok : 0;
}

template <int HD, bool FUSE>
__global__ __launch_bounds__(256) void window_attn_fwd_mfma_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ bias_table,
                                                                   bf16_t* __restrict__ out, AttnGeom g, int total_items, int iters,
                                                                   const uint32_t* rng, uint32_t stream, float p_attn,
                                                                   const bf16_t* __restrict__ wqkv, const float* __restrict__ bqkv) {
  // FUSE: `qkv` is the LayerNorm output a1 [M][C] and q / k / v of an item are projected here (HD == 16, C == 64: see qkv_project)
  constexpr int P = HD + 4, TILE = 16 * P;
  __shared__ __attribute__((aligned(16))) bf16_t tiles[4][3][TILE];
  __shared__ __attribute__((aligned(16))) float qkv_bias[FUSE ? 192 : 4];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // scalar: item, window and head arithmetic then runs on the SALU
  if constexpr (FUSE) {
    if (threadIdx.x < 192) qkv_bias[threadIdx.x] = bqkv[threadIdx.x];
    __syncthreads();
  }
  bf16_t* Qt = tiles[wave][0];
  bf16_t* Kt = tiles[wave][1];
  bf16_t* Vt = tiles[wave][2];
  const DropCtx dc = make_drop(rng, stream, p_attn);
  const bool drop_on = p_attn > 0.f;
  const int C = g.C;
  const TileIdx tA = make_tile_idx<true>(g, lane);
  const int slot = lane & 15;
  const SlotLane SL = slot_lane(g, slot);
  float biasA[4] = {0.f, 0.f, 0.f, 0.f};
  int h_cur = -1;
  // item state (uniform per wave) + this lane's token; the tiles of item it + 1 are in flight while item it is multiplied
  auto item_of = [&](int it, bool& live, int& win, int& h) {
    const int item = (it * gridDim.x + blockIdx.x) * 4 + wave;
    live = item < total_items;
    win = live ? (int)att_div((uint32_t)item, g.m_heads, (uint32_t)g.heads) : 0;
    h = live ? item - win * g.heads : 0;
  };
  TileRegs<HD> rq, rk, rv;
  RowRegs rx;
  QkvFrags qf;
  bool live_n; int win_n, h_n, reg_n = 0, tok_n = 0;
  item_of(0, live_n, win_n, h_n);
  tok_n = slot_token(g, win_n, SL, &reg_n);
  if constexpr (FUSE) {
    qf = qkv_frags(wqkv, qkv_bias, C, wave % g.heads, lane);  // item = (..) * 4 + wave and heads == 4: the wave's head never changes
    row_fetch(rx, qkv, C, tok_n, slot < g.N, lane);
  } else {
    tile_fetch<HD>(rq, qkv + h_n * HD, 3 * C, tok_n, g.N, lane);
    tile_fetch<HD>(rk, qkv + C + h_n * HD, 3 * C, tok_n, g.N, lane);
    tile_fetch<HD>(rv, qkv + 2 * C + h_n * HD, 3 * C, tok_n, g.N, lane);
  }
  for (int it = 0; it < iters; ++it) {
    const bool live = live_n;
    const int win = win_n, h = h_n, reg_own = reg_n, tok_own = tok_n;
    const bool edge = g.shifted;  // (restricting this to the windows that really mix mask regions costs more in index arithmetic than it saves)
    wave_lds_fence();  // previous iteration's fragment reads are issued before these tile writes
    if constexpr (FUSE) {
      qkv_project<P, false>(Qt, Kt, Vt, qf, rx, slot < g.N, lane);
    } else {
      tile_commit<HD>(Qt, rq, lane);
      tile_commit<HD>(Kt, rk, lane);
      tile_commit<HD>(Vt, rv, lane);
    }
    wave_lds_fence();
    if (it + 1 < iters) {
      item_of(it + 1, live_n, win_n, h_n);
      reg_n = 0;
      tok_n = slot_token(g, win_n, SL, &reg_n);
      if constexpr (FUSE) {
        row_fetch(rx, qkv, C, tok_n, slot < g.N, lane);
      } else {
        tile_fetch<HD>(rq, qkv + h_n * HD, 3 * C, tok_n, g.N, lane);
        tile_fetch<HD>(rk, qkv + C + h_n * HD, 3 * C, tok_n, g.N, lane);
        tile_fetch<HD>(rv, qkv + 2 * C + h_n * HD, 3 * C, tok_n, g.N, lane);
      }
    }
    f32x4 st = {0.f, 0.f, 0.f, 0.f};  // S^T: rows j (keys), col i (query)
#pragma unroll
    for (int kk = 0; kk < HD / 16; ++kk) st = mma16x16(frag_rows(Kt, P, kk, lane), frag_rows(Qt, P, kk, lane), st);
    if (h != h_cur) {
      h_cur = h;
      // bias + validity of this lane's four (i, j) as ONE additive constant per head (as the backward kernel): the relative-position bias inside
      // the window, 0 on a padded query row (kept finite), -1e30 on a padded key
#pragma unroll
      for (int r = 0; r < 4; ++r) biasA[r] = tA.ok[r] ? bias_table[tA.rel[r] + h] : (tA.qpad[r] ? 0.f : -1.0e30f);
    }
    float p[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) p[r] = fmaf(st[r], g.scale, biasA[r]);
    if (edge) {  // (wave-uniform) SwinModules.py:287: -100 between tokens of different mask regions; the query's region is the lane's own
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int reg_j = __shfl(reg_own, 4 * (lane >> 4) + r, 64);
        p[r] += (tA.ok[r] && reg_j != reg_own) ? -100.0f : 0.f;
      }
    }
    {  // softmax over the keys of query i = lane & 15: in-lane over r, then across the four row groups
      float m = fmaxf(fmaxf(p[0], p[1]), fmaxf(p[2], p[3]));
      m = rows4_max(m);
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) { p[r] = __expf(p[r] - m); sum += p[r]; }
      sum = rows4_sum(sum);
      const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
      for (int r = 0; r < 4; ++r) p[r] *= inv;
    }
    const int i = lane & 15;
    if (drop_on) {
      float dm[4];
      att_drop4(dc, att_drop_q((uint32_t)win * g.heads + h, i, lane >> 4), dm);
#pragma unroll
      for (int r = 0; r < 4; ++r) p[r] *= dm[r];
    }
    const bf16x4 pb = pack4(p);
#pragma unroll
    for (int db = 0; db < HD / 16; ++db) {
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
      o = mma16x16(frag_cols(Vt, P, db, lane), pb, o);  // O^T[d][i] = sum_j V[j][d] P[i][j]
      if (live && i < g.N) {
        const float ov[4] = {o[0], o[1], o[2], o[3]};
        br_st<bf16x4>(out, att_row(tok_own, C * 2) + (uint32_t)(h * HD + db * 16 + 4 * (lane >> 4)) * 2u) = pack4(ov);
      }
    }
  }
}

// Token of window slot t for the lanes that load it, once per item: the four tiles of an item (Q, K, V, dO) share their source rows.
template <int HD> struct TileRows { int tok[HD / 16]; };
template <int HD> __device__ __forceinline__ TileRows<HD> tile_rows(int tok_own, int lane) {
  constexpr int CPR = HD / 4;
  TileRows<HD> r;
#pragma unroll
  for (int q = 0; q < HD / 16; ++q) r.tok[q] = __shfl(tok_own, (lane + 64 * q) / CPR, 64);
  return r;
}
template <int HD>
__device__ __forceinline__ void tile_fetch_rows(TileRegs<HD>& r, const bf16_t* base, long row_stride, const TileRows<HD>& rows, int N, int lane) {
  constexpr int CPR = HD / 4;
#pragma unroll
  for (int q = 0; q < HD / 16; ++q) {
    const int c = lane + 64 * q, t = c / CPR, dc = c % CPR;
    r.v[q] = bf16x4{(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
    if (t < N) r.v[q] = br_ld<bf16x4>(base, att_row(rows.tok[q], (int)row_stride * 2) + dc * 8);
  }
}

// The backward works in the S^T orientation (rows j = keys, column i = query -- the forward kernel's): a lane holds four keys of ONE query,
// so the three reductions over keys (softmax maximum and sum, the dS row dot) are three in-lane operations and two cross-row exchanges
// each.  In the S orientation they were 12 sixteen-lane DPP reductions of 8 VALU instructions -- a quarter of an instruction stream that
// is what bounds this kernel (4 waves per SIMD, ~400 VALU instructions per (window, head) item: profiles/r3_attn_bwd_depth_ab.txt).
// dS^T is directly the B operand of dQ; dS and Pd in the other orientation (B operands of dK, dV) come from two bf16 tiles written
// [i][j] into wave-private LDS and read back with the hardware transpose read.
template <int HD, int NW, bool FUSE, bool PROJ>
__global__ __launch_bounds__(NW * 64) void window_attn_bwd_mfma_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ bias_table,
                                                                   const bf16_t* __restrict__ dout, bf16_t* __restrict__ dqkv,
                                                                   float* __restrict__ dbias_table, AttnGeom g, int total_items,
                                                                   int iters, const uint32_t* rng, uint32_t stream, float p_attn,
                                                                   const bf16_t* __restrict__ wqkv, const float* __restrict__ bqkv,
                                                                   const bf16_t* __restrict__ wproj) {
  // FUSE: `qkv` is the LayerNorm output a1 [M][C]; q / k / v are recomputed per item (qkv_project), nothing of them was saved
  // PROJ: `dout` is the gradient w.r.t. the proj Linear's OUTPUT [M][C] (already x the branch's mask); this head's slice of
  //       dO = dout . Wproj is computed per item (2 MFMAs) -- no dX launch of the proj layer, no [M][C] dO tensor
  constexpr int P = HD + 4, TILE = 16 * P;
  __shared__ __attribute__((aligned(16))) bf16_t tiles[NW][4][TILE];
  __shared__ __attribute__((aligned(16))) bf16_t trt[NW][2][16 * 20];  // dS^T / Pd^T tiles, written [i][j], read back transposed
  __shared__ float dbacc[256];  // (2wh-1)(2ww-1) x heads <= 256 entries
  __shared__ __attribute__((aligned(16))) float qkv_bias[FUSE ? 192 : 4];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // scalar: item, window and head arithmetic then runs on the SALU
  if constexpr (FUSE) {
    if (threadIdx.x < 192) qkv_bias[threadIdx.x] = bqkv[threadIdx.x];
  }
  bf16_t* Qt = tiles[wave][0];
  bf16_t* Kt = tiles[wave][1];
  bf16_t* Vt = tiles[wave][2];
  bf16_t* Gt = tiles[wave][3];  // dO
  const int table = (2 * g.wh - 1) * (2 * g.ww - 1) * g.heads;
  for (int t = threadIdx.x; t < 256; t += NW * 64) dbacc[t] = 0.f;
  __syncthreads();
  const DropCtx dc = make_drop(rng, stream, p_attn);
  const bool drop_on = p_attn > 0.f;
  const int C = g.C, grp = lane >> 4, col = lane & 15;
  const TileIdx tA = make_tile_idx<true>(g, lane);  // rows j = 4 grp + r, column i = col
  const int slot = lane & 15;
  const SlotLane SL = slot_lane(g, slot);
  // Relative-position-bias gradient: a lane owns the same (i, j) -- hence the same table row -- in every window, so it
  // accumulates in registers and touches LDS only when the head it works on changes (never, when NW is a multiple of heads).
  float dbreg[4] = {0.f, 0.f, 0.f, 0.f};
  int h_acc = -1;
  auto flush_dbias = [&]() {
    if (h_acc < 0) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (tA.ok[r]) atomicAdd(&dbacc[tA.rel[r] + h_acc], dbreg[r]);
      dbreg[r] = 0.f;
    }
  };
  // bias + validity of this lane's four (i, j) as ONE additive constant per head: score = s * scale + badd with badd = the relative-
  // position bias inside the window, 0 on a padded query row (kept finite), -1e30 on a padded key (the tiles are zero-filled there, so s = 0)
  float badd[4] = {0.f, 0.f, 0.f, 0.f};
  int h_cur = -1;
  auto item_of = [&](int it, bool& live, int& win, int& h) {
    const int item = (it * gridDim.x + blockIdx.x) * NW + wave;
    live = item < total_items;
    win = live ? (int)att_div((uint32_t)item, g.m_heads, (uint32_t)g.heads) : 0;
    h = live ? item - win * g.heads : 0;
  };
  TileRegs<HD> rq, rk, rv, rg;
  RowRegs rx, rgm;
  QkvFrags qf;
  bf16x8 wpt[2];  // PROJ: A fragments of Wproj^T for this wave's head: row d = l & 15 <-> column h 16 + d of Wproj, k = output channels 32 kk + 8 (l >> 4) ..
  bool live_n; int win_n, h_n, reg_n = 0, tok_n = 0;
  item_of(0, live_n, win_n, h_n);
  tok_n = slot_token(g, win_n, SL, &reg_n);
  {
    const TileRows<HD> rows = tile_rows<HD>(tok_n, lane);
    if constexpr (FUSE) {
      qf = qkv_frags(wqkv, qkv_bias, C, wave % g.heads, lane);  // NW % heads == 0: the wave's head never changes
      row_fetch(rx, qkv, C, tok_n, slot < g.N, lane);
    } else {
      tile_fetch_rows<HD>(rq, qkv + h_n * HD, 3 * C, rows, g.N, lane);
      tile_fetch_rows<HD>(rk, qkv + C + h_n * HD, 3 * C, rows, g.N, lane);
      tile_fetch_rows<HD>(rv, qkv + 2 * C + h_n * HD, 3 * C, rows, g.N, lane);
    }
    if constexpr (PROJ) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int e = 0; e < 8; ++e) wpt[kk][e] = wproj[(long)(32 * kk + 8 * grp + e) * C + (wave % g.heads) * 16 + col];
      row_fetch(rgm, dout, C, tok_n, slot < g.N, lane);
    } else {
      tile_fetch_rows<HD>(rg, dout + h_n * HD, C, rows, g.N, lane);
    }
  }
  for (int it = 0; it < iters; ++it) {
    const bool live = live_n;
    const int win = win_n, h = h_n, reg_own = reg_n, tok_own = tok_n;
    if (live && h != h_acc) {
      flush_dbias();
      h_acc = h;
    }
    wave_lds_fence();
    if constexpr (FUSE) {
      qkv_project<P>(Qt, Kt, Vt, qf, rx, slot < g.N, lane);
    } else {
      tile_commit<HD>(Qt, rq, lane);
      tile_commit<HD>(Kt, rk, lane);
      tile_commit<HD>(Vt, rv, lane);
    }
    if constexpr (PROJ) {  // dO^T[d][slot] = sum_c Wproj[c][h 16 + d] dout[token(slot)][c]: 4 consecutive d of one token per lane, as qkv_project
      f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wpt[0], rgm.v[0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wpt[1], rgm.v[1], acc, 0, 0, 0);
      const float o4[4] = {acc[0], acc[1], acc[2], acc[3]};
      *reinterpret_cast<bf16x4*>(Gt + col * P + 4 * grp) = pack4z(o4, slot < g.N);
    } else {
      tile_commit<HD>(Gt, rg, lane);
    }
    wave_lds_fence();
    if (it + 1 < iters) {  // the next item's tiles fly while this one is multiplied
      item_of(it + 1, live_n, win_n, h_n);
      reg_n = 0;
      tok_n = slot_token(g, win_n, SL, &reg_n);
      const TileRows<HD> rows = tile_rows<HD>(tok_n, lane);
      if constexpr (FUSE) {
        row_fetch(rx, qkv, C, tok_n, slot < g.N, lane);
      } else {
        tile_fetch_rows<HD>(rq, qkv + h_n * HD, 3 * C, rows, g.N, lane);
        tile_fetch_rows<HD>(rk, qkv + C + h_n * HD, 3 * C, rows, g.N, lane);
        tile_fetch_rows<HD>(rv, qkv + 2 * C + h_n * HD, 3 * C, rows, g.N, lane);
      }
      if constexpr (PROJ) row_fetch(rgm, dout, C, tok_n, slot < g.N, lane);
      else tile_fetch_rows<HD>(rg, dout + h_n * HD, C, rows, g.N, lane);
    }
    f32x4 st = {0.f, 0.f, 0.f, 0.f}, dt = st;
#pragma unroll
    for (int kk = 0; kk < HD / 16; ++kk) {
      const bf16x4 fq = frag_rows(Qt, P, kk, lane), fk = frag_rows(Kt, P, kk, lane);
      const bf16x4 fv = frag_rows(Vt, P, kk, lane), fg = frag_rows(Gt, P, kk, lane);
      st = mma16x16(fk, fq, st);  // S^T  : rows j, col i
      dt = mma16x16(fv, fg, dt);  // dPd^T: rows j, col i   (sum_d V[j][d] dO[i][d])
    }
    if (h != h_cur) {
      h_cur = h;
#pragma unroll
      for (int r = 0; r < 4; ++r) badd[r] = tA.ok[r] ? bias_table[tA.rel[r] + h] : (tA.qpad[r] ? 0.f : -1.0e30f);
    }
    // ---- softmax over the keys of query i = col: in-lane over r, then across the four row groups
    float p[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) p[r] = fmaf(st[r], g.scale, badd[r]);
    if (g.shifted) {  // (wave-uniform) SwinModules.py:287: -100 between tokens of different mask regions; the query's region is the lane's own
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int reg_j = __shfl(reg_own, 4 * grp + r, 64);
        p[r] += (tA.ok[r] && reg_j != reg_own) ? -100.0f : 0.f;
      }
    }
    float m = fmaxf(fmaxf(p[0], p[1]), fmaxf(p[2], p[3]));
    m = rows4_max(m);
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) { p[r] = __expf(p[r] - m); sum += p[r]; }
    sum = rows4_sum(sum);
    const float inv = __builtin_amdgcn_rcpf(sum);
    // ---- dP (through the dropout mask), the row dot, dS^T and Pd^T
    float dm[4] = {1.f, 1.f, 1.f, 1.f};
    if (drop_on) att_drop4(dc, att_drop_q((uint32_t)win * g.heads + h, col, grp), dm);  // (i = col, j = 4 grp + r)
    float dsT[4], pdT[4], dot = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      p[r] *= inv;
      const float mlt = tA.ok[r] ? dm[r] : 1.f;
      pdT[r] = p[r] * mlt;
      dsT[r] = dt[r] * mlt;      // dP
      dot += p[r] * dsT[r];
    }
    dot = rows4_sum(dot);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      dsT[r] = p[r] * (dsT[r] - dot);  // 0 on padded keys (p = 0); padded query columns meet zero rows of Q / dO below and are not stored
      if (live) dbreg[r] += dsT[r];
    }
    const bf16x4 bdsT = pack4(dsT), bpdT = pack4(pdT);
    bf16_t* tw = trt[wave][0];
    *reinterpret_cast<bf16x4*>(tw + col * 20 + 4 * grp) = bdsT;              // tile[i = col][j = 4 grp .. + 3]
    *reinterpret_cast<bf16x4*>(tw + 16 * 20 + col * 20 + 4 * grp) = bpdT;
    wave_lds_fence();
    const bf16x4 bds = frag_cols(tw, 20, 0, lane);             // lane (col j) <- rows i = 4 grp .. + 3 : dS
    const bf16x4 bpd = frag_cols(tw + 16 * 20, 20, 0, lane);   // Pd
    const bool st_ok = live && col < g.N;
    const uint32_t dst = att_row(tok_own, 6 * C) + (uint32_t)(h * HD + 4 * grp) * 2u;  // byte offset into dqkv [M][3C]
#pragma unroll
    for (int db = 0; db < HD / 16; ++db) {
      f32x4 z = {0.f, 0.f, 0.f, 0.f};
      const f32x4 dq = mma16x16(frag_cols(Kt, P, db, lane), bdsT, z);  // dQ^T[d][i] = sum_j K[j][d] dS[i][j]
      const f32x4 dk = mma16x16(frag_cols(Qt, P, db, lane), bds, z);   // dK^T[d][j] = sum_i Q[i][d] dS[i][j]
      const f32x4 dv = mma16x16(frag_cols(Gt, P, db, lane), bpd, z);   // dV^T[d][j] = sum_i dO[i][d] Pd[i][j]
      if (st_ok) {
        const float a[4] = {dq[0] * g.scale, dq[1] * g.scale, dq[2] * g.scale, dq[3] * g.scale};
        const float b[4] = {dk[0] * g.scale, dk[1] * g.scale, dk[2] * g.scale, dk[3] * g.scale};
        const float c[4] = {dv[0], dv[1], dv[2], dv[3]};
        br_st<bf16x4>(dqkv, dst + db * 32u) = pack4(a);
        br_st<bf16x4>(dqkv, dst + (uint32_t)(C + db * 16) * 2u) = pack4(b);
        br_st<bf16x4>(dqkv, dst + (uint32_t)(2 * C + db * 16) * 2u) = pack4(c);
      }
    }
  }
  flush_dbias();
  __syncthreads();
  for (int t = threadIdx.x; t < table; t += NW * 64) atomicAdd(dbias_table + t, dbacc[t]);
}

int focal_attn_mfma_fwd(const AttnGeom& g, const bf16_t* qkv, const float* bias_table, bf16_t* out, const uint32_t* rng, uint32_t stream_id,
                        float p_attn, hipStream_t st, const bf16_t* wqkv, const float* bqkv) {
  // (token rows are addressed with 24-bit multiplies and 32-bit byte offsets: att_row)
  if ((long)g.B * g.H * g.W >= (1L << 24) || (long)g.B * g.H * g.W * g.C * 6 >= (1L << 32)) {
    focal_set_error("window attention (MFMA): %ld tokens x %d channels exceed the kernels' 32-bit row addressing", (long)g.B * g.H * g.W, g.C);
    return FOCAL_EUNSUPPORTED;
  }
  const int items = g.B * g.nW * g.heads;
  int blocks = ceil_div(items, 4);
  if (blocks > 4096) blocks = 4096;
  const int iters = ceil_div(items, blocks * 4);
#define LAUNCH(HD, FUSE) FOCAL_LAUNCH((window_attn_fwd_mfma_kernel<HD, FUSE>), dim3(blocks), dim3(256), 0, st, qkv, bias_table, out, g, items, iters, rng, stream_id, p_attn, wqkv, bqkv)
  if (wqkv != nullptr) LAUNCH(16, true);  // (q / k / v projected in the kernel: C == 64, heads == 4, checked by the caller)
  else if (g.hd == 16) LAUNCH(16, false); else if (g.hd == 32) LAUNCH(32, false); else LAUNCH(64, false);
#undef LAUNCH
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}

int focal_attn_mfma_bwd(const AttnGeom& g, const bf16_t* qkv, const float* bias_table, const bf16_t* dout, bf16_t* dqkv, float* dbias_table,
                        const uint32_t* rng, uint32_t stream_id, float p_attn, hipStream_t st, const bf16_t* wqkv, const float* bqkv,
                        const bf16_t* wproj) {
  // (token rows are addressed with 24-bit multiplies and 32-bit byte offsets: att_row)
  if ((long)g.B * g.H * g.W >= (1L << 24) || (long)g.B * g.H * g.W * g.C * 6 >= (1L << 32)) {
    focal_set_error("window attention (MFMA): %ld tokens x %d channels exceed the kernels' 32-bit row addressing", (long)g.B * g.H * g.W, g.C);
    return FOCAL_EUNSUPPORTED;
  }
  const int items = g.B * g.nW * g.heads;
  // Every workgroup ends with one atomic per bias-table entry, and atomics onto one address are a serial chain
  // (~10 ns a link): the grid is kept small (workgroups loop over items) so the chain, not the math, does not set the
  // kernel's duration.
  // Measured (r1, B=256): 4-wave x 4096 blocks 125/83/69/59/36/21 us for the six stage geometries, 16-wave x 256
  // blocks 99/54/39/24/20/16 us.
  const int nw = 16, maxb = 256;  // 16-wave persistent workgroups, one per CU (8 waves / 512 workgroups: within +-0.6 %, profiles/r1_u_same_box_knob_sweep.log)
  int blocks = ceil_div(items, nw);
  if (blocks > maxb) blocks = maxb;
  const int iters = ceil_div(items, blocks * nw);
#define LAUNCH(HD, FUSE, PROJ) FOCAL_LAUNCH((window_attn_bwd_mfma_kernel<HD, 16, FUSE, PROJ>), dim3(blocks), dim3(16 * 64), 0, st, qkv, bias_table, dout, dqkv, dbias_table, g, items, iters, rng, stream_id, p_attn, wqkv, bqkv, wproj)
  if (wqkv != nullptr && wproj != nullptr) LAUNCH(16, true, true);
  else if (wqkv != nullptr) LAUNCH(16, true, false);
  else if (g.hd == 16) LAUNCH(16, false, false); else if (g.hd == 32) LAUNCH(32, false, false); else LAUNCH(64, false, false);
#undef LAUNCH
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}
