// The [1, k] "same" convolutions between 64-channel token tensors (DeepSense's inter-layer convolutions, their data gradients) as a
// weight-stationary kernel on an LDS-DMA ring of TOKEN ROWS (round 6):
//
//   out[m][n] = sum_t sum_c x[m + t - pad][c] * w[n][t * 64 + c]   (+ bias[n] | + resid[m][n]),   taps leaving m's interval of S tokens are zero
//
// Why not the GEMM with a window prologue (gemm.hpp PRO_CONV, the round-1 tile kernel): that kernel re-forms the k-fold overlapping operand
// element by element (a validity test per element and tap) and re-stages the 40 KB weight per 64-row tile; it ran the k = 5 layer of the
// step (102 400 rows: 13 MB in, 26 MB out) in 19.4 us forward, 27.5 us with the BatchNorm statistics, 24.7 us as the data gradient
// (tools/mb_conv.py).  Here a token row is read from memory ONCE into LDS (128 bytes) and serves its k taps as k row-shifted fragment reads of
// the same image; the weight lives in the consumer waves' registers for the whole launch (20 fragments per wave at k = 5); an interval
// boundary is one select per fragment address (a lane's 8 operand elements belong to one row and one tap: the lane reads a zero row instead).
//
// Workgroup = 4 consumer waves (2 x 2: 32 rows x 32 output channels each) + 1 loader wave, 64-row tiles; 168 registers and 31 KB of LDS let two
// workgroups share a CU, the launcher starts one per CU (the step runs another encoder beside this one: conv_ring_launch).  The loader
// keeps two tiles ahead of the consumers in a 3-buffer ring (a tile = 80 rows: 8 halo rows either side, 1 KB pieces of 8 rows, chunk c of row r
// at position c ^ ((r >> 1) & 7)) and publishes a buffer with a counted vmcnt in front of the tile's one s_barrier; the consumers' own loads
// (the residual of the data-gradient form) and stores are ordinary compiler-scheduled accesses -- no LDS-DMA is ever in flight in those waves.
// A workgroup walks a contiguous run of tiles inside one statistics group; the per-channel sums of the BatchNorm behind a forward convolution
// are kept in registers over the run and leave through the 16-slot / arrival-ticket scheme of gemm_body.inc (EPI_STORE_STATS).
#pragma once
#include "gemm_ring.hpp"

enum ConvRingEpi { CR_STORE = 0, CR_STORE_STATS = 1, CR_RESID = 2 };

struct ConvRingParams {
  const bf16_t* x;      // [rows][64]
  const bf16_t* w;      // [64][k * 64]   (k-major taps: column t * 64 + c)
  const float* bias;    // [64] | null
  const float* resid;   // [rows][64]     (CR_RESID; may alias out)
  float* out;           // [rows][64]
  int rows, S;
  uint32_t s_magic;     // 2^32 / S + 1: m % S by multiply-high (rows < 2^20, S < 2^12: the launcher checks)
  int tiles_per_group, wgs_per_group, run;   // 64-row tiles of a statistics group, workgroups per group, tiles per workgroup
  // CR_STORE_STATS (the layout of GemmParams' bn_* fields: gemm.hpp)
  float* bn_sums; float* bn_mean_rstd; float* bn_run_mean; float* bn_run_var; long bn_rows; float bn_eps, bn_momentum; int bn_groups;
};

template <int KT, int EPI>
__global__ __launch_bounds__(320, 3) void conv_ring_kernel(const ConvRingParams p) {
  constexpr int BM = 64, HALO = 8, TROWS = BM + 2 * HALO, BUF = TROWS * 128, NBUF = 3, PIECES = TROWS / 8, PAD = KT / 2, K = KT * 64, KS = KT * 2;
  extern __shared__ __attribute__((aligned(1024))) char cr_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, l15 = lane & 15;
  const int grp = blockIdx.x / p.wgs_per_group, wg = blockIdx.x % p.wgs_per_group;
  const int t_first = grp * p.tiles_per_group + wg * p.run;
  const int my = min(p.run, p.tiles_per_group - wg * p.run);  // >= 1: the launcher sizes the grid that way

  if (wave == 4) {
    // ---------------------------------------------------------------------------------------- loader
    const int prow = lane >> 3, pos = lane & 7;
    auto fill = [&](int it, int buf) {
      const long m0 = (long)(t_first + it) * BM - HALO;
#pragma unroll
      for (int q = 0; q < PIECES; ++q) {
        const int r = 8 * q + prow;
        long gr = m0 + r;
        gr = gr < 0 ? 0 : (gr >= p.rows ? p.rows - 1 : gr);  // (rows outside the tensor lie outside every interval: never multiplied)
        const int chunk = pos ^ ((r >> 1) & 7);
#if defined(CR_LAB_NOFILL)
        if (p.rows < 0)
#endif
        __builtin_amdgcn_global_load_lds((pipe_glb_ptr)(p.x + gr * 64 + chunk * 8), (pipe_lds_ptr)(cr_lds + buf * BUF + q * 1024), 16, 0, 0);
      }
    };
    fill(0, 0);
    if (my > 1) fill(1, 1);
    int fbuf = 2;
    for (int it = 0; it < my; ++it) {
      if (it + 1 < my) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_barrier" ::: "memory");  // tile `it` is in LDS; the consumers have left tile it - 1
      if (it + 2 < my) fill(it + 2, fbuf);
      fbuf = fbuf == NBUF - 1 ? 0 : fbuf + 1;
    }
  } else {
    // ---------------------------------------------------------------------------------------- consumers
    const int wm = wave >> 1, wn = wave & 1;
    bf16x8 wb[2][KS];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int kk = 0; kk < KS; ++kk)
        wb[j][kk] = *reinterpret_cast<const bf16x8*>(p.w + (long)(32 * wn + 16 * j + l15) * K + 32 * kk + 8 * g);
    f32x4 bias4[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bias4[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (EPI != CR_RESID && p.bias != nullptr) {
        const float4 b = *reinterpret_cast<const float4*>(p.bias + 32 * wn + 16 * j + 4 * g);
        bias4[j] = f32x4{b.x, b.y, b.z, b.w};
      }
    }
    f32x4 st_s[2], st_q[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) st_s[j] = st_q[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const uint32_t lds0 = pipe_lds_addr(cr_lds);
    const uint32_t zrow = lds0 + NBUF * BUF;  // 16 zero bytes behind the ring (written below, published by the first tile's barrier)
    if (wave == 0 && lane < 4) reinterpret_cast<uint32_t*>(cr_lds + NBUF * BUF)[lane] = 0u;
    int buf = 0;
    for (int it = 0; it < my; ++it) {
      asm volatile("s_barrier" ::: "memory");
      const int m0 = (t_first + it) * BM + 32 * wm;
      const uint32_t sb = lds0 + buf * BUF;
      buf = buf == NBUF - 1 ? 0 : buf + 1;
      int s_in[2];
      f32x4 res[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const uint32_t m = (uint32_t)(m0 + 16 * i + l15);
        s_in[i] = (int)(m - (uint32_t)p.S * __umulhi(m, p.s_magic));
        if (EPI == CR_RESID) {
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const float4 r = *reinterpret_cast<const float4*>(p.resid + (long)m * 64 + 32 * wn + 16 * j + 4 * g);
            res[i][j] = f32x4{r.x, r.y, r.z, r.w};
          }
        }
      }
      f32x4 acc[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#if !defined(CR_LAB_NOMMA)
      // tap t + 1's fragments are requested before tap t's MFMAs (two register sets, a counted lgkmcnt); a lane whose tap leaves its
      // interval reads the workgroup's zero row instead of being zeroed in registers (one select on the address for the lane's two k halves)
      bf16x8 xa[2][2][2];  // [set][k half][row fragment]
      auto request = [&](auto tc, auto bc) {
        constexpr int t = decltype(tc)::value, bset = decltype(bc)::value;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const bool ok = (unsigned)(s_in[i] + t - PAD) < (unsigned)p.S;
          const int R = HALO + 32 * wm + 16 * i + l15 + t - PAD;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const uint32_t a = sb + R * 128 + (((4 * h + g) ^ ((R >> 1) & 7)) << 4);
#if defined(CR_LAB_NOREAD)   // lab: no LDS read (the operand is a weight fragment)
            xa[bset][h][i] = wb[0][(t + h + i) % KS];
            if (!ok && a == 0x7fffffffu) xa[bset][h][i] = wb[1][0];
#else
            xa[bset][h][i] = pipe_lds_read128<0>(ok ? a : zrow);
#endif
          }
        }
      };
      // (k = 5 with a residual or statistics epilogue: the second register set spills at the 168 registers of two workgroups per CU -- 17 / 21
      //  spilled registers made those launches 34.7 / 29.2 us against 20.5 / 27.1 un-pipelined; they keep one set)
      constexpr bool PIPE = !(KT == 5 && EPI != CR_STORE);
      request(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
      pipe_static_for<0, KT>([&](auto tc) {
        constexpr int t = decltype(tc)::value, cur = PIPE ? (t & 1) : 0;
        if constexpr (!PIPE) {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xa[0][0][0]), "+v"(xa[0][0][1]), "+v"(xa[0][1][0]), "+v"(xa[0][1][1]));
        } else if constexpr (t + 1 < KT) {
          request(std::integral_constant<int, t + 1>{}, std::integral_constant<int, (t + 1) & 1>{});
          asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(xa[cur][0][0]), "+v"(xa[cur][0][1]), "+v"(xa[cur][1][0]), "+v"(xa[cur][1][1]));
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xa[cur][0][0]), "+v"(xa[cur][0][1]), "+v"(xa[cur][1][0]), "+v"(xa[cur][1][1]));
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
#if defined(CR_LAB_NOMFMA)  // lab: the reads without the matrix instructions
            for (int j = 0; j < 2; ++j) acc[i][j][j] += (float)xa[cur][h][i][2 * j] * (float)wb[j][2 * t + h][1];
#else
            for (int j = 0; j < 2; ++j) acc[i][j] = mma16(wb[j][2 * t + h], xa[cur][h][i], acc[i][j]);
#endif
        if constexpr (!PIPE && t + 1 < KT) {
          // the MFMAs above have read set 0 when the next tap's reads overwrite it: the fragment registers are operands of this statement
          asm volatile("" : "+v"(xa[0][0][0]), "+v"(xa[0][0][1]), "+v"(xa[0][1][0]), "+v"(xa[0][1][1]));
          request(std::integral_constant<int, t + 1>{}, std::integral_constant<int, 0>{});
        }
      });
#else
      acc[0][0][0] = (float)wb[0][0][0] + (float)wb[1][KS - 1][7] + (float)wb[0][KS / 2][3] + (float)s_in[0] + (float)s_in[1];
#endif
      // acc[i][j][e] = out[m0 + 16 i + l15][32 wn + 16 j + 4 g + e]
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          f32x4 v = acc[i][j] + (EPI == CR_RESID ? res[i][j] : bias4[j]);
#if defined(CR_LAB_NOSTORE)
          if (v[0] == 12345.678f && v[1] == 1.f && v[2] == 2.f && v[3] == 3.f)
#endif
          *reinterpret_cast<float4*>(p.out + (long)(m0 + 16 * i + l15) * 64 + 32 * wn + 16 * j + 4 * g) = make_float4(v[0], v[1], v[2], v[3]);
          if (EPI == CR_STORE_STATS) {
            st_s[j] += v;
            st_q[j] += v * v;
          }
        }
    }
    if (EPI == CR_STORE_STATS) {
      // this wave's column sums: rows live in l15 -> one 16-lane reduction per value; lanes l15 == 0 park them for the two row halves' sum
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) { st_s[j][e] = row16_sum(st_s[j][e]); st_q[j][e] = row16_sum(st_q[j][e]); }
    }
    if (EPI == CR_STORE_STATS) {
      asm volatile("s_barrier" ::: "memory");  // (1) every consumer has left the ring: its first bytes become the staging area
      float* red = reinterpret_cast<float*>(cr_lds);  // [wm][which][64]
      if (l15 == 0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          *reinterpret_cast<float4*>(red + (wm * 2 + 0) * 64 + 32 * wn + 16 * j + 4 * g) = make_float4(st_s[j][0], st_s[j][1], st_s[j][2], st_s[j][3]);
          *reinterpret_cast<float4*>(red + (wm * 2 + 1) * 64 + 32 * wn + 16 * j + 4 * g) = make_float4(st_q[j][0], st_q[j][1], st_q[j][2], st_q[j][3]);
        }
      }
    }
  }
  if (EPI != CR_STORE_STATS) return;
  if (wave == 4) asm volatile("s_barrier" ::: "memory");  // (1)
  __syncthreads();                                        // (2) the partial sums are in LDS
  {
    constexpr int N = 64;
    float* red = reinterpret_cast<float*>(cr_lds);
    float* const bn_sums = p.bn_sums + (size_t)grp * (BN_STAT_SLOTS * 2 * N + 1);
    float* slot = bn_sums + (size_t)(blockIdx.x & (BN_STAT_SLOTS - 1)) * 2 * N;
    if (p.bn_mean_rstd == nullptr) {
      // sums only: the BatchNorm launch behind this one adds the slots up (focal_bn_act_fwd_sums) -- fire-and-forget adds, the end of the
      // kernel publishes them
      if (tid < 2 * N) {
        const int which = tid / N, col = tid % N;
        atomicAdd(slot + which * N + col, red[(0 * 2 + which) * 64 + col] + red[(1 * 2 + which) * 64 + col]);
      }
      return;
    }
    if (tid < 2 * N) {
      const int which = tid / N, col = tid % N;
      const float tsum = red[(0 * 2 + which) * 64 + col] + red[(1 * 2 + which) * 64 + col];
      const float old = __hip_atomic_fetch_add(slot + which * N + col, tsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("" ::"v"(old));  // the returning form: its result arriving means the add is done at L2 (gemm_body.inc, bn.hip)
    }
    int* last_flag = reinterpret_cast<int*>(cr_lds + 2048);
    __syncthreads();
    unsigned int* counter = reinterpret_cast<unsigned int*>(bn_sums + (size_t)BN_STAT_SLOTS * 2 * N);
    if (tid == 0) *last_flag = atomicAdd(counter, 1u) == (unsigned int)p.wgs_per_group - 1u;
    __syncthreads();
    if (*last_flag) {
      __syncthreads();
      for (int i = tid; i < BN_STAT_SLOTS * 2 * N; i += 320) red[i] = atomicAdd(bn_sums + i, 0.f);  // [slot][which][N], coherent at L2
      __syncthreads();
      if (tid < N) {
        float sm = 0.f, sq = 0.f;
        for (int sl = 0; sl < BN_STAT_SLOTS; ++sl) { sm += red[sl * 2 * N + tid]; sq += red[sl * 2 * N + N + tid]; }
        const float n = (float)p.bn_rows;
        const float mean = sm / n;
        float var = sq / n - mean * mean;
        var = fmaxf(var, 0.f);
        float* const mr = p.bn_mean_rstd + (size_t)grp * 2 * N;
        mr[tid] = mean;
        mr[N + tid] = rsqrtf(var + p.bn_eps);
        if (p.bn_run_mean) {
          float* const rm = p.bn_run_mean + (size_t)grp * N;
          float* const rv = p.bn_run_var + (size_t)grp * N;
          rm[tid] = (1.f - p.bn_momentum) * rm[tid] + p.bn_momentum * mean;
          rv[tid] = (1.f - p.bn_momentum) * rv[tid] + p.bn_momentum * var * (n / fmaxf(n - 1.f, 1.f));
        }
      }
    }
  }
}

// Does the row-ring kernel take this convolution?  64 -> 64 channels, k = 3 / 5, bf16 operands, whole 64-row tiles (per statistics group),
// intervals of at least k tokens, 16-byte aligned operands.  FOCAL_CONV_RING=0 keeps the sliding-window GEMM (A/B runs, the tests' second path).
// Alone on the chip the kernel is 5 - 25 % faster than that GEMM per launch (k = 5 / 3 forward 16.3 / 12.1 vs ~20 / ~14 us, data gradient
// 21.2 / 18.9 vs ~24 / ~21; tools/prof_conv.sh); inside the replayed DeepSense step +1.4 ... +2.3 %, +2.0 ... +2.9 % with the sums-only
// statistics (tools/ab_conv_ring.sh, three interleaved repetitions; profiles/r6_conv_ring.txt -- which also records the build whose second
// fragment set spilled: that one LOST 2 - 4 % in the step).
static inline bool conv_ring_fits(const focal_conv_desc* d, int c_in, int c_out, const void* x, const void* w, int groups) {
  const char* sel = getenv("FOCAL_CONV_RING");  // (read per call: the tests switch paths inside one process)
  if ((sel != nullptr && sel[0] == '0') || d->dtype != FOCAL_BF16 || c_in != 64 || c_out != 64 || (d->k != 3 && d->k != 5)) return false;
  if (groups < 1 || d->rows % (64 * groups) != 0 || d->rows >= (1 << 20) || d->S < d->k || d->S >= (1 << 12)) return false;
  return ((uintptr_t)x % 16 == 0) && ((uintptr_t)w % 16 == 0);
}

template <int EPI>
static inline hipError_t conv_ring_launch(ConvRingParams& p, int k, int groups, hipStream_t stream) {
  constexpr int LDS_BYTES = 3 * 80 * 128 + 128;  // the ring + the zero row
  const int tiles = p.rows / 64;
  p.tiles_per_group = tiles / groups;
  static const int lab_slots = getenv("FOCAL_LAB_CONV_SLOTS") ? atoi(getenv("FOCAL_LAB_CONV_SLOTS")) : 0;
  // one workgroup per CU (229 workgroups of 7 tiles at the step's shape), although two fit: inside the step -- the other modality's encoder runs
  // beside this one -- 256 slots gave 136.9 / 138.7 / 139.3 k windows/s against 136.7 / 135.9 / 137.2 k for 512 (tools/ab_conv_slots.sh)
  const int slots = lab_slots > 0 ? lab_slots : focal_cu_count();
  const int per_group = slots / groups > 0 ? slots / groups : 1;
  // (an even cut over 512 slots -- 512 workgroups of 3 or 4 tiles instead of 400 of 4 -- was slower alone (statistics form 28.6 vs 26.3 us) and
  //  lost the step's gain: more weight-fragment loads, more slot adds, less room for the other encoder; tools/ab_conv_ring.sh)
  p.run = (p.tiles_per_group + per_group - 1) / per_group;
  p.wgs_per_group = (p.tiles_per_group + p.run - 1) / p.run;
  p.s_magic = 0xFFFFFFFFu / (uint32_t)p.S + 1u;
  const int wgs = p.wgs_per_group * groups;
  if (k == 5) FOCAL_LAUNCH((conv_ring_kernel<5, EPI>), dim3(wgs), dim3(320), LDS_BYTES, stream, p);
  else FOCAL_LAUNCH((conv_ring_kernel<3, EPI>), dim3(wgs), dim3(320), LDS_BYTES, stream, p);
  return hipGetLastError();
}
