// Shared device/host helpers for libfocal_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include "../../include/focal_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define FOCAL_WAVE 64

// ----------------------------------------------------------------------------------------------- errors
void focal_set_error(const char* fmt, ...);
void focal_note_kernel(const char* fmt, ...);
#define FOCAL_CHECK_ARG(cond, ...)                 \
  do {                                             \
    if (!(cond)) {                                 \
      focal_set_error(__VA_ARGS__);                \
      return FOCAL_EINVAL;                         \
    }                                              \
  } while (0)
#define FOCAL_LAUNCH_CHECK()                                                       \
  do {                                                                             \
    hipError_t e__ = hipGetLastError();                                            \
    if (e__ != hipSuccess) {                                                       \
      focal_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
      return FOCAL_EHIP;                                                           \
    }                                                                              \
  } while (0)

// ----------------------------------------------------------------------------------------------- launches
// Every kernel of the library goes out through FOCAL_LAUNCH: the plain launch, or -- while a launch trace is open (trace.cpp,
// focal_trace_begin: bench.py's in-step roofline) -- the same launch with a start / stop event pair on the dispatch itself.
extern int g_focal_trace_on;
int focal_trace_slot(const void* kernel, dim3 grid, dim3 block, hipStream_t st, hipEvent_t* e0, hipEvent_t* e1);
#define FOCAL_LAUNCH(kern, grid, block, sm, st, ...)                                                  \
  do {                                                                                                 \
    auto k__ = (kern);                                                                                 \
    hipStream_t st__ = (st);                                                                           \
    hipEvent_t e0__ = nullptr, e1__ = nullptr;                                                         \
    const int tm__ = __builtin_expect(g_focal_trace_on, 0) ? focal_trace_slot((const void*)k__, (grid), (block), st__, &e0__, &e1__) : 0; \
    if (tm__ == FOCAL_TRACE_DISPATCH) {                                                                \
      hipExtLaunchKernelGGL(k__, (grid), (block), (sm), st__, e0__, e1__, 0, __VA_ARGS__);             \
    } else {                                                                                           \
      if (tm__ == FOCAL_TRACE_EVENTS) (void)hipEventRecord(e0__, st__);                                \
      hipLaunchKernelGGL(k__, (grid), (block), (sm), st__, __VA_ARGS__);                               \
      if (tm__ == FOCAL_TRACE_EVENTS) (void)hipEventRecord(e1__, st__);                                \
    }                                                                                                  \
  } while (0)

// ----------------------------------------------------------------------------------------------- scalar conversions
__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16_t x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float x) { return (bf16_t)x; }

// ----------------------------------------------------------------------------------------------- counter RNG
// Dropout masks are a pure function of (seed word in device memory, stream id, element index) so that the
// backward pass regenerates the forward mask without storing it, and so that a captured hipGraph gets fresh
// masks on every replay (the seed word is bumped on the device by focal_rng_advance).
__device__ __forceinline__ uint32_t focal_mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
// The per-element hash of the dropout streams.  focal_mix32's two 32-bit multiplies issue at quarter rate (v_mul_lo_u32: 8 of the hash's
// 14 issue slots), and the element masks of the GEMM / MLP epilogues and the attention kernels draw one hash per element or pair inside
// vector-bound loops.  This one multiplies 24-bit operands (v_mad_u32_u24 / v_mul_u32_u24, full rate: 8 slots); the addend of the first round
// carries the bits the 24-bit operand drops.  Not a bijection (2^24 distinct values per key), which a keep / drop decision does not need:
// keep rates, neighbour / row / column correlations, byte chi-squares and avalanche (0.500 per input bit) are indistinguishable from
// focal_mix32's over 2 M consecutive indices.  Seeds and keys are still mixed with focal_mix32 (a bijection: the seed sequence must not cycle).
__device__ __forceinline__ uint32_t focal_hash24(uint32_t x) {
  uint32_t h = x ^ (x >> 16);
  h = __umul24(h, 0xE35A2Bu) + x;
  h ^= h >> 15;
  h = __umul24(h, 0xB5297Bu);
  return h ^ (h >> 16);
}
struct DropCtx {
  uint32_t key;     // mixed (seed, stream)
  uint32_t thresh;  // drop if (hash >> 8) < thresh  (24-bit resolution)
  float scale;      // 1 / (1 - p)
};
__device__ __forceinline__ DropCtx make_drop(const uint32_t* seed_ptr, uint32_t stream, float p) {
  DropCtx d;
  uint32_t seed = seed_ptr ? seed_ptr[0] : 0u;
  d.key = focal_mix32(seed * 0x9E3779B9U + stream * 0x85EBCA6BU + 0x1234567U);
  d.thresh = (uint32_t)(p * 16777216.0f);
  d.scale = 1.0f / (1.0f - p);
  return d;
}
// multiplier (0 or 1/(1-p)) for element `idx`
__device__ __forceinline__ float drop_mult(const DropCtx& d, uint32_t idx) {
  uint32_t h = focal_hash24(idx ^ d.key);
  return ((h >> 8) < d.thresh) ? 0.0f : d.scale;
}

// erf-form GELU and its derivative (nn.GELU(approximate="none")).  erf by Abramowitz-Stegun 7.1.26 (|err| < 1.5e-7,
// one v_exp + one v_rcp): libm erff costs ~3x more VALU and these sit in GEMM prologues / epilogues.
__device__ __forceinline__ void gelu_parts(float x, float& cdf, float& pdf) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);  // raw v_rcp_f32 (1 ulp); __frcp_rn expands to the ~10-instruction IEEE divide
  const float e = __expf(-z * z);  // = exp(-x^2 / 2)
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float erf_abs = 1.0f - poly * e;
  cdf = 0.5f * (1.0f + copysignf(erf_abs, x));
  pdf = 0.39894228040143268f * e;
}
// Two elements at a time on 2-wide vectors: the polynomial, the scalings and the final combinations compile to packed fp32
// (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32), half the issue slots of the scalar form; v_exp / v_rcp stay per element.
typedef float gelu_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gelu_parts2(gelu_f2 x, gelu_f2& cdf, gelu_f2& pdf) {
  const gelu_f2 ax = {fabsf(x.x), fabsf(x.y)};
  const gelu_f2 z = ax * 0.70710678118654752f;
  const gelu_f2 d = z * 0.3275911f + 1.0f;
  const gelu_f2 t = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  const gelu_f2 nz2 = -(z * z);
  const gelu_f2 e = {__expf(nz2.x), __expf(nz2.y)};
  const gelu_f2 poly = t * (t * (t * (t * (t * 1.061405429f - 1.453152027f) + 1.421413741f) - 0.284496736f) + 0.254829592f);
  const gelu_f2 h = poly * e * -0.5f + 0.5f;  // 0.5 * erf(|x| / sqrt 2)
  cdf = gelu_f2{copysignf(h.x, x.x), copysignf(h.y, x.y)} + 0.5f;
  pdf = e * 0.39894228040143268f;
}
__device__ __forceinline__ float gelu_f(float x) {
  float cdf, pdf;
  gelu_parts(x, cdf, pdf);
  return x * cdf;
}
__device__ __forceinline__ float gelu_grad_f(float x) {
  float cdf, pdf;
  gelu_parts(x, cdf, pdf);
  return cdf + x * pdf;
}

// All-reduce inside each row of 16 lanes with DPP modifiers (quad_perm xor 1, xor 2, row_half_mirror, row_mirror): four VALU
// instructions, no LDS.  __shfl_xor compiles to ds_bpermute_b32 -- an LDS-pipe round trip and an s_waitcnt each; the MFMA
// attention kernels did 74 of them per (window, head).
template <int CTRL> __device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_f32<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_f32<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_f32<0x141>(v);  // row_half_mirror (quads now hold equal values: i <-> 7 - i joins the two quads of a half)
  v += dpp_f32<0x140>(v);  // row_mirror      (i <-> 15 - i joins the two halves)
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_f32<0xB1>(v));
  v = fmaxf(v, dpp_f32<0x4E>(v));
  v = fmaxf(v, dpp_f32<0x141>(v));
  v = fmaxf(v, dpp_f32<0x140>(v));
  return v;
}

// v + (v of lane ^ 16) / v + (v of lane ^ 32) with the gfx950 row / half swaps: v_permlane16_swap(a, b) exchanges a's odd 16-lane rows with b's
// even rows -- with a = b = v the two results hold rows (R0, R0, R2, R2) and (R1, R1, R3, R3) --, v_permlane32_swap the 32-lane halves.  Vector
// instructions of a few cycles' latency where `v += __shfl_xor(v, 16 | 32)` is a ds_bpermute round trip through the LDS pipeline (hundreds of
// cycles under load, on the dependency chain of every LayerNorm row in the GEMM epilogues); same pairs, same value: bit-identical.
__device__ __forceinline__ float xadd16(float v) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(a[0]) + __uint_as_float(a[1]);
}
__device__ __forceinline__ float xadd32(float v) {
  auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(a[0]) + __uint_as_float(a[1]);
}
__device__ __forceinline__ float xmax16(float v) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
}
__device__ __forceinline__ float xmax32(float v) {
  auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
}

// v + (v of lane ^ o): the swaps for o = 16 / 32 (a wave-uniform or compile-time o), ds_bpermute otherwise
__device__ __forceinline__ float xadd(float v, int o) { return o == 16 ? xadd16(v) : (o == 32 ? xadd32(v) : v + __shfl_xor(v, o, 64)); }

__device__ __forceinline__ float wave_sum(float v) {
  v = xadd32(v);
  v = xadd16(v);
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  v = xmax32(v);
  v = xmax16(v);
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// The warp a plan record asks for: 0 unless it names one AND carries the knots of one (a record with fewer than 4 knots is not something
// focal_view_draw writes; every consumer -- the curve solve, the apply pass, the transform's source selection -- reads it through here)
__device__ __forceinline__ int focal_plan_warp(const focal_view_plan* pl) {
  const int w = pl->warp, n = pl->nknots;
  return (w != 0 && n >= 4 && n <= FOCAL_VIEW_MAX_KNOTS) ? w : 0;
}
