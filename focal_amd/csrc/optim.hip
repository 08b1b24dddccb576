// Fused AdamW over the flat parameter arena + bf16 shadow refresh + device-side step / seed bookkeeping.
// HBM-bound: 16 B read + 12 B written per parameter (+2 B shadow), float4 per lane.
#include "common.hpp"

__global__ void rng_advance_kernel(uint32_t* state) {
  state[0] = focal_mix32(state[0] + 0x9E3779B9U);
  state[1] += 1u;
}

extern "C" int focal_rng_advance(uint32_t* state, void* stream) {
  FOCAL_CHECK_ARG(state != nullptr, "rng_advance: null state");
  FOCAL_LAUNCH(rng_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state);
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}

// Diagnostic: slots[index] = the device's constant-rate wall clock (100 MHz) when this one-thread launch runs -- a marker that can sit INSIDE
// a captured hipGraph, where events cannot be read and rocprofv3 serialises the branches (tools/phase_marks.py).
__global__ void mark_kernel(unsigned long long* slots, int index) { slots[index] = wall_clock64(); }

extern "C" int focal_mark(unsigned long long* slots, int index, void* stream) {
  FOCAL_CHECK_ARG(slots != nullptr && index >= 0, "mark: null buffer or negative index");
  FOCAL_LAUNCH(mark_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, slots, index);
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}

// torch.optim.AdamW: p *= 1 - lr*wd; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
// p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// advance != 0 (focal_adamw_multi_advance): the update uses step count step_state[1] + 1, and the workgroup that finishes LAST -- by
// then every workgroup has read the count -- advances step_state and seed_state the way focal_rng_advance would (tickets in
// step_state[2] and step_state[8 .. 39]: FOCAL_STEP_STATE_WORDS = 40 zero-initialised words): the step's two one-thread bookkeeping launches ride on the optimizer kernel.
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, bf16_t* __restrict__ shadow, long n,
                                                    const float* __restrict__ lr_dev, uint32_t* rng_state, uint32_t* seed_state,
                                                    int advance, focal_adamw_desc d) {
  const float lr = lr_dev[0];
  // (an explicit relaxed atomic load: `count` feeds every store below and, through `ticket_base`, the ticket this workgroup takes -- a data
  // dependence, so the read cannot be ordered behind the ticket whatever the compiler does with the loop)
  const uint32_t count = __atomic_load_n(rng_state + 1, __ATOMIC_RELAXED);
  const float t = (float)(count + (advance ? 1u : 0u));
  const float bc1 = 1.0f - powf(d.beta1, t), bc2 = 1.0f - powf(d.beta2, t);
  // weight decay: decoupled (AdamW: p *= 1 - lr * wd) or, with d.l2_decay, torch.optim.Adam's L2 form (g += wd * p)
  const float step = lr / bc1, isq = rsqrtf(bc2), decay = d.l2_decay ? 1.0f : 1.0f - lr * d.weight_decay;
  const float l2 = d.l2_decay ? d.weight_decay : 0.0f;
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 pp = reinterpret_cast<float4*>(p)[i];
    const float4 gg = reinterpret_cast<const float4*>(g)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
#define ADAM1(c)                                              \
    const float g_##c = gg.c + l2 * pp.c;                     \
    mm.c = d.beta1 * mm.c + (1.0f - d.beta1) * g_##c;         \
    vv.c = d.beta2 * vv.c + (1.0f - d.beta2) * g_##c * g_##c; \
    pp.c = pp.c * decay - step * mm.c / (sqrtf(vv.c) * isq + d.eps);
    ADAM1(x) ADAM1(y) ADAM1(z) ADAM1(w)
#undef ADAM1
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
    if (shadow) {
      bf16x4 s;
      s[0] = (bf16_t)pp.x; s[1] = (bf16_t)pp.y; s[2] = (bf16_t)pp.z; s[3] = (bf16_t)pp.w;
      reinterpret_cast<bf16x4*>(shadow)[i] = s;
    }
  }
  if (advance == 2) {
    __syncthreads();
    if (threadIdx.x == 0) {
      // (no fence: a workgroup's read of the step count has returned -- its value went into every store above -- before it takes a
      // ticket, and the words written below are read by later kernels only.  __threadfence() here is an L2 write-back per workgroup:
      // it tripled the kernel's duration.)  Two ticket levels -- 32 groups of workgroups, then the groups -- because atomics onto ONE
      // word are a serial chain of ~10 ns links: 2048 workgroups on one ticket added 12 us to DeepSense's 26 us launch.
      const uint32_t grp = blockIdx.x & 31u, ngrp = gridDim.x < 32u ? gridDim.x : 32u;
      const uint32_t gsize = (gridDim.x - grp + 31u) >> 5;
      const uint32_t ticket_one = 1u + (count & 0u);  // == 1, computed from the count that was read
      if (atomicAdd(rng_state + 8 + grp, ticket_one) == gsize - 1u) {
        rng_state[8 + grp] = 0u;
        if (atomicAdd(rng_state + 2, 1u) == ngrp - 1u) {
          rng_state[2] = 0u;
          rng_state[0] = focal_mix32(rng_state[0] + 0x9E3779B9U);
          rng_state[1] += 1u;
          if (seed_state != nullptr) {
            seed_state[0] = focal_mix32(seed_state[0] + 0x9E3779B9U);
            seed_state[1] += 1u;
          }
        }
      }
    }
  }
}

static int adamw_launch(const focal_adamw_desc* d, int nseg, float* const* p, const float* const* g, float* const* m, float* const* v,
                        void* const* shadow_bf16, const long* n, const float* lr_dev, uint32_t* step_state, uint32_t* seed_state,
                        bool advance, hipStream_t st) {
  FOCAL_CHECK_ARG(d && p && g && m && v && n && lr_dev && step_state, "adamw_multi: null argument");
  int last = -1;
  for (int s = 0; s < nseg; ++s) {
    FOCAL_CHECK_ARG(n[s] % 4 == 0, "adamw_multi: segment %d length %ld is not a multiple of 4 (arena segments are padded)", s, n[s]);
    if (n[s] > 0) last = s;
  }
  FOCAL_CHECK_ARG(!advance || last >= 0, "adamw_multi_advance: no non-empty segment to carry the step-count advance");
  for (int s = 0; s < nseg; ++s) {
    if (n[s] == 0) continue;
    int blocks = ceil_div(n[s] / 4, 256);
    if (blocks > 2048) blocks = 2048;
    // segments run in stream order: every one of them reads the same step count, the last one advances it
    FOCAL_LAUNCH(adamw_kernel, dim3(blocks), dim3(256), 0, st, p[s], g[s], m[s], v[s], shadow_bf16 ? (bf16_t*)shadow_bf16[s] : nullptr,
                       n[s], lr_dev, step_state, (advance && s == last) ? seed_state : nullptr, advance ? (s == last ? 2 : 1) : 0, *d);
  }
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}

extern "C" int focal_adamw_multi(const focal_adamw_desc* d, int nseg, float* const* p, const float* const* g, float* const* m,
                                 float* const* v, void* const* shadow_bf16, const long* n, const float* lr_dev,
                                 const uint32_t* rng_state, void* stream) {
  return adamw_launch(d, nseg, p, g, m, v, shadow_bf16, n, lr_dev, const_cast<uint32_t*>(rng_state), nullptr, false, (hipStream_t)stream);
}

extern "C" int focal_adamw_multi_advance(const focal_adamw_desc* d, int nseg, float* const* p, const float* const* g, float* const* m,
                                         float* const* v, void* const* shadow_bf16, const long* n, const float* lr_dev,
                                         uint32_t* step_state, int step_state_words, uint32_t* seed_state, void* stream) {
  FOCAL_CHECK_ARG(step_state_words >= FOCAL_STEP_STATE_WORDS, "adamw_multi_advance: step_state holds %d words, the tickets need FOCAL_STEP_STATE_WORDS = %d "
                  "(the 4-word state of focal_rng_advance / focal_adamw_multi is not enough)", step_state_words, FOCAL_STEP_STATE_WORDS);
  return adamw_launch(d, nseg, p, g, m, v, shadow_bf16, n, lr_dev, step_state, seed_state, true, (hipStream_t)stream);
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long n) {
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const float4 a = reinterpret_cast<const float4*>(src)[i];
    bf16x4 s;
    s[0] = (bf16_t)a.x; s[1] = (bf16_t)a.y; s[2] = (bf16_t)a.z; s[3] = (bf16_t)a.w;
    reinterpret_cast<bf16x4*>(dst)[i] = s;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) dst[(n4 << 2) + threadIdx.x] = (bf16_t)src[(n4 << 2) + threadIdx.x];
}

extern "C" int focal_cast_bf16(const float* src, void* dst, long n, void* stream) {
  FOCAL_CHECK_ARG(src && dst && n >= 0, "cast_bf16: bad argument");
  if (n == 0) return FOCAL_OK;
  int blocks = ceil_div((n + 3) / 4, 256);
  if (blocks > 2048) blocks = 2048;
  FOCAL_LAUNCH(cast_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, n);
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}
