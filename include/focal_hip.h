/* libfocal_hip -- C ABI of the MI355X (gfx950) FOCAL pretraining hot path.
 *
 * The reference (tomoyoshki/focal) is pure PyTorch and has no FFI of its own; its boundary for this path is the
 * Python plugin surface (`src/train_utils/model_selection.py:14-59`, the modules under `src/models/`).  This library sits
 * *underneath* that surface: the modules under `focal_amd/src/models/` keep the reference's classes / state-dict names and calls
 * the entry points below through ctypes, one per ATen operator group the reference step executes
 * (the reference site each entry replaces is cited on the declaration).
 *
 * Conventions (all entry points):
 *   - plain C: pointers + sizes, no torch / C++ types;
 *   - every tensor pointer is a CALLER-OWNED DEVICE pointer, row-major contiguous unless a leading dimension is
 *     given; descriptor structs and pointer *arrays* live on the host;
 *   - work is enqueued asynchronously on `stream` (a hipStream_t passed as void*); nothing allocates, frees or
 *     synchronises, so every call is hipGraph-capturable;
 *   - returns FOCAL_OK (0) or a negative error code; `focal_last_error()` gives a thread-local message;
 *   - "dtype" is the storage type of activations and of the weight operand handed to the matrix cores:
 *     FOCAL_F32 (exact-fp32 MFMA, the 1e-3 parity mode) or FOCAL_BF16 (bf16 operands, fp32 accumulate).
 *     Residual streams, statistics, loss values, master weights, gradients and optimizer state are always fp32.
 */
#ifndef FOCAL_HIP_H
#define FOCAL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 7: launch trace (focal_trace_*), focal_adamw_multi_advance takes the step-state length; 2: BatchNorm scratch of 2C + 1 floats; 3: fused MLP, warps, Mixup; 4: row-sharded loss head, weight-gradient launch queries */
#define FOCAL_ABI_VERSION 13

enum { FOCAL_OK = 0, FOCAL_EINVAL = -1, FOCAL_EUNSUPPORTED = -2, FOCAL_EWORKSPACE = -3, FOCAL_EHIP = -4 };
enum { FOCAL_F32 = 0, FOCAL_BF16 = 1 };

int focal_abi_version(void);
/* diagnostics: the kernel template the GEMM-family entry points launched last on this thread ("" before the first one) */
const char* focal_last_kernel(void);
const char* focal_last_error(void);

/* ------------------------------------------------------------------------------------------------ launch trace (measurement only)
 * Per-kernel durations of eager steps for bench.py's in-step roofline.  Between focal_trace_begin and focal_trace_end
 * every kernel launch of the library is recorded: FOCAL_TRACE_DISPATCH attaches a start / stop event pair to the dispatch itself
 * (hipExtLaunchKernel: the duration is the dispatch's own begin -> end timestamps, as rocprofv3 --kernel-trace reports it),
 * FOCAL_TRACE_EVENTS records one event directly before and one directly behind the launch.  Launches onto a stream that is being
 * captured into a hipGraph are not recorded.  These calls own a pool of events and focal_trace_read synchronises on them: they are
 * the one exception to "nothing allocates or synchronises", and no product path calls them. */
enum { FOCAL_TRACE_DISPATCH = 1, FOCAL_TRACE_EVENTS = 2 };
typedef struct {
  char kernel[384];        /* symbol name of the launched kernel (mangled: one per template instantiation = one row of `rocprofv3 --stats -M`) */
  unsigned grid[3], block[3];
  void* stream;
  float us;                /* duration in microseconds */
} focal_trace_record;
int focal_trace_begin(int capacity, int mode); /* drops the records of an earlier trace, records up to `capacity` launches */
int focal_trace_end(void);
int focal_trace_count(void);
int focal_trace_read(int first, int n, focal_trace_record* out); /* waits for the launches; records stay until the next begin */

/* ------------------------------------------------------------------------------------------------ RNG state
 * Device-resident {seed, step} pair (uint32[4], two words used).  Dropout masks are pure functions of
 * (seed, stream id, element index): backward regenerates the forward mask, and a captured graph gets new masks
 * on every replay because `focal_rng_advance` bumps the words on the device.  Replaces torch's Philox stream
 * for nn.Dropout / DropPath (models/SwinModules.py:26,115,257; models/ConvModules.py:96). */
int focal_rng_advance(uint32_t* state, void* stream);
/* Diagnostic marker (round 6): slots[index] = the device wall clock (100 MHz ticks) at the moment a one-thread launch on `stream` runs.
 * Usable inside a captured hipGraph: the only way to time the BRANCHES of a replayed step (tools/phase_marks.py). */
int focal_mark(unsigned long long* slots, int index, void* stream);

typedef struct {
  const uint32_t* rng;   /* device RNG state or NULL (seed 0) */
  uint32_t stream_elem;  /* stream id of the element-wise dropout */
  float p_elem;          /* element-wise drop probability, 0 = off */
  uint32_t stream_path;  /* stream id of the per-sample stochastic depth (DropPath) */
  float p_path;          /* DropPath probability, 0 = off */
  int rows_per_sample;   /* rows of the [M, N] activation that belong to one sample */
} focal_drop_desc;

/* ------------------------------------------------------------------------------------------------ row 3: FFT
 * Augmenter.fft_preprocess (data_augmenter/Augmenter.py:141-158): full two-sided complex DFT of a real
 * [B, C, I, n] tensor along n, packed as [B, 2C, I, n] (channel 2c = Re, 2c+1 = Im).  n = n1 * n2 (four-step
 * DFT; n2 = 1 gives a direct DFT); `twiddle` is a device table of n {cos, -sin}(2 pi k / n) pairs. */
typedef struct { int B, C, I, n, n1, n2; } focal_fft_desc;
int focal_fft_realpack_fwd(const focal_fft_desc* d, const float* x, const float* twiddle, float* out, void* stream);
/* Same transform with one FOCAL view augmentation folded in (the data_augmenter package), applied to the whole
 * [B, C, I, n] tensor as the reference does per (location, modality):
 *   time domain, before the DFT:  x' = scale * x            NegationAugmenter.py:34 (scale = -1), ScalingAugmenter.py:35-36
 *                                 x'[i][s] = x[I-1-i][n-1-s]  HorizontalFlipAugmenter.py:34 (torch.flip dims 2, 3)
 *                                 x'[i] = x[perm[i]]          PermutationAugmenter.py:35-36 (one order for the whole batch)
 *   frequency domain, after it:   every bin times e^(i*angle)  PhaseShiftAugmenter.py:39-54
 * The reference applies exactly one of them per call; if several are set they compose as permute(flip(scale * x)), then the
 * rotation.  The random draws (coin flips, factor, order, angle) stay on the host, as in the reference. */
#define FOCAL_AUG_MAX_INTERVALS 32
typedef struct { float scale; int flip; int use_perm; int perm[FOCAL_AUG_MAX_INTERVALS]; float phase_cos, phase_sin; } focal_aug_desc;
int focal_augment_fft_fwd(const focal_fft_desc* d, const focal_aug_desc* a, const float* x, const float* twiddle, float* out,
                          void* stream);
/* n transforms (each with or without an augmentation: the two views x the modalities of a step) in ONE call: problems with short rows
 * (n <= 64 samples as a direct DFT, n2 == 1: the 20-sample sensor modalities) share one launch -- a thread per output bin, 256 / n
 * rows per workgroup pass, a problem table in the kernel arguments, up to 8 problems per launch -- the others are launched as by
 * focal_fft_realpack_fwd / focal_augment_fft_fwd.  Results are those of the single calls. */
typedef struct focal_view_plan focal_view_plan;
/* plan (round 5) non-NULL: the augmentation of this transform is read from that DEVICE record when the kernel runs (focal_view_draw
 * fills it inside the same captured step) instead of from `aug`; when the record says the view is warped, rows come from x_warped. */
typedef struct { focal_fft_desc d; int has_aug; focal_aug_desc aug; const float* x; const float* twiddle; float* out;
                 const focal_view_plan* plan; const float* x_warped; } focal_fft_problem;
int focal_fft_realpack_multi(int n, const focal_fft_problem* problems, void* stream);

/* ------------------------------------------------------------------------------------------------ view draws on the device (round 5)
 * The reference draws a view's augmentation in Python before every forward (data_augmenter/Augmenter.py:76-113 forward_random: ONE
 * augmenter of the pool per call, np.random.randint; each augmenter class then flips its coin per (location, modality), random() < p, and
 * draws its own parameters: ScalingAugmenter.py:35-36 N(1, std), PermutationAugmenter.py:35-36 torch.randperm(intervals),
 * PhaseShiftAugmenter.py:39-54 an angle in (-pi, pi), tsai's random-curve knots N(1, magnitude) for the two warps).  focal_view_draw
 * makes exactly these draws on the device from a counter RNG keyed by (seed word, stream id, view, slot) -- the seed word is the one the
 * optimizer advances every step -- and writes one focal_view_plan per (view, slot); slot = a (location, modality) pair.  Nothing of a
 * random view is left on the host: the draw, the warp tables (focal_warp_plan_fwd) and the transform (focal_fft_realpack_multi with
 * `plan`) are launches of fixed shape, so the whole view generation sits inside the captured step. */
#define FOCAL_VIEW_MAX_KNOTS 16
#define FOCAL_VIEW_MAX_POOL 8
#define FOCAL_VIEW_MAX_SLOTS 8
enum { FOCAL_VIEW_NONE = 0, FOCAL_VIEW_NEGATION = 1, FOCAL_VIEW_SCALING = 2, FOCAL_VIEW_HFLIP = 3, FOCAL_VIEW_PERMUTATION = 4,
       FOCAL_VIEW_PHASE_SHIFT = 5, FOCAL_VIEW_MAG_WARP = 6, FOCAL_VIEW_TIME_WARP = 7 };
struct focal_view_plan {
  focal_aug_desc aug;                  /* what the transform applies: the identity when the coin flip missed */
  int kind;                            /* the augmenter drawn for this (view, slot): FOCAL_VIEW_*, 0 = the coin flip missed */
  int pool_index;                      /* which pool entry the view drew (the same for all its slots) */
  int warp;                            /* 0, FOCAL_VIEW_MAG_WARP or FOCAL_VIEW_TIME_WARP: the transform reads the warped copy */
  int nknots;                          /* 3 (ord - 1) + 1 */
  float knots[FOCAL_VIEW_MAX_KNOTS];   /* the warp's random-curve ordinates */
};
typedef struct {
  int n_aug;                                                   /* pool size: one entry is drawn per view, uniformly */
  int kind[FOCAL_VIEW_MAX_POOL]; float prob[FOCAL_VIEW_MAX_POOL]; /* FOCAL_VIEW_* and the per-slot coin of each entry */
  float scaling_std, mag_magnitude, time_magnitude;
  int mag_order, time_order;                                   /* spline `ord` of the warps (tsai defaults 4 / 6) */
  int intervals[FOCAL_VIEW_MAX_SLOTS];                         /* per slot: the interval count the permutation shuffles */
} focal_view_pool;
/* plans: device [n_views][n_slots].  seed: the device seed word (NULL = 0); stream_id separates this draw from the dropout streams. */
int focal_view_draw(const focal_view_pool* pool, int n_views, int n_slots, const uint32_t* seed, uint32_t stream_id, focal_view_plan* plans,
                    void* stream);
/* The same draws from a state of their own (round 6): view_state = 4 device words {seed, draw count, 0, 0} that the kernel advances after
 * drawing (seed <- mix32(seed + golden), count += 1), so consecutive calls -- consecutive replays of a captured step -- draw different
 * views without any other kernel touching the words.  Under data parallelism every rank holds a copy started from ONE broadcast seed: the
 * ranks then draw identical plans, i.e. the global batch gets one augmenter / coin / permutation / scale / phase per view as the
 * reference's batch does (Augmenter.py:76-113, ScalingAugmenter.py:35 size=(1,), PermutationAugmenter.py:35-36), while the dropout
 * streams (keyed by the word the optimizer advances) stay per-rank. */
int focal_view_draw_shared(const focal_view_pool* pool, int n_views, int n_slots, uint32_t* view_state, uint32_t stream_id, focal_view_plan* plans,
                           void* stream);
/* The warps of up to 8 (view, slot) pairs whose plans may ask for one, as TWO launches (the tables of all of them, then the passes over
 * their windows); a problem whose plan says "no warp" costs nothing and leaves its y unwritten.  Per problem: the random curve through the
 * plan's knots and, for the time warp, its cumulative positions (one workgroup per problem; focal_amd/warp.py is the host statement of the
 * same arithmetic), then x [rows][L] -> y with the 24 resampling weights of focal_warp_fwd's tables formed per position by the pass.  end_coef: device fp32 [47][4][48], the
 * not-a-knot basis splines of a 48-sample end window (focal_amd.warp.end_window_coefficients(), a constant of the method); tables:
 * device workspace of 2 L floats per problem (multipliers, or floor | fraction of the positions), 16-byte aligned. */
typedef struct { int rows, L; const float* x; const focal_view_plan* plan; float* tables; float* y; } focal_warp_problem;
int focal_warp_plan_multi(int n, const focal_warp_problem* problems, const float* end_coef, void* stream);

/* TimeWarp / MagWarp (data_augmenter/TimeWarpAugmenter.py:18,44, MagWarpAugmenter.py:18,44 -> tsai 0.3.7 TSTimeWarp / TSMagWarp;
 * SURVEY 8f rank 1): one smooth random curve per call over the flattened (I*S) axis of [B, C, I, S], shared by batch and channels;
 * x, y fp32 [rows = B*C][L = I*S].  Exactly one of the two table sets:
 *   mult [L]                      y[r][n] = x[r][n] * mult[n]                                  (magnitude warp)
 *   k0 [L] int32, w [L][taps]     y[r][n] = sum_t w[n][t] * x[r][clamp(k0[n] + t, 0, L-1)]     (time warp: the interpolating cubic
 *                                 spline of each row evaluated at the warped positions, cardinal form; taps = 24)
 * The curve (a dozen Gaussian knots) and the tables are drawn / built on the host, as the reference draws them in numpy. */
int focal_warp_fwd(int rows, int L, const float* x, const float* mult, const int* k0, const float* w, int taps, float* y, void* stream);

/* Mixup / CutMix of the supervised `fixed` pipeline (data_augmenter/MixupAugmenter.py -> input_utils/mixup_utils.py:252-281, mode
 * "random_batch"): x, y fp32 [B, C, I, S] (y != x), perm int32 [B] = the ONE batch permutation of the call;
 *   cut = 0: y[b] = lam x[b] + (1 - lam) x[perm[b]];   cut = 1: y[b][:, yl:yh, xl:xh] = x[perm[b]][:, yl:yh, xl:xh], y[b] = x[b] elsewhere. */
int focal_mixup_fwd(int B, int C, int I, int S, const float* x, const int* perm, float lam, int cut, int yl, int yh, int xl, int xh,
                    float* y, void* stream);

/* ------------------------------------------------------------------------------------------------ row 8: embed
 * SW_Transformer.pad_input + PatchEmbed (models/SW_Transformer.py:184-208, models/SwinModules.py:547-558):
 * zero-pad [B, cin, I, S] to (Hp*1, Wp*pw), Conv2d(cin -> C0, kernel = stride = [1, pw]), flatten, LayerNorm.
 * Output tokens fp32 [B, Hp*Wp, C0].  Frozen in pretraining (general_utils/weight_utils.py:85-94): no backward. */
typedef struct { int B, cin, I, S, Hp, Wp, pw, C0; float eps; } focal_embed_desc;
int focal_pad_patch_embed_ln_fwd(const focal_embed_desc* d, const float* x, const float* w, const float* b,
                                 const float* gamma, const float* beta, float* tokens, void* stream);
/* The same, plus the LayerNorm that reads these tokens next (block 0's norm1, SwinModules.py:253): y_ln = LayerNorm(tokens;
 * gamma2, beta2, eps2) in `ln_dtype` and stats fp32 [tokens][2] = {mean, rstd} from the same kernel. */
int focal_pad_patch_embed_ln2_fwd(const focal_embed_desc* d, const float* x, const float* w, const float* b,
                                  const float* gamma, const float* beta, float* tokens, const float* gamma2, const float* beta2,
                                  float eps2, int ln_dtype, void* y_ln, float* stats, void* stream);

/* ------------------------------------------------------------------------------------------------ LayerNorm
 * nn.LayerNorm(eps 1e-5) of models/SwinModules.py:253,258,376.  x fp32 [rows, C]; y `dtype`; stats fp32
 * [rows, 2] = {mean, rstd}.  gather = 1 is PatchMerging's 2x2 gather fused in front of its LayerNorm
 * (SwinModules.py:388-399): x is [B, H, W, Cin], rows = B*(H/2)*(W/2), C = 4*Cin.
 * Backward: dx (fp32) += or = the input gradient (scattered back to [B, H, W, Cin] when gather = 1);
 * dgamma / dbeta (fp32 [C]) are accumulated (+=).
 * dx_masked / mask (both optional): additionally writes dtype(dx * mask) -- dx being the COMPLETED residual-stream
 * gradient -- where mask is the dropout x drop-path mask of the residual branch that consumes dx next (the forward
 * epilogue's index convention); that branch's dX / dW GEMMs then take it as a plain `dtype` dy (y_dtype = dtype, no
 * residual epilogue in their descriptor).  focal_mask_cast is the same product for a gradient no LayerNorm completes. */
typedef struct { int dtype; int rows, C; float eps; int gather; int B, H, W, Cin; } focal_ln_desc;
int focal_layernorm_fwd(const focal_ln_desc* d, const float* x, const float* gamma, const float* beta, void* y,
                        float* stats, void* stream);
int focal_layernorm_bwd(const focal_ln_desc* d, const void* dy, const float* x, const float* stats,
                        const float* gamma, float* dx, int accumulate_dx, float* dgamma, float* dbeta,
                        void* dx_masked, const focal_drop_desc* mask, void* stream);
int focal_mask_cast(int dtype, int rows, int C, const float* g, const focal_drop_desc* mask, void* out, void* stream);

/* ------------------------------------------------------------------------------------------------ Linear family
 * y[M, N] = epilogue( x[M, K] . w[N, K]^T + bias ), i.e. nn.Linear of models/SwinModules.py:22-34,113-116,375 and
 * models/SW_Transformer.py:121-124,157-161, with the element-wise neighbours of each call site fused:
 *   epilogue FOCAL_EPI_RESIDUAL  y = resid + out_drop(DropPath(.)): proj_drop / Mlp.drop + drop_path + shortcut
 *            FOCAL_EPI_GELU      y = out_drop(gelu_erf(.)) AND act_grad = d y / d(.)  (Mlp.act + Mlp.drop; the
 *                                derivative is emitted once here so that no pass ever re-evaluates erf)
 *            FOCAL_EPI_RELU      y = relu(.)
 *   act_in   FOCAL_ACT_GELU      x came out of an upstream FOCAL_EPI_GELU linear: bwd_data multiplies dx by that
 *                                linear's act_grad, which is passed in the `x` slot
 *            FOCAL_ACT_RELU_OUT  x is a ReLU output: bwd_data masks dx by x > 0
 * x_dtype / y_dtype are FOCAL_F32 or `dtype`; w is `dtype` (the bf16 shadow of the fp32 master in bf16 mode).
 * splits > 1 = split-K with fp32 atomics (y must be zeroed, fp32, epilogue NONE).
 * bwd_data  : dx = [(dy * out_mask) . w] (* act_grad | * (x > 0));  dy has y's dtype, dx has x's dtype.
 * bwd_weight: dw[N, K] += (dy * out_mask)^T . x;  dbias[N] += column sums (either may be NULL). */
enum { FOCAL_ACT_NONE = 0, FOCAL_ACT_GELU = 1, FOCAL_ACT_RELU_OUT = 2 };
enum { FOCAL_EPI_NONE = 0, FOCAL_EPI_RESIDUAL = 1, FOCAL_EPI_RELU = 2, FOCAL_EPI_GELU = 3 };
typedef struct {
  int dtype;
  int M, N, K;
  int x_dtype, y_dtype;
  int act_in, epilogue;
  int splits;
  focal_drop_desc out_drop;
  int dw_workgroups;  /* bwd_weight only: workgroups the token-split plan aims at; 0 = the default (~512: one launch fills the chip).  A
                       * caller that runs several passes side by side on their own streams (the DeepSense engine: two views x two
                       * modalities) asks for ~192: the launches share the chip anyway, and a third of the workgroups is a third of the
                       * fp32 atomics of their small outputs */
} focal_linear_desc;
int focal_linear_fwd(const focal_linear_desc* d, const void* x, const void* w, const float* bias, const float* resid,
                     void* y, void* act_grad, void* stream);
/* y = resid + out_drop(DropPath(x w^T + bias)) (epilogue FOCAL_EPI_RESIDUAL, fp32 y) AND, from the same kernel, the LayerNorm
 * that follows it in the Swin block (SwinModules.py:253,258: norm2 after the attention branch, the next block's norm1 after
 * the MLP branch): y_ln = LayerNorm(y; gamma, beta, eps) in `dtype`, stats fp32 [M][2] = {mean, rstd}.  One wave must own whole
 * rows: N = 64, or 128 / 256 in bf16 (focal_linear_resid_ln_supported); other widths: focal_linear_fwd followed by focal_layernorm_fwd.  Saves one pass over the residual
 * stream and one launch per LayerNorm. */
int focal_linear_resid_ln_fwd(const focal_linear_desc* d, const void* x, const void* w, const float* bias, const float* resid,
                              float* y, const float* gamma, const float* beta, float eps, void* y_ln, float* stats, void* stream);
/* row widths the fused form takes: N == 64 (any dtype), N == 128 / 256 with bf16 operands and K % 64 == 0 (csrc/gemm_pipe.hpp with
 * row-complete wave tiles) */
int focal_linear_resid_ln_supported(int dtype, int N, int K);
int focal_linear_bwd_data(const focal_linear_desc* d, const void* dy, const void* w, const void* x, void* dx,
                          void* stream);
/* focal_linear_bwd_data for a layer whose input x came out of a LayerNorm (qkv after norm1, fc1 after norm2: SwinModules.py:253,258)
 * TOGETHER with that LayerNorm's backward (focal_layernorm_bwd with accumulate_dx = 1): g (fp32 [M, K], the residual-stream gradient) +=
 * dLN(dy . w; ln_x, ln_stats, ln_gamma); dgamma / dbeta += ...; g_masked (optional, `dtype`) = dtype(g * mask).  The [M, K] product never
 * reaches memory.  bf16, K = 64 or 128 (focal_linear_bwd_data_ln_supported); d describes the LINEAR layer (plain `dtype` x / y). */
/* (g == NULL: only dgamma / dbeta are produced -- the LayerNorm's input is a leaf nobody differentiates, e.g. norm1 of the first block behind
 * the frozen patch embedding in FOCAL pretraining; the residual-stream gradient is then neither read, updated nor re-cast) */
int focal_linear_bwd_data_ln(const focal_linear_desc* d, const void* dy, const void* w, const float* ln_x, const float* ln_stats,
                             const float* ln_gamma, float* g, float* dgamma, float* dbeta, void* g_masked, const focal_drop_desc* mask,
                             void* stream);
int focal_linear_bwd_data_ln_supported(int dtype, int N, int K);
int focal_linear_bwd_weight(const focal_linear_desc* d, const void* dy, const void* x, float* dw, float* dbias,
                            void* stream);
/* Number of workgroups focal_linear_bwd_weight launches for this descriptor (output tiles x token splits): lets a caller match its
 * calls against the launch shapes of a profiler trace (bench.py's in-step roofline).  0 = invalid descriptor. */
int focal_linear_bwd_weight_workgroups(const focal_linear_desc* d);
/* Which kernel that launch runs: 1 = focal_gemm_kernel (register-staged, any dtype / loader), 2 = focal_dw_ring_kernel (the LDS-DMA
 * ring for bf16 shapes made of whole 64-tiles, given 16-byte aligned operands; 512-thread workgroups of two token slices each).  0 = invalid descriptor. */
int focal_linear_bwd_weight_kernel(const focal_linear_desc* d);

#define FOCAL_DW_GROUP_MAX_PROBLEMS 20
/* The weight gradients of up to FOCAL_DW_GROUP_MAX_PROBLEMS linear layers in ONE launch (the four linears of a Swin block: qkv, proj, fc1, fc2 --
 * of one block, or of several consecutive blocks whose operands the caller keeps alive: round 6): per problem
 * dw[N, K] += dy[M, N]^T . x[M, K], dbias[N] += column sums of dy (dbias may be NULL) -- n calls of focal_linear_bwd_weight with
 * plain `dtype` operands (bf16; dy already carries its dropout mask), as one kernel on 128 x 128 output tiles
 * (csrc/gemm_dw_group.hpp).  Needs M % 64 == 0 and N, K % 128 == 0 per problem (focal_linear_bwd_weight_group_supported), 16-byte
 * aligned tensors.  `exclusive` != 0 promises that nothing else adds to this dw while the launch runs (another stream, another
 * problem of the same call): tiles that one workgroup owns then leave with plain read-add-write instead of atomics. */
typedef struct {
  const void* dy; const void* x; float* dw; float* dbias;
  int M, N, K;
  int exclusive;
} focal_dw_problem;
int focal_linear_bwd_weight_group(int dtype, int n, const focal_dw_problem* problems, void* stream);
int focal_linear_bwd_weight_group_supported(int dtype, int M, int N, int K);
/* Narrower problems -- M % 64 == 0 and N, K % 64 == 0 for EVERY problem, at most 4 of them (the qkv [192, 64] and proj [64, 64]
 * gradients of a 64-channel block) -- run as one launch of the 64 x 64 ring tiles behind a problem table (csrc/gemm_dw_ring.hpp,
 * always atomics).  focal_linear_bwd_weight_group_kind: 2 = a shape for the 128 x 128 group, 1 = for the 64-tile group only, 0 = neither. */
int focal_linear_bwd_weight_group_kind(int dtype, int M, int N, int K);
/* workgroups that launch consists of (profiler-trace matching, as focal_linear_bwd_weight_workgroups); 0 = invalid */
int focal_linear_bwd_weight_group_workgroups(int dtype, int n, const focal_dw_problem* problems);
/* The same for linears whose dy and x are FP32 tensors (DeepSense's nn.GRU: weight_ih / weight_hh of a layer's two directions,
 * models/RecurrentModule.py:5-31 -- dgi, dgh, the hidden states and the layer input are fp32): up to 8 problems of any shape as ONE launch
 * of the 64 x 64 tiles focal_linear_bwd_weight uses for them (operands rounded to `compute` on load, fp32 atomics; `exclusive` is not
 * used).  workgroups: the launch's target over all problems (0 = default), cf. focal_linear_desc.dw_workgroups. */
int focal_linear_bwd_weight_group_f32(int compute, int n, const focal_dw_problem* problems, int workgroups, void* stream);
int focal_linear_bwd_weight_group_f32_workgroups(int compute, int n, const focal_dw_problem* problems, int workgroups);

/* Fused MLP branch of a Swin block (models/SwinModules.py:18-34 Mlp.forward + the residual / DropPath of :339-341), bf16,
 * C = 64 -> hidden = 256 -> C (Swin stage 0, where 2/3 of the model's hidden-activation bytes are; focal_mlp_supported says
 * whether a shape is built):
 *   fwd: y = resid + drop_out( drop_hidden(gelu(a w1^T + b1)) w2^T + b2 ); the hidden activation never reaches HBM.  With y_ln
 *        non-NULL the LayerNorm that reads y next is emitted too (y_ln bf16 [M, C], ln_stats fp32 [M][2] = {mean, rstd}).
 *   bwd: gm = dtype(dL/dy x drop_out mask) (what focal_layernorm_bwd / focal_mask_cast emit as dx_masked); the hidden
 *        activation and its derivative are RECOMPUTED from `a` (dropout mask regenerated), da = dL/da (bf16), dw1 / db1 / dw2 /
 *        db2 accumulated (+=) in fp32 -- one pass over gm and a instead of four GEMMs over two saved [M, 4C] tensors.
 *        (Round 4: every wave owns 16 hidden units for the whole launch; the recomputed h / du re-enter the matrix cores from
 *        registers for the weight gradients and only du crosses LDS, for dL/da: focal_amd/csrc/mlp_bwd.hip.)
 *        With ln_x non-NULL the norm2 backward is fused behind it: da is not written; instead g (fp32 [M, C], the residual-
 *        stream gradient) += LayerNorm-backward(da; ln_x, ln_stats, ln_gamma), gm_next (bf16 [M, C], optional) = g x next_mask,
 *        dgamma / dbeta accumulated.
 * drop_hidden is a per-(row, lane group) xorshift stream seeded from (seed word, stream id, row); the forward kernel writes its keep
 * decisions to mask_bits (uint32 [M][8], one bit per hidden unit: 32 bytes per token; ignored when drop_hidden.p_elem == 0;
 * otherwise focal_mlp_bwd requires it, and focal_mlp_fwd takes NULL only from a caller that never differentiates the call -- a
 * forward-only pass with dropout on) and the backward kernel reads them back: the saved state of the branch is `a` plus these words.
 * drop_out follows the element / DropPath convention of FOCAL_EPI_RESIDUAL (over [M, C]). */
typedef struct { int dtype; int M, C, hidden; focal_drop_desc drop_hidden, drop_out; float ln_eps; } focal_mlp_desc;
int focal_mlp_supported(int dtype, int C, int hidden);
int focal_mlp_fwd(const focal_mlp_desc* d, const void* a, const float* resid, const void* w1, const float* b1, const void* w2,
                  const float* b2, float* y, const float* ln_gamma, const float* ln_beta, void* y_ln, float* ln_stats,
                  uint32_t* mask_bits, void* stream);
/* Round 6: the attention branch's tail in front of the MLP, in the same launch -- x_mid = x + drop_proj(o wp^T + bp) (the 64-channel proj
 * Linear, its Dropout, DropPath and the residual add: SwinModules.py:147, :336-338), a2 = norm2(x_mid) (:339), then the MLP branch as
 * focal_mlp_fwd.  x_mid [M, C] fp32, a2 [M, C] dtype and st2 [M, 2] are OUTPUTS (the backward pass reads them): everything
 * focal_linear_resid_ln_fwd + focal_mlp_fwd produce, bit-identical with the masks off, in one launch instead of two.
 * FOCAL_MLP_PROJ=0 makes focal_mlp_proj_supported return 0. */
int focal_mlp_proj_supported(int dtype, int C, int hidden);
int focal_mlp_proj_fwd(const focal_mlp_desc* d, const void* o, const float* x, const void* wp, const float* bp, const focal_drop_desc* drop_proj,
                       const float* g2, const float* bt2, float* x_mid, void* a2, float* st2, const void* w1, const float* b1, const void* w2,
                       const float* b2, float* y, const float* ln_gamma, const float* ln_beta, void* y_ln, float* ln_stats,
                       uint32_t* mask_bits, void* stream);
int focal_mlp_bwd(const focal_mlp_desc* d, const void* gm, const void* a, const void* w1, const float* b1, const void* w2,
                  void* da, float* dw1, float* db1, float* dw2, float* db2,
                  const float* ln_x, const float* ln_stats, const float* ln_gamma, float* g, void* gm_next,
                  const focal_drop_desc* next_mask, float* dgamma, float* dbeta, const uint32_t* mask_bits, float* dw_partials, void* stream);
/* dw_partials (round 5; NULL = every workgroup adds its 128 KB of weight gradients with fp32 atomics, 33.5 MB per launch at the memory
 * side's 1.3 TB/s): a device workspace of focal_mlp_bwd_partials_floats(d) floats, 16-byte aligned -- the workgroups store their images
 * there with plain stores and a second, small launch on the same stream sums them into dw1 / dw2. */
long focal_mlp_bwd_partials_floats(const focal_mlp_desc* d);
/* The same branch at 128 / 256 channels (Swin stages 1-2; round 6): ONE launch instead of the fc1 and fc2 launches of focal_linear_fwd /
 * focal_linear_resid_ln_fwd.  The weights (256 KB / 1 MB) stream through an LDS ring 64 hidden units at a time; a wave keeps its 16 token
 * rows' [16, C] fc2 accumulators in registers across the slices and feeds each slice's activation to fc2 straight from the fc1
 * accumulators, so the hidden tensor's tiles never leave the chip between the two products.  h and hg ([M, hidden], dtype: the activation
 * and its derivative x dropout mask) are still WRITTEN -- once -- because the backward pass of these widths reads them (a recomputing
 * backward needs 1 MB of weight-gradient accumulators per workgroup).  y_ln / ln_* (C = 128 only): the LayerNorm that reads y next, as
 * focal_linear_resid_ln_fwd emits it.  Every output is bit-identical to the two-launch form (same k order, same element math, same mask
 * indices; with the masks on, to an ulp where hipcc contracts the residual expression differently).  Back to back on cold operands it only
 * ties the two launches (profiles/r6_mlp_wide.txt); inside the replayed step it is worth +1.5 % (tools/ab_wide.sh).  FOCAL_MLP_WIDE=0 makes
 * focal_mlp_wide_supported return 0 (the engine then issues the two launches), =128 / =256 restrict it to one width. */
int focal_mlp_wide_supported(int dtype, int C, int hidden);
int focal_mlp_wide_bwd_supported(int dtype, int C, int hidden);
int focal_mlp_wide_fwd(const focal_mlp_desc* d, const void* a, const float* resid, const void* w1, const float* b1, const void* w2,
                       const float* b2, float* y, void* h, void* hg, const float* ln_gamma, const float* ln_beta, void* y_ln,
                       float* ln_stats, void* stream);
/* focal_mlp_proj_fwd's form for these widths (round 6): proj + residual + DropPath + norm2 of the attention branch in front of the MLP, in its
 * launch -- the proj weight rides the same ring, 64 output channels per step; x_mid, a2, st2 are outputs.  Bit-identical to
 * focal_linear_resid_ln_fwd (128) / focal_linear_fwd + focal_layernorm_fwd (256) followed by focal_mlp_wide_fwd with the masks off. */
int focal_mlp_wide_proj_supported(int dtype, int C, int hidden);
int focal_mlp_wide_proj_fwd(const focal_mlp_desc* d, const void* o, const float* x, const void* wp, const float* bp, const focal_drop_desc* drop_proj,
                            const float* g2, const float* bt2, float* x_mid, void* a2, float* st2, const void* w1, const float* b1, const void* w2,
                            const float* b2, float* y, void* h, void* hg, const float* ln_gamma, const float* ln_beta, void* y_ln,
                            float* ln_stats, void* stream);
/* The data path of the same branch's backward pass as one launch: du = (gm w2) x hg ([M, hidden], written once: fc1's weight gradient reads
 * it), then dc = du w1 = dL/da2, either stored (dc, dtype [M, C]; ln_x == NULL) or -- C = 128 -- finished on the row as the backward of
 * the LayerNorm that produced a2 (norm2), exactly as focal_linear_bwd_data_ln does: g += dLN (in place), g_masked = dtype(g x mask),
 * dgamma / dbeta accumulate.  Replaces focal_linear_bwd_data (GELU derivative) + focal_linear_bwd_data / _ln; the two weights are read
 * through the hardware transpose from the same ring the forward kernel uses.  Equal to the two launches up to the order inside an MFMA's
 * 32-term sum.  Measured inside the replayed step: neutral at 128 channels, -1.2 % at 256 (tools/ab_wide_bwd.sh): NOT the default --
 * focal_mlp_wide_bwd_supported answers 1 only under FOCAL_MLP_WIDE_BWD=1 (or =128 / =256 for one width); the entry point itself always works. */
int focal_mlp_wide_bwd_data(const focal_mlp_desc* d, const void* gm, const void* hg, const void* w1, const void* w2, void* du, void* dc,
                            const float* ln_x, const float* ln_stats, const float* ln_gamma, float* g, void* g_masked,
                            const focal_drop_desc* mask, float* dgamma, float* dbeta, void* stream);

/* ------------------------------------------------------------------------------------------------ row 10: W-MSA
 * WindowAttention between its qkv and proj Linears (models/SwinModules.py:121-152) with the cyclic shift, window
 * partition / reverse of SwinTransformerBlock.forward (:294-334) folded into the addressing: qkv is the token-major
 * [B*H*W, 3C] output of the qkv Linear in ORIGINAL token order, out is [B*H*W, C] in original order.
 * (sh, sw) > 0 on both axes = rolled windows + the -100 region mask of :262-289.  Attention dropout via rng. */
typedef struct {
  int dtype;
  int B, H, W, C, heads;
  int wh, ww, sh, sw;
  float p_attn;
  const uint32_t* rng;
  uint32_t stream;
} focal_attn_desc;
int focal_window_attn_fwd(const focal_attn_desc* d, const void* qkv, const float* bias_table, void* out, void* stream);
int focal_window_attn_bwd(const focal_attn_desc* d, const void* qkv, const float* bias_table, const void* dout,
                          void* dqkv, float* dbias_table, void* stream);
/* The same two operators with the qkv Linear (models/SwinModules.py:113, 128-130) folded in, for 64-channel blocks (bf16, 4 heads of
 * 16: Swin stage 0; focal_window_attn_qkv_supported): a1 [B*H*W, C] is the output of norm1, wqkv [3C, C] / bqkv [3C] the qkv layer.  A
 * wave's head is fixed, so its 48 rows of wqkv are loop-invariant MFMA operands and q / k / v of a (window, head) item are projected
 * from the window's a1 rows inside the kernel: the [B*H*W, 3C] qkv tensor never reaches HBM -- the forward pass saves a1 only, the
 * backward pass recomputes (and still emits dqkv for the layer's weight / input gradients).  Results equal focal_linear_fwd +
 * focal_window_attn_fwd / _bwd up to the accumulation order of the projection. */
int focal_window_attn_qkv_supported(int dtype, int C, int heads, int window_tokens);
int focal_window_attn_qkv_fwd(const focal_attn_desc* d, const void* a1, const void* wqkv, const float* bqkv, const float* bias_table,
                              void* out, void* stream);
/* _bwd: wproj == NULL: dout = dL/d(attention output) as in focal_window_attn_bwd; wproj [C, C] (the proj Linear's weight): dout = dL/d(proj
 * output) [B*H*W, C] (already x the branch's dropout / drop-path mask) and the kernel forms its head's slice of dout . wproj per item itself --
 * the proj layer's input-gradient launch and the [B*H*W, C] tensor between the two are gone. */
int focal_window_attn_qkv_bwd(const focal_attn_desc* d, const void* a1, const void* wqkv, const float* bqkv, const float* bias_table,
                              const void* dout, const void* wproj, void* dqkv, float* dbias_table, void* stream);


/* ------------------------------------------------------------------------------------------------ row 5: DeepSense convs
 * ConvBlock (models/ConvModules.py:115-216) on CHANNEL-LAST tokens: activation [B*I*S, C] (row = (b, interval, s)).
 * in-conv: Conv2d(cin -> C, [1,k], stride [1,stride], zero pad `pad_left`) read straight from the reference's NCHW fp32
 * spectrum [B, cin, I, S_in]; fp32; its input is a leaf, so only the weight gradient exists.  w: [C, cin, 1, k]. */
typedef struct { int B, cin, I, S_in, S_out, k, stride, pad_left, C; } focal_conv_in_desc;
int focal_conv_in_fwd(const focal_conv_in_desc* d, const float* x, const float* w, const float* bias, float* z, void* stream);
int focal_conv_in_bwd_weight(const focal_conv_in_desc* d, const float* x, const void* dz, int dz_dtype, float* dw, float* dbias,
                             void* stream);
/* [1,k] "same" convs (k odd) = MFMA GEMMs over a sliding token window.  Operand orders: w_fwd [C_out][k][C_in],
 * w_bwd [C_in][k][C_out] with taps flipped, both `dtype`, produced from the reference layout [C_out][C_in][1][k] by
 * focal_permute_pack / focal_conv_pack_bwd (the flatten + Conv1d(1x1) output layer, ConvModules.py:207-216, is
 * focal_linear_* on weights re-ordered from (c*S + s) to (s*C + c) by focal_permute_pack).
 * fwd: z (fp32) = conv(x) + bias.  bwd_data: g_out = g_in + conv^T(dz) (fp32; the residual-gradient stream).
 * bwd_weight: dw_packed [C_out][k][C_in] += dz^T window(x); dbias += column sums; focal_permute_unpack_add folds
 * dw_packed back into the [C_out][C_in][1][k] gradient. */
typedef struct { int dtype; int rows, S, C_in, C_out, k; int dw_workgroups; /* bwd_weight: as focal_linear_desc.dw_workgroups */ } focal_conv_desc;
int focal_permute_pack(int A, int Bd, int Cd, const float* src, void* dst, int dtype, void* stream);      /* dst[a][c][b] = src[a][b][c] */
int focal_permute_unpack_add(int A, int Bd, int Cd, const float* src, float* dst, void* stream);          /* dst[a][b][c] += src[a][c][b] */
int focal_conv_pack_bwd(const focal_conv_desc* d, const float* w, void* w_bwd, void* stream);
/* Several re-orderings in one launch (an encoder's 13 per pass): kind FOCAL_PACK_PERMUTE = focal_permute_pack's dst[a][c][b] = src[a][b][c]
 * (A, B, C as there), FOCAL_PACK_CONV_BWD = focal_conv_pack_bwd's dst[ci][t][co] = src[co][ci][k-1-t] with (A, B, C) = (C_out, C_in, k);
 * src fp32, dst `dtype`.  focal_unpack_add_multi: dst[a][b][c] += src[a][c][b] per entry (fp32; focal_permute_unpack_add). */
enum { FOCAL_PACK_PERMUTE = 0, FOCAL_PACK_CONV_BWD = 1, FOCAL_PACK_FRAG = 2, FOCAL_PACK_FRAG_T = 3 };
/* FOCAL_PACK_FRAG: the matrix src[A][B] (C = 1; A % 16 == 0, B % 32 == 0) in MFMA-fragment order:
 *   dst[((r / 16) * (B / 32) + c / 32) * 512 + ((c % 32) / 8 * 16 + r % 16) * 8 + c % 8] = src[r][c];
 * FOCAL_PACK_FRAG_T: the same of its TRANSPOSE m[r][c] = src[c][r] (an [B][A] matrix: B % 16 == 0, A % 32 == 0). */
#define FOCAL_PACK_MAX 24
typedef struct { const void* src; void* dst; int A, B, C; int kind; } focal_pack_entry;
int focal_pack_multi(int dtype, int n, const focal_pack_entry* entries, void* stream);
int focal_unpack_add_multi(int n, const focal_pack_entry* entries, void* stream);
int focal_conv_fwd(const focal_conv_desc* d, const void* x, const void* w_fwd, const float* bias, float* z, void* stream);
int focal_conv_bwd_data(const focal_conv_desc* d, const void* dz, const void* w_bwd, const float* g_in, float* g_out, void* stream);
int focal_conv_bwd_weight(const focal_conv_desc* d, const void* dz, const void* x, float* dw_packed, float* dbias, void* stream);

/* BatchNorm2d(eps 1e-5, momentum 0.1) + GELU + Dropout2d + residual of ConvLayer2D (ConvModules.py:98-112,203-204).
 * focal_bn_stats: training -> batch mean / biased variance of z [rows, C] into mean_rstd [2C] and the momentum update of the
 * running buffers (unbiased variance); eval -> mean_rstd from the running buffers.  scratch: 2C + 1 floats (the sums and an arrival counter:
 * in FOCAL_BN_TRAIN mode the last workgroup of the statistics kernel finalises mean / rstd / running buffers itself).
 * focal_bn_act_fwd: y = resid + drop2d(gelu(gamma * zhat + beta)) (fp32), optional `dtype` copy y_cast for the next GEMM.
 * focal_bn_act_bwd: dz (`dtype`) from g = dL/dy (fp32); dgamma / dbeta accumulated (+=).
 * Data-parallel exact ("sync") statistics: the per-channel sums are exposed so that the caller can all-reduce the 2C
 * floats of `scratch` between the two halves of each call --
 *   focal_bn_stats   training = FOCAL_BN_PARTIAL: scratch <- local {sum z, sum z^2};   FOCAL_BN_FINALIZE: mean_rstd and
 *                    running buffers from scratch, normalising by d->stat_rows (the GLOBAL row count);
 *   focal_bn_act_bwd phase = FOCAL_BN_PARTIAL: scratch <- local {sum da, sum da*zhat}, dgamma / dbeta += the LOCAL sums;
 *                    FOCAL_BN_FINALIZE: dz from the (all-reduced) scratch and d->stat_rows.
 * stat_rows = 0 means rows. */
typedef struct { int dtype; int rows, C, rows_per_sample; float eps, momentum, p_drop; const uint32_t* rng; uint32_t stream; int stat_rows; int groups; } focal_bn_desc;
/* groups (0 or 1: one): the tensor's `rows` are `groups` equal, consecutive row ranges with batch statistics of their OWN -- the two
 * augmented views of a FOCAL step as one batch of 2B windows (models/FOCALModules.py:21-34 calls the backbone once per view, so
 * every BatchNorm2d normalises each view by itself, ConvModules.py:86).  With groups > 1 every per-call array is `groups` copies of its
 * one-group layout, group after group: scratch, mean_rstd [groups][2C], running_mean / running_var [groups][C] (sinks that RECORD each
 * group's statistics: d->momentum = 1, then focal_bn_running_combine); gamma / beta / dgamma / dbeta stay [C] (the gradients are the sum
 * over the groups).  Dropout2d samples are numbered through the whole tensor.  Training mode on one rank only (stat_rows = 0). */
enum { FOCAL_BN_EVAL = 0, FOCAL_BN_TRAIN = 1, FOCAL_BN_PARTIAL = 2, FOCAL_BN_FINALIZE = 3 };
/* OR into `training` / `phase`: the caller guarantees `scratch` holds zeros (e.g. a slice of a per-step zeroed pool), so the call
 * does not enqueue its own memset (one launch less per BatchNorm pass). */
#define FOCAL_BN_SCRATCH_ZEROED 16
int focal_bn_stats(const focal_bn_desc* d, const float* z, float* scratch, float* mean_rstd, float* running_mean, float* running_var,
                   int training, void* stream);
/* focal_conv_fwd AND the training-mode statistics of the BatchNorm that follows it (ConvModules.py:54-112: conv -> BatchNorm2d), from the
 * convolution's epilogue: what focal_bn_stats(bn, z, ..., FOCAL_BN_TRAIN) would compute, without reading z back and without its launch.
 * bf16 operands.  scratch: FOCAL_BN_STAT_SLOTS x 2C + 1 floats, ZERO on entry (column sums in 16 slots + the arrival counter of the last-
 * workgroup finalisation); mean_rstd [2C] out; running_mean / running_var updated with bn->momentum (both NULL: not touched). */
#define FOCAL_BN_STAT_SLOTS 16
int focal_conv_fwd_bn(const focal_conv_desc* d, const void* x, const void* w_fwd, const float* bias, float* z,
                      const focal_bn_desc* bn, float* scratch, float* mean_rstd, float* running_mean, float* running_var, void* stream);
/* Round 6: the sums-only form.  Where focal_conv_fwd_bn_sums_supported(...) != 0 (64 -> 64 channels, k = 3 / 5, bf16, whole 64-row tiles per
 * statistics group: csrc/conv_ring.hpp), focal_conv_fwd_bn may be called with mean_rstd = NULL (running buffers ignored): the launch then only
 * adds its per-channel sums into `scratch` and ends, and focal_bn_act_fwd_sums -- focal_bn_act_fwd reading `scratch` instead of mean_rstd --
 * finishes the statistics in its prologue (mean_rstd [groups x 2C] out for the backward pass, running buffers updated with d->momentum).  Same
 * arithmetic as the one-launch form; the convolution loses three dependent memory-side round trips (~9 us per launch). */
int focal_conv_fwd_bn_sums_supported(const focal_conv_desc* d, const focal_bn_desc* bn, const void* x, const void* w_fwd);
int focal_bn_act_fwd_sums(const focal_bn_desc* d, const float* z, const float* sums, float* mean_rstd, float* running_mean, float* running_var,
                          const float* gamma, const float* beta, const float* resid, float* y, void* y_cast, void* stream);
/* The running-buffer updates of TWO passes (the two augmented views of a FOCAL step), applied after both have run: each pass records
 * its batch statistics instead of updating the buffers (focal_bn_stats with d->momentum = 1 and a per-pass sink in place of the running
 * buffers), this applies r <- (1 - m) ((1 - m) r + m s1) + m s2 to n buffers of C values in one launch -- what the reference's two
 * sequential backbone calls leave behind (models/FOCALModules.py:21-34), without ordering the passes. */
#define FOCAL_BN_COMBINE_MAX 24
int focal_bn_running_combine(int n, float* const* running, const float* const* view1, const float* const* view2, int C, float momentum,
                             void* stream);
int focal_bn_act_fwd(const focal_bn_desc* d, const float* z, const float* mean_rstd, const float* gamma, const float* beta,
                     const float* resid, float* y, void* y_cast, void* stream);
int focal_bn_act_bwd(const focal_bn_desc* d, const float* z, const float* g, const float* mean_rstd, const float* gamma,
                     const float* beta, float* scratch, void* dz, float* dgamma, float* dbeta, int phase, void* stream);

/* ------------------------------------------------------------------------------------------------ row 6: bi-GRU
 * nn.GRU(2 layers, bidirectional) + time mean (models/RecurrentModule.py:5-31).  Matrix products go through
 * focal_linear_*; these are the per-step gate kernels.  gi: [B*T, 3H] (rows (b, t), b_ih included), gh: [B, 3H] (b_hh
 * included), out: [B, T, 2H], save: [4][B][H] per step (r, z, n, W_hn h + b_hn).
 * bwd: dh = scale * dout[b*ld_b + t*ld_t + dir_offset + j] + dh_rec + dhz_in; writes dgi rows (b, t), dgh [B, 3H], dhz_out. */
typedef struct { int B, T, H; int whh_frag; } focal_gru_desc;
/* whh_frag (focal_gru_seq_fwd / _bwd only; 0 = as documented below): the W_hh operands are given in MFMA-fragment order instead of
 * row-major -- 16-row x 32-column tiles, tile-major, inside a tile the 64 lanes' 16-byte pieces back to back
 * (FOCAL_PACK_FRAG / FOCAL_PACK_FRAG_T of focal_pack_multi): every load instruction of the kernels' prologue then reads one contiguous KB
 * instead of sixteen 64-byte row segments. */
int focal_gru_gate_fwd(const focal_gru_desc* d, int t, int dir_offset, const float* gi, const float* gh, const float* h_prev,
                       float* h_new, float* out, float* save, void* stream);
int focal_gru_gate_bwd(const focal_gru_desc* d, int t, int dir_offset, const float* dout, long ld_b, long ld_t, float scale,
                       const float* dh_rec, const float* dhz_in, const float* save, const float* h_prev, float* dgi, float* dgh,
                       float* dhz_out, void* stream);
/* Whole-sequence form (bf16 W_hh; H in {128, 256}): ONE launch runs all T steps of one layer for n_dir (1 or 2) directions
 * (direction 0 walks t = 0..T-1, direction 1 t = T-1..0 and writes out[..., H:2H]).  Per-direction HOST arrays of device
 * pointers: gi [B*T, 3H] (b_ih included), whh bf16 [3H, H], bhh [3H], hs [T+1, B, H] (hs[0] = h0, written from hs[1]),
 * save [T, 4, B, H] indexed by step s.  Backward: whh_t = bf16 [H, 3H] (the transposed W_hh), dgi [B*T, 3H], dgh [T, B, 3H]
 * (step-indexed) are written for the weight / input gradient GEMMs; dout as in focal_gru_gate_bwd.
 * Tensors must stay below 4 GB (32-bit byte offsets).  A batch that is not a multiple of the kernel's samples per workgroup (8 while
 * ceil(B / 8) * n_dir <= 128 at H = 256, else 16) is handled by recomputing the last sample in the spare lanes. */
int focal_gru_seq_fwd(const focal_gru_desc* d, int n_dir, const float* const* gi, const void* const* whh, const float* const* bhh,
                      float* const* hs, float* const* save, float* out, void* stream);
int focal_gru_seq_bwd(const focal_gru_desc* d, int n_dir, const float* dout, long ld_b, long ld_t, float scale,
                      const void* const* whh_t, const float* const* hs, const float* const* save, float* const* dgi,
                      float* const* dgh, void* stream);
int focal_mean_time(int B, int T, int D, const float* x, float* y, void* stream);                       /* y[b] = mean_t x[b][t] */
int focal_dropout(long n, const float* x, float* y, const uint32_t* rng, uint32_t stream_id, float p, void* stream); /* y = x * mask */
int focal_axpy(long n, float a, const float* x, float* y, void* stream);                                   /* y += a * x */
int focal_mul(long n, const float* a, float* y, void* stream);                                             /* y *= a (element-wise, fp32) */

/* ------------------------------------------------------------------------------------------------ classifier head (8f rank 4)
 * Finetuning path, `backbone(freq_x, class_head=True)` (models/SW_Transformer.py:269-276, models/FusionModules.py:61-140):
 * focal_fusion_attn_*: the attention core of TransformerFusionBlock -- one query per sample (the mean of the M fused tokens,
 * after in_proj) over the M modality tokens, nn.MultiheadAttention semantics with head_dim 64: q [B, E] (in_proj applied),
 * kv [B*M, 2E] rows (b, j) = {k | v}, out [B, E] (before out_proj); probs / weights [B, heads, M] = softmax and softmax x
 * attention-dropout mask (saved for backward).  Backward writes dq [B, E] and dkv [B*M, 2E].
 * focal_cross_entropy: nn.CrossEntropyLoss(), mean reduction: loss[0] and dlogits = d loss / d logits in one launch. */
int focal_fusion_attn_fwd(int B, int M, int E, int heads, const float* q, const float* kv, float* out, float* probs, float* weights,
                          const uint32_t* rng, uint32_t stream_id, float p_drop, void* stream);
int focal_fusion_attn_bwd(int B, int M, int E, int heads, const float* q, const float* kv, const float* probs, const float* weights,
                          const float* dout, float* dq, float* dkv, void* stream);
int focal_cross_entropy(int B, int C, const float* logits, const long* labels, float* loss, float* dlogits, void* stream);
/* The class layer nn.Linear(K -> n_cls) (a few output columns, fp32): y = x w^T + bias; backward accumulates dw / dbias (+=) and
 * writes dx when it is non-NULL. */
int focal_small_linear_fwd(int B, int N, int K, const float* x, const float* w, const float* bias, float* y, void* stream);
int focal_small_linear_bwd(int B, int N, int K, const float* dy, const float* x, const float* w, float* dw, float* dbias, float* dx,
                           void* stream);

/* ------------------------------------------------------------------------------------------------ rows 11-13: loss
 * FOCALLoss.forward (models/loss.py:139-218): 2 InfoNCE families on the shared / private halves, orthogonality,
 * temporal ranking.  feats / dfeats are HOST arrays of 2*n_mod device pointers, view-major
 * ([view0 mod0, view0 mod1, .., view1 mod0, ..]), each fp32 [B, dim]; B = b*seq whole subsequences.
 * terms (device fp32[5]) = {shared, private, orth, rank, weighted total}; dfeats receive d total / d feat.
 * Limit: n_mod <= 4 (the problem tables of the head's grouped launches travel as kernel arguments: n_mod^2 <= 16 InfoNCE problems and
 * n_mod (n_mod + 1) <= 20 orthogonality problems fit the 4 KB argument segment; a fifth modality is rejected with FOCAL_EINVAL
 * "too many modality pairs" -- the reference's datasets have 2 (MOD) and the authored HAR4 config 4). */
typedef struct {
  int n_mod, B, dim, seq;
  float temperature, margin;
  float w_shared, w_private, w_orth, w_rank;
  int no_private;  /* args.tag == "noPrivate": shared InfoNCE on the full features (loss.py:163-170) */
} focal_loss_desc;
size_t focal_loss_head_workspace(const focal_loss_desc* d);
int focal_loss_head(const focal_loss_desc* d, const float* const* feats, float* terms, float* const* dfeats,
                    void* workspace, size_t workspace_bytes, void* stream);
/* The same loss evaluated by `world` data-parallel ranks, each over the rows of its own samples (round 2).  `d` describes the GLOBAL
 * batch (B = world x the rank's batch, rank-major sample order as an all-gather leaves it; b = B / seq must divide by world); feats
 * are the gathered [B, dim] embeddings, identical on every rank.  Every cross-sample matrix of the loss is symmetric and a sample's
 * gradient needs only its own row of it, so rank r computes 1 / world of every product and row pass:
 *   _shard_a   zeroes dfeats / terms, evaluates the similarity rows, log-sum-exps, distances, block means and hinges of the rank's
 *              rows and the orthogonality term, and writes what other ranks need of them -- lse of its rows, diagonal block means,
 *              its partial loss terms -- into `chunk` (focal_loss_head_exchange_floats(d, world) floats);
 *   (caller)   all-gathers the chunks in rank order -> `chunks` [world][that many floats]   (the ONE collective of the head: ~70 KB);
 *   _shard_b   finishes: coefficient rows, dL/dz of the rank's own samples into their rows of dfeats (other rows stay zero), and the
 *              global terms[5] (sum of the partial terms).
 * Same workspace in both calls (focal_loss_head_workspace(d) bytes), untouched in between.  focal_loss_head is world = 1. */
size_t focal_loss_head_exchange_floats(const focal_loss_desc* d, int world);
int focal_loss_head_shard_a(const focal_loss_desc* d, int rank, int world, const float* const* feats, float* terms,
                            float* const* dfeats, float* chunk, void* workspace, size_t workspace_bytes, void* stream);
int focal_loss_head_shard_b(const focal_loss_desc* d, int rank, int world, const float* const* feats, float* terms,
                            float* const* dfeats, const float* chunks, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------ row 14: AdamW
 * torch.optim.AdamW (train_utils/optimizer.py:27-32) over `nseg` contiguous fp32 segments (the parameter arena's
 * trainable region; grad-less parameters are simply not listed, which is how torch skips them).  lr comes from a
 * device scalar and the step count from rng_state[1] (1-based after focal_rng_advance) so a captured graph stays
 * valid across epochs.  shadow (nullable) receives the bf16 copy of the updated weights for the matrix cores. */
typedef struct { float beta1, beta2, eps, weight_decay; int l2_decay; /* 0: decoupled decay (torch.optim.AdamW); 1: L2 form, g += wd * p (torch.optim.Adam, the finetune optimizer) */ } focal_adamw_desc;
int focal_adamw_multi(const focal_adamw_desc* d, int nseg, float* const* p, const float* const* g, float* const* m,
                      float* const* v, void* const* shadow_bf16, const long* n, const float* lr_dev,
                      const uint32_t* rng_state, void* stream);
/* The same update with the step's bookkeeping folded in: the step count used is step_state[1] + 1, and the workgroup that finishes last
 * advances step_state (as focal_rng_advance would have before the call) and, when non-NULL, seed_state (the dropout seed words of the
 * next forward pass) -- two one-thread launches less on the serial tail of every step.  step_state holds FOCAL_STEP_STATE_WORDS
 * 32-bit words: {seed, step count, ticket, -, -, -, -, -, 32 group tickets}; the tickets are zero between calls; step_state_words = the
 * length of the caller's buffer, checked against FOCAL_STEP_STATE_WORDS (the 4-word state of focal_rng_advance has the same pointer
 * type and would be written out of bounds).  seed_state: the 4 words of focal_rng_advance. */
#define FOCAL_STEP_STATE_WORDS 40
int focal_adamw_multi_advance(const focal_adamw_desc* d, int nseg, float* const* p, const float* const* g, float* const* m,
                              float* const* v, void* const* shadow_bf16, const long* n, const float* lr_dev,
                              uint32_t* step_state, int step_state_words, uint32_t* seed_state, void* stream);
/* fp32 -> bf16 cast of a weight segment (refreshing the shadow after load_state_dict) */
int focal_cast_bf16(const float* src, void* dst, long n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FOCAL_HIP_H */
