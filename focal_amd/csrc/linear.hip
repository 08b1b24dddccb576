// focal_linear_{fwd,bwd_data,bwd_weight}: the nn.Linear family of the SW_Transformer / projector stacks, mapped
// onto the MFMA GEMM templates of gemm.hpp.
#include <stdlib.h>
#include "gemm.hpp"
#include "gemm_dw_group.hpp"
#include "gemm_dw_ring.hpp"
#include "gemm_ring.hpp"

static MaskParams to_mask(const focal_drop_desc& d, int ncols) {
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
static MaskParams no_mask() {
  MaskParams m;
  memset(&m, 0, sizeof(m));
  return m;
}

static int check_desc(const focal_linear_desc* d) {
  FOCAL_CHECK_ARG(d != nullptr, "linear: null descriptor");
  FOCAL_CHECK_ARG(d->dtype == FOCAL_F32 || d->dtype == FOCAL_BF16, "linear: bad dtype %d", d->dtype);
  FOCAL_CHECK_ARG(d->M > 0 && d->N > 0 && d->K > 0, "linear: bad shape %d %d %d", d->M, d->N, d->K);
  const int ec = d->dtype == FOCAL_BF16 ? 8 : 4;
  FOCAL_CHECK_ARG(d->K % ec == 0 && d->N % ec == 0, "linear: N=%d and K=%d must be multiples of %d", d->N, d->K, ec);
  FOCAL_CHECK_ARG(d->x_dtype == FOCAL_F32 || d->x_dtype == d->dtype, "linear: x_dtype must be f32 or dtype");
  FOCAL_CHECK_ARG(d->y_dtype == FOCAL_F32 || d->y_dtype == d->dtype, "linear: y_dtype must be f32 or dtype");
  return FOCAL_OK;
}

extern "C" int focal_linear_fwd(const focal_linear_desc* d, const void* x, const void* w, const float* bias,
                                const float* resid, void* y, void* act_grad, void* stream) {
  if (int rc = check_desc(d)) return rc;
  FOCAL_CHECK_ARG(x && w && y, "linear_fwd: null tensor");
  GemmSpec s;
  s.compute = d->dtype;
  s.a_dtype = d->x_dtype; s.b_dtype = d->dtype; s.c_dtype = d->y_dtype;
  s.tra = false; s.trb = false;
  s.proA = PRO_NONE;
  s.proB = PRO_NONE;
  const int splits = d->splits > 1 ? d->splits : 1;
  if (splits > 1) {
    FOCAL_CHECK_ARG(d->epilogue == FOCAL_EPI_NONE && d->y_dtype == FOCAL_F32, "linear_fwd: split-K needs a plain fp32 output");
    s.epi = EPI_ATOMIC;
  } else {
    s.epi = d->epilogue == FOCAL_EPI_RESIDUAL ? EPI_RESID : d->epilogue == FOCAL_EPI_RELU ? EPI_RELU
          : d->epilogue == FOCAL_EPI_GELU ? EPI_GELU_FWD : EPI_STORE;
  }
  if (s.epi == EPI_RESID) FOCAL_CHECK_ARG(resid != nullptr && d->y_dtype == FOCAL_F32, "linear_fwd: residual epilogue needs resid and fp32 y");
  if (s.epi == EPI_GELU_FWD) FOCAL_CHECK_ARG(act_grad != nullptr && d->y_dtype == d->dtype && d->x_dtype == d->dtype, "linear_fwd: GELU epilogue needs act_grad and dtype-typed x / y");
  GemmParams p;
  memset(&p, 0, sizeof(p));
  p.M = d->M; p.N = d->N; p.K = d->K;
  p.A = x; p.lda = d->K;
  p.B = w; p.ldb = d->K;
  p.C = y; p.ldc = d->N;
  p.batch = 1; p.splits = splits; p.alpha = 1.f;
  p.bias = bias;
  p.resid = resid; p.ldr = d->N;
  p.aux_out = act_grad;
  p.proA = no_mask();
  p.proB = no_mask();
  p.epi = to_mask(d->out_drop, d->N);
  return focal_launch_gemm(s, p, (hipStream_t)stream);
}

extern "C" int focal_linear_resid_ln_supported(int dtype, int N, int K) {
  if (N == 64) return 1;
  return dtype == FOCAL_BF16 && (N == 128 || N == 256) && K % 64 == 0;
}

extern "C" int focal_linear_resid_ln_fwd(const focal_linear_desc* d, const void* x, const void* w, const float* bias,
                                         const float* resid, float* y, const float* gamma, const float* beta, float eps,
                                         void* y_ln, float* stats, void* stream) {
  if (int rc = check_desc(d)) return rc;
  FOCAL_CHECK_ARG(x && w && y && resid && gamma && beta && y_ln && stats, "linear_resid_ln_fwd: null tensor");
  FOCAL_CHECK_ARG(d->epilogue == FOCAL_EPI_RESIDUAL && d->y_dtype == FOCAL_F32 && d->x_dtype == d->dtype && d->splits <= 1,
                  "linear_resid_ln_fwd: needs the residual epilogue, fp32 y and dtype-typed x");
  FOCAL_CHECK_ARG(focal_linear_resid_ln_supported(d->dtype, d->N, d->K), "linear_resid_ln_fwd: N = %d, K = %d (the fused LayerNorm needs 64-column rows, or bf16 with 128 / 256 columns and K %% 64 == 0)", d->N, d->K);
  GemmSpec s;
  s.compute = d->dtype;
  s.a_dtype = d->x_dtype; s.b_dtype = d->dtype; s.c_dtype = FOCAL_F32;
  s.tra = false; s.trb = false; s.proA = PRO_NONE; s.proB = PRO_NONE; s.epi = EPI_RESID_LN;
  GemmParams p;
  memset(&p, 0, sizeof(p));
  p.M = d->M; p.N = d->N; p.K = d->K;
  p.A = x; p.lda = d->K;
  p.B = w; p.ldb = d->K;
  p.C = y; p.ldc = d->N;
  p.batch = 1; p.splits = 1; p.alpha = 1.f;
  p.bias = bias;
  p.resid = resid; p.ldr = d->N;
  p.aux_out = y_ln;
  p.ln_gamma = gamma; p.ln_beta = beta; p.ln_stats = stats; p.ln_eps = eps;
  p.proA = no_mask();
  p.proB = no_mask();
  p.epi = to_mask(d->out_drop, d->N);
  return focal_launch_gemm(s, p, (hipStream_t)stream);
}

extern "C" int focal_linear_bwd_data(const focal_linear_desc* d, const void* dy, const void* w, const void* x, void* dx,
                                     void* stream) {
  if (int rc = check_desc(d)) return rc;
  FOCAL_CHECK_ARG(dy && w && dx, "linear_bwd_data: null tensor");
  // dx[M, K] = (dy . out_mask)[M, N] . w[N, K]   (w read transposed: memory [r = n][i = k])
  GemmSpec s;
  s.compute = d->dtype;
  s.a_dtype = d->y_dtype; s.b_dtype = d->dtype; s.c_dtype = d->x_dtype;
  s.tra = false; s.trb = true;
  // fp32 output gradients feeding `dtype` operands always go through the masking loader (identity when p = 0)
  const bool masked = d->epilogue == FOCAL_EPI_RESIDUAL || (d->y_dtype == FOCAL_F32 && d->x_dtype != FOCAL_F32);
  s.proA = masked ? PRO_MASK : PRO_NONE;
  s.proB = PRO_NONE;
  s.epi = d->act_in == FOCAL_ACT_GELU ? EPI_MUL_AUX : d->act_in == FOCAL_ACT_RELU_OUT ? EPI_RELU_BWD : EPI_STORE;
  if (s.epi != EPI_STORE) FOCAL_CHECK_ARG(x != nullptr, "linear_bwd_data: activation derivative / relu output needed");
  GemmParams p;
  memset(&p, 0, sizeof(p));
  p.M = d->M; p.N = d->K; p.K = d->N;
  p.A = dy; p.lda = d->N;
  p.B = w; p.ldb = d->K;
  p.C = dx; p.ldc = d->K;
  p.batch = 1; p.splits = 1; p.alpha = 1.f;
  p.aux = x; p.ldaux = d->K;
  p.proA = masked ? to_mask(d->out_drop, d->N) : no_mask();
  p.proB = no_mask();
  p.epi = no_mask();
  return focal_launch_gemm(s, p, (hipStream_t)stream);
}

extern "C" int focal_linear_bwd_data_ln_supported(int dtype, int N, int K) {
  // K = the LayerNorm's width (the linear layer's input features), N = its output features (the contraction of the dX product)
  return dtype == FOCAL_BF16 && (K == 64 || K == 128 || K == 256) && N % 64 == 0 && N >= 64;
}

// dx of a linear layer whose input came out of a LayerNorm, and that LayerNorm's backward, in one kernel: the [M, K] product dy . w never
// reaches memory -- the GEMM's epilogue (row-complete wave tiles, gemm_pipe.hpp EPI_LN_BWD) turns each finished row into the LayerNorm's
// input gradient, adds it to the residual-stream gradient g, writes dtype(g * mask) for the next branch and accumulates dgamma / dbeta.
extern "C" int focal_linear_bwd_data_ln(const focal_linear_desc* d, const void* dy, const void* w, const float* ln_x, const float* ln_stats,
                                        const float* ln_gamma, float* g, float* dgamma, float* dbeta, void* g_masked,
                                        const focal_drop_desc* mask, void* stream) {
  if (int rc = check_desc(d)) return rc;
  FOCAL_CHECK_ARG(dy && w && ln_x && ln_stats && ln_gamma && dgamma && dbeta, "linear_bwd_data_ln: null tensor");
  FOCAL_CHECK_ARG(g || !g_masked, "linear_bwd_data_ln: g_masked without g");  // g == NULL: only dgamma / dbeta (nobody needs the input gradient)
  FOCAL_CHECK_ARG(focal_linear_bwd_data_ln_supported(d->dtype, d->N, d->K) && d->x_dtype == d->dtype && d->y_dtype == d->dtype &&
                  d->epilogue != FOCAL_EPI_RESIDUAL && d->act_in == FOCAL_ACT_NONE,  // (a GELU epilogue's derivative is already in dy)
                  "linear_bwd_data_ln: bf16 layers with 64 / 128 / 256 input features, plain `dtype` dy (N = %d, K = %d)", d->N, d->K);
  FOCAL_CHECK_ARG(((uintptr_t)dy % 16 == 0) && ((uintptr_t)w % 16 == 0), "linear_bwd_data_ln: operands must be 16-byte aligned");
  FOCAL_CHECK_ARG(!mask || g_masked, "linear_bwd_data_ln: mask without g_masked");
  GemmParams p;
  memset(&p, 0, sizeof(p));
  p.M = d->M; p.N = d->K; p.K = d->N;
  p.A = dy; p.lda = d->N;
  p.B = w; p.ldb = d->K;
  p.C = g; p.ldc = d->K;
  p.batch = 1; p.splits = 1; p.alpha = 1.f;
  p.resid = ln_x; p.ldr = d->K;
  p.ln_stats = const_cast<float*>(ln_stats);
  p.ln_gamma = ln_gamma;
  p.ln_dgamma = dgamma; p.ln_dbeta = dbeta;
  p.aux_out = g_masked;
  p.proA = no_mask();
  p.proB = no_mask();
  focal_drop_desc dd;
  memset(&dd, 0, sizeof(dd));
  if (mask) dd = *mask;
  p.epi = to_mask(dd, d->K);
  hipError_t e;
  if (d->K == 64 && d->M >= 4096 && !focal_ring_disabled() &&
      focal_ring_fits<float, EPI_LN_BWD, true, 128, 64, 6, true, 8, 1, 2, false>(p)) {  // (stage 0, 192 -> 64: the weight panel is 24 KB; lab 0.82)
    focal_note_kernel("focal_gemm_ring_kernel<f32, epi=%d (LayerNorm backward), trB=1, 128x64, 8x1 waves, ring 6, panel resident>", EPI_LN_BWD);
    e = focal_launch_gemm_ring<float, EPI_LN_BWD, true, 128, 64, 6, true, 8, 1, 2, false>(p, (hipStream_t)stream);
  } else if (d->K == 128 && d->M >= 4096 && !focal_ring_disabled() &&
             focal_ring_fits<float, EPI_LN_BWD, true, 64, 128, 4, false, 4, 1, 2, false, false>(p)) {  // (stage 1; lab 0.88 at 73 728 rows, 0.94-0.97 at 36 864)
    focal_note_kernel("focal_gemm_ring_kernel<f32, epi=%d (LayerNorm backward), trB=1, 64x128, 4x1 waves, ring 4>", EPI_LN_BWD);
    e = focal_launch_gemm_ring<float, EPI_LN_BWD, true, 64, 128, 4, false, 4, 1, 2, false, false>(p, (hipStream_t)stream);
  } else if (d->K == 256 && d->M >= 2048 && !focal_ring_disabled() && getenv("FOCAL_LAB_LN_BWD_RING256") != nullptr &&
             focal_ring_fits<float, EPI_LN_BWD, true, 64, 256, 2, false, 4, 1, 2, false, false>(p)) {  // (stage 2: lab, round 6)
    focal_note_kernel("focal_gemm_ring_kernel<f32, epi=%d (LayerNorm backward), trB=1, 64x256, 4x1 waves, ring 2>", EPI_LN_BWD);
    e = focal_launch_gemm_ring<float, EPI_LN_BWD, true, 64, 256, 2, false, 4, 1, 2, false, false>(p, (hipStream_t)stream);
  } else {
    focal_note_kernel("focal_gemm_pipe_kernel<f32, epi=%d (LayerNorm backward), trB=1, %dx%d, 4x%d waves>", EPI_LN_BWD, d->K == 256 ? 64 : 128,
                      d->K, d->K == 256 ? 2 : 1);
    e = d->K == 64    ? focal_launch_gemm_pipe<float, EPI_LN_BWD, true, 128, 64, 2, 4, 1>(p, (hipStream_t)stream)
        : d->K == 128 ? focal_launch_gemm_pipe<float, EPI_LN_BWD, true, 128, 128, 2, 4, 1>(p, (hipStream_t)stream)
                      : focal_launch_gemm_pipe<float, EPI_LN_BWD, true, 64, 256, 2, 4, 2>(p, (hipStream_t)stream);
  }
  if (e != hipSuccess) {
    focal_set_error("linear_bwd_data_ln: launch failed: %s", hipGetErrorString(e));
    return FOCAL_EHIP;
  }
  return FOCAL_OK;
}

extern "C" int focal_linear_bwd_weight(const focal_linear_desc* d, const void* dy, const void* x, float* dw, float* dbias,
                                       void* stream) {
  if (int rc = check_desc(d)) return rc;
  FOCAL_CHECK_ARG(dy && x && dw, "linear_bwd_weight: null tensor");
  // dw[N, K] += sum_m (dy . out_mask)[m][n] * act_in(x)[m][k]: both operands read transposed, reduction over M is
  // split across workgroups and combined with fp32 atomics straight into the gradient arena.
  GemmSpec s;
  s.compute = d->dtype;
  s.a_dtype = d->y_dtype; s.b_dtype = d->x_dtype; s.c_dtype = FOCAL_F32;
  s.tra = true; s.trb = true;
  // fp32 output gradients feeding `dtype` operands always go through the masking loader (identity when p = 0)
  const bool masked = d->epilogue == FOCAL_EPI_RESIDUAL || (d->y_dtype == FOCAL_F32 && d->x_dtype != FOCAL_F32);
  s.proA = masked ? PRO_MASK : PRO_NONE;
  s.proB = PRO_NONE;
  s.epi = EPI_ATOMIC;
  GemmParams p;
  memset(&p, 0, sizeof(p));
  p.M = d->N; p.N = d->K; p.K = d->M;
  p.A = dy; p.lda = d->N;
  p.B = x; p.ldb = d->K;
  p.C = dw; p.ldc = d->K;
  p.batch = 1; p.alpha = 1.f;
  p.splits = 1;  // chosen with the tile shape in gemm_dispatch.inc (launch_dw)
  p.dw_target = d->dw_workgroups;
  p.proA = masked ? to_mask(d->out_drop, d->N) : no_mask();
  p.proB = no_mask();
  p.epi = no_mask();
  p.colsumA = dbias;
  return focal_launch_gemm(s, p, (hipStream_t)stream);
}

// the launch plan of focal_linear_bwd_weight for this descriptor: (kernel: 1 = focal_gemm_kernel, 2 = focal_dw_ring_kernel; workgroups)
static bool dw_launch_plan(const focal_linear_desc* d, int* kernel, int* wgs) {
  if (check_desc(d) != FOCAL_OK) return false;
  int bm, bn, splits;
  focal_dw_plan(d->N, d->K, d->M, d->dw_workgroups, &bm, &bn, &splits);
  const bool masked = d->epilogue == FOCAL_EPI_RESIDUAL || (d->y_dtype == FOCAL_F32 && d->x_dtype != FOCAL_F32);
  const bool ring = d->dtype == FOCAL_BF16 && d->x_dtype == FOCAL_BF16 && d->y_dtype == FOCAL_BF16 && !masked && bm == 64 && bn == 64 &&
                    focal_dw_ring_shape(d->N, d->K, d->M);
  *kernel = ring ? 2 : 1;
  if (ring && splits >= 2) splits = focal_dw_ring_pairs(d->N, d->K, splits);  // two token slices per 8-wave workgroup (gemm_dw_ring.hpp)
  *wgs = ((d->N + bm - 1) / bm) * ((d->K + bn - 1) / bn) * splits;
  return true;
}

extern "C" int focal_linear_bwd_weight_workgroups(const focal_linear_desc* d) {
  int kernel, wgs;
  return dw_launch_plan(d, &kernel, &wgs) ? wgs : 0;
}

extern "C" int focal_linear_bwd_weight_kernel(const focal_linear_desc* d) {
  int kernel, wgs;
  return dw_launch_plan(d, &kernel, &wgs) ? kernel : 0;
}

// ---------------------------------------------------------------------------------------------- grouped weight gradients
static bool dw_group_fits(const focal_dw_problem& q) {
  return q.M > 0 && q.N > 0 && q.K > 0 && q.M % 64 == 0 && q.N % 128 == 0 && q.K % 128 == 0 && q.dy && q.x && q.dw &&
         ((uintptr_t)q.dy % 16 == 0) && ((uintptr_t)q.x % 16 == 0) && ((uintptr_t)q.dw % 16 == 0);
}

extern "C" int focal_linear_bwd_weight_group_supported(int dtype, int M, int N, int K) {
  return dtype == FOCAL_BF16 && M > 0 && M % 64 == 0 && N % 128 == 0 && K % 128 == 0;
}

static int dw_group_build(int dtype, int n, const focal_dw_problem* probs, DwGroupParams* gp, int* wgs) {
  FOCAL_CHECK_ARG(dtype == FOCAL_BF16, "linear_bwd_weight_group: bf16 operands only (dtype %d)", dtype);
  FOCAL_CHECK_ARG(n >= 1 && n <= DWG_MAX_PROBLEMS && probs, "linear_bwd_weight_group: 1 .. %d problems (got %d)", DWG_MAX_PROBLEMS, n);
  memset(gp, 0, sizeof(*gp));
  gp->nprob = n;
  for (int i = 0; i < n; ++i) {
    const focal_dw_problem& q = probs[i];
    FOCAL_CHECK_ARG(dw_group_fits(q), "linear_bwd_weight_group: problem %d (M=%d N=%d K=%d) needs M %% 64 == 0, N, K %% 128 == 0 and 16-byte aligned tensors", i, q.M, q.N, q.K);
    DwGroupProblem& p = gp->prob[i];
    p.A = reinterpret_cast<const bf16_t*>(q.dy); p.lda = q.N;
    p.B = reinterpret_cast<const bf16_t*>(q.x); p.ldb = q.K;
    p.C = q.dw; p.ldc = q.K;
    p.colsum = q.dbias;
    p.M = q.N; p.N = q.K; p.rows = q.M;
    p.exclusive = q.exclusive ? 1 : 0;
  }
  const int target = getenv("FOCAL_LAB_DWG_TARGET") ? atoi(getenv("FOCAL_LAB_DWG_TARGET")) : 256, min_steps = 8;  // ~one workgroup per CU, at least 8 ring stages each (swept in round 3: profiles/r3_dw_group.txt)
  *wgs = focal_dw_group_plan<64>(*gp, target, min_steps);
  return FOCAL_OK;
}

// 64-tile problems (a 64-channel block's qkv / proj gradients): the ring kernel's tiles behind a problem table (gemm_dw_ring.hpp)
static bool dw_ring_group_fits(const focal_dw_problem& q) {
  return q.M >= 64 && q.M % 64 == 0 && q.N > 0 && q.N % 64 == 0 && q.K > 0 && q.K % 64 == 0 && q.dy && q.x && q.dw &&
         ((uintptr_t)q.dy % 16 == 0) && ((uintptr_t)q.x % 16 == 0);
}
static bool dw_group_uses_ring(int n, const focal_dw_problem* probs) {
  if (!probs || n < 1 || n > DWR_MAX_PROBLEMS) return false;
  bool all128 = true, all64 = true;
  for (int i = 0; i < n; ++i) { all128 = all128 && dw_group_fits(probs[i]); all64 = all64 && dw_ring_group_fits(probs[i]); }
  return !all128 && all64;
}
static int dw_ring_group_build(int dtype, int n, const focal_dw_problem* probs, DwRingGroupParams* gp, int* wgs) {
  FOCAL_CHECK_ARG(dtype == FOCAL_BF16, "linear_bwd_weight_group: bf16 operands only (dtype %d)", dtype);
  memset(gp, 0, sizeof(*gp));
  gp->nprob = n;
  for (int i = 0; i < n; ++i) {
    const focal_dw_problem& q = probs[i];
    GemmParams& p = gp->prob[i];
    p.M = q.N; p.N = q.K; p.K = q.M;
    p.A = q.dy; p.lda = q.N;
    p.B = q.x; p.ldb = q.K;
    p.C = q.dw; p.ldc = q.K;
    p.batch = 1; p.alpha = 1.f; p.splits = 1;
    p.colsumA = q.dbias;
  }
  *wgs = focal_dw_ring_group_plan<2>(*gp);
  return FOCAL_OK;
}

extern "C" int focal_linear_bwd_weight_group_kind(int dtype, int M, int N, int K) {
  if (dtype != FOCAL_BF16 || M <= 0 || M % 64 != 0) return 0;
  if (N % 128 == 0 && K % 128 == 0) return 2;
  return (N > 0 && K > 0 && N % 64 == 0 && K % 64 == 0) ? 1 : 0;
}

extern "C" int focal_linear_bwd_weight_group(int dtype, int n, const focal_dw_problem* probs, void* stream) {
  if (dw_group_uses_ring(n, probs)) {
    DwRingGroupParams rp;
    int rw = 0;
    if (int rc = dw_ring_group_build(dtype, n, probs, &rp, &rw)) return rc;
    focal_note_kernel("focal_dw_ring_group_kernel<64, 4, 2>");
    hipError_t e = focal_launch_dw_ring_group<64, 4, 2>(rp, rw, (hipStream_t)stream);
    if (e != hipSuccess) {
      focal_set_error("linear_bwd_weight_group (64-tile ring): launch failed: %s", hipGetErrorString(e));
      return FOCAL_EHIP;
    }
    return FOCAL_OK;
  }
  DwGroupParams gp;
  int wgs = 0;
  if (int rc = dw_group_build(dtype, n, probs, &gp, &wgs)) return rc;
  hipError_t e = focal_launch_dw_group<64, 4>(gp, wgs, (hipStream_t)stream);
  if (e != hipSuccess) {
    focal_set_error("linear_bwd_weight_group: launch failed: %s", hipGetErrorString(e));
    return FOCAL_EHIP;
  }
  return FOCAL_OK;
}

// n weight gradients whose operands are fp32 tensors, one launch (gemm.hpp: focal_dw_tail_group_kernel)
static int dw_tail_build(int compute, int n, const focal_dw_problem* probs, int workgroups, DwTailGroup* g) {
  FOCAL_CHECK_ARG(compute == FOCAL_BF16 || compute == FOCAL_F32, "linear_bwd_weight_group_f32: bad compute dtype %d", compute);
  FOCAL_CHECK_ARG(n >= 1 && n <= DW_TAIL_MAX && probs, "linear_bwd_weight_group_f32: 1 .. %d problems (got %d)", DW_TAIL_MAX, n);
  memset(g, 0, sizeof(*g));
  g->n = n;
  // every problem gets its share of the launch's workgroup target (default ~4 workgroups per CU over the whole launch); a slice is never
  // shorter than focal_dw_plan's 256 reduction rows
  const int target = (workgroups > 0 ? workgroups : 1024) / n;
  int end = 0;
  for (int i = 0; i < n; ++i) {
    const focal_dw_problem& q = probs[i];
    FOCAL_CHECK_ARG(q.dy && q.x && q.dw && q.M > 0 && q.N > 0 && q.K > 0, "linear_bwd_weight_group_f32: problem %d: null tensor or empty shape", i);
    GemmParams& p = g->p[i];
    p.M = q.N; p.N = q.K; p.K = q.M;
    p.A = q.dy; p.lda = q.N;
    p.B = q.x; p.ldb = q.K;
    p.C = q.dw; p.ldc = q.K;
    p.batch = 1; p.alpha = 1.f;
    p.colsumA = q.dbias;
    int bm, bn, splits;
    focal_dw_plan(p.M, p.N, q.M, target < 1 ? 1 : target, &bm, &bn, &splits);
    p.splits = splits;
    end += ((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn) * splits;
    g->wg_end[i] = end;
  }
  for (int i = n; i < DW_TAIL_MAX; ++i) g->wg_end[i] = end;
  return FOCAL_OK;
}
extern "C" int focal_linear_bwd_weight_group_f32(int compute, int n, const focal_dw_problem* probs, int workgroups, void* stream) {
  DwTailGroup g;
  if (int rc = dw_tail_build(compute, n, probs, workgroups, &g)) return rc;
  return compute == FOCAL_F32 ? focal_launch_dw_tail_f32(g, (hipStream_t)stream) : focal_launch_dw_tail_bf16(g, (hipStream_t)stream);
}
extern "C" int focal_linear_bwd_weight_group_f32_workgroups(int compute, int n, const focal_dw_problem* probs, int workgroups) {
  DwTailGroup g;
  return dw_tail_build(compute, n, probs, workgroups, &g) == FOCAL_OK ? g.wg_end[n - 1] : 0;
}

extern "C" int focal_linear_bwd_weight_group_workgroups(int dtype, int n, const focal_dw_problem* probs) {
  if (dw_group_uses_ring(n, probs)) {
    DwRingGroupParams rp;
    int rw = 0;
    return dw_ring_group_build(dtype, n, probs, &rp, &rw) == FOCAL_OK ? rw : 0;
  }
  DwGroupParams gp;
  int wgs = 0;
  return dw_group_build(dtype, n, probs, &gp, &wgs) == FOCAL_OK ? wgs : 0;
}
