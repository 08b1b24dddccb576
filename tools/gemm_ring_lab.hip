// Ring-GEMM lab: focal_gemm_ring_kernel (persistent, loader waves, optional weight-stationary panel: csrc/gemm_ring.hpp) against the
// focal_gemm_pipe_kernel configuration the dispatcher uses today, at the Swin stage-1 / stage-2 shapes of the B = 256 step, per epilogue.
// Cold operands (every launch works on another buffer set, > 256 MB in rotation) unless LAB_WARM=1.  Outputs are compared bit for bit.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I focal_amd/csrc -I include tools/gemm_ring_lab.hip focal_amd/csrc/error.cpp focal_amd/csrc/trace.cpp -o build/gemm_ring_lab
// run:   build/gemm_ring_lab [filter substring]
#include <vector>
#include <string>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include "gemm_ring.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void fill_bf16(bf16_t* x, long n, uint32_t seed, float scale) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const uint32_t h = focal_mix32((uint32_t)idx * 2654435761u + seed);
  x[idx] = (bf16_t)(((h >> 8) * (1.0f / 16777216.0f) - 0.5f) * scale);
}
__global__ void fill_f32(float* x, long n, uint32_t seed, float scale, float shift) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const uint32_t h = focal_mix32((uint32_t)idx * 2654435761u + seed);
  x[idx] = ((h >> 8) * (1.0f / 16777216.0f) - 0.5f) * scale + shift;
}
__global__ void diff_words(const uint32_t* a, const uint32_t* b, long n, unsigned long long* cnt) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  if (a[idx] != b[idx]) atomicAdd(cnt, 1ull);
}
__global__ void maxrel(const float* a, const float* b, long n, float* out) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const float e = fabsf(a[idx] - b[idx]) / (fabsf(a[idx]) + 1e-3f);
  atomicMax(reinterpret_cast<int*>(out), __float_as_int(e));
}

template <typename T> __global__ void maxdiff(const T* a, const T* b, long n, float* out) {  // out[0] = max |a - b|, out[1] = max |a|
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const float x = (float)a[idx], y = (float)b[idx];
  atomicMax(reinterpret_cast<int*>(out), __float_as_int(fabsf(x - y)));
  atomicMax(reinterpret_cast<int*>(out) + 1, __float_as_int(fabsf(x)));
}

static const char* g_filter = "";
static bool g_warm = false;

template <typename F> float time_us(F f, int iters) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a, 0));
  for (int i = 0; i < iters; ++i) f();
  CK(hipEventRecord(b, 0));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
  return ms * 1000.f / iters;
}

struct Bufs {
  int NB;
  size_t a_bytes, c_bytes, r_bytes, x_bytes, o_bytes, s_bytes;
  char *A, *C, *R, *X, *O, *S;   // A operand, C output, resid (f32), aux (bf16 in), aux_out, stats
  char *Cref, *Oref, *Cinit;
  float *bias, *gamma, *beta, *dg, *db, *dgref, *dbref;
  uint32_t* seed;
  bf16_t* W;
};

template <typename TC, int EPI, bool TRB>
static void run_case(const char* name, int M, int N, int K) {
  char label[160];
  snprintf(label, sizeof(label), "%s epi%d%s %s [%d x %d] K=%d", name, EPI, TRB ? "T" : "", sizeof(TC) == 2 ? "bf16" : "f32", M, N, K);
  if (g_filter[0] && !strstr(label, g_filter)) return;
  constexpr bool LN = EPI == EPI_RESID_LN || EPI == EPI_LN_BWD;
  constexpr bool NEED_R = EPI == EPI_RESID || LN;
  constexpr bool NEED_X = EPI == EPI_MUL_AUX;
  constexpr bool NEED_O = EPI == EPI_GELU_FWD || LN;
  Bufs b;
  memset(&b, 0, sizeof(b));
  b.a_bytes = (size_t)M * K * 2; b.c_bytes = (size_t)M * N * sizeof(TC); b.r_bytes = NEED_R ? (size_t)M * N * 4 : 0;
  b.x_bytes = NEED_X ? (size_t)M * N * 2 : 0; b.o_bytes = NEED_O ? (size_t)M * N * (EPI == EPI_GELU_FWD ? sizeof(TC) : 2) : 0; b.s_bytes = LN ? (size_t)M * 8 : 0;
  const size_t set = b.a_bytes + b.c_bytes + b.r_bytes + b.x_bytes + b.o_bytes + b.s_bytes;
  b.NB = g_warm ? 1 : (int)((size_t)640 * 1024 * 1024 / set) + 2;
  if (b.NB > 24) b.NB = 24;
  auto alloc = [&](char** p, size_t bytes) { if (bytes) CK(hipMalloc(p, bytes * b.NB)); };
  alloc(&b.A, b.a_bytes); alloc(&b.C, b.c_bytes); alloc(&b.R, b.r_bytes); alloc(&b.X, b.x_bytes); alloc(&b.O, b.o_bytes); alloc(&b.S, b.s_bytes);
  CK(hipMalloc(&b.Cref, b.c_bytes)); CK(hipMalloc(&b.Cinit, b.c_bytes));
  if (b.o_bytes) CK(hipMalloc(&b.Oref, b.o_bytes));
  CK(hipMalloc(&b.W, (size_t)N * K * 2)); CK(hipMalloc(&b.bias, N * 4)); CK(hipMalloc(&b.gamma, N * 4)); CK(hipMalloc(&b.beta, N * 4));
  CK(hipMalloc(&b.dg, N * 4)); CK(hipMalloc(&b.db, N * 4)); CK(hipMalloc(&b.dgref, N * 4)); CK(hipMalloc(&b.dbref, N * 4)); CK(hipMalloc(&b.seed, 4));
  const uint32_t seedv = 12345u;
  CK(hipMemcpy(b.seed, &seedv, 4, hipMemcpyHostToDevice));
  auto fb = [&](void* p, size_t bytes, uint32_t sd, float sc) { if (bytes) fill_bf16<<<ceil_div((long)(bytes / 2), 256), 256>>>((bf16_t*)p, (long)(bytes / 2), sd, sc); };
  auto ff = [&](void* p, size_t bytes, uint32_t sd, float sc, float sh) { if (bytes) fill_f32<<<ceil_div((long)(bytes / 4), 256), 256>>>((float*)p, (long)(bytes / 4), sd, sc, sh); };
  fb(b.A, b.a_bytes * b.NB, 1u, 2.0f);
  fb(b.W, (size_t)N * K * 2, 2u, 0.25f);
  ff(b.R, b.r_bytes * b.NB, 3u, 2.0f, 0.1f);
  fb(b.X, b.x_bytes * b.NB, 4u, 2.0f);
  if (LN) {  // stats = {mean, rstd}: plausible values
    std::vector<float> st((size_t)M * 2);
    for (int i = 0; i < M; ++i) { st[2 * i] = 0.1f + 0.01f * (i % 7); st[2 * i + 1] = 1.5f + 0.01f * (i % 5); }
    for (int s = 0; s < b.NB; ++s) CK(hipMemcpy(b.S + s * b.s_bytes, st.data(), b.s_bytes, hipMemcpyHostToDevice));
  }
  ff(b.bias, N * 4, 5u, 0.2f, 0.f); ff(b.gamma, N * 4, 6u, 0.2f, 1.f); ff(b.beta, N * 4, 7u, 0.2f, 0.f);
  if (sizeof(TC) == 4) ff(b.Cinit, b.c_bytes, 8u, 1.0f, 0.f); else fb(b.Cinit, b.c_bytes, 8u, 1.0f);
  CK(hipDeviceSynchronize());

  GemmParams p;
  memset(&p, 0, sizeof(p));
  p.M = M; p.N = N; p.K = K; p.lda = K; p.B = b.W; p.ldb = TRB ? N : K; p.ldc = N; p.batch = 1; p.splits = 1; p.alpha = 1.f;
  p.bias = (EPI == EPI_LN_BWD) ? nullptr : b.bias;
  p.ldr = N; p.ldaux = N;
  if (EPI == EPI_RESID || EPI == EPI_GELU_FWD || LN) {
    p.epi.seed = b.seed; p.epi.stream_elem = 3; p.epi.p_elem = 0.1f; p.epi.stream_path = 4; p.epi.p_path = EPI == EPI_GELU_FWD ? 0.f : 0.1f;
    p.epi.rows_per_sample = 36; p.epi.ncols = N;
  }
  p.ln_gamma = b.gamma; p.ln_beta = b.beta; p.ln_eps = 1e-5f; p.ln_dgamma = b.dg; p.ln_dbeta = b.db;
  int rot = 0;
  auto bind = [&](int s) {
    p.A = b.A + s * b.a_bytes; p.C = b.C + s * b.c_bytes;
    p.resid = NEED_R ? (const float*)(b.R + s * b.r_bytes) : nullptr;
    p.aux = NEED_X ? (const void*)(b.X + s * b.x_bytes) : nullptr;
    p.aux_out = NEED_O ? (void*)(b.O + s * b.o_bytes) : nullptr;
    p.ln_stats = LN ? (float*)(b.S + s * b.s_bytes) : nullptr;
  };
  auto next = [&]() { rot = (rot + 1) % b.NB; bind(rot); };
  // one checked launch on set 0 from a defined state -> reference copies (first call) or comparison
  bool have_ref = false;
  auto checked = [&](const char* what, auto launch) {
    bind(0);
    CK(hipMemcpy(b.C, b.Cinit, b.c_bytes, hipMemcpyDeviceToDevice));  // EPI_LN_BWD reads and updates C
    if (b.o_bytes) CK(hipMemset(b.O, 0, b.o_bytes));
    CK(hipMemset(b.dg, 0, N * 4)); CK(hipMemset(b.db, 0, N * 4));
    launch();
    CK(hipDeviceSynchronize());
    if (!have_ref) {
      CK(hipMemcpy(b.Cref, b.C, b.c_bytes, hipMemcpyDeviceToDevice));
      if (b.o_bytes) CK(hipMemcpy(b.Oref, b.O, b.o_bytes, hipMemcpyDeviceToDevice));
      CK(hipMemcpy(b.dgref, b.dg, N * 4, hipMemcpyDeviceToDevice)); CK(hipMemcpy(b.dbref, b.db, N * 4, hipMemcpyDeviceToDevice));
      have_ref = true;
      return;
    }
    unsigned long long* cnt;
    float* mr;
    CK(hipMalloc(&cnt, 8)); CK(hipMalloc(&mr, 4));
    CK(hipMemset(cnt, 0, 8)); CK(hipMemset(mr, 0, 4));
    diff_words<<<ceil_div((long)(b.c_bytes / 4), 256), 256>>>((const uint32_t*)b.Cref, (const uint32_t*)b.C, (long)(b.c_bytes / 4), cnt);
    if (b.o_bytes) diff_words<<<ceil_div((long)(b.o_bytes / 4), 256), 256>>>((const uint32_t*)b.Oref, (const uint32_t*)b.O, (long)(b.o_bytes / 4), cnt);
    if (EPI == EPI_LN_BWD) { maxrel<<<ceil_div(N, 256), 256>>>(b.dgref, b.dg, N, mr); maxrel<<<ceil_div(N, 256), 256>>>(b.dbref, b.db, N, mr); }
    unsigned long long h = 0;
    float hm = 0.f;
    CK(hipMemcpy(&h, cnt, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hm, mr, 4, hipMemcpyDeviceToHost));
    if (h != 0 || hm > 1e-3f) {
      float* md;
      CK(hipMalloc(&md, 16)); CK(hipMemset(md, 0, 16));
      maxdiff<TC><<<ceil_div((long)(b.c_bytes / sizeof(TC)), 256), 256>>>((const TC*)b.Cref, (const TC*)b.C, (long)(b.c_bytes / sizeof(TC)), md);
      if (b.o_bytes) maxdiff<bf16_t><<<ceil_div((long)(b.o_bytes / 2), 256), 256>>>((const bf16_t*)b.Oref, (const bf16_t*)b.O, (long)(b.o_bytes / 2), md + 2);
      float hd[4];
      CK(hipMemcpy(hd, md, 16, hipMemcpyDeviceToHost));
      printf("\n    !! %s: %llu words differ from the pipe kernel's output: C max |diff| %g of max %g, aux_out %g of %g; dgamma/dbeta rel %g", what, h, hd[0], hd[1], hd[2], hd[3], hm);
      CK(hipFree(md));
    }
    CK(hipFree(cnt)); CK(hipFree(mr));
  };
  const int iters = g_warm ? 30 : 2 * b.NB;
  // -DRING_STAMPS: the kernels add per-wave cycle sums to a u64[16] buffer passed in p.colsumA (unused by these epilogues)
  unsigned long long* stamps = nullptr;
  constexpr size_t STAMP_WORDS = 4096 * 16 * 8;  // [workgroup][wave][8]: every wave writes its own record (of the LAST launch)
#ifdef RING_STAMPS
  CK(hipMalloc(&stamps, STAMP_WORDS * 8));
#endif
  auto stamps_begin = [&]() { if (stamps) { CK(hipMemset(stamps, 0, STAMP_WORDS * 8)); p.colsumA = reinterpret_cast<float*>(stamps); } };
  auto stamps_end = [&](int launches) {
    if (!stamps) return;
    std::vector<unsigned long long> h(STAMP_WORDS);
    CK(hipMemcpy(h.data(), stamps, STAMP_WORDS * 8, hipMemcpyDeviceToHost));
    p.colsumA = nullptr;
    double l[4] = {0, 0, 0, 0}, c[4] = {0, 0, 0, 0}, ln = 0, cn = 0, cmax = 0, cmin = 1e30;
    for (size_t r = 0; r < STAMP_WORDS / 8; ++r) {
      const unsigned long long* q = &h[r * 8];
      if (q[4] == 1) { for (int i = 0; i < 4; ++i) l[i] += q[i]; ln += 1; }
      if (q[4] == 2) { for (int i = 0; i < 4; ++i) c[i] += q[i]; cn += 1; if (q[3] > cmax) cmax = q[3]; if (q[3] < cmin) cmin = q[3]; }
    }
    if (ln == 0) ln = 1;
    if (cn == 0) cn = 1;
    printf("\n      [cycles per wave, last launch] loader (%.0f waves): fill wait %.0f barrier %.0f issue %.0f all %.0f | consumer (%.0f waves): k loop %.0f epilogue %.0f all %.0f (min %.0f max %.0f)\n      ",
           ln, l[0] / ln, l[1] / ln, l[2] / ln, l[3] / ln, cn, c[1] / cn, c[2] / cn, c[3] / cn, cmin, cmax);
    (void)launches;
  };
  size_t bytes = b.a_bytes + b.c_bytes + b.r_bytes + b.x_bytes + b.o_bytes + (EPI == EPI_LN_BWD ? b.c_bytes : 0);
  printf("%-64s %6.1f MB :", label, bytes * 1e-6);
  auto report = [&](const char* tag, float us) { printf(" %s %5.1f (%4.2f TB/s) |", tag, us, bytes / us * 1e-6); fflush(stdout); };

#define OLD(BM, BN, NST, WGM, WGN)                                                                                                  \
  {                                                                                                                                 \
    checked("old", [&] { CK((focal_launch_gemm_pipe<TC, EPI, TRB, BM, BN, NST, WGM, WGN>(p, 0))); });                               \
    report("pipe" #BM "x" #BN, time_us([&] { next(); CK((focal_launch_gemm_pipe<TC, EPI, TRB, BM, BN, NST, WGM, WGN>(p, 0))); }, iters)); \
  }
#define RING(BM, BN, R, WS, WGM, WGN, NL, ASY)                                                                                      \
  if ((focal_ring_fits<TC, EPI, TRB, BM, BN, R, WS, WGM, WGN, NL, ASY>(p))) {                                                       \
    char tag[64];                                                                                                                   \
    snprintf(tag, sizeof(tag), "%dx%d/%d%s w%dx%d+%d%s", BM, BN, R, WS ? "ws" : "", WGM, WGN, NL, ASY ? "a" : "");                   \
    checked(tag, [&] { CK((focal_launch_gemm_ring<TC, EPI, TRB, BM, BN, R, WS, WGM, WGN, NL, ASY>(p, 0))); });                       \
    stamps_begin();                                                                                                                 \
    report(tag, time_us([&] { next(); CK((focal_launch_gemm_ring<TC, EPI, TRB, BM, BN, R, WS, WGM, WGN, NL, ASY>(p, 0))); }, iters)); \
    stamps_end(iters + 3);                                                                                                          \
  }
  if constexpr (LN) {
    if (N == 128) {
      OLD(128, 128, 2, 4, 1)
      RING(64, 128, 4, false, 4, 1, 2, false)
      if constexpr (EPI == EPI_LN_BWD) {
        checked("64x128/4 w4x1+2 no hoist", [&] { CK((focal_launch_gemm_ring<TC, EPI, TRB, 64, 128, 4, false, 4, 1, 2, false, false>(p, 0))); });
        report("64x128/4 w4x1+2 nohoist", time_us([&] { next(); CK((focal_launch_gemm_ring<TC, EPI, TRB, 64, 128, 4, false, 4, 1, 2, false, false>(p, 0))); }, iters));
      }
      RING(64, 128, 5, false, 4, 1, 2, false)
    } else if (N == 64) {
      OLD(128, 64, 2, 4, 1)
      RING(128, 64, 6, true, 4, 1, 2, false)
      RING(128, 64, 6, true, 8, 1, 2, false)
      RING(128, 64, 6, true, 8, 1, 2, true)
      RING(64, 64, 8, true, 4, 1, 2, false)
    }
  } else if (N % 128 == 0 && N >= 384) {
    OLD(128, 128, 2, 2, 2)
    RING(128, 128, 4, true, 4, 2, 2, false)
    RING(128, 128, 3, true, 4, 2, 2, false)
    RING(128, 128, 3, false, 4, 2, 2, false)
    RING(64, 128, 6, true, 2, 4, 2, false)
    RING(64, 128, 8, true, 2, 4, 2, false)
  } else {
    OLD(128, 64, 2, 4, 1)
    RING(128, 64, 4, false, 8, 1, 2, false)
    RING(128, 64, 5, false, 4, 1, 2, false)
    RING(128, 128, 3, false, 4, 2, 2, false)
    RING(64, 128, 5, false, 2, 4, 2, false)
    RING(64, 128, 5, true, 2, 4, 2, false)
    RING(128, 256, 2, false, 2, 4, 2, false)
    RING(64, 256, 3, false, 2, 4, 2, false)
  }
  printf("\n");
  for (char* q : {b.A, b.C, b.R, b.X, b.O, b.S, b.Cref, b.Oref, b.Cinit}) if (q) CK(hipFree(q));
  CK(hipFree(b.W)); CK(hipFree(b.bias)); CK(hipFree(b.gamma)); CK(hipFree(b.beta)); CK(hipFree(b.dg)); CK(hipFree(b.db)); CK(hipFree(b.dgref)); CK(hipFree(b.dbref)); CK(hipFree(b.seed));
}

int main(int argc, char** argv) {
  if (argc > 1) g_filter = argv[1];
  g_warm = getenv("LAB_WARM") != nullptr;
  const int Ms2[2] = {18432, 9216}, Ms1[2] = {73728, 36864}, Ms0[2] = {294912, 147456};
  for (int i = 0; i < 2; ++i) {
    const int m2 = Ms2[i], m1 = Ms1[i], m0 = Ms0[i];
    // ---- forward
    run_case<bf16_t, EPI_STORE, false>("s2 qkv", m2, 768, 256);
    run_case<bf16_t, EPI_STORE, false>("s1 qkv", m1, 384, 128);
    run_case<bf16_t, EPI_GELU_FWD, false>("s2 fc1", m2, 1024, 256);
    run_case<bf16_t, EPI_GELU_FWD, false>("s1 fc1", m1, 512, 128);
    run_case<float, EPI_RESID, false>("s2 fc2", m2, 256, 1024);
    run_case<float, EPI_RESID, false>("s2 proj", m2, 256, 256);
    run_case<float, EPI_RESID_LN, false>("s1 fc2+ln", m1, 128, 512);
    run_case<float, EPI_RESID_LN, false>("s1 proj+ln", m1, 128, 128);
    run_case<float, EPI_STORE, false>("merge 1->2", m2, 256, 512);
    run_case<float, EPI_STORE, false>("merge 0->1", m1, 128, 256);
    // ---- data gradients
    run_case<bf16_t, EPI_MUL_AUX, true>("s2 dx fc2", m2, 1024, 256);
    run_case<bf16_t, EPI_MUL_AUX, true>("s1 dx fc2", m1, 512, 128);
    run_case<bf16_t, EPI_STORE, true>("s2 dx fc1", m2, 256, 1024);
    run_case<bf16_t, EPI_STORE, true>("s2 dx qkv", m2, 256, 768);
    run_case<bf16_t, EPI_STORE, true>("s2 dx proj", m2, 256, 256);
    run_case<bf16_t, EPI_STORE, true>("s1 dx proj", m1, 128, 128);
    run_case<float, EPI_LN_BWD, true>("s1 dx fc1+ln", m1, 128, 512);
    run_case<float, EPI_LN_BWD, true>("s1 dx qkv+ln", m1, 128, 384);
    run_case<float, EPI_LN_BWD, true>("s0 dx qkv+ln", m0, 64, 192);
  }
  return 0;
}
