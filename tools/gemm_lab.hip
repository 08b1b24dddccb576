// GEMM lab: times focal_gemm_pipe_kernel variants against the 64x64 kernel of gemm.hpp on the deep-stage shapes, in one process.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I focal_amd/csrc tools/gemm_lab.hip focal_amd/csrc/error.cpp -o build/gemm_lab
#include <vector>
#include <cstdlib>
#include <cmath>
#include "gemm_pipe.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void ref_kernel(const bf16_t* A, const bf16_t* W, const float* bias, float* C, int M, int N, int K) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)M * N) return;
  const int m = idx / N, n = idx % N;
  float s = 0.f;
  for (int k = 0; k < K; ++k) s += (float)A[(long)m * K + k] * (float)W[(long)n * K + k];
  C[idx] = s + bias[n];
}
__global__ void cmp_kernel(const float* ref, const bf16_t* out, long n, float* maxerr) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const float r = ref[idx], o = (float)out[idx];
  const float e = fabsf(r - o) / (fabsf(r) + 1.0f);
  atomicMax(reinterpret_cast<int*>(maxerr), __float_as_int(e));
}
__global__ void transpose_kernel(const bf16_t* W, bf16_t* Wt, int N, int K) {  // W [N][K] -> Wt [K][N]
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)N * K) return;
  const int n = idx / K, k = idx % K;
  Wt[(long)k * N + n] = W[idx];
}
__global__ void fill_kernel(bf16_t* x, long n, uint32_t seed, float scale) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const uint32_t h = focal_mix32((uint32_t)idx * 2654435761u + seed);
  x[idx] = (bf16_t)(((h >> 8) * (1.0f / 16777216.0f) - 0.5f) * scale);
}

template <typename F> float time_us(F f, int iters = 30) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 5; ++i) f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a, 0));
  for (int i = 0; i < iters; ++i) f();
  CK(hipEventRecord(b, 0));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms * 1000.f / iters;
}

int main() {
  struct Shape { const char* name; int M, N, K; };
  const Shape shapes[] = {
      {"s2a qkv", 18432, 768, 256}, {"s2a proj", 18432, 256, 256}, {"s2a fc1", 18432, 1024, 256}, {"s2a fc2", 18432, 256, 1024},
      {"s2s qkv", 9216, 768, 256},  {"s2s fc2", 9216, 256, 1024},
      {"s1a qkv", 73728, 384, 128}, {"s1a proj", 73728, 128, 128}, {"s1a fc1", 73728, 512, 128}, {"s1a fc2", 73728, 128, 512},
      {"s1s qkv", 36864, 384, 128}, {"s1s fc2", 36864, 128, 512},
  };
  for (const Shape& sh : shapes) {
    const int M = sh.M, N = sh.N, K = sh.K;
    bf16_t *A, *W, *C;
    float *bias, *ref, *maxerr;
    const int NB = getenv("LAB_COLD") ? 12 : 1;  // cold mode: every launch works on a different (A, C) pair, > 256 MB in rotation
    const size_t a_elems = (size_t)M * K, c_elems = (size_t)M * N;
    CK(hipMalloc(&A, a_elems * 2 * NB)); CK(hipMalloc(&W, (size_t)N * K * 2)); CK(hipMalloc(&C, c_elems * 2 * NB));
    for (int b = 1; b < NB; ++b) fill_kernel<<<ceil_div((long)M * K, 256), 256>>>(A + b * a_elems, (long)M * K, 1u, 2.0f);
    int rot = 0;
    auto next = [&](GemmParams& q) { rot = (rot + 1) % NB; q.A = A + rot * a_elems; q.C = C + rot * c_elems; };
    CK(hipMalloc(&bias, N * 4)); CK(hipMalloc(&ref, (size_t)M * N * 4)); CK(hipMalloc(&maxerr, 4));
    fill_kernel<<<ceil_div((long)M * K, 256), 256>>>(A, (long)M * K, 1u, 2.0f);
    fill_kernel<<<ceil_div((long)N * K, 256), 256>>>(W, (long)N * K, 2u, 0.25f);
    std::vector<float> hb(N);
    for (int i = 0; i < N; ++i) hb[i] = 0.01f * (i % 17);
    CK(hipMemcpy(bias, hb.data(), N * 4, hipMemcpyHostToDevice));
    ref_kernel<<<ceil_div((long)M * N, 256), 256>>>(A, W, bias, ref, M, N, K);
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.M = M; p.N = N; p.K = K; p.A = A; p.lda = K; p.B = W; p.ldb = K; p.C = C; p.ldc = N; p.batch = 1; p.splits = 1; p.alpha = 1.f; p.bias = bias;
    auto check = [&](const char* what) {
      CK(hipMemset(maxerr, 0, 4));
      cmp_kernel<<<ceil_div((long)M * N, 256), 256>>>(ref, C + rot * c_elems, (long)M * N, maxerr);
      float e;
      CK(hipMemcpy(&e, maxerr, 4, hipMemcpyDeviceToHost));
      if (!(e < 8e-3f) && p.splits == 1) printf("    !! %s max rel err %g\n", what, e);
      CK(hipMemset(C, 0, c_elems * 2 * NB));
    };
    const double gf = 2.0 * M * N * K * 1e-9;
    printf("%-9s [%6d x %4d] K=%4d :", sh.name, M, N, K);
    {
      const float us = time_us([&] {
        next(p);
        dim3 grid(ceil_div(M, 64) * ceil_div(N, 64));
        hipLaunchKernelGGL((focal_gemm_kernel<bf16_t, bf16_t, bf16_t, bf16_t, false, false, PRO_NONE, PRO_NONE, EPI_STORE, 64, 64, 1>), grid, dim3(256), 0, 0, p);
      });
      check("base64");
      printf("  base64 %5.1f (%4.0f TF) |", us, gf / us * 1e3);
    }
#define VARIANT(BM, BN, NST, WGM, WGN)                                                             \
    if (N % BN == 0 && M % BM == 0) {                                                              \
      const float us = time_us([&] { next(p); CK((focal_launch_gemm_pipe<bf16_t, EPI_STORE, false, BM, BN, NST, WGM, WGN>(p, 0))); }); \
      check(#BM "x" #BN "x" #NST);                                                                 \
      printf(" %dx%d/%d w%dx%d %5.1f |", BM, BN, NST, WGM, WGN, us);              \
    }
    VARIANT(128, 128, 2, 2, 2)
    VARIANT(128, 128, 3, 2, 2)
    VARIANT(128, 128, 3, 4, 2)
    VARIANT(128, 64, 2, 2, 2)
    VARIANT(128, 64, 3, 2, 2)
    VARIANT(128, 64, 4, 2, 2)
    VARIANT(64, 64, 2, 2, 2)
    VARIANT(64, 64, 4, 2, 2)
    if (getenv("LAB_FWD_ONLY")) { printf("\n"); continue; }
    // ---- dX form: W stored [k][n] -> same product with Wt = W^T laid out [K][N]
    {
      bf16_t* Wt;
      CK(hipMalloc(&Wt, (size_t)N * K * 2));
      transpose_kernel<<<ceil_div((long)N * K, 256), 256>>>(W, Wt, N, K);
      GemmParams pt = p;
      pt.B = Wt; pt.ldb = N;
      {
        const float us = time_us([&] {
          dim3 grid(ceil_div(M, 64) * ceil_div(N, 64));
          hipLaunchKernelGGL((focal_gemm_kernel<bf16_t, bf16_t, bf16_t, bf16_t, false, true, PRO_NONE, PRO_NONE, EPI_STORE, 64, 64, 1>), grid, dim3(256), 0, 0, pt);
        });
        check("base64 trB");
        printf(" || trB base64 %5.1f |", us);
      }
#define TVARIANT(BM, BN, NST, WGM, WGN)                                                            \
      if (N % BN == 0 && M % BM == 0) {                                                            \
        const float us = time_us([&] { CK((focal_launch_gemm_pipe<bf16_t, EPI_STORE, true, BM, BN, NST, WGM, WGN>(pt, 0))); }); \
        check("T" #BM "x" #BN "x" #NST);                                                           \
        printf(" T%dx%d/%d %5.1f |", BM, BN, NST, us);                                             \
      }
      TVARIANT(128, 128, 2, 2, 2)
      TVARIANT(128, 64, 2, 2, 2)
      CK(hipFree(Wt));
    }
    printf("\n");
    CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(C)); CK(hipFree(bias)); CK(hipFree(ref)); CK(hipFree(maxerr));
  }
  return 0;
}
