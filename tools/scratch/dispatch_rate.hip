// Workgroup dispatch rate on MI355X: empty-ish kernels of 256 threads with a given static LDS footprint.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LDS_BYTES>
__global__ __launch_bounds__(256) void k(float* out, int spin) {
  __shared__ float s[LDS_BYTES / 4];
  s[threadIdx.x] = threadIdx.x;
  __syncthreads();
  float v = s[(threadIdx.x + 1) & 255];
  for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
  if (v == 12345.f) out[blockIdx.x] = v;
}
template <int L> void run(const char* name, float* d) {
  for (int spin : {0, 2000}) {
    for (int n : {1024, 4096, 16384, 65536}) {
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<L>, dim3(n), dim3(256), 0, 0, d, spin);
      hipEventRecord(a);
      for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(k<L>, dim3(n), dim3(256), 0, 0, d, spin);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      printf("%s spin=%d blocks=%6d: %8.2f us  -> %7.1f WG/us\n", name, spin, n, ms * 100.f, n / (ms * 100.f));
    }
  }
}
int main() {
  float* d; hipMalloc(&d, 1 << 20);
  run<1024>("lds 1KB ", d);
  run<18432>("lds 18KB", d);
  run<36864>("lds 36KB", d);
  return 0;
}
