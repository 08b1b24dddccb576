// Real-input full-spectrum DFT along the last axis, packed as [B, 2C, I, n] (Re / Im channel pairs), fp32.
// n = n1 * n2 four-step DFT held entirely in LDS by one workgroup per row:
//   X[k1 + n1*k2] = sum_{m2} W_{n2}^{m2 k2} * ( W_n^{m2 k1} * sum_{m1} x[n2*m1 + m2] W_{n1}^{m1 k1} )
// (m = n2*m1 + m2, k = k1 + n1*k2).  For the MOD audio rows n = 1600 = 40 x 40; n2 = 1 degenerates to the direct
// DFT used for the 20-point seismic rows.  The twiddle table holds {cos, -sin}(2 pi j / n), j < n, computed in
// fp64 on the host.
#include "common.hpp"

__global__ __launch_bounds__(256) void fft_realpack_kernel(const float* __restrict__ x, const float* __restrict__ tw,
                                                           float* __restrict__ out, focal_fft_desc d, int rows) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int n = d.n, n1 = d.n1, n2 = d.n2;
  float* xs = smem;            // [n]
  float* yr = smem + n;        // [n2][n1]  stage-1 output, real
  float* yi = yr + n;          //           imag
  float* twc = yi + n;         // [n] cos
  float* tws = twc + n;        // [n] -sin
  const int tid = threadIdx.x;
  for (int i = tid; i < n; i += 256) {
    twc[i] = tw[2 * i];
    tws[i] = tw[2 * i + 1];
  }
  for (int row = blockIdx.x; row < rows; row += gridDim.x) {
    __syncthreads();
    const float* xr = x + (long)row * n;
    for (int i = tid; i < n; i += 256) xs[i] = xr[i];
    __syncthreads();
    // stage 1: for each m2, n1-point DFT over m1, then twiddle W_n^{m2 k1}
    for (int o = tid; o < n; o += 256) {
      const int m2 = o / n1, k1 = o % n1;
      float re = 0.f, im = 0.f;
      int ph = 0;  // (m1 * k1 mod n1) * n2 indexes W_{n1} inside the n-point table
      for (int m1 = 0; m1 < n1; ++m1) {
        const float v = xs[n2 * m1 + m2];
        re += v * twc[ph * n2];
        im += v * tws[ph * n2];
        ph += k1;
        if (ph >= n1) ph -= n1;
      }
      const int t = (m2 * k1) % n;
      const float c = twc[t], s = tws[t];
      yr[o] = re * c - im * s;
      yi[o] = re * s + im * c;
    }
    __syncthreads();
    // stage 2: for each k1, n2-point DFT over m2 -> X[k1 + n1*k2]
    const int ci = row % d.I, bc = row / d.I;  // row = (b*C + c)*I + i
    float* ore = out + ((long)(2 * bc) * d.I + ci) * n;
    float* oim = out + ((long)(2 * bc + 1) * d.I + ci) * n;
    for (int o = tid; o < n; o += 256) {
      const int k = o, k1 = k % n1, k2 = k / n1;
      float re = 0.f, im = 0.f;
      int ph = 0;  // (m2 * k2 mod n2) * n1 indexes W_{n2}
      for (int m2 = 0; m2 < n2; ++m2) {
        const float a = yr[m2 * n1 + k1], b = yi[m2 * n1 + k1];
        const float c = twc[ph * n1], s = tws[ph * n1];
        re += a * c - b * s;
        im += a * s + b * c;
        ph += k2;
        if (ph >= n2) ph -= n2;
      }
      ore[k] = re;
      oim[k] = im;
    }
  }
}

extern "C" int focal_fft_realpack_fwd(const focal_fft_desc* d, const float* x, const float* twiddle, float* out,
                                      void* stream) {
  FOCAL_CHECK_ARG(d && x && twiddle && out, "fft_realpack: null argument");
  FOCAL_CHECK_ARG(d->n1 >= 1 && d->n2 >= 1 && d->n1 * d->n2 == d->n, "fft_realpack: n1*n2 != n");
  const size_t sm = (size_t)5 * d->n * sizeof(float);
  FOCAL_CHECK_ARG(sm <= 64 * 1024, "fft_realpack: n=%d too long for the LDS-resident DFT", d->n);
  const int rows = d->B * d->C * d->I;
  int blocks = rows < 4096 ? rows : 4096;
  hipLaunchKernelGGL(fft_realpack_kernel, dim3(blocks), dim3(256), sm, (hipStream_t)stream, x, twiddle, out, *d, rows);
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}
