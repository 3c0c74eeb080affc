// Microbenchmark: what fp64 matrix-core rate can a kernel reach on this MI355X with operands in registers (no memory traffic)?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int CHAINS>
__global__ __launch_bounds__(256) void k_mfma(double* out, int iters, double a0, double b0) {
  d4 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) acc[c] = d4{0.0, 0.0, 0.0, 0.0};
  double a = a0 + threadIdx.x * 1e-9, b = b0 + threadIdx.x * 1e-9;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
  }
  double s = 0.0;
  for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  if (s == 12345.678) out[0] = s;
}
__global__ __launch_bounds__(256) void k_fma(double* out, int iters, double a0, double b0) {
  double acc[8];
  for (int c = 0; c < 8; ++c) acc[c] = c;
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = fma(a, acc[c], b);
  }
  double s = 0.0;
  for (int c = 0; c < 8; ++c) s += acc[c];
  if (s == 12345.678) out[0] = s;
}
template <typename F>
static float timeit(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  double* d; hipMalloc(&d, 64);
  for (int iters = 20000; iters <= 2000000; iters *= 10)
  for (int wg_per_cu = 1; wg_per_cu <= 8; wg_per_cu *= 2) {
    const int blocks = 256 * wg_per_cu;
    float m1 = timeit([&] { hipLaunchKernelGGL(k_mfma<1>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0, 1.0); });
    float m4 = timeit([&] { hipLaunchKernelGGL(k_mfma<4>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0, 1.0); });
    float mf = timeit([&] { hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0, 1.0); });
    const double fl1 = (double)blocks * 4 * iters * 2048.0, fl4 = fl1 * 4, flf = (double)blocks * 256 * iters * 8 * 2.0;
    printf("iters %7d (%.1f ms) waves/SIMD %d: mfma 1 chain %.1f TFLOP/s, 4 chains %.1f TFLOP/s; vector fma %.1f TFLOP/s\n", iters, m4, wg_per_cu, fl1 / m1 / 1e9, fl4 / m4 / 1e9, flf / mf / 1e9);
  }
  return 0;
}
