// Microbenchmark: what fp32 matrix-core rate can a kernel reach on this MI355X with operands in registers (no memory traffic)?  The NWD
// demixer kernel (cm_nwd.hip) is built on v_mfma_f32_16x16x4_f32; the guide's dense fp32 matrix peak is 157.3 TFLOP/s.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak_f32.hip -o /tmp/mfma_peak_f32 && /tmp/mfma_peak_f32
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
template <int CHAINS>
__global__ __launch_bounds__(256) void k_mfma16(float* out, int iters, float a0, float b0) {
  f4 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) acc[c] = f4{0.f, 0.f, 0.f, 0.f};
  float a = a0 + threadIdx.x * 1e-6f, b = b0 + threadIdx.x * 1e-6f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  if (s == 12345.678f) out[0] = s;
}
template <int CHAINS>
__global__ __launch_bounds__(256) void k_mfma32(float* out, int iters, float a0, float b0) {
  f16v acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c)
    for (int q = 0; q < 16; ++q) acc[c][q] = 0.f;
  float a = a0 + threadIdx.x * 1e-6f, b = b0 + threadIdx.x * 1e-6f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < CHAINS; ++c)
    for (int q = 0; q < 16; ++q) s += acc[c][q];
  if (s == 12345.678f) out[0] = s;
}
template <typename F>
static float timeit(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  float* d; hipMalloc(&d, 64);
  const int iters = 400000;
  for (int wg_per_cu = 1; wg_per_cu <= 8; wg_per_cu *= 2) {
    const int blocks = 256 * wg_per_cu;
    float m1 = timeit([&] { hipLaunchKernelGGL(k_mfma16<1>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.f, 1.f); });
    float m4 = timeit([&] { hipLaunchKernelGGL(k_mfma16<4>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.f, 1.f); });
    float n1 = timeit([&] { hipLaunchKernelGGL(k_mfma32<1>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.f, 1.f); });
    float n2 = timeit([&] { hipLaunchKernelGGL(k_mfma32<2>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.f, 1.f); });
    const double f16 = (double)blocks * 4 * iters * (16.0 * 16 * 4 * 2), f32 = (double)blocks * 4 * iters * (32.0 * 32 * 2 * 2);
    printf("waves/SIMD %d: v_mfma_f32_16x16x4 1 chain %.1f, 4 chains %.1f TFLOP/s (%.1f ms); v_mfma_f32_32x32x2 1 chain %.1f, 2 chains %.1f TFLOP/s\n", wg_per_cu,
           f16 / m1 / 1e9, f16 * 4 / m4 / 1e9, m4, f32 / n1 / 1e9, f32 * 2 / n2 / 1e9);
  }
  return 0;
}
