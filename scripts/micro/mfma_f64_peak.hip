// Sustained rate of v_mfma_f64_16x16x4_f64 (register operands, independent accumulators): the denominator the fp64
// kernels of this build should be read against.   hipcc --offload-arch=gfx950 -O3 mfma_f64_peak.hip -o mfma_f64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters) {
  double4_t acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = double4_t{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC> void run(int waves_per_simd) {
  const int blocks = 256 * waves_per_simd, iters = 20000;
  double* out; hipMalloc(&out, (size_t)blocks * 256 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * 4 /*waves*/ * iters * NACC * 2048.0;
  printf("acc/wave %d, waves/SIMD %d: %.1f TFLOP/s (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", NACC, waves_per_simd,
         flops / ms / 1e9, 2.4e9 * ms * 1e-3 / ((double)iters * NACC * waves_per_simd));
  hipFree(out);
}
int main() { run<1>(1); run<4>(1); run<8>(1); run<4>(2); return 0; }
