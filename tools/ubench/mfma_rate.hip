// Microbenchmark: issue interval of v_mfma_f32_32x32x2_f32 per SIMD with 1..4 waves per SIMD and 1 or 2 independent
// accumulator chains per wave (s_memtime cycles per MFMA per SIMD, and the TFLOP/s that corresponds to on 256 CUs).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CHAINS>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *rec, int iters) {
  f32x16 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c)
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  const float a = threadIdx.x * 1e-3f, b = 1.0001f;
  const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  float r = 0;
  for (int c = 0; c < CHAINS; ++c)
    for (int i = 0; i < 16; ++i) r += acc[c][i];
  if (r == 1.2345f) out[0] = r;
  if ((threadIdx.x & 63) == 0) { rec[(blockIdx.x * 4 + threadIdx.x / 64) * 2] = c1 - c0; rec[(blockIdx.x * 4 + threadIdx.x / 64) * 2 + 1] = w1 - w0; }
}
template <int CHAINS> void run(float *out, unsigned long long *rec, int blocks_per_cu) {
  const int blocks = 256 * blocks_per_cu, iters = 20000;
  hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(256), 0, 0, out, rec, iters);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(256), 0, 0, out, rec, iters);
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks * 8);
  (void)hipMemcpy(h.data(), rec, h.size() * 8, hipMemcpyDeviceToHost);
  double c = 0, w = 0;
  for (int i = 0; i < blocks * 4; ++i) { c += h[2 * i]; w += h[2 * i + 1]; }
  const double per_wave = c / (blocks * 4) / (iters * 8.0 * CHAINS);
  const double flops = (double)blocks * 4 * iters * 8 * CHAINS * 4096.0;
  printf("%d waves/SIMD, %d chain(s): %6.1f cycles per MFMA per wave = %5.1f per SIMD; clock %.2f GHz; kernel %.2f ms -> %.1f TFLOP/s\n",
         blocks_per_cu, CHAINS, per_wave, per_wave / blocks_per_cu, c / w * 0.1, ms, flops / ms / 1e9);
}
int main() {
  float *out; unsigned long long *rec;
  (void)hipMalloc(&out, 64); (void)hipMalloc(&rec, 1024 * 8 * 8);
  for (int b : {1, 2, 3, 4}) { run<1>(out, rec, b); run<2>(out, rec, b); }
  return 0;
}
