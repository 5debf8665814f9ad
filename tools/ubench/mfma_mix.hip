// Microbenchmark: how many VALU instructions hide behind v_mfma_f32_32x32x2_f32 on gfx950.
// Each wave runs a chain of MFMAs on NACC accumulators with FILL independent v_fma_f32 after each one;
// 1, 2 or 3 waves per SIMD.  Reports shader cycles per MFMA per SIMD (the matrix pipe needs 64).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int FILL, int NACC>
__global__ void k(float* out, unsigned long long* ticks, int iters, float seed) {
  float a = seed + threadIdx.x, b = 1.0001f;
  float s[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) s[i] = (float)i;
  f32x16 acc[NACC];
#pragma unroll
  for (int n = 0; n < NACC; ++n)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      acc[t % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t % NACC], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < FILL; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s[i % 16]) : "v"(a), "v"(b));
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float r = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) r += s[i];
#pragma unroll
  for (int n = 0; n < NACC; ++n) r += acc[n][0] + acc[n][7];
  if (r == 1.2345f) out[0] = r;
  if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}
template <int FILL, int NACC> void run(float* d_out, unsigned long long* d_t) {
  printf("fill %2d VALU/MFMA, %d accumulator(s):", FILL, NACC);
  for (int waves : {4, 8, 12}) {
    const int iters = 500;
    hipLaunchKernelGGL((k<FILL, NACC>), dim3(256), dim3(64 * waves), 0, 0, d_out, d_t, iters, 1.0f);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * waves);
    (void)hipMemcpy(h.data(), d_t, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += v;
    const double per = s / h.size() / (iters * 8.0);
    printf("   %d w/SIMD: %6.1f cyc/MFMA/SIMD", waves / 4, per / (waves / 4.0));
  }
  printf("\n");
}
int main() {
  float* d_out; unsigned long long* d_t;
  (void)hipMalloc(&d_out, 64); (void)hipMalloc(&d_t, 4096 * 8);
  run<0, 1>(d_out, d_t); run<4, 1>(d_out, d_t); run<8, 1>(d_out, d_t); run<12, 1>(d_out, d_t); run<16, 1>(d_out, d_t); run<24, 1>(d_out, d_t);
  run<0, 2>(d_out, d_t); run<8, 2>(d_out, d_t); run<16, 2>(d_out, d_t);
  return 0;
}
