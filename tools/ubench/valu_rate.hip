// Microbenchmark: issue rate of v_fma_f32 / v_pk_fma_f32 (with and without op_sel swizzles) on gfx950,
// one wave per SIMD and 4 waves per SIMD.  Reports cycles per wave-instruction (s_memtime ticks).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float* out, unsigned long long* ticks, int iters, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 * 0.5f, b0 = 1.0001f, b1 = 0.9999f;
  f32x2 pa = {a0, a1}, pb = {b0, b1};
  float s[16]; f32x2 p[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { s[i] = (float)i; p[i] = (f32x2){(float)i, (float)-i}; }
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (MODE == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s[i]) : "v"(a0), "v"(b0));
      if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(pa), "v"(pb));
      if (MODE == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[0,0,1]" : "+v"(p[i]) : "v"(pa), "v"(pb));
      if (MODE == 3) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(s[i]) : "v"(a0), "v"(b0));
      if (MODE == 4) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p[i]) : "v"(pa), "v"(pb));
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float acc = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc += s[i] + p[i].x + p[i].y;
  if (acc == 1.2345f) out[0] = acc;
  if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}
template <int MODE> void run(const char* name, float* d_out, unsigned long long* d_t) {
  for (int waves : {4, 8, 16}) {
    const int iters = 2000;
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(64 * waves), 0, 0, d_out, d_t, iters, 1.0f);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * waves);
    (void)hipMemcpy(h.data(), d_t, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += v;
    const double per = s / h.size() / (iters * 16.0);
    printf("%-34s waves/CU=%2d  ticks per wave-instr = %6.2f   per SIMD-instr = %5.2f\n", name, waves, per, per / (waves / 4.0));
  }
}
int main() {
  float* d_out; unsigned long long* d_t;
  (void)hipMalloc(&d_out, 64); (void)hipMalloc(&d_t, 4096 * 8);
  run<0>("v_fma_f32", d_out, d_t);
  run<3>("v_fmac_f32", d_out, d_t);
  run<1>("v_pk_fma_f32", d_out, d_t);
  run<2>("v_pk_fma_f32 op_sel swizzle", d_out, d_t);
  run<4>("v_pk_mul_f32", d_out, d_t);
  return 0;
}
