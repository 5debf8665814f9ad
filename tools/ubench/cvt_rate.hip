// Microbenchmark: issue rate of the instructions of the fp32 -> 3 x bf16 operand split on gfx950 (v_cvt_pk_bf16_f32 against the
// bit operations that could replace it), 1 / 2 / 4 waves per SIMD.  Reports s_memtime ticks per wave-instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float* out, unsigned long long* ticks, int iters, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 * 0.5f;
  f32x2 pa = {a0, a1}, pb = {1.0001f, 0.9999f};
  unsigned s[16]; f32x2 p[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { s[i] = i; p[i] = (f32x2){(float)i, (float)-i}; }
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (MODE == 0) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(s[i]) : "v"(a0), "v"(a1));
      if (MODE == 1) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(s[i]) : "v"(a0));
      if (MODE == 2) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(s[i]) : "v"(a0), "v"(a1), "v"(0x07060302u));
      if (MODE == 3) asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(p[i]) : "v"(pa), "v"(pb));
      if (MODE == 4) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(s[i]) : "v"(a0));
      if (MODE == 5) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p[i]) : "v"(pa), "v"(pb));
      if (MODE == 6) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(p[i].x) : "v"(a0), "v"(a1));
      if (MODE == 7) asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(s[i]) : "v"(a0), "v"(0xffff0000u), "v"(a1));
      if (MODE == 8) asm volatile("v_bfi_b32 %0, %1, %2, %3" : "=v"(s[i]) : "v"(0xffff0000u), "v"(a0), "v"(a1));
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float acc = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc += (float)s[i] + p[i].x + p[i].y;
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
  run<6>("v_fma_f32 (reference)", d_out, d_t);
  run<0>("v_cvt_pk_bf16_f32", d_out, d_t);
  run<1>("v_and_b32", d_out, d_t);
  run<2>("v_perm_b32", d_out, d_t);
  run<4>("v_lshlrev_b32", d_out, d_t);
  run<7>("v_and_or_b32", d_out, d_t);
  run<8>("v_bfi_b32", d_out, d_t);
  run<3>("v_pk_add_f32 (neg)", d_out, d_t);
  run<5>("v_pk_mul_f32", d_out, d_t);
  return 0;
}
