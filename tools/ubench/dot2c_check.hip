#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
__global__ void k(float* o, const float* x) {
  float x0 = x[threadIdx.x], x1 = x[threadIdx.x + 64];
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 v = {x0, x1};
  unsigned hp = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2));
  float r0 = x0, r1 = x1;
  asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(r0) : "s"(0x0000bf80u), "v"(hp));   // {lo = -1.0, hi = 0}
  asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(r1) : "s"(0xbf800000u), "v"(hp));   // {lo = 0, hi = -1.0}
  o[threadIdx.x] = r0; o[threadIdx.x + 64] = r1;
}
int main() {
  float *x, *o; hipMalloc(&x, 512); hipMalloc(&o, 512);
  float h[128]; for (int i = 0; i < 128; ++i) h[i] = 1.2345678f * (i + 1) * (i % 3 ? 1.f : -0.001f);
  hipMemcpy(x, h, 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, x);
  float r[128]; hipMemcpy(r, o, 512, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 128; ++i) {
    unsigned u; memcpy(&u, &h[i], 4); unsigned ub = u + 0x7fffu + ((u >> 16) & 1u); ub &= 0xffff0000u; float hf; memcpy(&hf, &ub, 4);
    if (r[i] != h[i] - hf) { if (bad < 5) printf("i=%d x=%g got %g want %g\n", i, h[i], r[i], h[i] - hf); ++bad; }
  }
  printf("dot2c residual check: %d mismatches of 128\n", bad);
  return 0;
}
