// Microbenchmark / check: the residual x - float(bf16(x)) of four values per lane by ONE v_mfma_f32_4x4x4_16B_bf16 with a constant selector
// (A = -identity per 4 x 4 block: lane l holds -1.0 in K slot l % 4), against the VALU form, bit for bit; and the issue cost of that
// instruction next to v_mfma_f32_16x16x32_bf16 (cycles per instruction, one wave and four waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void check(const float *x, float *r_mfma, float *r_valu) {
  const int l = threadIdx.x;
  f32x4 c = {x[4 * l], x[4 * l + 1], x[4 * l + 2], x[4 * l + 3]};
  const f32x2 p0 = {c[0], c[1]}, p1 = {c[2], c[3]};
  const unsigned w0 = __builtin_bit_cast(unsigned, __builtin_convertvector(p0, bf2)), w1 = __builtin_bit_cast(unsigned, __builtin_convertvector(p1, bf2));
  unsigned hw[2] = {w0, w1};
  s16x4 h; memcpy(&h, hw, 8);
  unsigned sw[2] = {0u, 0u};
  sw[(l & 3) >> 1] = 0xBF80u << (16 * (l & 1));
  s16x4 sel; memcpy(&sel, sw, 8);
  f32x4 d = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(sel, h, c, 0, 0, 0);
  for (int i = 0; i < 4; ++i) r_mfma[4 * l + i] = d[i];
  for (int i = 0; i < 4; ++i) {
    const unsigned hb = ((i < 2 ? w0 : w1) >> (16 * (i & 1))) << 16;
    r_valu[4 * l + i] = c[i] - __builtin_bit_cast(float, hb);
  }
}
template <int MODE> __global__ __launch_bounds__(1024) void rate(float *out, unsigned long long *cyc, int iters) {
  f32x4 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  s16x4 x4 = {1, 2, 3, 4}, y4 = {5, 6, 7, 8};
  bf16x8 x8, y8;
  for (int e = 0; e < 8; ++e) { x8[e] = (__bf16)(e + threadIdx.x * 0.01f); y8[e] = (__bf16)(e * 0.5f); }
  const unsigned long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
      a0 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(x4, y4, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(x4, y4, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(x4, y4, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(x4, y4, a3, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(x4, y4, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(x4, y4, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(x4, y4, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(x4, y4, a3, 0, 0, 0);
    } else {
      a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x8, y8, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x8, y8, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x8, y8, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x8, y8, a3, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x8, y8, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x8, y8, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x8, y8, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x8, y8, a3, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const unsigned long long t1 = clock64();
  float s = 0; for (int i = 0; i < 4; ++i) s += a0[i] + a1[i] + a2[i] + a3[i];
  if (s == 1.2345e30f) out[0] = s;
  if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}
int main() {
  float hx[256], hm[256], hv[256];
  for (int i = 0; i < 256; ++i) hx[i] = (i % 7 == 0 ? -1.f : 1.f) * (0.37f + i * 1.618033f) * powf(10.f, (i % 13) - 6);
  float *x, *rm, *rv; unsigned long long *cyc; float *out;
  hipMalloc(&x, 1024); hipMalloc(&rm, 1024); hipMalloc(&rv, 1024); hipMalloc(&cyc, 1024); hipMalloc(&out, 64);
  hipMemcpy(x, hx, 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(check, dim3(1), dim3(64), 0, 0, x, rm, rv);
  hipMemcpy(hm, rm, 1024, hipMemcpyDeviceToHost); hipMemcpy(hv, rv, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 256; ++i) if (memcmp(&hm[i], &hv[i], 4)) { if (bad < 6) printf("  lane %d reg %d: x %.9g mfma %.9g valu %.9g\n", i / 4, i % 4, hx[i], hm[i], hv[i]); ++bad; }
  printf("selector v_mfma_f32_4x4x4_16B_bf16 residual against the VALU residual: %d of 256 values differ\n", bad);
  const int iters = 4000;
  for (int mode = 0; mode < 2; ++mode)
    for (int w : {1, 2, 4}) {
      for (int rep = 0; rep < 2; ++rep) {
        if (mode == 0) hipLaunchKernelGGL(rate<0>, dim3(256), dim3(256 * w), 0, 0, out, cyc, iters);
        else hipLaunchKernelGGL(rate<1>, dim3(256), dim3(256 * w), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
      }
      unsigned long long h[16]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
      double m = 0; for (int i = 0; i < 4 * w; ++i) m += (double)h[i]; m /= 4 * w * iters * 8.0;
      printf("%-28s waves/SIMD %d: %6.1f cycles per instruction per wave (%5.1f per SIMD)\n", mode == 0 ? "v_mfma_f32_4x4x4_16B_bf16" : "v_mfma_f32_16x16x32_bf16", w, m, m / w);
    }
  return 0;
}
