// Microbenchmark: the shader clock a kernel actually runs at.  Each wave reads the shader-cycle counter (s_memtime) and
// the constant 100 MHz wall clock (s_memrealtime) around a long loop of (a) fp32 MFMAs, (b) packed fp32 FMAs,
// (c) ds_read_b128, (d) global loads; f = d(cycles) / d(wall).  The peaks in MI355X_MICROARCH.md are quoted at 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, const float *in, unsigned long long *rec, int iters) {
  float *lds = reinterpret_cast<float *>(lds_raw);
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = 1.0f;
  __syncthreads();
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  f32x2 p[8];
  for (int i = 0; i < 8; ++i) p[i] = (f32x2){0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  const f32x2 pa = {a, b}, pb = {b, a};
  float4 v = make_float4(0, 0, 0, 0);
  const int addr = (threadIdx.x & 255) * 16;
  const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    } else if (MODE == 1) {
#pragma unroll
      for (int u = 0; u < 32; ++u) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[u & 7]) : "v"(pa), "v"(pb));
    } else if (MODE == 2) {
#pragma unroll
      for (int u = 0; u < 16; ++u) { asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory"); }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      a += v.x;
    } else {
#pragma unroll
      for (int u = 0; u < 8; ++u) a += in[((size_t)(it * 8 + u) * 65536 + blockIdx.x * 256 + threadIdx.x) & ((1u << 26) - 1)];
    }
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  float r = a;
  for (int i = 0; i < 16; ++i) r += acc[i];
  for (int i = 0; i < 8; ++i) r += p[i].x + p[i].y;
  if (r == 1.2345f) out[0] = r;
  if ((threadIdx.x & 63) == 0) { rec[(blockIdx.x * 4 + threadIdx.x / 64) * 2] = c1 - c0; rec[(blockIdx.x * 4 + threadIdx.x / 64) * 2 + 1] = w1 - w0; }
}
template <int MODE> void run(const char *name, float *out, float *in, unsigned long long *rec, int iters) {
  const int blocks = 768;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 16384, 0, out, in, rec, iters);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 8);
    (void)hipMemcpy(h.data(), rec, h.size() * 8, hipMemcpyDeviceToHost);
    double c = 0, w = 0;
    for (int i = 0; i < blocks * 4; ++i) { c += h[2 * i]; w += h[2 * i + 1]; }
    printf("%-26s run %d: %8.2f ms per wave, shader clock %.3f GHz\n", name, rep, w / (blocks * 4) / 1e5, c / w * 0.1);
  }
}
int main() {
  float *out, *in; unsigned long long *rec;
  (void)hipMalloc(&out, 64); (void)hipMalloc(&in, (size_t)1 << 28); (void)hipMalloc(&rec, 768 * 8 * 8);
  (void)hipMemset(in, 0, (size_t)1 << 28);
  run<0>("fp32 MFMA 32x32x2", out, in, rec, 40000);
  run<1>("v_pk_fma_f32", out, in, rec, 100000);
  run<2>("ds_read_b128", out, in, rec, 100000);
  run<3>("global loads", out, in, rec, 4000);
  return 0;
}
