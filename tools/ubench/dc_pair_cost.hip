// Microbenchmark: cycles per channel pair of the deformable-conv main loop for two GEMM formulations, with the same
// gather traffic and VALU load as the kernel.  MODE 0: exact fp32 (9 x v_mfma_f32_32x32x2_f32 + 16 ds_read_b32 + NV VALU);
// MODE 1: bf16 x 3 split, 6 products (7 x v_mfma_f32_32x32x16_bf16 per pair incl. the amortised tap-8 group,
// 3 ds_read_b128 of split weights, 16 ds_read_b32, NV + 52 VALU for the splitting).  3 or 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];

template <int MODE, int NV>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *ticks, int iters) {
  float *lds = reinterpret_cast<float *>(lds_raw);
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = 1.0f + (float)(i & 7) * 1e-3f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int base = wave * 4096 + lane * 4;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float s[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) s[i] = (float)i;
  float g[16];
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(g[i]) : "v"(base), "n"(256 * 0) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < NV + (MODE ? 52 : 0); ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s[i & 15]) : "v"(g[i & 15]), "v"(g[(i + 1) & 15]));
    if (MODE == 0) {
#pragma unroll
      for (int t = 0; t < 9; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(s[t], g[t], acc, 0, 0, 0);
    } else {
      float4 a0, a1, a2;
      asm volatile("ds_read_b128 %0, %1" : "=v"(a0) : "v"(base * 4) : "memory");
      asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(a1) : "v"(base * 4) : "memory");
      asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(a2) : "v"(base * 4) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const float4 b0 = make_float4(s[0], s[1], s[2], s[3]), b1 = make_float4(s[4], s[5], s[6], s[7]), b2 = make_float4(s[8], s[9], s[10], s[11]);
#define MF(A, B) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), acc, 0, 0, 0)
      MF(a2, b0); MF(a1, b1); MF(a0, b2); MF(a1, b0); MF(a0, b1); MF(a0, b0); MF(a1, b2);
#undef MF
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float r = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) r += acc[i] + s[i];
  if (r == 1.2345f) out[0] = r;
  if (lane == 0) ticks[blockIdx.x * 4 + wave] = t1 - t0;
}
template <int MODE, int NV> void run(const char *name, float *d_out, unsigned long long *d_t, int blocks_per_cu) {
  const int iters = 2000, blocks = 256 * blocks_per_cu;
  hipLaunchKernelGGL((k<MODE, NV>), dim3(blocks), dim3(256), 40960, 0, d_out, d_t, iters);
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks * 4);
  (void)hipMemcpy(h.data(), d_t, h.size() * 8, hipMemcpyDeviceToHost);
  double s = 0; for (auto v : h) s += v;
  const double per = s / h.size() / iters;
  printf("%-50s %d waves/SIMD: cycles per pair per wave = %7.1f  -> per SIMD-pair = %6.1f\n", name, blocks_per_cu, per, per / blocks_per_cu);
}
int main() {
  float *d_out; unsigned long long *d_t;
  (void)hipMalloc(&d_out, 64); (void)hipMalloc(&d_t, 4096 * 8);
  for (int bpc : {1, 2, 3}) {
    run<0, 58>("fp32 MFMA 9 x 32x32x2 + 16 lds + 58 VALU", d_out, d_t, bpc);
    run<1, 58>("bf16x3 7 x 32x32x16 + 16 lds + 3 b128 + 110 VALU", d_out, d_t, bpc);
    run<1, 20>("bf16x3 7 x 32x32x16 + 16 lds + 3 b128 + 72 VALU", d_out, d_t, bpc);
  }
  return 0;
}
