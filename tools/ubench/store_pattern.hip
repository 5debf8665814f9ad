// Microbenchmark: what the cost volume's STORE pattern alone costs on gfx950 (no loads, no arithmetic).
//   pattern 0  "rows":  a wave instruction writes 64 x 16 B = 8 runs of 128 B (corr_dma_kernel: 32-px tile rows, 9 planes per lane)
//   pattern 1  "band":  a wave instruction writes 36 x 16 B = 18 runs of 32 B in 9 different planes (corr_gram_kernel:
//               lane (g, n0) -> plane dy*9 + n0-4h, row yy, 16 B at x0+4h; buffer_store with out-of-band lanes range-dropped)
// Both write the 8 x 81 x 96 x 128 fp32 volume (31.85 MB) exactly once; policies plain / nt / sc0 sc1.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr int N = 8, H = 96, W = 128, D = 9, PLANE = H * W;

template <int POL> __device__ __forceinline__ void st(i32x4 rsrc, unsigned voff, unsigned soff, f32x4 v) {
  if (POL == 2) asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen sc0 sc1" ::"v"(v), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
  else if (POL == 1) asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen nt" ::"v"(v), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
  else asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen" ::"v"(v), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ i32x4 mk(const void* p, unsigned n) {
  const unsigned long long a = (unsigned long long)p;
  i32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xFFFFu));
  r.z = (int)n; r.w = 0x00020000;
  return r;
}
// band pattern: one wave = 8-px strip x ROWS rows of one image, as corr_gram_kernel
template <int POL, int ROWS> __global__ __launch_bounds__(256) void band(float* out, float val) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x;
  const int q = gridDim.x >> 3, xcd = b & 7, i = b >> 3;
  const int bid = xcd * q + i;   // XCD k gets image k
  const int bxs = bid % 4, rest = bid / 4, seg = rest % (H / ROWS), n = rest / (H / ROWS);
  const int x0 = (bxs * 4 + wave) * 8, ys = seg * ROWS;
  const int g = lane >> 4, n0 = lane & 15, h = g & 1, yy = g >> 1;
  const int dxi = n0 - 4 * h;
  const bool ok = dxi >= 0 && dxi < D;
  const unsigned voffS = ok ? (unsigned)(((1 - yy) * D + dxi) * PLANE + yy * W + x0 + 4 * h) * 4u : 0xFFFFFF00u;
  const unsigned vup = yy == 0 ? voffS : 0xFFFFFF00u, vlo = yy == 1 ? voffS : 0xFFFFFF00u;
  float* outn = out + (size_t)n * D * D * PLANE;
  const f32x4 v = {val, val + 1, val + 2, val + 3};
  for (int t = 0; t < ROWS / 2; ++t) {
    const i32x4 rs = mk(outn + ((long long)(ys + 2 * t) * W - (long long)D * PLANE), 0x80000000u);
#pragma unroll
    for (int e = 0; e < 10; ++e) st<POL>(rs, e == 0 ? vup : (e == 9 ? vlo : voffS), (unsigned)e * D * PLANE * 4u, v);
  }
}
// row pattern: one block of 5 waves = 32 x 4 px tile, lanes 0-31 displacement row 2w, 32-63 row 2w+1, 9 planes each
template <int POL> __global__ __launch_bounds__(320) void rows(float* out, float val) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x;
  const int q = gridDim.x >> 3, xcd = b & 7, i = b >> 3;
  const int bid = xcd * q + i;
  const int n = bid / 96, t = bid % 96, ty = t / 4, tx = t % 4;
  const int dyi = wave * 2 + (lane >> 5);
  const int l = lane & 31, row = l >> 3, gx = l & 7;
  if (dyi >= D) return;
  float* dst = out + (size_t)n * D * D * PLANE + (size_t)(dyi * D) * PLANE + (size_t)(ty * 4 + row) * W + tx * 32 + 4 * gx;
  const f32x4 v = {val, val + 1, val + 2, val + 3};
  const i32x4 rs = mk(dst - (size_t)lane * 0, 0x80000000u);
  (void)rs;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    float* p = dst + (size_t)d * PLANE;
    if (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    else if (POL == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
  }
}
template <class F> static double time_us(F launch, int reps) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int i = 0; i < 20; ++i) launch();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a, 0);
  for (int i = 0; i < reps; ++i) launch();
  (void)hipEventRecord(b, 0);
  (void)hipEventSynchronize(b);
  float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
  return ms * 1000.0 / reps;
}
int main() {
  const size_t bytes = (size_t)N * D * D * PLANE * 4;
  float* out; (void)hipMalloc(&out, bytes + (1 << 20));
  (void)hipMemset(out, 0, bytes);
  const double mb = bytes / 1e6;
  printf("volume %.2f MB; time per launch (back to back, incl. ~1.5 us launch boundary), GB/s\n", mb);
#define RUN(name, expr) { const double us = time_us([&] { expr; }, 200); printf("%-34s %7.2f us  %7.0f GB/s\n", name, us, mb / us * 1e3); }
  RUN("rows  plain", hipLaunchKernelGGL(rows<0>, dim3(768), dim3(320), 0, 0, out, 1.f));
  RUN("rows  nt", hipLaunchKernelGGL(rows<1>, dim3(768), dim3(320), 0, 0, out, 1.f));
  RUN("rows  sc0 sc1", hipLaunchKernelGGL(rows<2>, dim3(768), dim3(320), 0, 0, out, 1.f));
  RUN("band  plain    rows/item 6", hipLaunchKernelGGL((band<0, 6>), dim3(512), dim3(256), 0, 0, out, 1.f));
  RUN("band  nt       rows/item 6", hipLaunchKernelGGL((band<1, 6>), dim3(512), dim3(256), 0, 0, out, 1.f));
  RUN("band  sc0 sc1  rows/item 6", hipLaunchKernelGGL((band<2, 6>), dim3(512), dim3(256), 0, 0, out, 1.f));
  RUN("band  plain    rows/item 12", hipLaunchKernelGGL((band<0, 12>), dim3(256), dim3(256), 0, 0, out, 1.f));
  RUN("band  sc0 sc1  rows/item 12", hipLaunchKernelGGL((band<2, 12>), dim3(256), dim3(256), 0, 0, out, 1.f));
  // check coverage of the band pattern: every element written once
  (void)hipMemset(out, 0, bytes);
  hipLaunchKernelGGL((band<0, 6>), dim3(512), dim3(256), 0, 0, out, 1.f);
  (void)hipDeviceSynchronize();
  std::vector<float> h(bytes / 4);
  (void)hipMemcpy(h.data(), out, bytes, hipMemcpyDeviceToHost);
  size_t zeros = 0; for (float f : h) zeros += f == 0.f;
  printf("band coverage: %zu of %zu elements unwritten\n", zeros, h.size());
  return 0;
}
