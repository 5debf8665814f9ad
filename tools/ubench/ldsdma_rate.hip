// Microbenchmark: how fast ONE CU can pull L2-resident data into its LDS, by transfer kind and by the number of waves
// asking -- the bound of the staged kernels at the coarse levels (a level-3 correlation tile streams 196 KB through LDS,
// a level-3 deformable-conv CU 600 KB) is not HBM but this number.
//   dma   : buffer_load_dwordx4 ... lds (16 B per lane, 1 KB per wave instruction, no VGPRs)
//   regs  : global_load_dwordx4 into VGPRs (8 in flight per lane), no LDS write
//   regs+w: the same + ds_write_b128 of every quad (what a register-staged kernel does)
// One block per CU (256 blocks), W waves per block, the block's waves interleave over a per-block 64 KB region (32 blocks x 64 KB = 2 MB per XCD: L2 resident)
// (L2 / L1 resident after the first touch) `iters` times.  Reported: GB/s per CU and B/clk/CU at the measured clock.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef int rsrc_t __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ rsrc_t make_rsrc(const void *p, unsigned nbytes) {
  const unsigned long long a = (unsigned long long)p;
  rsrc_t r;
  r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xFFFFu));
  r.z = __builtin_amdgcn_readfirstlane((int)nbytes);
  r.w = 0x00020000;
  return r;
}
extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
template <int MODE>
__global__ __launch_bounds__(1024) void k(const float *src, float *sink, int iters, int region_bytes) {
  float *lds = reinterpret_cast<float *>(lds_raw);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  const char *base = (const char *)src + (size_t)blockIdx.x * region_bytes;
  const int per_wave = region_bytes / 1024 / nw;      // 1 KB wave instructions per wave per pass over the region (multiple of 4)
  float acc = 0.f;
  const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds + wave * 1024));  // 4 KB of LDS per wave
  if (MODE == 0) {
    const rsrc_t r = make_rsrc(base, (unsigned)region_bytes);
    for (int it = 0; it < iters; ++it)
      for (int i = 0; i < per_wave; ++i) {
        const unsigned voff = (unsigned)((i * nw + wave) * 1024 + lane * 16);
        const unsigned la = lds_addr + (unsigned)((i & 3) * 1024);
        asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(la), "v"(voff), "s"(r) : "memory", "m0");
        if ((i & 3) == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc = lds[wave * 1024 + lane];
  } else {
    const unsigned la = lds_addr + (unsigned)(lane * 16);
    for (int it = 0; it < iters; ++it)
      for (int i = 0; i < per_wave; i += 4) {
        f4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const f4 *p = (const f4 *)(base + (size_t)((i + u) * nw + wave) * 1024) + lane;
          asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[u]) : "v"(p) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3])::"memory");
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (MODE == 2) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(la), "v"(v[u]), "n"(0) : "memory");
          else acc += v[u].x;
        }
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (MODE == 2) acc += lds[wave * 1024 + lane];
  }
  if (acc == 1.2345e30f) sink[0] = acc;
}
int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  const int nblk = 256, region = 64 << 10;
  float *src, *sink; CK(hipMalloc(&src, (size_t)nblk * region)); CK(hipMalloc(&sink, 256));
  CK(hipMemsetAsync(src, 0, (size_t)nblk * region, s));
  int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
  printf("%-8s %6s %12s %12s %14s\n", "kind", "waves", "us", "GB/s per CU", "TB/s chip");
  const int iters = 64;
  for (int mode = 0; mode < 3; ++mode)
    for (int nw : {1, 2, 4, 8, 16}) {
      auto launch = [&] {
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(nblk), dim3(nw * 64), 65536, s, src, sink, iters, region);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(nblk), dim3(nw * 64), 65536, s, src, sink, iters, region);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(nblk), dim3(nw * 64), 65536, s, src, sink, iters, region);
      };
      for (int i = 0; i < 3; ++i) launch();
      CK(hipStreamSynchronize(s));
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < 10; ++i) launch();
      CK(hipStreamSynchronize(s));
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 10.0;
      const double bytes_cu = (double)region * iters;
      printf("%-8s %6d %12.1f %12.1f %14.2f\n", mode == 0 ? "dma" : (mode == 1 ? "regs" : "regs+w"), nw, us, bytes_cu / us / 1e3,
             bytes_cu * nblk / us / 1e6);
    }
  return 0;
}
