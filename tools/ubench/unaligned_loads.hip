// Microbenchmark: vector-memory instruction throughput of one CU-filling grid by ACCESS SHAPE of a 64-lane load, all L1/L2 hits.
//   A  dwordx4, lane l at byte 16 l          (aligned, disjoint: 1 KB per instruction)
//   B  dwordx4, lane l at byte 4 l           (dword-aligned, neighbouring lanes overlap by 12 B: a pixel row's 4-wide windows)
//   C  dwordx4, 8 lanes x 4 B apart, rows of 8 lanes 512 B apart, two channel planes (the weight-gradient producer's shape)
//   D  as C with four dword loads per lane instead of one dwordx4
//   E  as C with two ALIGNED dwordx4 per lane covering the window (then selected in registers)
// Build: hipcc --offload-arch=gfx950 -O3 -o unaligned_loads unaligned_loads.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f4a __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k(const float *buf, float *out, int iters, int plane) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float *base = buf + (size_t)((blockIdx.x * 4 + wave) % 64) * 4096;
  float acc = 0.f;
  for (int i = 0; i < iters; ++i) {
    const float *p = base + (i & 15) * 128;
    if (MODE == 0) { f4a v = *(const f4a *)(p + lane * 4); acc += v.x + v.y + v.z + v.w; }
    if (MODE == 1) { f4u v = *(const f4u *)(p + lane + 1); acc += v.x + v.y + v.z + v.w; }
    const int j = lane & 31, half = lane >> 5;
    const float *q = p + half * plane + (j >> 3) * 128 + (j & 7) + 1;
    if (MODE == 2) { f4u v = *(const f4u *)q; acc += v.x + v.y + v.z + v.w; }
    if (MODE == 3) { acc += q[0] + q[1] + q[2] + q[3]; }
    if (MODE == 4) {
      const float *qa = p + half * plane + (j >> 3) * 128 + ((j & 7) & ~3);
      f4a v = *(const f4a *)qa, w = *(const f4a *)(qa + 4);
      acc += v.x + v.y + v.z + v.w + w.x + w.y + w.z + w.w;
    }
    asm volatile("" ::: "memory");
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int MODE> void run(const char *name) {
  float *d, *o;
  const int plane = 96 * 128;
  (void)hipMalloc(&d, (size_t)(64 * 4096 + 2 * plane + 4096) * 4); (void)hipMemset(d, 0, (size_t)(64 * 4096 + 2 * plane + 4096) * 4);
  (void)hipMalloc(&o, 1024 * 256 * 4);
  const int blocks = 1024, iters = 2048;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, o, iters, plane);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, o, iters, plane);
  (void)hipEventRecord(e1, 0); (void)hipDeviceSynchronize();
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  const double per_cu = (double)blocks * 4 * iters / 256.0;   // wave-level iterations per CU
  printf("%-64s %7.3f ms  %6.1f shader cycles (2.1 GHz) per wave iteration per CU\n", name, ms, ms * 1e-3 * 2.1e9 / per_cu);
}
int main() {
  run<0>("A dwordx4 aligned, disjoint (1 KB / instr)");
  run<1>("B dwordx4 at 4-byte steps (overlapping)");
  run<2>("C dwordx4 unaligned: 4 rows x 8 lanes x 2 planes");
  run<3>("D the same as 4 dword loads");
  run<4>("E the same as 2 aligned dwordx4");
  return 0;
}
