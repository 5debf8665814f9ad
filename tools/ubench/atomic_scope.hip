// Microbenchmark: rate of scattered fp32 atomic adds on gfx950 by memory scope.  Every block adds to the eighth of the
// buffer that belongs to ITS XCD (block b runs on XCD b % 8), so that a workgroup-scope atomic -- performed in that
// XCD's L2 -- is still correct: the kernel-end release writes the lines back.  Checks the sums and the XCD assumption.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int SCOPE>
__global__ __launch_bounds__(256) void k(float *buf, size_t per_xcd, int iters, unsigned *bad) {
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const unsigned region = blockIdx.x & 7;
  if ((xcc & 0xF) != region && threadIdx.x == 0) atomicAdd(bad, 1u);
  float *base = buf + (size_t)region * per_xcd;
  unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
  for (int i = 0; i < iters; ++i) {
    s = s * 1664525u + 1013904223u;
    float *p = base + (s >> 8) % per_xcd;
    if (SCOPE == 0) atomicAdd(p, 1.0f);
    else if (SCOPE == 1) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  }
}
template <int SCOPE> void run(const char *name, size_t per_xcd) {
  float *d; unsigned *bad;
  const size_t n = per_xcd * 8;
  (void)hipMalloc(&d, n * 4); (void)hipMalloc(&bad, 4);
  (void)hipMemset(d, 0, n * 4); (void)hipMemset(bad, 0, 4);
  const int blocks = 2048, iters = 256;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<SCOPE>, dim3(blocks), dim3(256), 0, 0, d, per_xcd, iters, bad);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<SCOPE>, dim3(blocks), dim3(256), 0, 0, d, per_xcd, iters, bad);
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<float> h(n); unsigned hb = 0;
  (void)hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
  (void)hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
  double sum = 0; for (float v : h) sum += v;
  const double total = 2.0 * blocks * 256.0 * iters;
  printf("%-34s %6.1f MB buffer: %7.3f ms  %7.1f G atomics/s   sum %.0f of %.0f %s   blocks off their XCD: %u\n", name, n * 4 / 1e6, ms,
         blocks * 256.0 * iters / ms / 1e6, sum, total, sum == total ? "OK" : "** WRONG **", hb);
  (void)hipFree(d); (void)hipFree(bad);
}
int main() {
  for (size_t per : {(size_t)1 << 16, (size_t)1 << 19, (size_t)1 << 22}) {
    run<0>("agent scope (atomicAdd)", per);
    run<1>("workgroup scope", per);
    run<2>("wavefront scope", per);
  }
  return 0;
}
