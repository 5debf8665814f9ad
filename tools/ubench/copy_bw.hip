// Microbenchmark: what a plain streaming kernel achieves on MI355X as a function of the bytes it moves --
// the practical ceiling for the 50-60 MB kernels of the pass (warp, level-2 correlation).  read -> write copy,
// one float (256 B per wave instruction, the shape of the warp kernel's accesses) or one float4 per thread,
// and a read-only / write-only split; K back-to-back launches captured in a hipGraph, wall clock / K.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <class T> __global__ __launch_bounds__(256) void copy_k(const T* __restrict__ a, T* __restrict__ b, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) b[i] = a[i];
}
__global__ __launch_bounds__(256) void read_k(const float4* __restrict__ a, float* sink, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) { float4 v = a[i]; if (v.x == 1.2345e30f) sink[0] = v.y; }
}
__global__ __launch_bounds__(256) void write_k(float4* b, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) b[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
template <class F> double timeit(hipStream_t s, F enqueue, int K) {
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
  for (int i = 0; i < K; ++i) enqueue();
  hipStreamEndCapture(s, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  for (int i = 0; i < 3; ++i) hipGraphLaunch(ge, s);
  hipStreamSynchronize(s);
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < 10; ++i) hipGraphLaunch(ge, s);
  hipStreamSynchronize(s);
  double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (10.0 * K);
  hipGraphExecDestroy(ge); hipGraphDestroy(g);
  return us;
}
int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  const size_t maxb = (size_t)1 << 30;
  float *a, *b, *sink; CK(hipMalloc(&a, maxb)); CK(hipMalloc(&b, maxb)); CK(hipMalloc(&sink, 256));
  CK(hipMemsetAsync(a, 0, maxb, s)); CK(hipMemsetAsync(b, 0, maxb, s));
  printf("%10s %28s %28s %20s %20s\n", "MB moved", "copy float (us, TB/s)", "copy float4 (us, TB/s)", "read only", "write only");
  const double mbs[] = {6.3, 12.6, 25.2, 50.3, 100.7, 201.3, 402.7, 805.3, 1610.6};
  for (double mb : mbs) {
    const size_t half = (size_t)(mb * 1e6 / 2) / 16 * 16;   // bytes read = bytes written
    if (half > maxb) break;
    const size_t n1 = half / 4, n4 = half / 16;
    const int K = 20;
    double u1 = timeit(s, [&] { copy_k<float><<<dim3((unsigned)((n1 + 255) / 256)), 256, 0, s>>>(a, b, n1); }, K);
    double u4 = timeit(s, [&] { copy_k<float4><<<dim3((unsigned)((n4 + 255) / 256)), 256, 0, s>>>((const float4*)a, (float4*)b, n4); }, K);
    double ur = timeit(s, [&] { read_k<<<dim3((unsigned)((2 * n4 + 255) / 256)), 256, 0, s>>>((const float4*)a, sink, 2 * n4 > maxb / 16 ? maxb / 16 : 2 * n4); }, K);
    double uw = timeit(s, [&] { write_k<<<dim3((unsigned)((2 * n4 + 255) / 256)), 256, 0, s>>>((float4*)b, 2 * n4 > maxb / 16 ? maxb / 16 : 2 * n4); }, K);
    // the same float4 copy with every launch of the graph on its own slice of the 1 GB buffers (K * half <= 1 GB): what a launch of this
    // size costs when neither its input nor its output is in the 256 MB Infinity Cache (bench.py's hbm_rotated case)
    double urot = -1.0;
    if ((size_t)K * half <= maxb) {
      int it = 0;
      urot = timeit(s, [&] { const size_t o = (size_t)(it++ % K) * (half / 16);
                             copy_k<float4><<<dim3((unsigned)((n4 + 255) / 256)), 256, 0, s>>>((const float4*)a + o, (float4*)b + o, n4); }, K);
    }
    const double bytes = 2.0 * half;
    printf("%10.1f %16.2f %10.2f %16.2f %10.2f %10.2f %8.2f %10.2f %8.2f\n", bytes / 1e6, u1, bytes / u1 / 1e6, u4, bytes / u4 / 1e6,
           ur, bytes / ur / 1e6, uw, bytes / uw / 1e6);
    if (urot > 0) printf("%10s %16s %10s %16.2f %10.2f   <- float4 copy, every launch on its own slice (cache-cold)\n", "", "", "", urot, bytes / urot / 1e6);
  }
  return 0;
}
