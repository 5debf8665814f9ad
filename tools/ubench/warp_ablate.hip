// Microbenchmark: where the time of the full-resolution warp (N=8, C=3, 384x512, smooth flow) goes.
// Variants of the product kernel's body (kernels/warp.h pieces) with parts removed, K launches in a hipGraph.
#include "../../maskflownet_amd/csrc/kernels/warp.h"
#include <chrono>
#include <cmath>
#include <cstdio>
#include <vector>
using namespace mfn;
// MODE 0 full; 1 no tap loads (weights only); 2 no flow load (identity grid); 3 4-byte taps; 4 copy (x -> out, reads flow too)
template <int MODE>
__global__ __launch_bounds__(256) void k(WarpParams p, unsigned total) {
  const unsigned idx = blockIdx.x * 256u + threadIdx.x;
  if (idx >= total) return;
  const unsigned W = (unsigned)p.W, H = (unsigned)p.H;
  const unsigned row = idx / W, x = idx - row * W;
  const unsigned n = row / H, y = row - n * H;
  const size_t plane = (size_t)H * W;
  const unsigned pix = y * W + x;
  const float *fl = p.flow + (size_t)n * 2 * plane + pix;
  float fy = 0.25f, fx = 0.5f;
  if (MODE != 2) { fy = fl[0]; fx = fl[plane]; }
  float gx, gy;
  warp_grid(fx, fy, (int)x, (int)y, p.H, p.W, p.clip, gx, gy);
  const Taps t = sampler_taps(gx, gy, p.H, p.W);
  const float *xin = p.x + (size_t)n * p.C * plane;
  float *o = p.out + (size_t)n * p.C * plane + pix;
  if (MODE == 4) {
    for (int c = 0; c < 3; ++c) o[(size_t)c * plane] = xin[(size_t)c * plane + pix] + fy * 0.f + fx * 0.f;
    return;
  }
  f2u a[3], b[3];
  float r[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float *pl = xin + (size_t)c * plane;
    if (MODE == 1) r[c] = t.w00 + 2.f * t.w01 + 3.f * t.w10 + 4.f * t.w11 + (float)c;
    else if (MODE == 3) r[c] = sample(pl, t);
    else { a[c] = mfn_load2u(pl + t.p0); b[c] = mfn_load2u(pl + t.p1); }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    if (MODE != 1 && MODE != 3) r[c] = combine_pairs(a[c], b[c], t);
    o[(size_t)c * plane] = r[c];
  }
}
// XCD-aware: workgroup b runs on XCD b % 8; give every XCD one contiguous band of rows so that the two
// workgroups that read a source row (output rows y-1.. y) share an L2.  ROWS output rows per thread (same x).
template <int ROWS, bool XCD>
__global__ __launch_bounds__(256) void kx(WarpParams p, unsigned nrows_g) {
  unsigned b = blockIdx.x;
  if (XCD) {
    const unsigned nb = gridDim.x, q = nb / 8, r = nb % 8, xcd = b % 8, i = b / 8;
    b = xcd * q + min(xcd, r) + i;
  }
  const unsigned W = (unsigned)p.W, H = (unsigned)p.H;
  const unsigned bpr = (W + 255) / 256;           // blocks per row group
  const unsigned rg = b / bpr, x = (b - rg * bpr) * 256 + threadIdx.x;
  if (x >= W || rg >= nrows_g) return;
  const unsigned hg = (H + ROWS - 1) / ROWS;      // row groups per image
  const unsigned n = rg / hg, y0 = (rg - n * hg) * ROWS;
  const size_t plane = (size_t)H * W;
  const float *xin = p.x + (size_t)n * p.C * plane;
  float fy[ROWS], fx[ROWS];
#pragma unroll
  for (int k = 0; k < ROWS; ++k) {
    const unsigned y = min(y0 + k, H - 1);
    const float *fl = p.flow + (size_t)n * 2 * plane + y * W + x;
    fy[k] = fl[0]; fx[k] = fl[plane];
  }
  Taps t[ROWS];
#pragma unroll
  for (int k = 0; k < ROWS; ++k) {
    float gx, gy;
    warp_grid(fx[k], fy[k], (int)x, (int)min(y0 + k, H - 1), p.H, p.W, p.clip, gx, gy);
    t[k] = sampler_taps(gx, gy, p.H, p.W);
  }
  for (int c = 0; c < p.C; ++c) {
    const float *pl = xin + (size_t)c * plane;
    f2u a[ROWS], bb[ROWS];
#pragma unroll
    for (int k = 0; k < ROWS; ++k) { a[k] = mfn_load2u(pl + t[k].p0); bb[k] = mfn_load2u(pl + t[k].p1); }
#pragma unroll
    for (int k = 0; k < ROWS; ++k)
      if (y0 + k < H) p.out[((size_t)n * p.C + c) * plane + (y0 + k) * W + x] = combine_pairs(a[k], bb[k], t[k]);
  }
}
template <class F> double timeit(hipStream_t s, F enqueue, int K) {
  hipGraph_t g; hipGraphExec_t ge;
  (void)hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
  for (int i = 0; i < K; ++i) enqueue();
  (void)hipStreamEndCapture(s, &g);
  (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  for (int i = 0; i < 3; ++i) (void)hipGraphLaunch(ge, s);
  (void)hipStreamSynchronize(s);
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < 10; ++i) (void)hipGraphLaunch(ge, s);
  (void)hipStreamSynchronize(s);
  double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (10.0 * K);
  (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
  return us;
}
int main() {
  const int N = 8, C = 3, H = 384, W = 512;
  hipStream_t s; (void)hipStreamCreate(&s);
  const size_t plane = (size_t)H * W;
  std::vector<float> hx(N * C * plane), hf(N * 2 * plane);
  for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f;
  for (int n = 0; n < N; ++n)
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) {
        hf[(n * 2 + 0) * plane + y * W + x] = 8.f * sinf(x / 50.f + n) * cosf(y / 40.f);
        hf[(n * 2 + 1) * plane + y * W + x] = 8.f * cosf(x / 45.f) * sinf(y / 60.f + n);
      }
  float *x, *f, *o;
  (void)hipMalloc(&x, hx.size() * 4); (void)hipMalloc(&f, hf.size() * 4); (void)hipMalloc(&o, hx.size() * 4);
  (void)hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(f, hf.data(), hf.size() * 4, hipMemcpyHostToDevice);
  WarpParams p{x, f, o, N, C, H, W, 0, 0};
  const unsigned total = (unsigned)(N * plane);
  const dim3 g((total + 255) / 256);
  printf("full                 %7.2f us\n", timeit(s, [&] { k<0><<<g, 256, 0, s>>>(p, total); }, 20));
  printf("no tap loads         %7.2f us\n", timeit(s, [&] { k<1><<<g, 256, 0, s>>>(p, total); }, 20));
  printf("no flow load         %7.2f us\n", timeit(s, [&] { k<2><<<g, 256, 0, s>>>(p, total); }, 20));
  printf("4-byte taps          %7.2f us\n", timeit(s, [&] { k<3><<<g, 256, 0, s>>>(p, total); }, 20));
  printf("plain copy + flow    %7.2f us\n", timeit(s, [&] { k<4><<<g, 256, 0, s>>>(p, total); }, 20));
  auto runx = [&](auto kern, int rows, const char *nm) {
    const unsigned hg = (H + rows - 1) / rows, nrg = N * hg, nb = nrg * ((W + 255) / 256);
    printf("%-20s %7.2f us\n", nm, timeit(s, [&] { kern<<<dim3(nb), 256, 0, s>>>(p, nrg); }, 20));
  };
  runx(kx<1, false>, 1, "rows=1");
  runx(kx<1, true>, 1, "rows=1 xcd");
  runx(kx<2, false>, 2, "rows=2");
  runx(kx<2, true>, 2, "rows=2 xcd");
  runx(kx<4, false>, 4, "rows=4");
  runx(kx<4, true>, 4, "rows=4 xcd");
  runx(kx<8, true>, 8, "rows=8 xcd");
  return 0;
}
