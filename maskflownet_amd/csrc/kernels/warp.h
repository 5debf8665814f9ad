// warp.h -- flow warp (GridGenerator 'warp' + BilinearSampler fused) and the two MXNet operators
// on their own, for gfx950.
//
// Replaces /root/reference/network/layer.py:14-18 (Reconstruction2D) and :26-30
// (Reconstruction2DSmooth); semantics as oracle/mfn_ref_body.inc warp_fwd / bilinear_sampler_fwd.
// Gather-bound (4*N*H*W*(2C+2) bytes): one thread owns a pixel, computes the 4 tap
// addresses/weights once and reuses them for every channel.  The grid never exists in memory.
// warp_fwd_fast_kernel is what the pass runs (a wave is a 16 x 4 pixel tile where W % 64 == 0 and H % 4 == 0, 64 pixels of
// a row elsewhere); warp_fwd_kernel<1> is the general form (any size, W = 1).
#pragma once
#include "../mfn_rt.h"

#ifndef MFN_WARP_1D
#define MFN_WARP_1D 0   // measurement hook: 1 = a wave is 64 pixels of one row everywhere
#endif

namespace mfn {

struct WarpParams {
  const float *x;
  const float *flow;  // (N,2,H,W): ch0 = dy, ch1 = dx
  float *out;
  int N, C, H, W;
  int clip;
  int st_policy;  // cache policy of the output stores (mfn_store1_stream; fast kernel)
};

struct Taps {
  int i00, i01, i10, i11;  // element offsets inside one channel plane (clamped when masked)
  float w00, w01, w10, w11;
  // paired form: the two taps of a row are adjacent in memory, so one 8-byte load at column
  // pb = clamp(tx, 0, W-2) serves both.  sel0 / sel1: the left / right tap is the pair's second / first element
  int p0, p1;        // element offsets of the two pairs (rows cy0, cy1)
  bool sel0, sel1;   // left tap = pair.y (tx == W-1);  right tap = pair.x (tx == -1)
};

// grid value -> taps, exactly the BilinearSamplerForward arithmetic (fp32 round trip included)
__device__ __forceinline__ Taps sampler_taps(float gx, float gy, int iH, int iW) {
  const float y_real = (gy + 1.f) * (float)(iH - 1) / 2.f;
  const float x_real = (gx + 1.f) * (float)(iW - 1) / 2.f;
  const float fy = floorf(y_real), fx = floorf(x_real);
  // saturating float->int keeps absurd coordinates (|flow| ~ 1e9) out of range instead of UB
  const int ty = (int)fminf(fmaxf(fy, -2.f), (float)iH + 1.f);
  const int tx = (int)fminf(fmaxf(fx, -2.f), (float)iW + 1.f);
  const float wy = 1.f - (y_real - fy);
  const float wx = 1.f - (x_real - fx);
  const bool y0 = ty >= 0 && ty <= iH - 1, y1 = ty + 1 >= 0 && ty + 1 <= iH - 1;
  const bool x0 = tx >= 0 && tx <= iW - 1, x1 = tx + 1 >= 0 && tx + 1 <= iW - 1;
  Taps t;
  t.w00 = (y0 && x0) ? wy * wx : 0.f;
  t.w01 = (y0 && x1) ? wy * (1.f - wx) : 0.f;
  t.w10 = (y1 && x0) ? (1.f - wy) * wx : 0.f;
  t.w11 = (y1 && x1) ? (1.f - wy) * (1.f - wx) : 0.f;
  const int cy0 = min(max(ty, 0), iH - 1), cy1 = min(max(ty + 1, 0), iH - 1);
  const int cx0 = min(max(tx, 0), iW - 1), cx1 = min(max(tx + 1, 0), iW - 1);
  t.i00 = cy0 * iW + cx0;
  t.i01 = cy0 * iW + cx1;
  t.i10 = cy1 * iW + cx0;
  t.i11 = cy1 * iW + cx1;
  const int pb = min(max(tx, 0), max(iW - 2, 0));
  t.p0 = cy0 * iW + pb;
  t.p1 = cy1 * iW + pb;
  t.sel0 = tx != pb;      // left tap sits in the pair's second slot (only when tx == W-1; masked otherwise)
  t.sel1 = tx + 1 == pb;  // right tap sits in the pair's first slot (only when tx == -1; masked otherwise)
  return t;
}

__device__ __forceinline__ float sample(const float *plane, const Taps &t) {
  // masked taps have weight 0 and a clamped (valid) address; select instead of multiply so that
  // a non-finite value at the clamped address cannot leak into the result
  const float v00 = t.w00 != 0.f ? plane[t.i00] : 0.f;
  const float v01 = t.w01 != 0.f ? plane[t.i01] : 0.f;
  const float v10 = t.w10 != 0.f ? plane[t.i10] : 0.f;
  const float v11 = t.w11 != 0.f ? plane[t.i11] : 0.f;
  return v00 * t.w00 + v01 * t.w01 + v10 * t.w10 + v11 * t.w11;
}

// same result from two 8-byte loads (dword aligned, allowed on gfx950) instead of four 4-byte gathers: half the
// texture-addresser work of the gather-bound warp.  Needs W >= 2.
__device__ __forceinline__ float combine_pairs(const f2u a, const f2u b, const Taps &t) {
  const float v00 = t.w00 != 0.f ? (t.sel0 ? a.y : a.x) : 0.f;
  const float v01 = t.w01 != 0.f ? (t.sel1 ? a.x : a.y) : 0.f;
  const float v10 = t.w10 != 0.f ? (t.sel0 ? b.y : b.x) : 0.f;
  const float v11 = t.w11 != 0.f ? (t.sel1 ? b.x : b.y) : 0.f;
  return v00 * t.w00 + v01 * t.w01 + v10 * t.w10 + v11 * t.w11;
}
__device__ __forceinline__ float sample_pairs(const float *plane, const Taps &t) {
  return combine_pairs(mfn_load2u(plane + t.p0), mfn_load2u(plane + t.p1), t);
}

// GridGenerator kWarp for one pixel: (flow + index) / ((size-1)/2) - 1
__device__ __forceinline__ void warp_grid(float fx, float fy, int x, int y, int H, int W, int clip, float &gx,
                                          float &gy) {
  const float nx = (float)((W - 1) / 2.0), ny = (float)((H - 1) / 2.0);
  gx = (fx + (float)x) / nx - 1.f;
  gy = (fy + (float)y) / ny - 1.f;
  if (clip) {
    gx = fminf(fmaxf(gx, -1.f), 1.f);
    gy = fminf(fmaxf(gy, -1.f), 1.f);
  }
}

template <int VEC>
__global__ __launch_bounds__(256) void warp_fwd_kernel(WarpParams p) {
  const int W = p.W, H = p.H;
  const int wv = W / VEC;  // vectors per row (VEC divides W)
  const size_t total = (size_t)p.N * H * wv;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int xv = (int)(idx % wv);
  const int y = (int)((idx / wv) % H);
  const int n = (int)(idx / ((size_t)wv * H));
  const int x0 = xv * VEC;
  const size_t plane = (size_t)H * W;
  const float *fl = p.flow + (size_t)n * 2 * plane + (size_t)y * W + x0;
  float fy[VEC], fx[VEC];
  if (VEC == 4) {
    const float4 a = *reinterpret_cast<const float4 *>(fl);
    const float4 b = *reinterpret_cast<const float4 *>(fl + plane);
    fy[0] = a.x; fy[1 % VEC] = a.y; fy[2 % VEC] = a.z; fy[3 % VEC] = a.w;
    fx[0] = b.x; fx[1 % VEC] = b.y; fx[2 % VEC] = b.z; fx[3 % VEC] = b.w;
  } else {
    MFN_UNROLL
    for (int k = 0; k < VEC; ++k) { fy[k] = fl[k]; fx[k] = fl[plane + k]; }
  }
  Taps t[VEC];
  MFN_UNROLL
  for (int k = 0; k < VEC; ++k) {
    float gx, gy;
    warp_grid(fx[k], fy[k], x0 + k, y, H, W, p.clip, gx, gy);
    t[k] = sampler_taps(gx, gy, H, W);
  }
  const float *xin = p.x + (size_t)n * p.C * plane;
  float *o = p.out + (size_t)n * p.C * plane + (size_t)y * W + x0;
  for (int c = 0; c < p.C; ++c) {
    const float *pl = xin + (size_t)c * plane;
    float r[VEC];
    MFN_UNROLL
    for (int k = 0; k < VEC; ++k) r[k] = (VEC == 1 && W >= 2) ? sample_pairs(pl, t[k]) : sample(pl, t[k]);
    if (VEC == 4) {
      *reinterpret_cast<float4 *>(o + (size_t)c * plane) = make_float4(r[0], r[1 % VEC], r[2 % VEC], r[3 % VEC]);
    } else {
      MFN_UNROLL
      for (int k = 0; k < VEC; ++k) o[(size_t)c * plane + k] = r[k];
    }
  }
}

// The kernel of the pass (W >= 2, N*H*W < 2^32).  One pixel per thread, a block = 256 consecutive pixels, and
//  * blocks XCD-remapped: the source rows a block gathers are also gathered by the blocks of the output rows above
//    and below; in dispatch order those sit on other XCDs and every L2 fetched its own copy over the fabric
//    (8x3x384x512: 15.2 us; 11.6 us with the remap; a plain 4-byte-per-lane copy of the same bytes takes 9.3);
//  * 32-bit index arithmetic;
//  * the channel loop in groups of G with every tap load of a group issued before the first use, so a pixel costs
//    two memory round trips (flow, taps) per group instead of one per channel.
template <int G, bool T2D>
__global__ __launch_bounds__(256) void warp_fwd_fast_kernel(WarpParams p, unsigned total) {
  const unsigned W = (unsigned)p.W, H = (unsigned)p.H;
  unsigned x, y, n;
  if (T2D) {
    // a wave is a 16 x 4 pixel tile, a block four of them side by side (W % 64 == 0, H % 4 == 0): full 64-byte lines of flow and
    // output per tile row, and a footprint of 4 + 2 r rows x (16 + 2 r) / 16 lines under a flow of radius r where 64 pixels
    // of one row touch 1 + 2 r rows x (64 + 2 r) / 16 lines
    const unsigned b = mfn_xcd_remap(blockIdx.x, gridDim.x), bw = W >> 6, bh = H >> 2;
    const unsigned bxi = b % bw, rest = b / bw, byi = rest % bh;
    n = rest / bh;
    x = bxi * 64u + (threadIdx.x >> 6) * 16u + (threadIdx.x & 15u);
    y = byi * 4u + ((threadIdx.x >> 4) & 3u);
  } else {
    const unsigned idx = mfn_xcd_remap(blockIdx.x, gridDim.x) * 256u + threadIdx.x;
    if (idx >= total) return;
    const unsigned row = idx / W;
    x = idx - row * W;
    n = row / H;
    y = row - n * H;
  }
  const size_t plane = (size_t)H * W;
  const unsigned pix = y * W + x;
  const float *fl = p.flow + (size_t)n * 2 * plane + pix;
  const float fy = fl[0], fx = fl[plane];
  float gx, gy;
  warp_grid(fx, fy, (int)x, (int)y, p.H, p.W, p.clip, gx, gy);
  const Taps t = sampler_taps(gx, gy, p.H, p.W);
  const float *xin = p.x + (size_t)n * p.C * plane;
  float *o = p.out + (size_t)n * p.C * plane + pix;
  int c = 0;
  for (; c + G <= p.C; c += G) {
    f2u a[G], b[G];
    MFN_UNROLL
    for (int k = 0; k < G; ++k) {
      const float *pl = xin + (size_t)(c + k) * plane;
      a[k] = mfn_load2u(pl + t.p0);
      b[k] = mfn_load2u(pl + t.p1);
    }
    MFN_UNROLL
    for (int k = 0; k < G; ++k) mfn_store1_stream(o + (size_t)(c + k) * plane, combine_pairs(a[k], b[k], t), p.st_policy);
  }
  for (; c < p.C; ++c) mfn_store1_stream(o + (size_t)c * plane, sample_pairs(xin + (size_t)c * plane, t), p.st_policy);
}

inline int warp_fwd_launch(WarpParams p, hipStream_t stream) {
  // one pixel per thread keeps the gathers of a wave on adjacent addresses (measured 28 us vs 46 us for 4 px / thread on
  // 8x3x384x512 with noisy flow; 2 / 4 strided pixels per thread were no faster either: both forms removed in round 3)
  const size_t total = (size_t)p.N * p.H * p.W;
  if (total == 0) return 0;
  const dim3 grid((unsigned)((total + 255) / 256));
  if (total < ((size_t)1 << 32) - 256 && p.W >= 2) {   // the fast kernel's 32-bit indices and 8-byte tap pairs apply
    const bool t2d = !MFN_WARP_1D && p.W % 64 == 0 && p.H % 4 == 0;
    const bool g3 = p.C % 4 != 0 && p.C % 3 == 0;
    if (t2d) return g3 ? launch("warp_fwd_fast", warp_fwd_fast_kernel<3, true>, grid, dim3(256), 0, stream, p, (unsigned)total)
                       : launch("warp_fwd_fast", warp_fwd_fast_kernel<4, true>, grid, dim3(256), 0, stream, p, (unsigned)total);
    return g3 ? launch("warp_fwd_fast", warp_fwd_fast_kernel<3, false>, grid, dim3(256), 0, stream, p, (unsigned)total)
              : launch("warp_fwd_fast", warp_fwd_fast_kernel<4, false>, grid, dim3(256), 0, stream, p, (unsigned)total);
  }
  return launch("warp_fwd_v1", warp_fwd_kernel<1>, grid, dim3(256), 0, stream, p);
}

// ---- the MXNet operators on their own ----------------------------------------------------------
struct GridWarpParams { const float *flow_xy; float *grid; int N, H, W; };
__global__ __launch_bounds__(256) void grid_warp_kernel(GridWarpParams p) {
  const size_t plane = (size_t)p.H * p.W;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)p.N * plane) return;
  const int x = (int)(idx % p.W), y = (int)((idx / p.W) % p.H);
  const size_t n = idx / plane, pix = idx - n * plane;
  float gx, gy;
  warp_grid(p.flow_xy[n * 2 * plane + pix], p.flow_xy[n * 2 * plane + plane + pix], x, y, p.H, p.W, 0, gx, gy);
  p.grid[n * 2 * plane + pix] = gx;
  p.grid[n * 2 * plane + plane + pix] = gy;
}

struct GridAffineParams { const float *theta; float *grid; int N, H, W; };
__global__ __launch_bounds__(256) void grid_affine_kernel(GridAffineParams p) {
  const size_t plane = (size_t)p.H * p.W;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)p.N * plane) return;
  const int x = (int)(idx % p.W), y = (int)((idx / p.W) % p.H);
  const size_t n = idx / plane, pix = idx - n * plane;
  const float *t = p.theta + n * 6;
  const float xn = -1.f + (float)x * (float)(2.0 / (p.W - 1));
  const float yn = -1.f + (float)y * (float)(2.0 / (p.H - 1));
  p.grid[n * 2 * plane + pix] = t[0] * xn + t[1] * yn + t[2];
  p.grid[n * 2 * plane + plane + pix] = t[3] * xn + t[4] * yn + t[5];
}

struct SamplerParams { const float *data; const float *grid; float *out; int N, C, iH, iW, oH, oW; };
template <bool T2D>
__global__ __launch_bounds__(256) void bilinear_sampler_kernel(SamplerParams p) {
  const size_t oplane = (size_t)p.oH * p.oW, iplane = (size_t)p.iH * p.iW;
  size_t n, pix;
  if (T2D) {  // a wave is a 16 x 4 tile of the output (oW % 64 == 0, oH % 4 == 0), as in warp_fwd_fast_kernel
    const unsigned bw = (unsigned)p.oW >> 6, bh = (unsigned)p.oH >> 2;
    const unsigned bxi = blockIdx.x % bw, rest = blockIdx.x / bw, byi = rest % bh;
    n = rest / bh;
    pix = (size_t)(byi * 4u + ((threadIdx.x >> 4) & 3u)) * p.oW + bxi * 64u + (threadIdx.x >> 6) * 16u + (threadIdx.x & 15u);
  } else {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)p.N * oplane) return;
    n = idx / oplane;
    pix = idx - n * oplane;
  }
  const float gx = p.grid[n * 2 * oplane + pix], gy = p.grid[n * 2 * oplane + oplane + pix];
  const Taps t = sampler_taps(gx, gy, p.iH, p.iW);
  for (int c = 0; c < p.C; ++c)
    p.out[(n * p.C + c) * oplane + pix] = sample(p.data + (n * p.C + c) * iplane, t);
}

inline int grid_warp_launch(GridWarpParams p, hipStream_t s) {
  const size_t total = (size_t)p.N * p.H * p.W;
  if (!total) return 0;
  return launch("grid_generator_warp", grid_warp_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p);
}
inline int grid_affine_launch(GridAffineParams p, hipStream_t s) {
  const size_t total = (size_t)p.N * p.H * p.W;
  if (!total) return 0;
  return launch("grid_generator_affine", grid_affine_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p);
}
inline int bilinear_sampler_launch(SamplerParams p, hipStream_t s) {
  const size_t total = (size_t)p.N * p.oH * p.oW;
  if (!total) return 0;
  const dim3 grid((unsigned)((total + 255) / 256));
  if (!MFN_WARP_1D && p.oW % 64 == 0 && p.oH % 4 == 0 && total < ((size_t)1 << 32))
    return launch("bilinear_sampler", bilinear_sampler_kernel<true>, grid, dim3(256), 0, s, p);
  return launch("bilinear_sampler", bilinear_sampler_kernel<false>, grid, dim3(256), 0, s, p);
}

}  // namespace mfn
