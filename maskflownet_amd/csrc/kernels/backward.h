// backward.h -- gradients of the hot-path operators for gfx950 (the training
// step of /root/reference/network/pipeline.py:112-113 reaches them through MXNet autograd).
// Semantics as oracle/mfn_ref_body.inc: correlation_bwd, bilinear_sampler_bwd + grid generator backward
// (warp_bwd), deform_conv_bwd (deformable_col2im / col2im_coord / weight + bias gradients).
//
// Round-1 scope: correct and parallel, not yet tuned (the forward path is the benchmarked one):
//   * correlation: both gradients are GATHERS for the reference configuration (kernel_size=1, strides 1,
//     pad == max_displacement) -- one thread per input element, no atomics, deterministic; any other
//     parameter set scatters with atomics exactly like MXNet's GPU kernels;
//   * warp: data gradient = 4-tap atomic scatter, flow gradient accumulated per pixel in registers;
//   * deformable conv: input/offset gradients in one kernel (column gradient formed on the fly from
//     W^T x gout, never written), weight gradient by a block-reduction kernel, bias by a row reduction.
// req semantics (MXNet OpReqType): 0 = skip, 1 = write, 3 = add to the existing contents.
#pragma once
#include "../mfn_rt.h"
#include "correlation.h"
#include "deform_conv.h"
#include "warp.h"

namespace mfn {

struct FillParams { float *p; size_t n; };
__global__ __launch_bounds__(256) void fill_zero_kernel(FillParams f) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < f.n) f.p[i] = 0.f;
}
inline int fill_zero_launch(float *p, size_t n, hipStream_t s) {
  if (!n) return 0;
  return launch("fill_zero", fill_zero_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, FillParams{p, n});
}

// up to four buffers in one launch (the write-mode gradients of a deformable-conv backward: gx, goffset, gw, gbias);
// a thread zeroes four consecutive floats of one buffer
struct Fill4Params { float *p[4]; size_t n[4]; unsigned qend[4]; };  // qend: running total of 4-float groups
__global__ __launch_bounds__(256) void fill_zero4_kernel(Fill4Params f) {
  const unsigned q = blockIdx.x * 256u + threadIdx.x;
  MFN_UNROLL
  for (int k = 0; k < 4; ++k) {
    const unsigned q0 = k ? f.qend[k - 1] : 0u;
    if (q >= q0 && q < f.qend[k]) {
      const size_t b = (size_t)(q - q0) * 4;
      if (b + 3 < f.n[k] && (reinterpret_cast<size_t>(f.p[k]) & 15) == 0) {  // one 16-byte store
        *reinterpret_cast<float4 *>(f.p[k] + b) = make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        MFN_UNROLL
        for (int e = 0; e < 4; ++e)
          if (b + e < f.n[k]) f.p[k][b + e] = 0.f;
      }
    }
  }
}
inline int fill_zero4_launch(float *const ptr[4], const size_t cnt[4], hipStream_t s) {
  Fill4Params f;
  size_t tot = 0;
  for (int k = 0; k < 4; ++k) {
    f.p[k] = ptr[k];
    f.n[k] = ptr[k] ? cnt[k] : 0;
    tot += (f.n[k] + 3) / 4;
    f.qend[k] = (unsigned)tot;
  }
  if (!tot) return 0;
  if (tot >= 0xffffff00ull) {  // beyond 32-bit group indices: one launch per buffer
    for (int k = 0; k < 4; ++k)
      if (int rc = fill_zero_launch(f.p[k], f.n[k], s)) return rc;
    return 0;
  }
  return launch("fill_zero4", fill_zero4_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, f);
}

// ---- elementwise helpers of the f-row backward passes (round 3) -------------------------------------------
// LeakyReLU(0.1) backward from the forward OUTPUT y (sign(y) = sign of the pre-activation): gin = gout * (y > 0 ? 1 : 0.1),
// as MXNet's LeakyReLU (xelu_grad: x > 0 ? 1 : slope)
struct LeakyBwdParams { const float *gout, *y; float *gin; size_t n; float slope; };
__global__ __launch_bounds__(256) void leaky_bwd_kernel(LeakyBwdParams p) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < p.n) p.gin[i] = p.y[i] > 0.f ? p.gout[i] : p.gout[i] * p.slope;
}
inline int leaky_bwd_launch(LeakyBwdParams p, hipStream_t s) {
  if (!p.n) return 0;
  return launch("leaky_bwd", leaky_bwd_kernel, dim3((unsigned)((p.n + 255) / 256)), dim3(256), 0, s, p);
}
struct AccumParams { float *dst; const float *src; size_t n; };
__global__ __launch_bounds__(256) void accumulate_kernel(AccumParams p) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < p.n) p.dst[i] += p.src[i];
}
inline int accumulate_launch(float *dst, const float *src, size_t n, hipStream_t s) {
  if (!n) return 0;
  return launch("accumulate", accumulate_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, AccumParams{dst, src, n});
}
// data gradient of a stride-1 convolution = convolution of gout with wf[c][o][i][j] = w[o][c][kh-1-i][kw-1-j]
struct FlipParams { const float *w; float *wf; int Cout, Cin, kh, kw; };
__global__ __launch_bounds__(256) void conv_flip_weights_kernel(FlipParams p) {
  const size_t total = (size_t)p.Cout * p.Cin * p.kh * p.kw;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int j = (int)(idx % p.kw), i = (int)((idx / p.kw) % p.kh);
  const int o = (int)((idx / ((size_t)p.kw * p.kh)) % p.Cout), c = (int)(idx / ((size_t)p.kw * p.kh * p.Cout));
  p.wf[idx] = p.w[(((size_t)o * p.Cin + c) * p.kh + (p.kh - 1 - i)) * p.kw + (p.kw - 1 - j)];
}
inline int conv_flip_weights_launch(FlipParams p, hipStream_t s) {
  const size_t total = (size_t)p.Cout * p.Cin * p.kh * p.kw;
  if (!total) return 0;
  return launch("conv_flip_weights", conv_flip_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p);
}
// Data gradient of the network's 4x4 / stride 2 / pad 1 Deconvolution = a 4x4 / stride 2 / pad 1 CONVOLUTION of the output gradient;
// on the pixel-unshuffled gradient t[(py,px,co)][Y][X] = g[co][2Y+py][2X+px] that is a 3x3 / stride 1 / pad 1 convolution (the MFMA
// kernels' shape): kernel row ky reads input row 2y-1+ky = 2(y+r-1)+p with (p, r) = (1,0) (0,1) (1,1) (0,2) for ky = 0..3 -- four of
// the nine taps of a parity are used, the others are zero weights.  The generic kernel took 1.6 ms per such layer.
struct S2dParams { const float *g; float *t; int N, C, H, W; };   // g: (N, C, 2H, 2W) -> t: (N, 4C, H, W)
__global__ __launch_bounds__(256) void conv_s2d_kernel(S2dParams p) {
  const size_t total = (size_t)p.N * 4 * p.C * p.H * p.W;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int X = (int)(idx % p.W), Y = (int)((idx / p.W) % p.H);
  const size_t rest = idx / ((size_t)p.W * p.H);
  const int cc = (int)(rest % (4 * p.C)), n = (int)(rest / (4 * p.C));
  const int par = cc / p.C, co = cc - par * p.C, py = par >> 1, px = par & 1;
  p.t[idx] = p.g[(((size_t)n * p.C + co) * (2 * p.H) + 2 * Y + py) * (2 * p.W) + 2 * X + px];
}
struct S2dWeightParams { const float *w; float *w3; int Cin, Cout; };   // w: (Cin, Cout, 4, 4) -> w3: (Cin, 4 Cout, 3, 3)
__global__ __launch_bounds__(256) void conv_s2d_weights_kernel(S2dWeightParams p) {
  const size_t total = (size_t)p.Cin * 4 * p.Cout * 9;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int sx = (int)(idx % 3), r = (int)((idx / 3) % 3);
  const int cc = (int)((idx / 9) % (4 * p.Cout)), ci = (int)(idx / ((size_t)9 * 4 * p.Cout));
  const int par = cc / p.Cout, co = cc - par * p.Cout, py = par >> 1, px = par & 1;
  auto tap = [](int par1, int r1) { return par1 ? (r1 == 0 ? 0 : (r1 == 1 ? 2 : -1)) : (r1 == 1 ? 1 : (r1 == 2 ? 3 : -1)); };
  const int ky = tap(py, r), kx = tap(px, sx);
  p.w3[idx] = (ky >= 0 && kx >= 0) ? p.w[(((size_t)ci * p.Cout + co) * 4 + ky) * 4 + kx] : 0.f;
}
// per-channel sum over (n, pixel) of a (N, C, plane) tensor (bias gradient)
struct ChanSumParams { const float *g; float *out; int N, C; size_t plane; int add; };
__global__ __launch_bounds__(256) void channel_sum_kernel(ChanSumParams p) {
  MFN_DYN_SHARED(float, red);
  const int o = blockIdx.x;
  float s = 0.f;
  for (size_t q = threadIdx.x; q < (size_t)p.N * p.plane; q += 256) {
    const size_t n = q / p.plane, pix = q - n * p.plane;
    s += p.g[(n * p.C + o) * p.plane + pix];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int st = 128; st >= 1; st >>= 1) {
    if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) p.out[o] = (p.add ? p.out[o] : 0.f) + red[0];
}
// The same sum with the (n, pixel) range of a channel cut into `S` slices over blockIdx.y (a block per channel leaves most CUs idle:
// 107 us for the 50 MB of a level-2 layer): partial[c][s] here, the slices added in index order by channel_sum_final_kernel --
// deterministic, no atomics.  plane % 4 == 0 and 16-byte aligned `g`: float4 loads (a quad never straddles two images).
struct ChanSumPartParams { const float *g; float *partial; int N, C, S; size_t plane, chunk; int vec; };
__global__ __launch_bounds__(256) void channel_sum_partial_kernel(ChanSumPartParams p) {
  MFN_DYN_SHARED(float, red);
  const int o = blockIdx.x, sl = blockIdx.y;
  const size_t total = (size_t)p.N * p.plane;
  const size_t q0 = (size_t)sl * p.chunk, q1 = q0 + p.chunk < total ? q0 + p.chunk : total;
  float s = 0.f;
  if (p.vec) {
    for (size_t q = q0 + 4 * (size_t)threadIdx.x; q < q1; q += 1024) {
      const size_t n = q / p.plane, pix = q - n * p.plane;
      const float4 v = *reinterpret_cast<const float4 *>(p.g + (n * p.C + o) * p.plane + pix);
      s += (v.x + v.y) + (v.z + v.w);
    }
  } else {
    for (size_t q = q0 + threadIdx.x; q < q1; q += 256) {
      const size_t n = q / p.plane, pix = q - n * p.plane;
      s += p.g[(n * p.C + o) * p.plane + pix];
    }
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int st = 128; st >= 1; st >>= 1) {
    if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) p.partial[(size_t)o * p.S + sl] = red[0];
}
struct ChanSumFinalParams { const float *partial; float *out; int C, S, add; };
__global__ __launch_bounds__(256) void channel_sum_final_kernel(ChanSumFinalParams p) {
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= p.C) return;
  float s = 0.f;
  for (int k = 0; k < p.S; ++k) s += p.partial[(size_t)o * p.S + k];
  p.out[o] = (p.add ? p.out[o] : 0.f) + s;
}
// Upsample(factor) backward (MaskFlownet.py:35-62 is linear: its adjoint).  One thread per INPUT pixel gathers the at most
// (2f-1)^2 outputs it fed, with the forward's weights (upsample.h): row i receives weight ka0(r) from output rows i*f + r and
// ka1(r) from rows whose lower neighbour min(iy0 + 1, H - 1) is i (the edge pad makes the last row its own neighbour).
struct UpsampleBwdParams { const float *gout; float *gx; int N, C, H, W, f, add; };
__global__ __launch_bounds__(256) void upsample_bwd_kernel(UpsampleBwdParams p) {
  const size_t total = (size_t)p.N * p.C * p.H * p.W;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int f = p.f, cc = f - 1;
  const int ix = (int)(idx % p.W), iy = (int)((idx / p.W) % p.H);
  const size_t nc = idx / ((size_t)p.W * p.H);
  const int Hout = p.H * f, Wout = p.W * f;
  const float *g = p.gout + nc * (size_t)Hout * Wout;
  auto tri = [&](int a) { return 1.f - fabsf((float)(cc - a)) / (float)(cc + 1); };
  // weight of input line `i` (of `n` lines) in output line `o`
  auto wline = [&](int o, int i, int n) {
    const int i0 = o / f, r = o - i0 * f;
    float w = 0.f;
    if (i0 == i) w += tri(r + f - 1);
    if (r && min(i0 + 1, n - 1) == i) w += tri(r - 1);
    return w;
  };
  float s = 0.f;
  const int oy_lo = max((iy - 1) * f + 1, 0), oy_hi = min((iy + 1) * f - 1, Hout - 1);
  const int ox_lo = max((ix - 1) * f + 1, 0), ox_hi = min((ix + 1) * f - 1, Wout - 1);
  for (int oy = oy_lo; oy <= oy_hi; ++oy) {
    const float wy = wline(oy, iy, p.H);
    if (wy == 0.f) continue;
    for (int ox = ox_lo; ox <= ox_hi; ++ox) s += g[(size_t)oy * Wout + ox] * (wy * wline(ox, ix, p.W));
  }
  p.gx[idx] = p.add ? p.gx[idx] + s : s;
}
// Large factors (the multiscale loss upsamples the coarse predictions by 8 .. 64: an input pixel feeds up to 127 x 127 outputs):
// one BLOCK per input pixel, its threads stride over the footprint, partial sums meet in LDS in a fixed tree -- the
// thread-per-pixel form above runs 768 threads of 16 k serial terms each at factor 64 (345 us per call in the training step).
__global__ __launch_bounds__(256) void upsample_bwd_block_kernel(UpsampleBwdParams p) {
  MFN_DYN_SHARED(float, red);
  const size_t idx = blockIdx.x;
  const int f = p.f, cc = f - 1;
  const int ix = (int)(idx % p.W), iy = (int)((idx / p.W) % p.H);
  const size_t nc = idx / ((size_t)p.W * p.H);
  const int Hout = p.H * f, Wout = p.W * f;
  const float *g = p.gout + nc * (size_t)Hout * Wout;
  auto tri = [&](int a) { return 1.f - fabsf((float)(cc - a)) / (float)(cc + 1); };
  auto wline = [&](int o, int i, int n) {
    const int i0 = o / f, r = o - i0 * f;
    float w = 0.f;
    if (i0 == i) w += tri(r + f - 1);
    if (r && min(i0 + 1, n - 1) == i) w += tri(r - 1);
    return w;
  };
  const int oy_lo = max((iy - 1) * f + 1, 0), oy_hi = min((iy + 1) * f - 1, Hout - 1);
  const int ox_lo = max((ix - 1) * f + 1, 0), ox_hi = min((ix + 1) * f - 1, Wout - 1);
  const int nx = ox_hi - ox_lo + 1, ny = oy_hi - oy_lo + 1;
  float s = 0.f;
  for (int e = threadIdx.x; e < nx * ny; e += 256) {
    const int ry = e / nx, oy = oy_lo + ry, ox = ox_lo + (e - ry * nx);
    s += g[(size_t)oy * Wout + ox] * (wline(oy, iy, p.H) * wline(ox, ix, p.W));
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int st = 128; st >= 1; st >>= 1) {
    if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) p.gx[idx] = p.add ? p.gx[idx] + red[0] : red[0];
}

// ---- correlation -----------------------------------------------------------------------------------
struct CorrBwdParams {
  const float *gout, *f1, *f2;
  float *g1, *g2;
  int N, C, H, W, md, D;  // D = 2*md+1 (kernel 1, strides 1, pad == md: top_h = H, top_w = W)
  int req1, req2;
  int st_policy;  // cache policy of the gradient stores (mfn_store4_stream; block kernel)
};
// g1[n,c,y,x] = 1/C sum_d gout[n,d,y,x]       * f2[n,c,y+dy,x+dx]
// g2[n,c,y,x] = 1/C sum_d gout[n,d,y-dy,x-dx] * f1[n,c,y-dy,x-dx]      (terms outside the image vanish)
__global__ __launch_bounds__(256) void corr_bwd_gather_kernel(CorrBwdParams p) {
  const size_t plane = (size_t)p.H * p.W;
  const size_t total = (size_t)p.N * p.C * plane;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int x = (int)(idx % p.W), y = (int)((idx / p.W) % p.H);
  const int c = (int)((idx / plane) % p.C), n = (int)(idx / (plane * p.C));
  const float *go = p.gout + (size_t)n * p.D * p.D * plane;
  const float *a = p.f1 + ((size_t)n * p.C + c) * plane;
  const float *b = p.f2 + ((size_t)n * p.C + c) * plane;
  const float sumelems = (float)p.C;
  float s1 = 0.f, s2 = 0.f;
  for (int iy = 0; iy < p.D; ++iy) {
    const int dy = iy - p.md;
    const int y2 = y + dy, ys = y - dy;
    for (int ix = 0; ix < p.D; ++ix) {
      const int dx = ix - p.md;
      const int x2 = x + dx, xs = x - dx;
      const float *gd = go + (size_t)(iy * p.D + ix) * plane;
      if (y2 >= 0 && y2 < p.H && x2 >= 0 && x2 < p.W) s1 += gd[(size_t)y * p.W + x] * b[(size_t)y2 * p.W + x2] / sumelems;
      if (ys >= 0 && ys < p.H && xs >= 0 && xs < p.W) s2 += gd[(size_t)ys * p.W + xs] * a[(size_t)ys * p.W + xs] / sumelems;
    }
  }
  if (p.req1) p.g1[idx] = (p.req1 == 3 ? p.g1[idx] : 0.f) + s1;
  if (p.req2) p.g2[idx] = (p.req2 == 3 ? p.g2[idx] : 0.f) + s2;
}

// Register-blocked form for W % 4 == 0: a thread owns 4 adjacent pixels x 4 channels.  Per displacement row it
// loads the nine gout quads once for all four channels and, per channel, one 12-wide window of the other feature
// map (three aligned 16-byte loads, whole quads outside the image read as zero) that serves all nine dx --
// 0.15 sixteen-byte loads per FMA instead of 2 scalar loads, and one multiply by 1/C at the end instead of a
// division per term.  For g2 the roles swap: the window slides over gout's displaced row.
// PART 0: both gradients (gridDim.y == 2 hands them to different blocks), 1: g1 only, 2: g2 only.
// __launch_bounds__(256, 3): hipcc left to itself takes 210 VGPRs (two waves per SIMD) for load hoisting that does not pay;
// capped at three waves (154 VGPRs, no spills) every level is 15-20 % faster (level 2: 66 -> 56 us); at four it spills.
template <int PART>
__global__ __launch_bounds__(256, 3) void corr_bwd_block_kernel(CorrBwdParams p) {
  constexpr int CB = 4;
  const int W = p.W, H = p.H, C = p.C, D = p.D, md = p.md;
  const int QW = W >> 2, CG = (C + CB - 1) / CB;
  const size_t plane = (size_t)H * W;
  const size_t total = (size_t)p.N * CG * H * QW;
  // blocks of neighbouring rows read the same 9 rows of gout / f2: one XCD per contiguous range (mfn_xcd_remap)
  const size_t idx = (size_t)mfn_xcd_remap(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
  if (idx >= total) return;
  const int qx = (int)(idx % QW), y = (int)((idx / QW) % H);
  const int cg = (int)((idx / ((size_t)QW * H)) % CG), n = (int)(idx / ((size_t)QW * H * CG));
  const int x = 4 * qx, c0 = cg * CB;
  const float *go = p.gout + (size_t)n * D * D * plane;
  const float *f1n = p.f1 + ((size_t)n * C) * plane, *f2n = p.f2 + ((size_t)n * C) * plane;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  // 12-wide window [x-4, x+8) of one row; quads outside [0, W) are zero (W % 4 == 0: a quad is in or out as a whole)
  // every load is unconditional (clamped address) and zeroed by a select afterwards: a branch per load would make
  // hipcc wait for each one before issuing the next (cdna_hip_programming.md section 5, trap c)
  const int xl = max(x - 4, 0), xr = min(x + 4, W - 4);
  const bool okl = x >= 4, okr = x + 4 < W;
  auto window = [&](const float *row, bool row_ok, float (&v)[12]) {
    const float4 a = zero_unless(row_ok && okl, *reinterpret_cast<const float4 *>(row + xl));
    const float4 b = zero_unless(row_ok, *reinterpret_cast<const float4 *>(row + x));
    const float4 c = zero_unless(row_ok && okr, *reinterpret_cast<const float4 *>(row + xr));
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    v[8] = c.x; v[9] = c.y; v[10] = c.z; v[11] = c.w;
  };
  float s1[CB][4], s2[CB][4];
  MFN_UNROLL
  for (int k = 0; k < CB; ++k)
    MFN_UNROLL
    for (int q = 0; q < 4; ++q) { s1[k][q] = 0.f; s2[k][q] = 0.f; }

  // gridDim.y == 2: the two gradients go to different blocks (two independent chains of D dependent load rounds; at the
  // coarse levels the launch is nothing but that latency chain: 27 us for 0.3 MB)
  const bool do1 = PART != 2 && p.req1 && (gridDim.y == 1 || blockIdx.y == 0);
  const bool do2 = PART != 1 && p.req2 && (gridDim.y == 1 || blockIdx.y == 1);
  if (do1) {  // g1[c,y,x+q] = sum_d gout[d,y,x+q] * f2[c,y+dy,x+q+dx]
    for (int iy = 0; iy < D; ++iy) {
      const int y2u = y + iy - md;
      const bool rok = y2u >= 0 && y2u < H;
      const int y2 = min(max(y2u, 0), H - 1);
      float4 g[9];
      MFN_UNROLL
      for (int ix = 0; ix < 9; ++ix)
        g[ix] = zero_unless(ix < D, *reinterpret_cast<const float4 *>(go + (size_t)(iy * D + min(ix, D - 1)) * plane +
                                                                      (size_t)y * W + x));
      MFN_UNROLL
      for (int k = 0; k < CB; ++k) {
        float bv[12];
        window(f2n + (size_t)min(c0 + k, C - 1) * plane + (size_t)y2 * W, rok && c0 + k < C, bv);
        // window index of pixel q displaced by dx = ix - md: q + dx + 4.  D == 9 and D == 5 (md = 4, 2) are the two
        // compile-time shapes; entries ix >= D carry zero gout
        MFN_UNROLL
        for (int ix = 0; ix < 9; ++ix) {
          const int o9 = ix, o5 = min(ix + 2, 8);
          const float b0 = D == 9 ? bv[o9 + 0] : bv[o5 + 0], b1 = D == 9 ? bv[o9 + 1] : bv[o5 + 1];
          const float b2 = D == 9 ? bv[o9 + 2] : bv[o5 + 2], b3 = D == 9 ? bv[o9 + 3] : bv[o5 + 3];
          s1[k][0] = fmaf(g[ix].x, b0, s1[k][0]);
          s1[k][1] = fmaf(g[ix].y, b1, s1[k][1]);
          s1[k][2] = fmaf(g[ix].z, b2, s1[k][2]);
          s1[k][3] = fmaf(g[ix].w, b3, s1[k][3]);
        }
      }
    }
  }
  if (do2) {  // g2[c,y,x+q] = sum_d gout[d,y-dy,x+q-dx] * f1[c,y-dy,x+q-dx]
    for (int iy = 0; iy < D; ++iy) {
      const int ysu = y - (iy - md);
      const bool rok = ysu >= 0 && ysu < H;
      const int ys = min(max(ysu, 0), H - 1);
      float av[CB][12];
      MFN_UNROLL
      for (int k = 0; k < CB; ++k)
        window(f1n + (size_t)min(c0 + k, C - 1) * plane + (size_t)ys * W, rok && c0 + k < C, av[k]);
      MFN_UNROLL
      for (int ix = 0; ix < 9; ++ix) {
        float gv[12];
        window(go + (size_t)(iy * D + min(ix, D - 1)) * plane + (size_t)ys * W, rok && ix < D, gv);
        // window index of source pixel q - dx = q + 4 - (ix - md): 8 - ix for md = 4, 6 - ix for md = 2
        const int o9 = 8 - ix, o5 = max(6 - ix, 0);
        MFN_UNROLL
        for (int k = 0; k < CB; ++k)
          MFN_UNROLL
          for (int q = 0; q < 4; ++q) {
            const float gq = D == 9 ? gv[o9 + q] : gv[o5 + q], aq = D == 9 ? av[k][o9 + q] : av[k][o5 + q];
            s2[k][q] = fmaf(gq, aq, s2[k][q]);
          }
      }
    }
  }
  const float inv = 1.0f / (float)C;
  MFN_UNROLL
  for (int k = 0; k < CB; ++k) {
    if (c0 + k >= C) break;
    const size_t o = ((size_t)n * C + c0 + k) * plane + (size_t)y * W + x;
    if (do1) {
      float4 r = make_float4(s1[k][0] * inv, s1[k][1] * inv, s1[k][2] * inv, s1[k][3] * inv);
      if (p.req1 == 3) { const float4 old = *reinterpret_cast<const float4 *>(p.g1 + o); r.x += old.x; r.y += old.y; r.z += old.z; r.w += old.w; }
      mfn_store4_stream(p.g1 + o, r.x, r.y, r.z, r.w, p.st_policy);
    }
    if (do2) {
      float4 r = make_float4(s2[k][0] * inv, s2[k][1] * inv, s2[k][2] * inv, s2[k][3] * inv);
      if (p.req2 == 3) { const float4 old = *reinterpret_cast<const float4 *>(p.g2 + o); r.x += old.x; r.y += old.y; r.z += old.z; r.w += old.w; }
      mfn_store4_stream(p.g2 + o, r.x, r.y, r.z, r.w, p.st_policy);
    }
  }
}

// LDS-staged form for W = 8, 16, ... 256 (every level of the network).  The block kernel above is bound by the L1: 21 (g1) / 39 (g2) sixteen-
// byte loads per lane and displacement row against 144 FMAs -- 0.6 GB through the vector caches for one gradient at level 2,
// 46 B/clk/CU of the 64 the L1 delivers; halving its instruction count (no selects) changed nothing.  Here a block is
// 256 / (W/4) image rows x 4 channels and first copies the rows of the OTHER feature map it will touch -- its own rows +- md,
// with four zero columns on either side and zero rows outside the image -- into LDS once: every feature value leaves the L1
// once instead of ~27 times (3 overlapping windows x 9 displacement rows), the 12-wide windows come out of LDS (three
// ds_read_b128, no selects: the border is in the copy), and the vector caches are left with the nine gout quads per
// displacement row.  g2 reads those as UNALIGNED quads at x - dx (one load per displacement instead of a 12-wide window of
// which a third is used); what an edge lane's quad takes from the neighbouring row is zeroed by a select (lane masks okl /
// okr), rows outside the image are skipped by per-lane loop bounds.  Same terms in the same order per output as the block
// kernel.  blockIdx.y = which of the requested gradients.  At the coarse levels (a handful of waves, each a chain of D
// dependent rounds) the gain is the shorter round: no selects, gout of the next round already on its way.
// Both gradients, cfg2 levels 6..2: 8.1 / 8.8 / 11.5 / 18.3 / 33.8 us against 14.5 / 15.1 / 20.5 / 29.9 / 54.6.
struct CorrBwdLdsParams {
  CorrBwdParams b;
  int rows_per_block, row_blocks;  // 256 / (W/4); cdiv(H, rows_per_block)
};
inline size_t corr_bwd_lds_bytes(int H, int W, int md) {
  const int rb = 256 / (W / 4);
  return (size_t)4 * ((rb < H ? rb : H) + 2 * md) * (W + 8) * sizeof(float);
}
template <int MD>
__global__ __launch_bounds__(256, 3) void corr_bwd_lds_kernel(CorrBwdLdsParams pp) {
  const CorrBwdParams &p = pp.b;
  constexpr int CB = 4, D = 2 * MD + 1;
  MFN_DYN_SHARED(float, lds);
  const int W = p.W, H = p.H, C = p.C;
  // (an image lower than a block's rows: only its rows are copied)
  const int QW = W >> 2, CG = (C + CB - 1) / CB, RB = pp.rows_per_block, RS = min(RB, H) + 2 * MD, PW = W + 8, PQ = QW + 2;
  const size_t plane = (size_t)H * W;
  // block -> (image, channel group, row block); neighbouring row blocks read overlapping rows: one XCD per contiguous range
  const unsigned bx = mfn_xcd_remap(blockIdx.x, gridDim.x);
  const int rb = (int)(bx % (unsigned)pp.row_blocks), cg = (int)((bx / (unsigned)pp.row_blocks) % (unsigned)CG);
  const int n = (int)(bx / ((unsigned)pp.row_blocks * (unsigned)CG));
  const int y0 = rb * RB, c0 = cg * CB;
  // which gradient this block forms
  const bool second = p.req1 && p.req2 ? blockIdx.y == 1 : p.req2 != 0;
  const float *src = (second ? p.f1 : p.f2) + (size_t)n * C * plane;
  // ---- the copy: [4 channels][RS rows y0 - MD ...][PW columns -4 ... W + 3], zero outside the image
  for (int e = threadIdx.x; e < CB * RS * PQ; e += 256) {
    const int qc = e % PQ, rr = (e / PQ) % RS, k = e / (PQ * RS);
    const int yy = y0 - MD + rr, xx = 4 * (qc - 1);
    const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
    const float4 v = zero_unless(ok, *reinterpret_cast<const float4 *>(src + (size_t)min(c0 + k, C - 1) * plane +
                                                                       (size_t)min(max(yy, 0), H - 1) * W + min(max(xx, 0), W - 4)));
    *reinterpret_cast<float4 *>(lds + ((size_t)k * RS + rr) * PW + 4 * qc) = v;
  }
  MFN_LDS_BARRIER();
  const int r = (int)threadIdx.x / QW, qx = (int)threadIdx.x - r * QW;
  const int y = y0 + r, x = 4 * qx;
  if (y >= H) return;
  const float *go = p.gout + (size_t)n * D * D * plane;
  // 12-wide window [x - 4, x + 8) of copied row rr (columns are shifted by 4 in the copy)
  auto window = [&](int k, int rr, float (&v)[12]) {
    const float *row = lds + ((size_t)k * RS + rr) * PW + x;
    const float4 a = *reinterpret_cast<const float4 *>(row), b = *reinterpret_cast<const float4 *>(row + 4),
                 c = *reinterpret_cast<const float4 *>(row + 8);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    v[8] = c.x; v[9] = c.y; v[10] = c.z; v[11] = c.w;
  };
  float acc[CB][4];
  MFN_UNROLL
  for (int k = 0; k < CB; ++k)
    MFN_UNROLL
    for (int q = 0; q < 4; ++q) acc[k][q] = 0.f;
  if (!second) {  // g1[c,y,x+q] = sum_d gout[d,y,x+q] * f2[c,y+dy,x+q+dx]; copied row of y + dy: r + iy
    // the D gout quads of displacement row iy + 1 are requested before row iy is multiplied (left alone hipcc turns the loops
    // inside out and waits for one quad at a time: nine round trips per row)
    const float *gp = go + (size_t)y * W + x;
    float4 g[D], gn[D];
    MFN_UNROLL
    for (int ix = 0; ix < D; ++ix) gn[ix] = *reinterpret_cast<const float4 *>(gp + (size_t)ix * plane);
    MFN_UNROLL
    for (int iy = 0; iy < D; ++iy) {
      MFN_UNROLL
      for (int ix = 0; ix < D; ++ix) g[ix] = gn[ix];
      if (iy + 1 < D) {
        MFN_UNROLL
        for (int ix = 0; ix < D; ++ix) gn[ix] = *reinterpret_cast<const float4 *>(gp + (size_t)((iy + 1) * D + ix) * plane);
      }
      MFN_COMPILER_FENCE();
      MFN_UNROLL
      for (int k = 0; k < CB; ++k) {
        float bv[12];
        window(k, r + iy, bv);
        MFN_UNROLL
        for (int ix = 0; ix < D; ++ix) {  // window index of pixel q displaced by dx = ix - MD: q + dx + 4
          constexpr int base = 4 - MD;
          acc[k][0] = fmaf(g[ix].x, bv[base + ix + 0], acc[k][0]);
          acc[k][1] = fmaf(g[ix].y, bv[base + ix + 1], acc[k][1]);
          acc[k][2] = fmaf(g[ix].z, bv[base + ix + 2], acc[k][2]);
          acc[k][3] = fmaf(g[ix].w, bv[base + ix + 3], acc[k][3]);
        }
      }
    }
  } else {  // g2[c,y,x+q] = sum_d gout[d,y-dy,x+q-dx] * f1[c,y-dy,x+q-dx]; copied row of y - dy: r + 2 MD - iy
    const bool okl = x >= 4, okr = x + 8 <= W;
    const int lo = max(0, y + MD - H + 1), hi = min(D, y + MD + 1);  // source rows inside the image
    for (int iy = lo; iy < hi; ++iy) {
      const int ys = y - (iy - MD);
      float av[CB][12];
      MFN_UNROLL
      for (int k = 0; k < CB; ++k) window(k, r + 2 * MD - iy, av[k]);
      MFN_UNROLL
      for (int ix = 0; ix < D; ++ix) {
        // the four source pixels x + q - dx of this displacement: one unaligned quad (it stays inside the tensor: a displacement
        // to the left never belongs to the first plane, one to the right never to the last); window index j = q + 4 - dx
        const f4u gl = mfn_load4u(go + (size_t)(iy * D + ix) * plane + (size_t)ys * W + x - (ix - MD));
        constexpr int base = 4 + MD;
        float gq[4] = {gl.x, gl.y, gl.z, gl.w};
        MFN_UNROLL
        for (int q = 0; q < 4; ++q) {
          const int j = base - ix + q;
          if (j < 4) gq[q] = okl ? gq[q] : 0.f;
          if (j >= 8) gq[q] = okr ? gq[q] : 0.f;
        }
        MFN_UNROLL
        for (int k = 0; k < CB; ++k)
          MFN_UNROLL
          for (int q = 0; q < 4; ++q) acc[k][q] = fmaf(gq[q], av[k][base - ix + q], acc[k][q]);
      }
    }
  }
  const float inv = 1.0f / (float)C;
  float *gdst = second ? p.g2 : p.g1;
  const int req = second ? p.req2 : p.req1;
  MFN_UNROLL
  for (int k = 0; k < CB; ++k) {
    if (c0 + k >= C) break;
    const size_t o = ((size_t)n * C + c0 + k) * plane + (size_t)y * W + x;
    float4 rv = make_float4(acc[k][0] * inv, acc[k][1] * inv, acc[k][2] * inv, acc[k][3] * inv);
    if (req == 3) { const float4 old = *reinterpret_cast<const float4 *>(gdst + o); rv.x += old.x; rv.y += old.y; rv.z += old.z; rv.w += old.w; }
    mfn_store4_stream(gdst + o, rv.x, rv.y, rv.z, rv.w, p.st_policy);
  }
}

struct CorrBwdGenericParams {
  const float *gout, *f1, *f2;
  float *g1, *g2;
  int N, C, H, W, md, kernel, s1, s2, pad, is_multiply, top_c, top_h, top_w, radius, gw;
  int req1, req2;
};
// any MXNet-valid parameter set: one thread per (n, top_channel, i, j), atomic scatter over channels
__global__ __launch_bounds__(256) void corr_bwd_scatter_kernel(CorrBwdGenericParams p) {
  const size_t total = (size_t)p.N * p.top_c * p.top_h * p.top_w;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int j = (int)(idx % p.top_w), i = (int)((idx / p.top_w) % p.top_h);
  const int tc = (int)((idx / ((size_t)p.top_w * p.top_h)) % p.top_c);
  const int n = (int)(idx / ((size_t)p.top_w * p.top_h * p.top_c));
  const int x1 = j * p.s1 + p.md, y1 = i * p.s1 + p.md;
  const int x2 = x1 + (tc % p.gw - p.radius) * p.s2, y2 = y1 + (tc / p.gw - p.radius) * p.s2;
  const size_t plane = (size_t)p.H * p.W;
  const float g = p.gout[idx];
  const float sumelems = (float)(p.kernel * p.kernel * p.C);
  for (int h = 0; h < p.kernel; ++h)
    for (int w = 0; w < p.kernel; ++w) {
      const int ya = y1 + h - p.pad, xa = x1 + w - p.pad, yb = y2 + h - p.pad, xb = x2 + w - p.pad;
      const bool ina = ya >= 0 && ya < p.H && xa >= 0 && xa < p.W;
      const bool inb = yb >= 0 && yb < p.H && xb >= 0 && xb < p.W;
      for (int c = 0; c < p.C; ++c) {
        const size_t base = ((size_t)n * p.C + c) * plane;
        const float va = ina ? p.f1[base + (size_t)ya * p.W + xa] : 0.f;
        const float vb = inb ? p.f2[base + (size_t)yb * p.W + xb] : 0.f;
        float c1, c2;
        if (p.is_multiply) { c1 = g * vb / sumelems; c2 = g * va / sumelems; }
        else { const float sg = (va - vb) >= 0.f ? 1.f : -1.f; c1 = g * sg / sumelems; c2 = -c1; }
        if (ina && p.req1) atomicAdd(p.g1 + base + (size_t)ya * p.W + xa, c1);
        if (inb && p.req2) atomicAdd(p.g2 + base + (size_t)yb * p.W + xb, c2);
      }
    }
}

// ---- warp ---------------------------------------------------------------------------------------------
struct WarpBwdParams {
  const float *gout, *x, *flow;
  float *gx, *gflow;
  int N, C, H, W, clip, req_x, req_flow;
};
__global__ __launch_bounds__(256) void warp_bwd_kernel(WarpBwdParams p) {
  const size_t plane = (size_t)p.H * p.W;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)p.N * plane) return;
  const int x = (int)(idx % p.W), y = (int)((idx / p.W) % p.H);
  const size_t n = idx / plane, pix = idx - n * plane;
  const float fy = p.flow[n * 2 * plane + pix], fx = p.flow[n * 2 * plane + plane + pix];
  float gxr, gyr;  // unclipped grid (for the clip mask)
  warp_grid(fx, fy, x, y, p.H, p.W, 0, gxr, gyr);
  float gxc = gxr, gyc = gyr;
  if (p.clip) { gxc = fminf(fmaxf(gxr, -1.f), 1.f); gyc = fminf(fmaxf(gyr, -1.f), 1.f); }
  // BilinearSamplerBackward
  const float y_real = (gyc + 1.f) * (float)(p.H - 1) / 2.f, x_real = (gxc + 1.f) * (float)(p.W - 1) / 2.f;
  const float fyr = floorf(y_real), fxr = floorf(x_real);
  const int ty = (int)fminf(fmaxf(fyr, -2.f), (float)p.H + 1.f), tx = (int)fminf(fmaxf(fxr, -2.f), (float)p.W + 1.f);
  const float wy = 1.f - (y_real - fyr), wx = 1.f - (x_real - fxr);
  const bool y0 = ty >= 0 && ty <= p.H - 1, y1 = ty + 1 >= 0 && ty + 1 <= p.H - 1;
  const bool x0 = tx >= 0 && tx <= p.W - 1, x1 = tx + 1 >= 0 && tx + 1 <= p.W - 1;
  float gwy = 0.f, gwx = 0.f;
  const int cy0 = min(max(ty, 0), p.H - 1), cy1 = min(max(ty + 1, 0), p.H - 1);
  const int cx0 = min(max(tx, 0), p.W - 1), cx1 = min(max(tx + 1, 0), p.W - 1);
  const int i00 = cy0 * p.W + cx0, i01 = cy0 * p.W + cx1, i10 = cy1 * p.W + cx0, i11 = cy1 * p.W + cx1;
  for (int c = 0; c < p.C; ++c) {
    const size_t cb = (n * p.C + c) * plane;
    const float g = p.gout[cb + pix];
    const float *pl = p.x + cb;
    // unconditional loads at clamped addresses, masked by a select afterwards: a branch per load would make the
    // compiler wait for each before issuing the next
    const float vtl = pl[i00], vtr = pl[i01], vbl = pl[i10], vbr = pl[i11];
    const float tl = (y0 && x0) ? vtl : 0.f;
    const float tr = (y0 && x1) ? vtr : 0.f;
    const float bl = (y1 && x0) ? vbl : 0.f;
    const float br = (y1 && x1) ? vbr : 0.f;
    if (p.req_x) {
      float *gp = p.gx + cb;
      if (y0 && x0) atomicAdd(gp + (size_t)ty * p.W + tx, g * wy * wx);
      if (y0 && x1) atomicAdd(gp + (size_t)ty * p.W + tx + 1, g * wy * (1.f - wx));
      if (y1 && x0) atomicAdd(gp + (size_t)(ty + 1) * p.W + tx, g * (1.f - wy) * wx);
      if (y1 && x1) atomicAdd(gp + (size_t)(ty + 1) * p.W + tx + 1, g * (1.f - wy) * (1.f - wx));
    }
    gwy -= g * (tr - br + (tl - tr - bl + br) * wx);
    gwx -= g * (bl - br + (tl - tr - bl + br) * wy);
  }
  if (p.req_flow) {
    float ggy = gwy * (float)(p.H - 1) / 2.f, ggx = gwx * (float)(p.W - 1) / 2.f;
    if (p.clip) {  // clip backward: gradient passes only inside [-1, 1]
      if (!(gxr >= -1.f && gxr <= 1.f)) ggx = 0.f;
      if (!(gyr >= -1.f && gyr <= 1.f)) ggy = 0.f;
    }
    const float nx = (float)((p.W - 1) / 2.0), ny = (float)((p.H - 1) / 2.0);
    float *gfy = p.gflow + n * 2 * plane + pix, *gfx = gfy + plane;  // channel 0 = dy, 1 = dx
    const float vy = ggy / ny, vx = ggx / nx;
    *gfy = (p.req_flow == 3 ? *gfy : 0.f) + vy;
    *gfx = (p.req_flow == 3 ? *gfx : 0.f) + vx;
  }
}

// ---- BilinearSampler / GridGenerator('warp') backward on their own ---------------------------------------------
// What MXNet's autograd reaches when the reference's operator PAIR (layer.py:17-18) is differentiated: the full model
// trains through c40 = warp(c20, Upsample(4)(flow2) * scale) (MaskFlownet.py:311, block_grad=False).  Semantics:
// oracle/mfn_ref_body.inc bilinear_sampler_bwd (BilinearSamplerBackward of bilinear_sampler.cc) and
// grid_generator_warp_bwd.  One thread per output pixel, taps once for all channels; the data gradient is a scatter
// (fp32 atomics, like MXNet's GPU kernel), the grid gradient a per-pixel sum over the channels.
struct SamplerBwdParams {
  const float *gout, *data, *grid;
  float *gdata, *ggrid;
  int N, C, iH, iW, oH, oW, req_data, req_grid;
};
__global__ __launch_bounds__(256) void bilinear_sampler_bwd_kernel(SamplerBwdParams p) {
  const size_t oplane = (size_t)p.oH * p.oW, iplane = (size_t)p.iH * p.iW;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)p.N * oplane) return;
  const size_t n = idx / oplane, pix = idx - n * oplane;
  const float gxv = p.grid[n * 2 * oplane + pix], gyv = p.grid[n * 2 * oplane + oplane + pix];
  const float y_real = (gyv + 1.f) * (float)(p.iH - 1) / 2.f, x_real = (gxv + 1.f) * (float)(p.iW - 1) / 2.f;
  const float fyr = floorf(y_real), fxr = floorf(x_real);
  const int ty = (int)fminf(fmaxf(fyr, -2.f), (float)p.iH + 1.f), tx = (int)fminf(fmaxf(fxr, -2.f), (float)p.iW + 1.f);
  const float wy = 1.f - (y_real - fyr), wx = 1.f - (x_real - fxr);
  const bool y0 = ty >= 0 && ty <= p.iH - 1, y1 = ty + 1 >= 0 && ty + 1 <= p.iH - 1;
  const bool x0 = tx >= 0 && tx <= p.iW - 1, x1 = tx + 1 >= 0 && tx + 1 <= p.iW - 1;
  const int cy0 = min(max(ty, 0), p.iH - 1), cy1 = min(max(ty + 1, 0), p.iH - 1);
  const int cx0 = min(max(tx, 0), p.iW - 1), cx1 = min(max(tx + 1, 0), p.iW - 1);
  const int i00 = cy0 * p.iW + cx0, i01 = cy0 * p.iW + cx1, i10 = cy1 * p.iW + cx0, i11 = cy1 * p.iW + cx1;
  float gwy = 0.f, gwx = 0.f;
  for (int c = 0; c < p.C; ++c) {
    const float g = p.gout[(n * p.C + c) * oplane + pix];
    const float *pl = p.data + (n * p.C + c) * iplane;
    const float vtl = pl[i00], vtr = pl[i01], vbl = pl[i10], vbr = pl[i11];  // unconditional, masked by selects
    const float tl = (y0 && x0) ? vtl : 0.f, tr = (y0 && x1) ? vtr : 0.f;
    const float bl = (y1 && x0) ? vbl : 0.f, br = (y1 && x1) ? vbr : 0.f;
    if (p.req_data) {
      float *gp = p.gdata + (n * p.C + c) * iplane;
      if (y0 && x0) atomicAdd(gp + i00, g * wy * wx);
      if (y0 && x1) atomicAdd(gp + i01, g * wy * (1.f - wx));
      if (y1 && x0) atomicAdd(gp + i10, g * (1.f - wy) * wx);
      if (y1 && x1) atomicAdd(gp + i11, g * (1.f - wy) * (1.f - wx));
    }
    gwy -= g * (tr - br + (tl - tr - bl + br) * wx);
    gwx -= g * (bl - br + (tl - tr - bl + br) * wy);
  }
  if (p.req_grid) {
    float *ggx = p.ggrid + n * 2 * oplane + pix, *ggy = ggx + oplane;  // channel 0 = x
    const float vx = gwx * (float)(p.iW - 1) / 2.f, vy = gwy * (float)(p.iH - 1) / 2.f;
    *ggx = (p.req_grid == 3 ? *ggx : 0.f) + vx;
    *ggy = (p.req_grid == 3 ? *ggy : 0.f) + vy;
  }
}

struct GridWarpBwdParams { const float *ggrid; float *gflow; int N, H, W, req; };
__global__ __launch_bounds__(256) void grid_warp_bwd_kernel(GridWarpBwdParams p) {
  const size_t plane = (size_t)p.H * p.W;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;  // over N*2*plane
  if (idx >= (size_t)p.N * 2 * plane) return;
  const bool is_y = ((idx / plane) & 1) != 0;
  const float nrm = is_y ? (float)((p.H - 1) / 2.0) : (float)((p.W - 1) / 2.0);
  const float v = p.ggrid[idx] / nrm;
  p.gflow[idx] = (p.req == 3 ? p.gflow[idx] : 0.f) + v;
}

// ---- deformable convolution -------------------------------------------------------------------------------
struct DcBwdParams {
  const float *gout, *x, *offset, *w;
  float *gx, *goffset, *gw, *gbias;
  int N, Cin, H, W, Cout, Ho, Wo, kh, kw, sh, sw, ph, pw, dh, dw, groups, dg;
  int req_x, req_offset, req_w, req_bias;
  int cchunk;  // input channels per thread of the input/offset gradient kernel
};

// get_gradient_weight of deformable_im2col.cuh
__device__ __forceinline__ float dc_gradient_weight(float ah, float aw, int h, int w, int H, int W) {
  if (ah < 0.f || ah > (float)H || aw < 0.f || aw > (float)W) return 0.f;
  ah = fmaxf(ah, 0.f);
  aw = fmaxf(aw, 0.f);
  int hl = (int)ah, wl = (int)aw, hh, wh;
  if (hl >= H - 1) { hh = hl = H - 1; ah = (float)hl; } else hh = hl + 1;
  if (wl >= W - 1) { wh = wl = W - 1; aw = (float)wl; } else wh = wl + 1;
  float wt = 0.f;
  if (h == hl) {
    if (w == wl) wt = ((float)(h + 1) - ah) * ((float)(w + 1) - aw);
    else if (w == wh) wt = ((float)(h + 1) - ah) * (aw + 1.f - (float)w);
  } else if (h == hh) {
    if (w == wl) wt = (ah + 1.f - (float)h) * ((float)(w + 1) - aw);
    else if (w == wh) wt = (ah + 1.f - (float)h) * (aw + 1.f - (float)w);
  }
  return wt;
}
// get_coordinate_weight of deformable_im2col.cuh
__device__ __forceinline__ float dc_coordinate_weight(float ah, float aw, int H, int W, const float *im, int bp_dir) {
  if (ah < 0.f || ah > (float)H || aw < 0.f || aw > (float)W) return 0.f;
  int hl = (int)ah, wl = (int)aw, hh, wh;
  if (hl >= H - 1) { hh = hl = H - 1; ah = (float)hl; } else hh = hl + 1;
  if (wl >= W - 1) { wh = wl = W - 1; aw = (float)wl; } else wh = wl + 1;
  const float v11 = im[(size_t)hl * W + wl], v12 = im[(size_t)hl * W + wh];
  const float v21 = im[(size_t)hh * W + wl], v22 = im[(size_t)hh * W + wh];
  float wt = 0.f;
  if (bp_dir == 0) {
    wt += -1.f * ((float)(wl + 1) - aw) * v11;
    wt += -1.f * (aw - (float)wl) * v12;
    wt += ((float)(wl + 1) - aw) * v21;
    wt += (aw - (float)wl) * v22;
  } else {
    wt += -1.f * ((float)(hl + 1) - ah) * v11;
    wt += ((float)(hl + 1) - ah) * v12;
    wt += -1.f * (ah - (float)hl) * v21;
    wt += (ah - (float)hl) * v22;
  }
  return wt;
}

// input + offset gradients: one thread per (n, output pixel, chunk of input channels).  The column
// gradient cg[(c,k),p] = sum_o W[o,c,k] * gout[n,o,p] is formed on the fly; gx is an atomic scatter
// (deformable_col2im), goffset accumulates in registers over the chunk, then one atomic per offset channel.
__global__ __launch_bounds__(256) void dc_bwd_input_kernel(DcBwdParams p) {
  const size_t oplane = (size_t)p.Ho * p.Wo, plane = (size_t)p.H * p.W;
  const int T = p.kh * p.kw;
  const int nchunks = (p.Cin + p.cchunk - 1) / p.cchunk;
  const size_t total = (size_t)p.N * oplane * nchunks;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const size_t pix = idx % oplane;
  const int chunk = (int)((idx / oplane) % nchunks);
  const int n = (int)(idx / (oplane * nchunks));
  const int wo = (int)(pix % p.Wo), ho = (int)(pix / p.Wo);
  const int h_in = ho * p.sh - p.ph, w_in = wo * p.sw - p.pw;
  const int cpg = p.Cin / p.groups, opg = p.Cout / p.groups, cpd = p.Cin / p.dg;
  const float *go = p.gout + (size_t)n * p.Cout * oplane + pix;
  const int c_lo = chunk * p.cchunk, c_hi = min(p.Cin, c_lo + p.cchunk);
  for (int t = 0; t < T; ++t) {
    const int ti = t / p.kw, tj = t - ti * p.kw;
    // a chunk may straddle deformable groups: accumulate per group
    int cur_dg = -1;
    float acc_h = 0.f, acc_w = 0.f, oh = 0.f, ow = 0.f;
    for (int c = c_lo; c < c_hi; ++c) {
      const int dgi = c / cpd;
      if (dgi != cur_dg) {
        if (cur_dg >= 0 && p.req_offset) {
          float *gof = p.goffset + ((size_t)n * p.dg + cur_dg) * 2 * T * oplane + pix;
          atomicAdd(gof + (size_t)(2 * t) * oplane, acc_h);
          atomicAdd(gof + (size_t)(2 * t + 1) * oplane, acc_w);
        }
        cur_dg = dgi;
        acc_h = acc_w = 0.f;
        const float *op = p.offset + ((size_t)n * p.dg + dgi) * 2 * T * oplane + pix;
        oh = op[(size_t)(2 * t) * oplane];
        ow = op[(size_t)(2 * t + 1) * oplane];
      }
      // column gradient of (c, t) at this pixel
      const int g = c / cpg, cl = c - g * cpg;
      float cg = 0.f;
      for (int ol = 0; ol < opg; ++ol) {
        const int o = g * opg + ol;
        cg = fmaf(p.w[((size_t)o * cpg + cl) * T + t], go[(size_t)o * oplane], cg);
      }
      const float inv_h = (float)(h_in + ti * p.dh) + oh, inv_w = (float)(w_in + tj * p.dw) + ow;
      const float *im = p.x + ((size_t)n * p.Cin + c) * plane;
      if (p.req_offset) {
        float ch_ = inv_h, cw_ = inv_w;
        if (inv_h < 0.f || inv_w < 0.f || inv_h >= (float)p.H || inv_w >= (float)p.W) ch_ = cw_ = -1.f;
        acc_h += dc_coordinate_weight(ch_, cw_, p.H, p.W, im, 0) * cg;
        acc_w += dc_coordinate_weight(ch_, cw_, p.H, p.W, im, 1) * cg;
      }
      if (p.req_x) {
        const int cur_h = (int)inv_h, cur_w = (int)inv_w;  // truncation, as deformable_col2im
        float *gim = p.gx + ((size_t)n * p.Cin + c) * plane;
        for (int dy = -2; dy <= 2; ++dy)
          for (int dx = -2; dx <= 2; ++dx) {
            const int hy = cur_h + dy, wx = cur_w + dx;
            if (hy >= 0 && hy < p.H && wx >= 0 && wx < p.W && fabsf(inv_h - (float)hy) < 1.f &&
                fabsf(inv_w - (float)wx) < 1.f) {
              const float wt = dc_gradient_weight(inv_h, inv_w, hy, wx, p.H, p.W);
              if (wt != 0.f) atomicAdd(gim + (size_t)hy * p.W + wx, wt * cg);
            }
          }
      }
    }
    if (cur_dg >= 0 && p.req_offset) {
      float *gof = p.goffset + ((size_t)n * p.dg + cur_dg) * 2 * T * oplane + pix;
      atomicAdd(gof + (size_t)(2 * t) * oplane, acc_h);
      atomicAdd(gof + (size_t)(2 * t + 1) * oplane, acc_w);
    }
  }
}

// weight gradient: block per (input channel c (within group), tap t, chunk of 16 filters of ONE group);
// threads stride over (n, pixel), rebuild the column value once and keep 16 partial sums; LDS tree
// reduction.  gw[o, cl, t] (+)= sum_{n,p} gout[n,o,p] * col[n,(c,t),p]
constexpr int DC_BW_OC = 16;
__global__ __launch_bounds__(256) void dc_bwd_weight_kernel(DcBwdParams p) {
  MFN_DYN_SHARED(float, red);  // [DC_BW_OC][256]
  const int T = p.kh * p.kw;
  const int cpg = p.Cin / p.groups, opg = p.Cout / p.groups, cpd = p.Cin / p.dg;
  const int ochunks = (opg + DC_BW_OC - 1) / DC_BW_OC;
  int b = blockIdx.x;
  const int oc = b % ochunks; b /= ochunks;
  const int t = b % T; b /= T;
  const int c = b;  // global input channel
  const int g = c / cpg, cl = c - g * cpg, dgi = c / cpd;
  const int o0 = g * opg + oc * DC_BW_OC;
  const int no = min(DC_BW_OC, g * opg + opg - o0);
  const int ti = t / p.kw, tj = t - ti * p.kw;
  const size_t oplane = (size_t)p.Ho * p.Wo, plane = (size_t)p.H * p.W;
  float s[DC_BW_OC];
  MFN_UNROLL
  for (int k = 0; k < DC_BW_OC; ++k) s[k] = 0.f;
  const size_t npix = (size_t)p.N * oplane;
  for (size_t q = threadIdx.x; q < npix; q += 256) {
    const int n = (int)(q / oplane);
    const size_t pix = q - (size_t)n * oplane;
    const int wo = (int)(pix % p.Wo), ho = (int)(pix / p.Wo);
    const float *op = p.offset + ((size_t)n * p.dg + dgi) * 2 * T * oplane + pix;
    const DcTap tp = dc_make_tap(op[(size_t)(2 * t) * oplane], op[(size_t)(2 * t + 1) * oplane], ho * p.sh - p.ph,
                                 wo * p.sw - p.pw, ti * p.dh, tj * p.dw, p.H, p.W, true);
    const float *pl = p.x + ((size_t)n * p.Cin + c) * plane;
    const int bb = tp.base & 0x3FFFFFFF, dwi = (tp.base >> 30) & 1;
    const float col = tp.w1 * pl[bb] + tp.w2 * pl[bb + dwi] + tp.w3 * pl[bb + tp.dhW] + tp.w4 * pl[bb + tp.dhW + dwi];
    const float *go = p.gout + ((size_t)n * p.Cout + o0) * oplane + pix;
    MFN_UNROLL
    for (int k = 0; k < DC_BW_OC; ++k)
      if (k < no) s[k] = fmaf(go[(size_t)k * oplane], col, s[k]);
  }
  MFN_UNROLL
  for (int k = 0; k < DC_BW_OC; ++k) red[k * 256 + threadIdx.x] = s[k];
  __syncthreads();
  for (int st = 128; st >= 1; st >>= 1) {
    if ((int)threadIdx.x < st) {
      MFN_UNROLL
      for (int k = 0; k < DC_BW_OC; ++k) red[k * 256 + threadIdx.x] += red[k * 256 + threadIdx.x + st];
    }
    __syncthreads();
  }
  if ((int)threadIdx.x < no) {
    float *dst = p.gw + ((size_t)(o0 + threadIdx.x) * cpg + cl) * T + t;
    *dst = (p.req_w == 3 ? *dst : 0.f) + red[threadIdx.x * 256];
  }
}

// ---- weight gradient on the fp32 MFMA (groups == 1, one deformable group) -----------------------------------
// gw[o, k] (+)= sum_p gout[o, p] * col[k, p],  k = c*T + t  (the layout of gw itself),  p over all (n, ho, wo).
// v_mfma_f32_32x32x2_f32 with the PIXELS as the reduction dimension: A[i=o][k=p] comes from an LDS tile of gout,
// B[k=p][j=combo] is the deformable-im2col value of combo j at pixel p, which lane (j, p&1) forms itself from a
// per-tile geometry table in LDS (4 bilinear weights + base index per (pixel, tap): any per-tap offsets) and 4
// gathers.  A block = 4 waves = 4 x 32 combos x MTO filter tiles over one slice of the pixels; slices and combo
// groups are spread over the grid and the partial sums meet in gw through fp32 atomics (as goffset / gx do).
struct DcBwdWParams {
  const float *gout, *x, *offset;
  float *gw;
  float *gbias;            // or NULL: the bias gradient (row sums of gout) rides along in the combo-group-0 blocks
  int N, Cin, H, W, Cout, Ho, Wo, kh, kw, sh, sw, ph, pw, dh, dw;
  int P, K2, T;            // N*Ho*Wo, Cin*T, kh*kw
  int tiles_per_block;     // 32-pixel tiles each block walks
};
// LDS floats of the weight-gradient kernel: geometry [T][7][32] + gout tile [MTO*32][33] + column tiles 4 x [32][33]
constexpr size_t dc_bwd_weight_lds_floats(int T, int mto) { return (size_t)T * 7 * 32 + (size_t)mto * 32 * 33 + (size_t)4 * 32 * 33; }
template <int MTO>
__global__ __launch_bounds__(256) void dc_bwd_weight_mfma_kernel(DcBwdWParams p) {
  constexpr int GF = 7;   // fields of a (tap, pixel) geometry entry, stored field-major [T][GF][32 pixels]: w1..w4, base,
                          // dhW, image offset -- lanes that are neighbouring PIXELS read neighbouring words (no conflicts)
  constexpr int GS = 33;  // tile row stride (32 pixels + 1: the 32 rows of an MFMA operand on distinct banks)
  MFN_DYN_SHARED(float, lds);
  float *geom = lds;                                   // [T][GF][32]
  float *gs = lds + p.T * GF * 32;                     // [MTO*32][GS]   gout tile
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = MFN_UNIFORM(tid >> 6);
  float *colt = gs + MTO * 32 * GS + wave * 32 * GS;   // [32 combos][GS] this wave's im2col tile
  const int j = lane & 31, half = lane >> 5;
  const int T = p.T;
  const size_t oplane = (size_t)p.Ho * p.Wo, plane = (size_t)p.H * p.W;
  const int o0 = blockIdx.z * (MTO * 32);
  const int k0 = (blockIdx.y * 4 + wave) * 32;     // the wave's first combo
  const int k = k0 + j;                            // this lane's combo as MFMA column
  const bool k_ok = k < p.K2;
  f32x16 acc[MTO];
  MFN_UNROLL
  for (int m = 0; m < MTO; ++m)
    MFN_UNROLL
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  // column build: lane (j, half) = PIXEL j of the tile, combos k0 + 2q + half, q < 16 -- for one (channel, tap) the 32
  // pixels of a tile gather neighbouring addresses (smooth flow), where one lane per combo gathered 64 cache lines
  int qc[16], qt[16];
  MFN_UNROLL
  for (int q = 0; q < 16; ++q) {
    const int kk = min(k0 + 2 * q + half, p.K2 - 1);
    qc[q] = kk / T;
    qt[q] = kk - qc[q] * T;
  }

  float bsum = 0.f;  // thread tid < MTO*32 of a combo-group-0 block: sum of gout row o0 + tid over this block's pixels
  const int tile0 = blockIdx.x * p.tiles_per_block;
  // What a thread stages for a tile -- the offsets of its (tap, pixel) entries and its gout values -- is loaded one tile
  // ahead: the loads of tile i+1 are issued right after tile i's staging barrier and travel during its column build and
  // MFMAs.  (One block alone on a CU needed ~15 k cycles per 32-pixel tile: offsets -> table -> gathers -> MFMA, each a
  // full memory round trip behind the other.)
  constexpr int NE = 4;  // entries per thread: taps tid/32 + 8 i (T <= 25)
  struct Stage {
    float oh[NE], ow[NE], gv[MTO * 4];
    int n, rem, ho, wo;
    bool ok;
  };
  const int pp = tid & 31, tt0 = tid >> 5;
  auto prefetch = [&](int pbase, Stage &q) {
    const int pl = pbase + pp;
    q.ok = pl < p.P;
    const int pc = q.ok ? pl : 0;
    q.n = pc / (int)oplane;
    q.rem = pc - q.n * (int)oplane;
    q.ho = q.rem / p.Wo;
    q.wo = q.rem - q.ho * p.Wo;
    const float *op = p.offset + (size_t)q.n * 2 * T * oplane + q.rem;
    MFN_UNROLL
    for (int i = 0; i < NE; ++i) {
      const int tt = min(tt0 + 8 * i, T - 1);
      q.oh[i] = op[(size_t)(2 * tt) * oplane];
      q.ow[i] = op[(size_t)(2 * tt + 1) * oplane];
    }
    const float *gp = p.gout + (size_t)q.n * p.Cout * oplane + q.rem;
    MFN_UNROLL
    for (int i = 0; i < MTO * 4; ++i) q.gv[i] = gp[(size_t)min(o0 + tt0 + 8 * i, p.Cout - 1) * oplane];
  };
  // one tile ahead only with one filter tile per wave: with two the extra live registers cost a wave per SIMD and more
  // than the prefetch returns (measured: levels 5..3 +8 / +10 / +25 us with it, level 2 -25 us)
  constexpr bool AHEAD = MTO == 1;
  Stage cur;
  if (AHEAD) prefetch(tile0 * 32 < p.P ? tile0 * 32 : 0, cur);
  for (int tl = 0; tl < p.tiles_per_block; ++tl) {
    const int pbase = (tile0 + tl) * 32;
    if (pbase >= p.P) break;  // uniform
    __syncthreads();          // the previous tile's readers are done
    if (!AHEAD) prefetch(pbase, cur);
    // geometry table: one entry per (tap, pixel)
    MFN_UNROLL
    for (int i = 0; i < NE; ++i) {
      const int tt = tt0 + 8 * i;
      if (tt < T) {
        const int ti = tt / p.kw, tj = tt - ti * p.kw;
        const DcTap tp = dc_make_tap(cur.oh[i], cur.ow[i], cur.ho * p.sh - p.ph, cur.wo * p.sw - p.pw, ti * p.dh, tj * p.dw,
                                     p.H, p.W, cur.ok);
        float *g = geom + (size_t)tt * GF * 32 + pp;
        g[0] = tp.w1; g[32] = tp.w2; g[64] = tp.w3; g[96] = tp.w4;
        reinterpret_cast<int *>(g)[128] = tp.base;                     // bit 30 = dwi
        reinterpret_cast<int *>(g)[160] = tp.dhW;
        reinterpret_cast<int *>(g)[192] = cur.n * p.Cin * (int)plane;  // element offset of image n (checked < 2^31)
      }
    }
    // gout tile [filter][pixel]
    MFN_UNROLL
    for (int i = 0; i < MTO * 4; ++i) {
      const int ol = tt0 + 8 * i;
      gs[ol * GS + pp] = (cur.ok && o0 + ol < p.Cout) ? cur.gv[i] : 0.f;
    }
    __syncthreads();
    if (AHEAD) {
      const int nb = pbase + 32;
      prefetch((tl + 1 < p.tiles_per_block && nb < p.P) ? nb : pbase, cur);  // the next tile's staging loads, in flight from here
    }
    if (p.gbias && blockIdx.y == 0 && tid < MTO * 32) {
      MFN_UNROLL
      for (int q2 = 0; q2 < 32; ++q2) bsum += gs[tid * GS + q2];
    }
    // the wave's column tile: col[combo][pixel] (zero for combos past K2 and pixels past P: their weights are zero).
    // All 16 values are formed before the first is stored: a store between them would keep the next combo's table
    // reads and gathers (possible aliases, for the compiler) from being issued until it retires.
    float colv[16];
    MFN_UNROLL
    for (int q = 0; q < 16; ++q) {
      const float *g = geom + (size_t)qt[q] * GF * 32 + j;
      const int base = reinterpret_cast<const int *>(g)[128], dhW = reinterpret_cast<const int *>(g)[160];
      const int ioff = reinterpret_cast<const int *>(g)[192];
      const int bb = base & 0x3FFFFFFF, dwi = (base >> 30) & 1;
      const float *pl = p.x + (size_t)ioff + (size_t)qc[q] * plane;
      const float v1 = pl[bb], v2 = pl[bb + dwi], v3 = pl[bb + dhW], v4 = pl[bb + dhW + dwi];
      const float col = g[0] * v1 + g[32] * v2 + g[64] * v3 + g[96] * v4;
      colv[q] = (k0 + 2 * q + half < p.K2) ? col : 0.f;
    }
    MFN_UNROLL
    for (int q = 0; q < 16; ++q) colt[(2 * q + half) * GS + j] = colv[q];
    MFN_WAIT_LGKM0();  // the wave's own tile: written above, read below (no block barrier)
    MFN_UNROLL
    for (int s = 0; s < 16; ++s) {
      const int pp = 2 * s + half;
      const float col = colt[j * GS + pp];
      MFN_UNROLL
      for (int m = 0; m < MTO; ++m) acc[m] = MFN_MFMA_32x32x2(gs[(m * 32 + j) * GS + pp], col, acc[m]);
    }
    MFN_WAIT_LGKM0();  // ... and all read before the next tile's build overwrites it
  }
  if (p.gbias && blockIdx.y == 0 && tid < MTO * 32 && o0 + tid < p.Cout) atomicAdd(p.gbias + o0 + tid, bsum);
  // D reg r of lane (j, half): filter row (r&3)+8*(r>>2)+4*half, combo j
  if (!k_ok) return;
  MFN_UNROLL
  for (int m = 0; m < MTO; ++m)
    MFN_UNROLL
    for (int r = 0; r < 16; ++r) {
      const int o = o0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (o < p.Cout) atomicAdd(p.gw + (size_t)o * p.K2 + k, acc[m][r]);
    }
}
template <int MTO>
inline int dc_bwd_weight_mfma_launch(DcBwdWParams p, int pixel_slices, hipStream_t stream) {
  const int tiles = cdiv(p.P, 32);
  p.tiles_per_block = cdiv(tiles, pixel_slices);
  const dim3 grid(cdiv(tiles, p.tiles_per_block), cdiv(cdiv(p.K2, 32), 4), cdiv(p.Cout, MTO * 32));
  const size_t lds = dc_bwd_weight_lds_floats(p.T, MTO) * sizeof(float);
  return launch("dc_bwd_weight_mfma", dc_bwd_weight_mfma_kernel<MTO>, grid, dim3(256), lds, stream, p);
}

// ---- input + offset gradients on the fp32 MFMA with LDS-privatised scatter (groups == 1, dg == 1, stride 1) -----
// cg[(c,t), p] = sum_o W[o,c,t] * gout[o,p] is a GEMM: D[pixel][channel] per tap with K = Cout, A = gout (coalesced),
// B = W (read in place).  The scatter of cg into gx (deformable_col2im) is what bounds MXNet's kernel and the simple
// one here: 36 global fp32 atomics per (pixel, channel) at ~57 G/s.  LDS float atomics are no way out either (measured
// ~3 cycles per lane).  So the scatter is arranged to need NO atomics: a wave owns a 2x16 pixel strip and 32 channels,
// lane j owns channel j (D column), walks the strip's pixels (D rows) and adds cg * bilinear weight into ITS OWN
// channel plane of a wave-private LDS window (10 rows x 28 columns, placed by the offset of the block's centre pixel)
// with plain read-add-write; the two half-waves hold different pixels of the same channel and take turns.  The four
// strips of a block (8x16 pixels) are merged and flushed once with ~3 global atomics per (pixel, channel); only
// contributions outside a window (rough flows) go to global memory directly.  goffset is reduced over the 32 channels
// with a DPP scan.  The bilinear corner weights are the forward's (dc_axis): MXNet's get_gradient_weight equals them.
struct DcBwdIParams {
  const float *gout, *x, *offset, *w;
  float *gx, *goffset;
  int N, Cin, H, W, Cout, kh, kw, ph, pw, dh, dw;  // stride 1: Ho == H, Wo == W
  int T, tiles_x, tiles_y;
  int req_x, req_offset;
  unsigned long long *timeline;  // measurement only: per block {geometry, MFMA, scatter, total} shader cycles
  // flow mode (mfn_deform_conv_shared_bwd; dc_backward.h: DcBwdPParams): offsets from flow[n][dir][pixel], d/dflow instead of goffset
  const float *flow;
  float *gflow;
  float flow_scale, flow_stride;
};
constexpr int DCI_TH = 8, DCI_TW = 16, DCI_WR = 10, DCI_WC = 28, DCI_GW = 16;
constexpr int DCI_PLANE = DCI_WR * DCI_WC + 1;  // odd channel-plane stride: the 32 lanes (channels) hit distinct banks
constexpr size_t dc_bwd_input_lds_bytes() { return ((size_t)4 * 32 * DCI_PLANE + (size_t)4 * 32 * DCI_GW) * sizeof(float); }

__global__ __launch_bounds__(256) void dc_bwd_input_tile_kernel(DcBwdIParams p) {
  constexpr int TH = DCI_TH, TW = DCI_TW, WR = DCI_WR, WC = DCI_WC, GW = DCI_GW, PL = DCI_PLANE;
  MFN_DYN_SHARED(float, lds);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = MFN_UNIFORM(tid >> 6);
  float *win = lds + (size_t)wave * 32 * PL;                      // [32 channels][WR][WC] (+1), this wave's strip
  float *geom = lds + (size_t)4 * 32 * PL + (size_t)wave * 32 * GW;  // [32 pixels][GW], rebuilt for every tap
  const int j = lane & 31, half = lane >> 5;
  const int T = p.T, H = p.H, W = p.W;
  const size_t plane = (size_t)H * W;
  const int tpi = p.tiles_x * p.tiles_y;
  const int bx = gridDim.y == 1 ? (int)mfn_xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;  // neighbours share an L2
  const int n = bx / tpi, rt = bx - n * tpi;
  const int ty0 = (rt / p.tiles_x) * TH, tx0 = (rt % p.tiles_x) * TW;
  const int cb = blockIdx.y * 32;
  // window origin of the block: follows the offset of the tile's centre pixel (centre tap); strip w sits 2w rows lower
  int wy0, wx0;
  {
    const int cy = min(ty0 + TH / 2, H - 1), cx = min(tx0 + TW / 2, W - 1);
    float oh, ow;
    if (p.flow) {
      const float *fp = p.flow + (size_t)n * 2 * plane + (size_t)cy * W + cx;
      oh = fp[0] * p.flow_scale / p.flow_stride;
      ow = fp[plane] * p.flow_scale / p.flow_stride;
    } else {
      const float *op = p.offset + (size_t)n * 2 * T * plane + (size_t)cy * W + cx;
      oh = op[(size_t)(2 * (T / 2)) * plane];
      ow = op[(size_t)(2 * (T / 2) + 1) * plane];
    }
    const float fh = fminf(fmaxf(floorf(oh), -1.0e6f), 1.0e6f), fw = fminf(fmaxf(floorf(ow), -1.0e6f), 1.0e6f);
    wy0 = MFN_UNIFORM(ty0 - p.ph + (int)fh - (WR - 5) / 2 + 2 * wave);  // a strip needs 2 + dh*(kh-1) + 1 = 5 rows: centred
    wx0 = MFN_UNIFORM(tx0 - p.pw + (int)fw - (WC - (TW + p.dw * (p.kw - 1) + 1)) / 2);
  }
  for (int e = lane; e < 32 * PL; e += 64) win[e] = 0.f;

  // this lane as a PIXEL of the strip (geometry build, MFMA A operand) ...
  const int py = ty0 + 2 * wave + (j >> 4), px = tx0 + (j & 15);
  const bool pix_ok = py < H && px < W;
  const int pyc = min(py, H - 1), pxc = min(px, W - 1);
  const size_t pix = (size_t)pyc * W + pxc;
  // ... and as a CHANNEL (MFMA B operand, D column, owner of one window plane)
  const int c = cb + j;
  const bool c_ok = c < p.Cin;
  float *wpl = win + (size_t)j * PL;
  int half_o = half;
  MFN_OPAQUE(half_o);
  float *gim = p.gx + ((size_t)n * p.Cin + (c_ok ? c : 0)) * plane;
  const float *im = p.x + ((size_t)n * p.Cin + (c_ok ? c : 0)) * plane;

  unsigned long long tk0 = MFN_CYCLES(), tk_geo = 0, tk_mma = 0, tk_sc = 0, tka = tk0;
  for (int t = 0; t < T; ++t) {
    // ---- geometry of the strip's 32 pixels for tap t (lanes 0..31 write, everyone reads it back as broadcasts)
    MFN_WAIT_LGKM0();  // the previous tap's readers are done (wave-private table: no block barrier)
    if (half == 0) {
      float oh, ow;
      if (p.flow) {
        const float *fp = p.flow + (size_t)n * 2 * plane + pix;
        oh = fp[0] * p.flow_scale / p.flow_stride;
        ow = fp[plane] * p.flow_scale / p.flow_stride;
      } else {
        const float *op = p.offset + (size_t)n * 2 * T * plane + pix;
        oh = op[(size_t)(2 * t) * plane];
        ow = op[(size_t)(2 * t + 1) * plane];
      }
      const int ti = t / p.kw, tj = t - ti * p.kw;
      const int h_in = pyc - p.ph, w_in = pxc - p.pw;
      bool vh, vw;
      int hl, hh, wl, wh;
      float lh, lw;
      dc_axis(oh, h_in, ti * p.dh, H, vh, hl, hh, lh);
      dc_axis(ow, w_in, tj * p.dw, W, vw, wl, wh, lw);
      const bool valid = vh && vw && pix_ok;
      const int cy = h_in + hl, cx = w_in + wl, dyi = hh - hl, dxi = wh - wl;
      const int ry = cy - wy0, rx = cx - wx0;
      const bool iny0 = ry >= 0 && ry < WR, iny1 = ry + dyi >= 0 && ry + dyi < WR;
      const bool inx0 = rx >= 0 && rx < WC, inx1 = rx + dxi >= 0 && rx + dxi < WC;
      float *g = geom + (size_t)j * GW;
      int *gi = reinterpret_cast<int *>(g);
      g[0] = valid ? (1.f - lh) * (1.f - lw) : 0.f;
      g[1] = valid ? (1.f - lh) * lw : 0.f;
      g[2] = valid ? lh * (1.f - lw) : 0.f;
      g[3] = valid ? lh * lw : 0.f;
      gi[4] = valid ? ry * WC + rx : 0;                                        // cell of corner (0,0) in the window
      gi[5] = valid ? (dyi << 5) | (dxi << 4) | ((iny0 && inx0) ? 1 : 0) | ((iny0 && inx1) ? 2 : 0) |
                          ((iny1 && inx0) ? 4 : 0) | ((iny1 && inx1) ? 8 : 0) : 0;  // corner steps, in-window bits
      gi[6] = valid ? cy * W + cx : 0;                                         // the same corner in the image plane
      // deformable_col2im_coord at (inv_h, inv_w): 4 samples with MXNet's clamping, weights of d/dh and d/dw
      float ah = valid ? (float)(h_in + ti * p.dh) + oh : 0.f, aw = valid ? (float)(w_in + tj * p.dw) + ow : 0.f;
      int chl = (int)ah, cwl = (int)aw, chh, cwh;
      if (chl >= H - 1) { chh = chl = H - 1; ah = (float)chl; } else chh = chl + 1;
      if (cwl >= W - 1) { cwh = cwl = W - 1; aw = (float)cwl; } else cwh = cwl + 1;
      gi[7] = chl * W + cwl;
      gi[8] = ((chh - chl) << 1) | (cwh - cwl) | (valid ? 4 : 0);
      g[9] = (float)(cwl + 1) - aw;   // fw0
      g[10] = aw - (float)cwl;        // fw1
      g[11] = (float)(chl + 1) - ah;  // fh0
      g[12] = ah - (float)chl;        // fh1
    }
    MFN_WAIT_LGKM0();
    if (p.timeline) { const unsigned long long n_ = MFN_CYCLES(); tk_geo += n_ - tka; tka = n_; }
    // ---- D[pixel][channel] = sum_o gout[o][pixel] * W[o][channel][t]
    f32x16 acc;
    MFN_UNROLL
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float *ga = p.gout + (size_t)n * p.Cout * plane + pix;
    const float *wb = p.w + ((size_t)(c_ok ? c : 0) * T + t);
    for (int s2 = 0; s2 < p.Cout; s2 += 16) {  // eight k-steps per trip: sixteen unconditional loads in flight
      float a[8], b[8];
      MFN_UNROLL
      for (int u = 0; u < 8; ++u) {
        const int oc = min(s2 + 2 * u + half, p.Cout - 1);
        a[u] = ga[(size_t)oc * plane];
        b[u] = wb[(size_t)oc * p.Cin * T];
      }
      MFN_UNROLL
      for (int u = 0; u < 8; ++u) {
        const bool ook = s2 + 2 * u + half < p.Cout;
        acc = MFN_MFMA_32x32x2((ook && pix_ok) ? a[u] : 0.f, (ook && c_ok) ? b[u] : 0.f, acc);
      }
    }
    if (p.timeline) { MFN_OPAQUE(acc[0]); const unsigned long long n_ = MFN_CYCLES(); tk_mma += n_ - tka; tka = n_; }
    // ---- scatter: D reg r of lane (j, half) = pixel (r&3)+8*(r>>2)+4*half of the strip, channel j.
    // Two passes so that nothing that stores sits between the loads of different pixels (the compiler may then keep
    // all 16 pixels' geometry reads and x loads in flight): first the offset gradient (loads only; its atomics are
    // deferred to the end of the pass), then the gx read-add-write pass.
    if (p.req_offset) {
      float sh_[16], sw_[16];
      MFN_UNROLL
      for (int r = 0; r < 16; ++r) {
        const int pp = (r & 3) + 8 * (r >> 2) + 4 * half;
        const float *g = geom + (size_t)pp * GW;
        const int *gi = reinterpret_cast<const int *>(g);
        const float cg = c_ok ? acc[r] : 0.f;
        const int i11 = gi[7], st = gi[8];
        const int dh_ = (st >> 1) & 1 ? W : 0, dw_ = st & 1;
        const float v11 = im[i11], v12 = im[i11 + dw_], v21 = im[i11 + dh_], v22 = im[i11 + dh_ + dw_];
        const float on = (st & 4) ? cg : 0.f;
        sh_[r] = mfn_half_sum_top((-g[9] * v11 - g[10] * v12 + g[9] * v21 + g[10] * v22) * on);
        sw_[r] = mfn_half_sum_top((-g[11] * v11 + g[11] * v12 - g[12] * v21 + g[12] * v22) * on);
      }
      if (j == 31) {  // the half-wave's top lane holds the sums; other channel blocks add to the same entries
        MFN_UNROLL
        for (int r = 0; r < 16; ++r) {
          const int pp = (r & 3) + 8 * (r >> 2) + 4 * half;
          const int y = ty0 + 2 * wave + (pp >> 4), x = tx0 + (pp & 15);
          if (y < H && x < W && (sh_[r] != 0.f || sw_[r] != 0.f)) {
            if (p.flow) {
              float *gf = p.gflow + (size_t)n * 2 * plane + (size_t)y * W + x;
              const float ratio = p.flow_scale / p.flow_stride;
              atomicAdd(gf, sh_[r] * ratio);
              atomicAdd(gf + plane, sw_[r] * ratio);
            } else {
              float *gof = p.goffset + ((size_t)n * 2 * T + 2 * t) * plane + (size_t)y * W + x;
              atomicAdd(gof, sh_[r]);
              atomicAdd(gof + plane, sw_[r]);
            }
          }
        }
      }
    }
    if (p.req_x) {
      MFN_UNROLL
      for (int r = 0; r < 16; ++r) {
        const int pp = (r & 3) + 8 * (r >> 2) + 4 * half;
        const float *g = geom + (size_t)pp * GW;
        const int *gi = reinterpret_cast<const int *>(g);
        const float cg = c_ok ? acc[r] : 0.f;
        const int cell = gi[4], fl = gi[5], gb_ = gi[6];
        const int dyc = (fl >> 5) & 1 ? WC : 0, dxc = (fl >> 4) & 1;
        const int dyg = (fl >> 5) & 1 ? W : 0;
        const float c1 = g[0] * cg, c2 = g[1] * cg, c3 = g[2] * cg, c4 = g[3] * cg;
        // The two half-waves hold different pixels of the SAME channel plane: they take turns; the LDS pipe executes
        // a wave's accesses in order, so the second turn (and the next pixel) sees the first one's writes without a
        // wait.  half_o is opaque so that the compiler cannot fold the two turns into one unordered pass.  Within a
        // turn the four cells are distinct (clamped corners have dxc / dyc == 0 and zero weight -> folded below), so
        // the four reads go out together: one LDS round trip per turn.
        const bool all_in = (fl & 15) == 15 && dxc && dyc;
        MFN_UNROLL
        for (int hs = 0; hs < 2; ++hs) {
          if (half_o == hs) {
            if (all_in) {
              const float o1 = wpl[cell], o2 = wpl[cell + 1], o3 = wpl[cell + WC], o4 = wpl[cell + WC + 1];
              wpl[cell] = o1 + c1; wpl[cell + 1] = o2 + c2; wpl[cell + WC] = o3 + c3; wpl[cell + WC + 1] = o4 + c4;
            } else {
              if (fl & 1) wpl[cell] += c1; else if (c1 != 0.f) atomicAdd(gim + gb_, c1);
              if (fl & 2) wpl[cell + dxc] += c2; else if (c2 != 0.f) atomicAdd(gim + gb_ + dxc, c2);
              if (fl & 4) wpl[cell + dyc] += c3; else if (c3 != 0.f) atomicAdd(gim + gb_ + dyg, c3);
              if (fl & 8) wpl[cell + dyc + dxc] += c4; else if (c4 != 0.f) atomicAdd(gim + gb_ + dyg + dxc, c4);
            }
          }
          MFN_WAVE_SYNC_EMU();
        }
      }
    }
    if (p.timeline) { MFN_WAIT_LGKM0(); const unsigned long long n_ = MFN_CYCLES(); tk_sc += n_ - tka; tka = n_; }
  }
  if (p.timeline) {
    MFN_WAIT_LGKM0();
    const unsigned long long n_ = MFN_CYCLES();
    if (tid == 0) {
      unsigned long long *b_ = p.timeline + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4;
      b_[0] = tk_geo; b_[1] = tk_mma; b_[2] = tk_sc; b_[3] = n_ - tk0;
    }
  }
  if (!p.req_x) return;
  __syncthreads();
  // ---- merge the four strips' windows (strip w covers block-window rows 2w .. 2w+7) and flush once
  const int bwy0 = wy0 - 2 * wave;  // the block window's first row (uniform across the block)
  for (int e = tid; e < 32 * (WR + 6) * WC; e += 256) {
    const int cl = e / ((WR + 6) * WC), rem = e - cl * ((WR + 6) * WC);
    const int R = rem / WC, cc = rem - R * WC;
    float v = 0.f;
    MFN_UNROLL
    for (int w2 = 0; w2 < 4; ++w2) {
      const int rr = R - 2 * w2;
      if (rr >= 0 && rr < WR) v += lds[(size_t)w2 * 32 * PL + (size_t)cl * PL + rr * WC + cc];
    }
    const int yy = bwy0 + R, xx = wx0 + cc;
    if (v != 0.f && cb + cl < p.Cin && yy >= 0 && yy < H && xx >= 0 && xx < W)
      atomicAdd(p.gx + ((size_t)n * p.Cin + cb + cl) * plane + (size_t)yy * W + xx, v);
  }
}


// ---- geometry record of a pixel whose nine taps share one offset (used by the lane = pixel kernels of dc_backward.h) ------------
constexpr int DCS_GS = 40;
// words of a pixel's geometry record
enum { DCS_AY = 0, DCS_BY = 3, DCS_AX = 6, DCS_BX = 9, DCS_M9 = 12, DCS_FH0 = 21, DCS_FH1 = 24, DCS_FW0 = 27, DCS_FW1 = 30,
       DCS_CELL = 36, DCS_LY0 = 37, DCS_LX0 = 38, DCS_FL = 39 };
// one axis of the record: forward weights of tap row i on lines i / i+1 (dc_axis), validity, and the weights of
// deformable_col2im_coord's absolute-coordinate interpolation; *ok is cleared when the floors leave the regular pattern
__device__ __forceinline__ void dcs_axis(float off, int in0, int dim, int lo0, float *g, int a_, int b_, int f0_, int f1_,
                                         float vf[3], bool &ok) {
  MFN_UNROLL
  for (int i = 0; i < 3; ++i) {
    bool v; int lo, hi; float l;
    dc_axis(off, in0, i, dim, v, lo, hi, l);
    if (v && lo != lo0 + i) ok = false;
    g[a_ + i] = 1.f - l;
    g[b_ + i] = l;
    vf[i] = v ? 1.f : 0.f;
    float a = v ? (float)(in0 + i) + off : 0.f;
    int cl = (int)a;
    if (cl >= dim - 1) { cl = dim - 1; a = (float)cl; }
    if (v && cl != min(max(in0 + lo0 + i, 0), dim - 1)) ok = false;
    g[f0_ + i] = (float)(cl + 1) - a;
    g[f1_ + i] = a - (float)cl;
  }
}

// bias gradient: block per filter, sum over (n, pixel)
__global__ __launch_bounds__(256) void dc_bwd_bias_kernel(DcBwdParams p) {
  MFN_DYN_SHARED(float, red);
  const int o = blockIdx.x;
  const size_t oplane = (size_t)p.Ho * p.Wo;
  float s = 0.f;
  for (size_t q = threadIdx.x; q < (size_t)p.N * oplane; q += 256) {
    const size_t n = q / oplane, pix = q - n * oplane;
    s += p.gout[(n * p.Cout + o) * oplane + pix];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int st = 128; st >= 1; st >>= 1) {
    if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) p.gbias[o] = (p.req_bias == 3 ? p.gbias[o] : 0.f) + red[0];
}

}  // namespace mfn
