// correlation_direct.h -- cost volume of the tiniest pyramid levels in ONE launch without LDS staging.
//
// Same operator as correlation.h (MXNet Correlation, kernel_size=1, stride1=stride2=1, pad=md, multiply;
// /root/reference/network/MaskFlownet.py:193-195); semantics as oracle/mfn_ref_body.inc correlation_fwd.
// (Rounds 1 / 2 also had a row-band kernel for levels 5 / 4 here; round 3 replaced it with corr_dma_kernel's
// displacement-row split, correlation.h.)
#pragma once
#include "../mfn_rt.h"
#include "correlation.h"

// ---- corr_direct_kernel: the tiniest levels (level 6: 6x8 pixels, 196 channels) ---------------------------------
// Everything fits in L1/L2 and the job is launch- and latency-bound, so no LDS staging at all: a workgroup owns
// (image, displacement row, band of output rows) and up to 1024 threads = (4-pixel quad, dx) x S channel slices; a
// thread walks its channels with one 16-byte load of f1 and four (clamped, masked) loads of f2, the slices meet in
// LDS in index order (deterministic), slice 0 normalises and stores.  One launch, no workspace.
namespace mfn {
// a / b for 0 <= a < 2^22 without the ~40-instruction integer division sequence: the float quotient is off by at
// most one, one correction step makes it exact
__device__ __forceinline__ void corr_divmod(int a, int b, float inv_b, int &q, int &r) {
  q = (int)((float)a * inv_b);
  r = a - q * b;
  if (r < 0) { --q; r += b; }
  if (r >= b) { ++q; r -= b; }
}

struct CorrDirectParams {
  const float *f1, *f2;
  float *out;
  size_t out_nstride;
  int st_policy;
  int N, C, H, W;
  int R, bands, S, Q;   // rows per band, bands per image, channel slices, outputs (quad, dx, row) per slice
  int cps;              // channels per slice
  int max_slices;       // cap on S: the serial slice sum in the epilogue grows with it
  float inv_sumelems, sumelems;
  int exact_div, leaky;
  float inv_bands, inv_rq, inv_q;  // 1/bands, 1/(R*QW), 1/QW for the division-free index decode
};

template <int D>
__global__ __launch_bounds__(1024) void corr_direct_kernel(CorrDirectParams p) {
  constexpr int MD = (D - 1) / 2;
  MFN_DYN_SHARED(float, red);  // [S-1][Q][4]
  const int tid = threadIdx.x;
  const int H = p.H, W = p.W, C = p.C, QW = W >> 2;
  const size_t plane = (size_t)H * W;
  // block -> (image, displacement row, band)
  int b = (int)mfn_xcd_remap(blockIdx.x, gridDim.x), n, rem, dyi, band;  // the 9 rows of an image on one XCD
  corr_divmod(b, D * p.bands, 1.0f / (float)(D * p.bands), n, rem);
  corr_divmod(rem, p.bands, p.inv_bands, dyi, band);
  // thread -> (slice, dx, row, quad)
  const int s = tid / p.Q, q = tid - s * p.Q;
  const bool active = s < p.S;
  int dxi, rq, r, qx;
  corr_divmod(active ? q : 0, p.R * QW, p.inv_rq, dxi, rq);
  corr_divmod(rq, QW, p.inv_q, r, qx);
  const int y = band * p.R + r, x = 4 * qx;
  const int y2 = y + dyi - MD;
  const bool row_ok = active && y < H && y2 >= 0 && y2 < H;
  const int yc = min(y, H - 1), y2c = min(max(y2, 0), H - 1);
  const int xs = x + dxi - MD;  // first f2 column of the quad
  int xi[4];
  bool xo[4];
  MFN_UNROLL
  for (int e = 0; e < 4; ++e) { xo[e] = xs + e >= 0 && xs + e < W; xi[e] = min(max(xs + e, 0), W - 1); }
  const int c0 = s * p.cps, c1 = min(C, c0 + p.cps);
  const float *a = p.f1 + ((size_t)n * C + c0) * plane + (size_t)yc * W + x;
  const float *bp = p.f2 + ((size_t)n * C + c0) * plane + (size_t)y2c * W;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (row_ok) {
    MFN_NOUNROLL
    for (int c = c0; c < c1; c += 4) {  // four channels per trip: 20 unconditional loads in flight (clamped addresses)
      float4 av[4];
      float bv[4][4];
      MFN_UNROLL
      for (int u = 0; u < 4; ++u) {
        const size_t o = (size_t)min(u, c1 - 1 - c) * plane;
        av[u] = *reinterpret_cast<const float4 *>(a + o);
        MFN_UNROLL
        for (int e = 0; e < 4; ++e) bv[u][e] = bp[o + xi[e]];
      }
      MFN_UNROLL
      for (int u = 0; u < 4; ++u) {
        const bool cok = c + u < c1;
        acc[0] = fmaf(cok ? av[u].x : 0.f, xo[0] ? bv[u][0] : 0.f, acc[0]);
        acc[1] = fmaf(cok ? av[u].y : 0.f, xo[1] ? bv[u][1] : 0.f, acc[1]);
        acc[2] = fmaf(cok ? av[u].z : 0.f, xo[2] ? bv[u][2] : 0.f, acc[2]);
        acc[3] = fmaf(cok ? av[u].w : 0.f, xo[3] ? bv[u][3] : 0.f, acc[3]);
      }
      a += 4 * plane;
      bp += 4 * plane;
    }
  }
  if (active && s > 0) {
    float *dst = red + ((size_t)(s - 1) * p.Q + q) * 4;
    dst[0] = acc[0]; dst[1] = acc[1]; dst[2] = acc[2]; dst[3] = acc[3];
  }
  __syncthreads();
  if (!active || s != 0 || y >= H) return;
  for (int s2 = 1; s2 < p.S; ++s2) {
    const float *src = red + ((size_t)(s2 - 1) * p.Q + q) * 4;
    acc[0] += src[0]; acc[1] += src[1]; acc[2] += src[2]; acc[3] += src[3];
  }
  const float slope = p.leaky ? 0.1f : 1.f;
  float v[4];
  MFN_UNROLL
  for (int e = 0; e < 4; ++e) {
    const float rr = p.exact_div ? acc[e] / p.sumelems : acc[e] * p.inv_sumelems;
    v[e] = fmaxf(rr, slope * rr);
  }
  mfn_store4_stream(p.out + (size_t)n * p.out_nstride + (size_t)(dyi * D + dxi) * plane + (size_t)y * W + x, v[0], v[1], v[2],
                    v[3], p.st_policy);
}

// plan: the band height R that gives the most channel slices within 1024 threads (latency is what counts here)
template <int D>
inline int corr_direct_launch(CorrDirectParams p, hipStream_t stream) {
  const int QW = p.W / 4;
  int bestR = 0, bestS = 0;
  for (int R = 1; R <= p.H; ++R) {
    const int Q = R * QW * D;
    if (Q > 1024) break;
    int S = 1024 / Q;
    if (S > (p.C + 3) / 4) S = (p.C + 3) / 4;
    if (S > p.max_slices) S = p.max_slices;
    // prefer many slices (short channel loops), then tall bands (fewer workgroups re-reading rows)
    if (S > bestS || (S == bestS && R > bestR && (long)p.N * D * ((p.H + R - 1) / R) >= 256)) { bestS = S; bestR = R; }
  }
  if (!bestR) return -1;
  p.R = bestR; p.S = bestS; p.Q = bestR * QW * D;
  p.bands = (p.H + p.R - 1) / p.R;
  p.cps = (((p.C + p.S - 1) / p.S) + 3) / 4 * 4;
  p.S = (p.C + p.cps - 1) / p.cps;
  p.inv_bands = 1.0f / (float)p.bands;
  p.inv_rq = 1.0f / (float)(p.R * QW);
  p.inv_q = 1.0f / (float)QW;
  const int threads = ((p.S * p.Q + 63) / 64) * 64;
  const size_t lds = (size_t)(p.S > 1 ? p.S - 1 : 1) * p.Q * 4 * sizeof(float);
  return launch("corr_direct", corr_direct_kernel<D>, dim3(p.N * D * p.bands), dim3(threads), lds, stream, p);
}
}  // namespace mfn
