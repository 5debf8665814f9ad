// upsample.h -- Upsample(factor) of the flow / mask between pyramid levels.
//
// Replaces the Gluon block at /root/reference/network/MaskFlownet.py:35-62 (call sites :228-229, :246-247, :264-265,
// :282-283, :308, :311): edge-pad one row/column at the bottom/right, Deconvolution with the separable triangle
// kernel k[a] = 1 - |f-1-a|/f (kernel 2f-1, stride f, pad f-1), drop the last row/column.  Semantics as
// oracle/mfn_ref_body.inc upsample.  Written as a gather: output (oy,ox) has at most 2x2 contributing inputs,
//   iy0 = oy / f with weight 1 - r/f  and  iy0 + 1 (clamped to H-1: the edge pad) with weight r/f,   r = oy % f,
// accumulated in the oracle's raster order with separately rounded multiplies and adds (fp contraction off), so
// results are bit-identical.
// HBM-bound: 4*N*C*H*W*(1 + f*f) bytes; one thread writes 4 adjacent outputs with one 16-byte store.
#pragma once
#include "../mfn_rt.h"

namespace mfn {

struct UpsampleParams {
  const float *x;
  float *out;
  int N, C, H, W, f;
  int st_policy;  // cache policy of the output stores (mfn_store4_stream)
};

// hipcc contracts a*b+c into FMAs by default (and HIP's __fmul_rn / __fadd_rn are plain operators, not barriers);
// the oracle (gcc -ffp-contract=off) rounds every product and sum separately, so contraction is switched off here.
__device__ __forceinline__ float upsample_tri(int cc, int a) {
#pragma clang fp contract(off)
  return 1.f - fabsf((float)(cc - a)) / (float)(cc + 1);  // the reference's _kernel2d entry, same rounding
}

template <int VEC>
__global__ __launch_bounds__(256) void upsample_kernel(UpsampleParams p) {
#pragma clang fp contract(off)
  const int f = p.f, cc = f - 1;
  const int Hout = p.H * f, Wout = p.W * f;
  const int wv = Wout / VEC;
  const size_t total = (size_t)p.N * p.C * Hout * wv;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int xv = (int)(idx % wv);
  const int oy = (int)((idx / wv) % Hout);
  const size_t nc = idx / ((size_t)wv * Hout);
  const float *src = p.x + nc * (size_t)p.H * p.W;
  const int iy0 = oy / f, ry = oy - iy0 * f;
  const float ka0 = upsample_tri(cc, ry + f - 1);               // row iy0
  const float ka1 = ry ? upsample_tri(cc, ry - 1) : 0.f;        // row iy0 + 1 (absent when r == 0)
  const float *r0 = src + (size_t)iy0 * p.W;
  const float *r1 = src + (size_t)min(iy0 + 1, p.H - 1) * p.W;
  float o[VEC];
  MFN_UNROLL
  for (int k = 0; k < VEC; ++k) {
    const int ox = xv * VEC + k;
    const int ix0 = ox / f, rx = ox - ix0 * f;
    const int ix1 = min(ix0 + 1, p.W - 1);
    const float kb0 = upsample_tri(cc, rx + f - 1);
    const float kb1 = rx ? upsample_tri(cc, rx - 1) : 0.f;
    // dst += v * (ka * kb) in raster order of the contributing inputs; no contraction into FMAs
    float acc = r0[ix0] * (ka0 * kb0);
    if (rx) acc = acc + r0[ix1] * (ka0 * kb1);
    if (ry) {
      acc = acc + r1[ix0] * (ka1 * kb0);
      if (rx) acc = acc + r1[ix1] * (ka1 * kb1);
    }
    o[k] = acc;
  }
  float *dst = p.out + nc * (size_t)Hout * Wout + (size_t)oy * Wout + (size_t)xv * VEC;
  if (VEC == 4) {
    mfn_store4_stream(dst, o[0], o[1 % VEC], o[2 % VEC], o[3 % VEC], p.st_policy);
  } else {
    MFN_UNROLL
    for (int k = 0; k < VEC; ++k) dst[k] = o[k];
  }
}

inline int upsample_launch(UpsampleParams p, hipStream_t stream) {
  const int Wout = p.W * p.f;
  const bool vec4 = (Wout % 4 == 0) && (((uintptr_t)p.out) % 16 == 0);
  const size_t total = (size_t)p.N * p.C * p.H * p.f * (vec4 ? Wout / 4 : Wout);
  if (total == 0) return 0;
  const dim3 grid((unsigned)((total + 255) / 256));
  if (vec4) return launch("upsample_v4", upsample_kernel<4>, grid, dim3(256), 0, stream, p);
  return launch("upsample_v1", upsample_kernel<1>, grid, dim3(256), 0, stream, p);
}

}  // namespace mfn
