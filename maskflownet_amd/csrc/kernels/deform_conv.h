// deform_conv.h -- DeformableConvolution forward for gfx950 as a fused gather + fp32-MFMA GEMM.
//
// Replaces MXNet contrib.DeformableConvolution at /root/reference/network/layer.py:117-124;
// semantics as oracle/mfn_ref_body.inc deform_conv_fwd (deformable_im2col + GEMM + bias).
//
// out[o, p] = sum_kk W[o, kk] * col[kk, p],  kk = c*9 + tap,  col = bilinear gather of x.
// The im2col buffer (9x the input, SURVEY.md Appendix B "col-buffer MB") is never written:
//   * one wave owns a 32-pixel x (32*MT)-filter output tile and a slice of the input channels;
//     v_mfma_f32_32x32x2_f32 (exact fp32, k-ordered fma chain) wants B[k][j] in lane j+32k, so
//     lane (j, half) gathers the 9 taps of channel 2*cp+half for pixel j and that register IS the
//     B operand of k-step `tap`; A comes from a [channel-pair][tap][half][filter] re-layout of the
//     weights (dc_pack_weights_kernel) so one wave reads two full 128-byte lines per k-step;
//   * the 4 waves of a block split K (channel pairs) of one pixel tile -- no barrier in the
//     main loop -- and reduce through LDS at the end (KS=4), or take 4/KS different pixel tiles;
//   * tap geometry (validity, clamp, 4 weights) is computed once per pixel and reused for all
//     channels.  When all taps of a pixel share one offset (the only way the reference calls it,
//     MaskFlownet.py:230) and the floor pattern is regular, the 9 taps read one 4x4
//     neighbourhood: 16 loads + separable interpolation instead of 36 loads (wave-uniform vote).
// dc_generic_kernel covers everything else MXNet accepts (groups, deformable groups, any
// kernel size), one thread per output element.
#pragma once
#include "../mfn_rt.h"

namespace mfn {

struct DeformParams {
  const float *x;
  const float *offset;  // (N, 2*kh*kw*dg, Ho, Wo), or NULL in shared-flow mode
  const float *flow;    // shared-flow mode: (N,2,Ho,Wo); offset of every tap = flow*scale/stride
  float flow_scale, flow_stride;
  const float *w;   // original (Cout, Cin/groups, kh, kw)   [generic kernel]
  const float *wt;  // packed [ceil(Cin/2)][kh*kw][2][CoutP]  [MFMA kernel]
  const float *bias;
  float *out;
  int N, Cin, H, W, Cout, CoutP, Ho, Wo;
  int kh, kw, sh, sw, ph, pw, dh, dw, groups, dg;
  int P;  // N*Ho*Wo
  int allow_fast;  // tuning: 0 forces the per-tap path
  int ablate;      // measurement only: 1 = no gather loads, 2 = no weight loads, 3 = no MFMA
};

// weights (Cout, Cin, T) -> wt[((cp*T + t)*2 + half)*CoutP + o], zero padded in c and o
struct PackParams { const float *w; float *wt; int Cin, Cout, CoutP, T; };
__global__ __launch_bounds__(256) void dc_pack_weights_kernel(PackParams p) {
  const int ncp = (p.Cin + 1) / 2;
  const size_t total = (size_t)ncp * p.T * 2 * p.CoutP;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int o = (int)(idx % p.CoutP);
  const int half = (int)((idx / p.CoutP) & 1);
  const int t = (int)((idx / ((size_t)2 * p.CoutP)) % p.T);
  const int cp = (int)(idx / ((size_t)2 * p.CoutP * p.T));
  const int c = 2 * cp + half;
  p.wt[idx] = (c < p.Cin && o < p.Cout) ? p.w[((size_t)o * p.Cin + c) * p.T + t] : 0.f;
}

// One tap of deformable_im2col: validity on h_im/w_im, bilinear on the (h_in,w_in)-relative map_h,
// clamp-to-last.  Returns 4 weights (0 when invalid) and the in-plane offsets of the 4 corners.
struct DcTap {
  float w1, w2, w3, w4;
  int base;  // (h_in+h_low)*W + (w_in+w_low), bit30 = dw (w_high-w_low), bit31 handled via dhW
  int dhW;   // (h_high-h_low)*W
};
__device__ __forceinline__ void dc_axis(float off, int in0, int tap_d, int dim, bool &valid, int &low, int &high,
                                        float &l) {
  const float im = (float)(in0 + tap_d) + off;  // h_in + i*dilation + offset
  valid = (im >= 0.f) && (im < (float)dim);
  float m = (float)tap_d + off;  // map_h
  const int cur = dim - in0;     // cur_height
  const float fl = floorf(m);
  int lo = (int)fminf(fmaxf(fl, -1.0e6f), 1.0e6f);
  if (lo >= cur - 1) { high = lo = cur - 1; m = (float)lo; }
  else high = lo + 1;
  l = m - (float)lo;
  low = lo;
}
__device__ __forceinline__ DcTap dc_make_tap(float off_h, float off_w, int h_in, int w_in, int i_d, int j_d, int H,
                                             int W, bool px_valid) {
  bool vh, vw;
  int hl, hh_i, wl, wh_i;
  float lh, lw;
  dc_axis(off_h, h_in, i_d, H, vh, hl, hh_i, lh);
  dc_axis(off_w, w_in, j_d, W, vw, wl, wh_i, lw);
  const bool valid = vh && vw && px_valid;
  const float hh = 1.f - lh, hw = 1.f - lw;
  DcTap t;
  t.w1 = valid ? hh * hw : 0.f;
  t.w2 = valid ? hh * lw : 0.f;
  t.w3 = valid ? lh * hw : 0.f;
  t.w4 = valid ? lh * lw : 0.f;
  t.base = valid ? ((h_in + hl) * W + (w_in + wl)) | ((wh_i - wl) << 30) : 0;
  t.dhW = valid ? (hh_i - hl) * W : 0;
  return t;
}

// per-axis descriptor of the shared-offset fast path: weights of tap row i on slots i and i+1
struct DcAxis3 { float a[3], b[3]; int idx[4]; };

template <int MT, int KS>
__global__ __launch_bounds__(256) void dc_mfma_kernel(DeformParams p) {
  constexpr int T = 9;
  constexpr int TILES_PER_BLOCK = 4 / KS;
  MFN_DYN_SHARED(float, red);  // [4 waves][MT*16][64] partial accumulators (KS > 1 only)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int half = lane >> 5, j = lane & 31;
  const int ks = wave % KS;
  const int tile = blockIdx.x * TILES_PER_BLOCK + wave / KS;
  const int m0 = blockIdx.y * (32 * MT);

  const int H = p.H, W = p.W, Ho = p.Ho, Wo = p.Wo;
  const size_t plane = (size_t)H * W;
  const size_t oplane = (size_t)Ho * Wo;
  const int plin = tile * 32 + j;
  const bool px_valid = plin < p.P;
  const int pc = px_valid ? plin : 0;
  const int n = pc / (Ho * Wo);
  const int rem = pc - n * (Ho * Wo);
  const int ho = rem / Wo, wo = rem - ho * Wo;
  const int h_in = ho * p.sh - p.ph, w_in = wo * p.sw - p.pw;

  // ---- offsets of the 9 taps ---------------------------------------------------------------------
  float offh[T], offw[T];
  bool shared = true;
  if (p.offset) {
    const float *op = p.offset + (size_t)n * 2 * T * oplane + (size_t)ho * Wo + wo;
    MFN_UNROLL
    for (int t = 0; t < T; ++t) {
      offh[t] = op[(size_t)(2 * t) * oplane];
      offw[t] = op[(size_t)(2 * t + 1) * oplane];
    }
    MFN_UNROLL
    for (int t = 1; t < T; ++t) shared = shared && (offh[t] == offh[0]) && (offw[t] == offw[0]);
  } else {
    const float *fp = p.flow + (size_t)n * 2 * oplane + (size_t)ho * Wo + wo;
    const float oh = fp[0] * p.flow_scale / p.flow_stride;       // MaskFlownet.py:230
    const float ow = fp[oplane] * p.flow_scale / p.flow_stride;
    MFN_UNROLL
    for (int t = 0; t < T; ++t) { offh[t] = oh; offw[t] = ow; }
  }

  // ---- fast-path descriptors: regular floor pattern on both axes -----------------------------------
  DcAxis3 ay, ax;
  bool regular = shared && (p.dh == 1) && (p.dw == 1) && (p.allow_fast != 0);
  {
    int lo0 = 0;
    MFN_UNROLL
    for (int i = 0; i < 3; ++i) {
      bool v; int lo, hi; float l;
      dc_axis(offh[0], h_in, i, H, v, lo, hi, l);
      v = v && px_valid;
      // unclamped floor for the regularity test
      const int ulo = (int)fminf(fmaxf(floorf((float)i + offh[0]), -1.0e6f), 1.0e6f);
      if (i == 0) lo0 = ulo; else regular = regular && (ulo == lo0 + i);
      ay.a[i] = v ? 1.f - l : 0.f;
      ay.b[i] = v ? l : 0.f;
    }
    MFN_UNROLL
    for (int m = 0; m < 4; ++m) ay.idx[m] = min(max(h_in + lo0 + m, 0), H - 1) * W;
    MFN_UNROLL
    for (int i = 0; i < 3; ++i) {
      bool v; int lo, hi; float l;
      dc_axis(offw[0], w_in, i, W, v, lo, hi, l);
      v = v && px_valid;
      const int ulo = (int)fminf(fmaxf(floorf((float)i + offw[0]), -1.0e6f), 1.0e6f);
      if (i == 0) lo0 = ulo; else regular = regular && (ulo == lo0 + i);
      ax.a[i] = v ? 1.f - l : 0.f;
      ax.b[i] = v ? l : 0.f;
    }
    MFN_UNROLL
    for (int m = 0; m < 4; ++m) ax.idx[m] = min(max(w_in + lo0 + m, 0), W - 1);
  }
  const bool fast = __all(regular || !px_valid) != 0;  // wave-uniform

  f32x16 acc[MT];
  MFN_UNROLL
  for (int mt = 0; mt < MT; ++mt)
    MFN_UNROLL
    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

  const int ncp = (p.Cin + 1) / 2;
  const float *xn = p.x + (size_t)n * p.Cin * plane;
  const float *wt_lane = p.wt + (size_t)half * p.CoutP + m0 + j;
  const size_t wt_step = (size_t)2 * p.CoutP;  // between taps

  if (fast) {
    for (int cp = ks; cp < ncp; cp += KS) {
      const int c = 2 * cp + half;
      float colv[T];
      if (c < p.Cin) {
        const float *pl = xn + (size_t)c * plane;
        float v[4][4];
        MFN_UNROLL
        for (int m = 0; m < 4; ++m)
          MFN_UNROLL
          for (int q = 0; q < 4; ++q) v[m][q] = p.ablate == 1 ? 1.f : pl[ay.idx[m] + ax.idx[q]];
        float tr[4][3];
        MFN_UNROLL
        for (int m = 0; m < 4; ++m)
          MFN_UNROLL
          for (int q = 0; q < 3; ++q) tr[m][q] = ax.a[q] * v[m][q] + ax.b[q] * v[m][q + 1];
        MFN_UNROLL
        for (int i = 0; i < 3; ++i)
          MFN_UNROLL
          for (int q = 0; q < 3; ++q) colv[i * 3 + q] = ay.a[i] * tr[i][q] + ay.b[i] * tr[i + 1][q];
      } else {
        MFN_UNROLL
        for (int t = 0; t < T; ++t) colv[t] = 0.f;
      }
      const float *wp = wt_lane + (size_t)cp * T * wt_step;
      MFN_UNROLL
      for (int t = 0; t < T; ++t) {
        MFN_UNROLL
        for (int mt = 0; mt < MT; ++mt) {
          const float a = p.ablate == 2 ? 1.f : wp[t * wt_step + mt * 32];
          if (p.ablate == 3) acc[mt][t] += a * colv[t];
          else acc[mt] = MFN_MFMA_32x32x2(a, colv[t], acc[mt]);
        }
      }
    }
  } else {
    // per-tap path (arbitrary offsets; never taken by the reference network).  Tap geometry is
    // recomputed per channel pair on purpose: keeping 9 descriptors live would cost ~50 VGPRs
    // and halve the occupancy of the fast path that shares this kernel.
    for (int cp = ks; cp < ncp; cp += KS) {
      const int c = 2 * cp + half;
      const bool cvalid = c < p.Cin;
      const float *pl = xn + (size_t)(cvalid ? c : 0) * plane;
      float colv[T];
      MFN_UNROLL
      for (int t = 0; t < T; ++t) {
        float oh = offh[t], ow = offw[t];
        MFN_OPAQUE(oh);  // stops LICM from hoisting the geometry out of the channel loop
        const DcTap tp = dc_make_tap(oh, ow, h_in, w_in, (t / 3) * p.dh, (t % 3) * p.dw, H, W, px_valid && cvalid);
        const int b = tp.base & 0x3FFFFFFF, dwi = (tp.base >> 30) & 1;
        const float v1 = pl[b], v2 = pl[b + dwi], v3 = pl[b + tp.dhW], v4 = pl[b + tp.dhW + dwi];
        colv[t] = tp.w1 * v1 + tp.w2 * v2 + tp.w3 * v3 + tp.w4 * v4;
      }
      const float *wp = wt_lane + (size_t)cp * T * wt_step;
      MFN_UNROLL
      for (int t = 0; t < T; ++t) {
        MFN_UNROLL
        for (int mt = 0; mt < MT; ++mt) acc[mt] = MFN_MFMA_32x32x2(wp[t * wt_step + mt * 32], colv[t], acc[mt]);
      }
    }
  }

  // ---- split-K reduction across the waves of the block ---------------------------------------------
  if (KS > 1) {
    if (ks != 0) {
      MFN_UNROLL
      for (int mt = 0; mt < MT; ++mt)
        MFN_UNROLL
        for (int r = 0; r < 16; ++r) red[(size_t)((wave * MT + mt) * 16 + r) * 64 + lane] = acc[mt][r];
    }
    __syncthreads();
    if (ks != 0) return;
    for (int k = 1; k < KS; ++k) {
      MFN_UNROLL
      for (int mt = 0; mt < MT; ++mt)
        MFN_UNROLL
        for (int r = 0; r < 16; ++r) acc[mt][r] += red[(size_t)(((wave + k) * MT + mt) * 16 + r) * 64 + lane];
    }
  }

  // ---- epilogue: + bias, store.  D reg r of lane (j,half): filter row (r&3)+8*(r>>2)+4*half, pixel j
  if (!px_valid) return;
  float *on = p.out + (size_t)n * p.Cout * oplane + (size_t)ho * Wo + wo;
  MFN_UNROLL
  for (int mt = 0; mt < MT; ++mt)
    MFN_UNROLL
    for (int r = 0; r < 16; ++r) {
      const int o = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (o < p.Cout) on[(size_t)o * oplane] = acc[mt][r] + (p.bias ? p.bias[o] : 0.f);
    }
}

template <int MT, int KS>
inline int dc_mfma_launch(const DeformParams &p, hipStream_t stream, const char *name) {
  const int tiles = cdiv(p.P, 32);
  const int bx = cdiv(tiles, 4 / KS);
  const int by = p.CoutP / (32 * MT);
  if (bx <= 0 || by <= 0) return 0;
  const size_t lds = KS > 1 ? (size_t)4 * MT * 16 * 64 * sizeof(float) : 0;
  return launch(name, dc_mfma_kernel<MT, KS>, dim3(bx, by), dim3(256), lds, stream, p);
}

inline int dc_pack_launch(PackParams pp, hipStream_t stream) {
  const size_t total = (size_t)((pp.Cin + 1) / 2) * pp.T * 2 * pp.CoutP;
  return launch("dc_pack_weights", dc_pack_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                stream, pp);
}

// ---- generic fallback: groups / deformable groups / any kernel size -----------------------------------
__global__ __launch_bounds__(256) void dc_generic_kernel(DeformParams p) {
  const size_t oplane = (size_t)p.Ho * p.Wo;
  const size_t total = (size_t)p.N * p.Cout * oplane;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int wo = (int)(idx % p.Wo), ho = (int)((idx / p.Wo) % p.Ho);
  const int o = (int)((idx / oplane) % p.Cout);
  const int n = (int)(idx / (oplane * p.Cout));
  const int T = p.kh * p.kw;
  const int cpg = p.Cin / p.groups, opg = p.Cout / p.groups, cpd = p.Cin / p.dg;
  const int g = o / opg;
  const int h_in = ho * p.sh - p.ph, w_in = wo * p.sw - p.pw;
  const size_t plane = (size_t)p.H * p.W;
  float s = 0.f;
  for (int cl = 0; cl < cpg; ++cl) {
    const int c = g * cpg + cl;
    const int dgi = c / cpd;
    const float *pl = p.x + ((size_t)n * p.Cin + c) * plane;
    for (int t = 0; t < T; ++t) {
      float oh, ow;
      if (p.offset) {
        const float *op = p.offset + ((size_t)n * p.dg + dgi) * 2 * T * oplane + (size_t)ho * p.Wo + wo;
        oh = op[(size_t)(2 * t) * oplane];
        ow = op[(size_t)(2 * t + 1) * oplane];
      } else {
        const float *fp = p.flow + (size_t)n * 2 * oplane + (size_t)ho * p.Wo + wo;
        oh = fp[0] * p.flow_scale / p.flow_stride;
        ow = fp[oplane] * p.flow_scale / p.flow_stride;
      }
      const DcTap tp = dc_make_tap(oh, ow, h_in, w_in, (t / p.kw) * p.dh, (t % p.kw) * p.dw, p.H, p.W, true);
      const int b = tp.base & 0x3FFFFFFF, dwi = (tp.base >> 30) & 1;
      const float val = tp.w1 * pl[b] + tp.w2 * pl[b + dwi] + tp.w3 * pl[b + tp.dhW] + tp.w4 * pl[b + tp.dhW + dwi];
      s = fmaf(p.w[((size_t)o * cpg + cl) * T + t], val, s);
    }
  }
  p.out[idx] = s + (p.bias ? p.bias[o] : 0.f);
}

inline int dc_generic_launch(const DeformParams &p, hipStream_t stream) {
  const size_t total = (size_t)p.N * p.Cout * p.Ho * p.Wo;
  if (!total) return 0;
  return launch("dc_generic", dc_generic_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, p);
}

// ---- offset builder (MaskFlownet.py:230) ----------------------------------------------------------------
struct OffsetsParams { const float *flow; float *offset; int N, H, W, taps; float scale, stride; };
__global__ __launch_bounds__(256) void offsets_from_flow_kernel(OffsetsParams p) {
  const size_t plane = (size_t)p.H * p.W;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;  // over N*2*plane
  if (idx >= (size_t)p.N * 2 * plane) return;
  const size_t n = idx / (2 * plane), r = idx - n * 2 * plane;
  const size_t t = r / plane, pix = r - t * plane;
  const float v = p.flow[idx] * p.scale / p.stride;
  float *o = p.offset + n * 2 * p.taps * plane + t * plane + pix;
  for (int k = 0; k < p.taps; ++k) o[(size_t)2 * k * plane] = v;
}
inline int offsets_from_flow_launch(OffsetsParams p, hipStream_t stream) {
  const size_t total = (size_t)p.N * 2 * p.H * p.W;
  if (!total) return 0;
  return launch("offsets_from_flow", offsets_from_flow_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                stream, p);
}

}  // namespace mfn
