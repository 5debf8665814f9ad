// correlation_gramk.h -- the cost volume of the COARSE pyramid levels (6 x 8 ... 24 x 32 pixels, 96 ... 196 channels) as a banded
// Gram matrix on the bf16 matrix cores, one launch, the channels of a block spread over its waves.
//
// Replaces MXNet Correlation at /root/reference/network/MaskFlownet.py:193-195 (md=4) and :440-441 (md=2) where the image
// is small and the channel count large (levels 6 / 5 / 4 of the reference's pyramid); semantics as oracle/mfn_ref_body.inc
// correlation_fwd.  Formulation and operand split as correlation_gram.h (M = an 8 x 2 pixel block of f1, N = a 16-pixel row
// segment of f2 starting 4 columns left of it, K = 32 channels per v_mfma_f32_16x16x32_bf16, every fp32 operand as three
// bf16 terms, six products, fp32 accumulate).  What differs is the shape of the work:
//   * these launches are LATENCY-bound (profiles/r03_corr_pmc.md; 7.1-7.8 us each in the pass where a node of the graph costs
//     2.5 us and the data moves in < 1 us): the fp32-FMA kernels walk the channels in a loop of dependent LDS-DMA stages or
//     global loads.  Here a block = (f1 block, RG of the 2*md+2 f2 rows it meets) and wave w = channels 32w .. 32w+31: every
//     wave issues ALL its tiles at once (2 + 2*RG LDS-DMA instructions), waits once, converts, runs RG chains of 6 matrix
//     instructions, and the waves' partial 16 x 16 results meet in LDS in wave order (deterministic) -- ONE memory round
//     trip, one barrier.
//   * channels past C (196 = 6 x 32 + 4) are lanes of the LDS-DMA with an out-of-range offset: zeros.
//   * 1/C is applied to the sum as the fp32 kernels do (a division where C is no power of two).
#pragma once
#include "../mfn_rt.h"
#include "correlation_gram.h"

namespace mfn {

struct CorrGramKParams {
  const float *f1;
  const float *f2;
  float *out;
  int N, C, H, W;
  int strips, brows, groups;   // ceil(W / 8), ceil(H / 2), (2*md+2) / RG
  size_t out_nstride;
  int store_policy;
  int leaky;
  int xcd_swizzle;
  float inv_sumelems, sumelems;
  int exact_div;
};

template <int D, int NW, int RG>
__global__ __launch_bounds__(NW * 64) void corr_gramk_kernel(CorrGramKParams p) {
  constexpr int MD = (D - 1) / 2;
  constexpr int S = 2 * MD + 2;          // f2 rows an 8 x 2 block meets
  constexpr int XOFF = 4;
  constexpr int SLOT_F = 512;            // floats per raw tile: 32 channels x 16 px
  constexpr unsigned INVALID = 0xFFFFFF00u;
  static_assert(S % RG == 0 && MD >= 1 && MD <= 4, "row groups divide the window");
  MFN_DYN_SHARED(float, lds_all);
  const int lane = threadIdx.x & 63;
  const int wave = MFN_UNIFORM(threadIdx.x >> 6);
  float *ring = lds_all + (size_t)wave * (1 + RG) * SLOT_F;
  float *red = lds_all + (size_t)NW * (1 + RG) * SLOT_F;   // [RG][NW][64 lanes] x 4 floats
  int bid = blockIdx.x;
  if (p.xcd_swizzle) bid = (int)mfn_xcd_remap((unsigned)bid, gridDim.x);
  const int grp = bid % p.groups;
  int rest = bid / p.groups;
  const int xs = rest % p.strips;
  rest /= p.strips;
  const int yb = rest % p.brows;
  const int n = rest / p.brows;
  const int H = p.H, W = p.W, C = p.C;
  const int plane = H * W;
  const int x0 = xs * 8, y0 = yb * 2;
  const float *f1n = p.f1 + (size_t)n * C * plane;
  const float *f2n = p.f2 + (size_t)n * C * plane;
  float *outn = p.out + (size_t)n * p.out_nstride;
  const unsigned img_bytes = (unsigned)(C * plane) * 4u;
  const int c0 = wave * 32;

  // ---- all tiles of this wave in one go -----------------------------------------------------------------------------------
  {
    const int xq = x0 - XOFF + 4 * (lane & 3);
    const bool okN = xq >= 0 && xq < W;
    const int xm = x0 + 4 * (lane & 1);
    const int rowm = (lane >> 1) & 1;
    const bool okM = xm < W && y0 + rowm < H;
    unsigned voffN[2], voffM[2];
    MFN_UNROLL
    for (int j = 0; j < 2; ++j) {
      const int c = c0 + (lane >> 2) + 16 * j;
      voffN[j] = (okN && c < C) ? (unsigned)(c * plane + xq) * 4u : INVALID;
      voffM[j] = (okM && c < C) ? (unsigned)(c * plane + rowm * W + xm) * 4u : INVALID;
    }
    const unsigned soffM = (unsigned)(y0 * W) * 4u;
    mfn_dma16_row(f1n, img_bytes, soffM, true, ring, voffM[0]);
    mfn_dma16_row(f1n, img_bytes, soffM, true, ring + 256, voffM[1]);
    MFN_UNROLL
    for (int i = 0; i < RG; ++i) {
      const int row = y0 - MD + grp * RG + i;
      const bool in = row >= 0 && row < H;           // MXNet's pad_size border: zeros
      const unsigned soff = (unsigned)((in ? row : 0) * W) * 4u;
      float *slot = ring + (1 + i) * SLOT_F;
      mfn_dma16_row(f2n, img_bytes, soff, in, slot, voffN[0]);
      mfn_dma16_row(f2n, img_bytes, soff, in, slot + 256, voffN[1]);
    }
  }
  // store geometry while the tiles fly (as corr_gram_wave's wave-private form): after the row shifts lane (g, n0) owns
  // displacement dx = n0 - XOFF - 4h for the pixels x0+4h .. +3 of block row yy (h = g&1, yy = g>>1)
  unsigned voffS_up, voffS_lo, voffS;
  {
    const int g = lane >> 4, n0 = lane & 15, h = g & 1, yy = g >> 1;
    const int dxi = n0 - XOFF - 4 * h + MD;
    const bool ok = dxi >= 0 && dxi < D && x0 + 4 * h < W;
    const unsigned v = ok ? (unsigned)(((1 - yy) * D + dxi) * plane + yy * W + x0 + 4 * h) * 4u : INVALID;
    const bool r1 = y0 + 1 < H;
    voffS_up = yy == 0 ? v : INVALID;
    voffS_lo = (yy == 1 && r1) ? v : INVALID;
    voffS = (yy == 0 || r1) ? v : INVALID;
  }
  const mfn_rsrc_t rs = mfn_make_rsrc(outn + ((long long)y0 * W - (long long)D * plane), 0x80000000u);
  const unsigned dplane4 = (unsigned)(D * plane) * 4u;
  const int rdoff = (lane >> 4) * 16 + (lane & 15);

  MFN_WAIT_VM(0);
  auto operand = [&](int slot, GramOp &o) {
    const float *su = ring + slot * SLOT_F + rdoff;
    float raw[8];
    MFN_UNROLL
    for (int j = 0; j < 8; ++j) raw[j] = su[64 * j];
    GramWords w;
    MFN_UNROLL
    for (int q = 0; q < 4; ++q) gram_split_pair<3>(raw[2 * q], raw[2 * q + 1], w, q);
    gram_words_to_op(w, o);
  };
  GramOp Mo;
  operand(0, Mo);
  f32x4 acc[RG];
  MFN_UNROLL
  for (int i = 0; i < RG; ++i) {
    GramOp No;
    operand(1 + i, No);
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    // smallest terms first: l*h, h*l, m*m, m*h, h*m, h*h
    a = MFN_MFMA_16x16x32_BF16(Mo.l, No.h, a);
    a = MFN_MFMA_16x16x32_BF16(Mo.h, No.l, a);
    a = MFN_MFMA_16x16x32_BF16(Mo.m, No.m, a);
    a = MFN_MFMA_16x16x32_BF16(Mo.m, No.h, a);
    a = MFN_MFMA_16x16x32_BF16(Mo.h, No.m, a);
    a = MFN_MFMA_16x16x32_BF16(Mo.h, No.h, a);
    acc[i] = a;
  }
  // the waves' partial tiles meet in LDS
  if (NW > 1) {
    MFN_UNROLL
    for (int i = 0; i < RG; ++i) *reinterpret_cast<f32x4 *>(red + ((size_t)(i * NW + wave) * 64 + lane) * 4) = acc[i];
    MFN_LDS_BARRIER();
  }
  MFN_UNROLL
  for (int i = 0; i < RG; ++i) {
    if (NW > 1 && (i % NW) != wave) continue;     // tile i is finished by wave i % NW
    f32x4 sum;
    if (NW > 1) {
      sum = *reinterpret_cast<const f32x4 *>(red + ((size_t)(i * NW) * 64 + lane) * 4);
      MFN_UNROLL
      for (int w = 1; w < NW; ++w) {
        const f32x4 t = *reinterpret_cast<const f32x4 *>(red + ((size_t)(i * NW + w) * 64 + lane) * 4);
        sum[0] += t[0]; sum[1] += t[1]; sum[2] += t[2]; sum[3] += t[3];
      }
    } else sum = acc[i];
    MFN_UNROLL
    for (int q = 0; q < 4; ++q) sum[q] = p.exact_div ? sum[q] / p.sumelems : sum[q] * p.inv_sumelems;
    // de-skew: register q of lane n holds (x = 4h+q, dx = n-XOFF-4h-q); lane n0 collects dx0 = n0-XOFF-4h from lanes n0+q
    f32x4 v;
    v[0] = sum[0];
    v[1] = mfn_dpp_row_shl<1>(sum[1], sum[1]);
    v[2] = mfn_dpp_row_shl<2>(sum[2], sum[2]);
    v[3] = mfn_dpp_row_shl<3>(sum[3], sum[3]);
    if (p.leaky) {
      MFN_UNROLL
      for (int q = 0; q < 4; ++q) v[q] = mfn_leaky01(v[q]);
    }
    // f2 row e = grp*RG + i of the window: block row 0 is displacement row e (exists while e < D), block row 1 is e-1
    const int e = grp * RG + i;
    const unsigned vo = e == 0 ? voffS_up : (e == 2 * MD + 1 ? voffS_lo : voffS);
    mfn_bstore4_so(rs, vo, (unsigned)e * dplane4, v, p.store_policy);
  }
}

inline bool corr_variant_gramk(int v) { return v == 44 || v == 45; }
// f2 rows per block: 44 = two (5 / 3 blocks per f1 block for md = 4 / 2), 45 = half the window.  (All of the window in one block,
// measured as variant 46 and removed: 13.1 / 5.2 / 10.2 us at levels 6 / 5 / 4 against 3.3 / 3.4 / 6.4.)
inline int corr_gramk_rg(int variant, int D) {
  if (D == 9) return variant == 44 ? 2 : 5;
  return variant == 44 ? 2 : 3;
}
inline size_t corr_gramk_lds_bytes(int nw, int rg) { return ((size_t)nw * (1 + rg) * 512 + (nw > 1 ? (size_t)rg * nw * 256 : 0)) * sizeof(float); }
inline bool corr_gramk_shape_ok(int C, int W, int variant, int D) {
  const int nw = cdiv(C, 32);
  return W % 8 == 0 && nw >= 1 && nw <= 8 && corr_gramk_lds_bytes(nw, corr_gramk_rg(variant, D)) <= 150 * 1024;
}

template <int D, int NW, int RG>
inline int corr_gramk_launch(CorrGramKParams p, hipStream_t stream) {
  p.strips = cdiv(p.W, 8);
  p.brows = cdiv(p.H, 2);
  p.groups = (D + 1) / RG;
  const long nblk = (long)p.N * p.brows * p.strips * p.groups;
  if (nblk <= 0) return 0;
  return launch("corr_gramk", corr_gramk_kernel<D, NW, RG>, dim3((unsigned)nblk), dim3(NW * 64), corr_gramk_lds_bytes(NW, RG), stream, p);
}
template <int D, int RG>
inline int corr_gramk_nw(const CorrGramKParams &p, hipStream_t s) {
  switch (cdiv(p.C, 32)) {
    case 1: return corr_gramk_launch<D, 1, RG>(p, s);
    case 2: return corr_gramk_launch<D, 2, RG>(p, s);
    case 3: return corr_gramk_launch<D, 3, RG>(p, s);
    case 4: return corr_gramk_launch<D, 4, RG>(p, s);
    case 5: return corr_gramk_launch<D, 5, RG>(p, s);
    case 6: return corr_gramk_launch<D, 6, RG>(p, s);
    case 7: return corr_gramk_launch<D, 7, RG>(p, s);
    default: return corr_gramk_launch<D, 8, RG>(p, s);
  }
}
template <int D>
inline int corr_gramk_variant(const CorrGramKParams &p, int variant, hipStream_t s) {
  constexpr int S = D + 1;
  if (variant == 44) return corr_gramk_nw<D, 2>(p, s);
  return corr_gramk_nw<D, S / 2>(p, s);
}

// ---- corr_gramr_kernel: the same band with the operands loaded STRAIGHT INTO REGISTERS ------------------------------------------
// A lane of the B operand holds 8 channels of ONE pixel: with the K index <-> channel map {4j + lane/16} those are 8
// buffer_load_dword whose 16 lanes of a row cover a 64-byte run of one channel plane -- the same bytes per instruction as the
// LDS-DMA form's 16-byte lanes, without the LDS round trip, the M0 traffic and the hand-counted wait, and without the 2 KB of
// LDS per tile that bound how many tiles (and waves) a CU could hold.  That makes a taller work item affordable:
// an item = T blocks of 8 x 2 pixels stacked vertically (2T rows) x one of G groups of the 2T + 2md f2 rows they meet, every
// f2 row converted once for up to md+1 blocks -- the coarse kernel above converts each f2 row for every block it meets, which
// is what made it lose at levels 4 / 3.  The group is a template parameter of the body (one copy of the code per group), so the
// chain <-> block relation is static and the matrix instructions of a row's chains interleave.
template <int D, int T, int G>
struct GramRSched {
  static constexpr int MD = (D - 1) / 2;
  static constexpr int S = 2 * T + 2 * MD;
  static constexpr int RG = S / G;
  static_assert(S % G == 0, "row groups divide the window");
  static constexpr bool active(int s, int t) { return s - 2 * t >= 0 && s - 2 * t <= 2 * MD + 1; }
  static constexpr int qidx(int grp, int i, int t) {   // position of chain (row i of the group, block t) among the group's chains
    int q = 0;
    for (int ii = 0; ii < RG; ++ii)
      for (int tt = 0; tt < T; ++tt) {
        if (ii == i && tt == t) return active(grp * RG + ii, tt) ? q : -1;
        if (active(grp * RG + ii, tt)) ++q;
      }
    return -1;
  }
  static constexpr int nq(int grp) {
    int q = 0;
    for (int ii = 0; ii < RG; ++ii)
      for (int tt = 0; tt < T; ++tt) if (active(grp * RG + ii, tt)) ++q;
    return q;
  }
  static constexpr int nq_max() { int m = 0; for (int g = 0; g < G; ++g) m = nq(g) > m ? nq(g) : m; return m; }
};

template <int D, int NW, int T, int G, int GRP>
__device__ __forceinline__ void corr_gramr_body(const CorrGramKParams &p, float *red, int lane, int wave, int n, int y0, int x0) {
  using SC = GramRSched<D, T, G>;
  constexpr int MD = SC::MD, RG = SC::RG;
  constexpr int XOFF = 4;
  constexpr unsigned INVALID = 0xFFFFFF00u;
  const int H = p.H, W = p.W, C = p.C;
  const int plane = H * W;
  const float *f1n = p.f1 + (size_t)n * C * plane;
  const float *f2n = p.f2 + (size_t)n * C * plane;
  float *outn = p.out + (size_t)n * p.out_nstride;
  const unsigned img_bytes = (unsigned)(C * plane) * 4u;
  const int g = lane >> 4, idx = lane & 15;
  const int c0 = wave * 32 + g;                        // this lane's channels: c0 + 4j (past C: past the image, zeros)
  const unsigned chs = (unsigned)(4 * plane) * 4u;

  // ---- every operand of the item at once ------------------------------------------------------------------------------------
  float rawA[T][8], rawB[RG][8];
  {
    const int mr = idx >> 3, mc = idx & 7;
    const unsigned voffA = x0 + mc < W ? (unsigned)(c0 * plane + mr * W + x0 + mc) * 4u : INVALID;
    MFN_UNROLL
    for (int t = 0; t < T; ++t) {
      const int row = y0 + 2 * t;
      const bool in = row < H;
      const unsigned vo = row + mr < H ? voffA : INVALID;
      MFN_UNROLL
      for (int j = 0; j < 8; ++j) rawA[t][j] = mfn_bload1_row(f1n, img_bytes, (unsigned)(row * W) * 4u + j * chs, in, vo);
    }
    const int xq = x0 - XOFF + idx;
    const unsigned voffB = (xq >= 0 && xq < W) ? (unsigned)(c0 * plane + xq) * 4u : INVALID;
    MFN_UNROLL
    for (int i = 0; i < RG; ++i) {
      const int row = y0 - MD + GRP * RG + i;
      const bool in = row >= 0 && row < H;           // MXNet's pad_size border: zeros
      MFN_UNROLL
      for (int j = 0; j < 8; ++j) rawB[i][j] = mfn_bload1_row(f2n, img_bytes, (unsigned)((in ? row : 0) * W) * 4u + j * chs, in, voffB);
    }
  }
  // store geometry (corr_gram_wave's): after the row shifts lane (g, n0) owns displacement dx = n0 - XOFF - 4h for the pixels
  // x0+4h .. +3 of block row yy (h = g&1, yy = g>>1); relative to plane -D of output row y0, the chain adds (e D plane + 2t W)
  const int h = g & 1, yy = g >> 1;
  unsigned voffS;
  {
    const int dxi = idx - XOFF - 4 * h + MD;
    const bool ok = dxi >= 0 && dxi < D && x0 + 4 * h < W;
    voffS = ok ? (unsigned)(((1 - yy) * D + dxi) * plane + yy * W + x0 + 4 * h) * 4u : INVALID;
  }
  const mfn_rsrc_t rs = mfn_make_rsrc(outn + ((long long)y0 * W - (long long)D * plane), 0x80000000u);

  auto to_op = [&](const float *raw, GramOp &o) {
    GramWords w;
    MFN_UNROLL
    for (int q = 0; q < 4; ++q) gram_split_pair<3>(raw[2 * q], raw[2 * q + 1], w, q);
    gram_words_to_op(w, o);
  };
  auto finish = [&](f32x4 sum, int t, int e) {
    MFN_UNROLL
    for (int q = 0; q < 4; ++q) sum[q] = p.exact_div ? sum[q] / p.sumelems : sum[q] * p.inv_sumelems;
    f32x4 v;
    v[0] = sum[0];
    v[1] = mfn_dpp_row_shl<1>(sum[1], sum[1]);
    v[2] = mfn_dpp_row_shl<2>(sum[2], sum[2]);
    v[3] = mfn_dpp_row_shl<3>(sum[3], sum[3]);
    if (p.leaky) {
      MFN_UNROLL
      for (int q = 0; q < 4; ++q) v[q] = mfn_leaky01(v[q]);
    }
    // block row 0 is displacement row e (exists while e < D), block row 1 is e-1 (exists from e = 1); rows past the image: nothing
    const bool rowok = yy == 0 ? (e < D && y0 + 2 * t < H) : (e >= 1 && y0 + 2 * t + 1 < H);
    mfn_bstore4_so(rs, rowok ? voffS : INVALID, (unsigned)(e * D * plane + 2 * t * W) * 4u, v, p.store_policy);
  };
  GramOp Mo[T];
  MFN_UNROLL
  for (int t = 0; t < T; ++t) to_op(rawA[t], Mo[t]);
  mfn_static_for<RG>([&](auto i_c) __attribute__((always_inline)) {
    constexpr int i = decltype(i_c)::value;
    constexpr int s = GRP * RG + i;
    GramOp No;
    to_op(rawB[i], No);
    f32x4 acc[T];
    MFN_UNROLL
    for (int t = 0; t < T; ++t) { acc[t][0] = 0.f; acc[t][1] = 0.f; acc[t][2] = 0.f; acc[t][3] = 0.f; }
    // the chains of this row interleaved product by product, smallest terms first: l*h, h*l, m*m, m*h, h*m, h*h
    MFN_UNROLL
    for (int k = 0; k < 6; ++k) {
      MFN_UNROLL
      for (int t = 0; t < T; ++t) {
        if (SC::active(s, t)) {
          const mfn_bf16x8 &a = k == 0 ? Mo[t].l : (k == 2 || k == 3 ? Mo[t].m : Mo[t].h);
          const mfn_bf16x8 &b = k == 1 ? No.l : (k == 2 || k == 4 ? No.m : No.h);
          acc[t] = MFN_MFMA_16x16x32_BF16(a, b, acc[t]);
        }
      }
    }
    mfn_static_for<T>([&](auto t_c) __attribute__((always_inline)) {
      constexpr int t = decltype(t_c)::value;
      if constexpr (SC::active(s, t)) {
        constexpr int q = SC::qidx(GRP, i, t);
        if (NW > 1) *reinterpret_cast<f32x4 *>(red + ((size_t)(q * NW + wave) * 64 + lane) * 4) = acc[t];
        else finish(acc[t], t, s - 2 * t);
      }
    });
  });
  if (NW > 1) {
    MFN_LDS_BARRIER();
    mfn_static_for<RG>([&](auto i_c) __attribute__((always_inline)) {
      constexpr int i = decltype(i_c)::value;
      constexpr int s = GRP * RG + i;
      mfn_static_for<T>([&](auto t_c) __attribute__((always_inline)) {
        constexpr int t = decltype(t_c)::value;
        if constexpr (SC::active(s, t)) {
          constexpr int q = SC::qidx(GRP, i, t);
          if (q % NW == wave) {                      // chain q is finished by wave q % NW: the waves' partial tiles in wave order
            f32x4 sum = *reinterpret_cast<const f32x4 *>(red + ((size_t)(q * NW) * 64 + lane) * 4);
            MFN_UNROLL
            for (int w = 1; w < NW; ++w) {
              const f32x4 v = *reinterpret_cast<const f32x4 *>(red + ((size_t)(q * NW + w) * 64 + lane) * 4);
              sum[0] += v[0]; sum[1] += v[1]; sum[2] += v[2]; sum[3] += v[3];
            }
            finish(sum, t, s - 2 * t);
          }
        }
      });
    });
  }
}

template <int D, int NW, int T, int G>
__global__ __launch_bounds__(NW * 64) void corr_gramr_kernel(CorrGramKParams p) {
  MFN_DYN_SHARED(float, red);
  const int lane = threadIdx.x & 63;
  const int wave = MFN_UNIFORM(threadIdx.x >> 6);
  int bid = blockIdx.x;
  if (p.xcd_swizzle) bid = (int)mfn_xcd_remap((unsigned)bid, gridDim.x);
  const int grp = bid % G;
  int rest = bid / G;
  const int xs = rest % p.strips;
  rest /= p.strips;
  const int seg = rest % p.brows;
  const int n = rest / p.brows;
  mfn_static_for<G>([&](auto g_c) __attribute__((always_inline)) {
    constexpr int GRP = decltype(g_c)::value;
    if (grp == GRP) corr_gramr_body<D, NW, T, G, GRP>(p, red, lane, wave, n, seg * 2 * T, xs * 8);
  });
}

// corr.variant 46 .. 50: (T blocks per item, G row groups) = md 4: (1,5) (1,2) (2,2) (3,2) (2,3); md 2: (1,3) (1,2) (2,2) (3,2) (2,4)
inline bool corr_variant_gramr(int v) { return v >= 46 && v <= 50; }
inline bool corr_gramr_shape_ok(int C, int W) {
  const int nw = cdiv(C, 32);
  return W % 8 == 0 && (nw == 2 || nw == 3 || nw == 4 || nw == 7);
}
template <int D, int NW, int T, int G>
inline int corr_gramr_launch(CorrGramKParams p, hipStream_t stream) {
  p.strips = cdiv(p.W, 8);
  p.brows = cdiv(p.H, 2 * T);
  p.groups = G;
  const long nblk = (long)p.N * p.brows * p.strips * G;
  if (nblk <= 0) return 0;
  const size_t lds = NW > 1 ? (size_t)GramRSched<D, T, G>::nq_max() * NW * 256 * sizeof(float) : 0;
  return launch("corr_gramr", corr_gramr_kernel<D, NW, T, G>, dim3((unsigned)nblk), dim3(NW * 64), lds, stream, p);
}
template <int D, int T, int G>
inline int corr_gramr_nw(const CorrGramKParams &p, hipStream_t s) {
  switch (cdiv(p.C, 32)) {
    case 2: return corr_gramr_launch<D, 2, T, G>(p, s);
    case 3: return corr_gramr_launch<D, 3, T, G>(p, s);
    case 4: return corr_gramr_launch<D, 4, T, G>(p, s);
    default: return corr_gramr_launch<D, 7, T, G>(p, s);
  }
}
template <int D>
inline int corr_gramr_variant(const CorrGramKParams &p, int variant, hipStream_t s) {
  if (D == 9) {
    switch (variant) {
      case 46: return corr_gramr_nw<9, 1, 5>(p, s);
      case 47: return corr_gramr_nw<9, 1, 2>(p, s);
      case 48: return corr_gramr_nw<9, 2, 2>(p, s);
      case 49: return corr_gramr_nw<9, 3, 2>(p, s);
      default: return corr_gramr_nw<9, 2, 3>(p, s);
    }
  }
  switch (variant) {
    case 46: return corr_gramr_nw<5, 1, 3>(p, s);
    case 47: return corr_gramr_nw<5, 1, 2>(p, s);
    case 48: return corr_gramr_nw<5, 2, 2>(p, s);
    case 49: return corr_gramr_nw<5, 3, 2>(p, s);
    default: return corr_gramr_nw<5, 2, 4>(p, s);
  }
}

}  // namespace mfn
