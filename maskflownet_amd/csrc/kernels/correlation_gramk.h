// correlation_gramk.h -- the cost volume of the COARSE pyramid levels (6 x 8 ... 28 x 64 pixels, 96 ... 196 channels) as a banded
// Gram matrix on the bf16 matrix cores: one launch, the channels of a block spread over its waves, ONE memory round trip.
//
// Replaces MXNet Correlation at /root/reference/network/MaskFlownet.py:193-195 (md=4) and :440-441 (md=2) where the image
// is small and the channel count large (levels 6 / 5 / 4 of the reference's pyramid); semantics as oracle/mfn_ref_body.inc
// correlation_fwd.  Formulation and operand split as correlation_gram.h: M = an 8 x 2 pixel block of f1, N = a 16-pixel row
// segment of f2 starting 4 columns left of it, K = 32 channels per v_mfma_f32_16x16x32_bf16, every fp32 operand as three
// bf16 terms, six products, fp32 accumulate.  What differs is the shape of the work:
//   * these launches are LATENCY-bound (profiles/r03_corr_pmc.md: 7.1-7.8 us each inside the pass, where a node of the graph costs
//     2.5 us and the data moves in < 1 us): the fp32-FMA kernels walk the channels in a loop of dependent LDS-DMA stages
//     (corr_dma_kernel) or global loads (corr_direct_kernel).  Here a block = (f1 block, RG of the 2 md + 2 f2 rows it meets)
//     and wave w = channels 32w .. 32w+31: every wave issues ALL its operand loads at once, converts, runs RG chains of 6
//     matrix instructions, and the waves' partial 16 x 16 tiles meet in LDS in wave order (deterministic): one round trip,
//     one barrier.  Level 6 / 5 / 4 of 384x512 at batch 8: 7.85 / 7.18 / 7.02 -> 3.4 / 3.4 / 6.0 us (profiles/r04_corr_gramk.txt).
//   * operands go STRAIGHT INTO REGISTERS: a lane of the B operand holds 8 channels of one pixel, with the K index <-> channel map
//     {4j + lane/16} that is 8 buffer_load_dword whose 16 lanes of a row cover a 64-byte run of one channel plane -- the same
//     bytes per instruction as an LDS-DMA of 16-byte lanes, without the LDS round trip, M0 and hand-counted waits (the first
//     form of this kernel staged 2 KB tiles through wave-private LDS like corr_gram_kernel: the same time at levels 6 / 5,
//     6.3 against 6.0 us at level 4).  The row and the channel step sit in the instruction's soffset; zero padding, the ragged
//     last strip and channels past C (196 = 6 x 32 + 4) are the descriptor's range check.
//   * 1/C is applied to the sum as the fp32 kernels do (a division where C is no power of two).
// Measured and not kept (same file, profiles/r04_corr_gramk.txt): the whole window in one block (13.1 / 5.2 / 10.2 us at levels
// 6 / 5 / 4); items of 2 or 3 stacked f1 blocks that convert every f2 row once for all the blocks it meets (level 3: 10.1-11.2 us
// against the FMA kernel's 9.65, slower than single blocks at levels 6 / 5 / 4) -- level 3 keeps corr_dma_kernel.
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
  constexpr unsigned INVALID = 0xFFFFFF00u;
  static_assert(S % RG == 0 && MD >= 1 && MD <= 4, "row groups divide the window");
  MFN_DYN_SHARED(float, red);            // [RG][NW][64 lanes] x 4 floats
  const int lane = threadIdx.x & 63;
  const int wave = MFN_UNIFORM(threadIdx.x >> 6);
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
  const int g = lane >> 4, idx = lane & 15;
  const int c0 = wave * 32 + g;                        // this lane's channels: c0 + 4j (past C = past the image: zeros)
  const unsigned chs = (unsigned)(4 * plane) * 4u;

  // ---- every operand of the block at once --------------------------------------------------------------------------------------
  float rawA[8], rawB[RG][8];
  {
    const int mr = idx >> 3, mc = idx & 7;
    const unsigned voffA = (x0 + mc < W && y0 + mr < H) ? (unsigned)(c0 * plane + mr * W + x0 + mc) * 4u : INVALID;
    MFN_UNROLL
    for (int j = 0; j < 8; ++j) rawA[j] = mfn_bload1_row(f1n, img_bytes, (unsigned)(y0 * W) * 4u + j * chs, true, voffA);
    const int xq = x0 - XOFF + idx;
    const unsigned voffB = (xq >= 0 && xq < W) ? (unsigned)(c0 * plane + xq) * 4u : INVALID;
    MFN_UNROLL
    for (int i = 0; i < RG; ++i) {
      const int row = y0 - MD + grp * RG + i;
      const bool in = row >= 0 && row < H;           // MXNet's pad_size border: zeros
      MFN_UNROLL
      for (int j = 0; j < 8; ++j) rawB[i][j] = mfn_bload1_row(f2n, img_bytes, (unsigned)((in ? row : 0) * W) * 4u + j * chs, in, voffB);
    }
  }
  // store geometry while the loads fly (as corr_gram_wave's wave-private form): after the row shifts lane (g, n0) owns
  // displacement dx = n0 - XOFF - 4h for the pixels x0+4h .. +3 of block row yy (h = g&1, yy = g>>1)
  const int h = g & 1, yy = g >> 1;
  unsigned voffS;
  {
    const int dxi = idx - XOFF - 4 * h + MD;
    const bool ok = dxi >= 0 && dxi < D && x0 + 4 * h < W && y0 + yy < H;
    voffS = ok ? (unsigned)(((1 - yy) * D + dxi) * plane + yy * W + x0 + 4 * h) * 4u : INVALID;
  }
  const mfn_rsrc_t rs = mfn_make_rsrc(outn + ((long long)y0 * W - (long long)D * plane), 0x80000000u);
  const unsigned dplane4 = (unsigned)(D * plane) * 4u;

  auto to_op = [&](const float *raw, GramOp &o) {
    GramWords w;
    MFN_UNROLL
    for (int q = 0; q < 4; ++q) gram_split_pair<3>(raw[2 * q], raw[2 * q + 1], w, q);
    gram_words_to_op(w, o);
  };
  GramOp Mo;
  to_op(rawA, Mo);
  f32x4 acc[RG];
  MFN_UNROLL
  for (int i = 0; i < RG; ++i) {
    GramOp No;
    to_op(rawB[i], No);
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
    // f2 row e = grp*RG + i of the window: block row 0 is displacement row e (exists while e < D), block row 1 is e-1 (from e = 1)
    const int e = grp * RG + i;
    const bool rowok = yy == 0 ? e < D : e >= 1;
    mfn_bstore4_so(rs, rowok ? voffS : INVALID, (unsigned)e * dplane4, v, p.store_policy);
  }
}

inline bool corr_variant_gramk(int v) { return v == 44 || v == 45; }
// f2 rows per block: 44 = two (5 / 3 blocks per f1 block for md = 4 / 2), 45 = half the window
inline int corr_gramk_rg(int variant, int D) {
  if (D == 9) return variant == 44 ? 2 : 5;
  return variant == 44 ? 2 : 3;
}
inline bool corr_gramk_shape_ok(int C, int W) { return W % 8 == 0 && C >= 1 && C <= 256; }

template <int D, int NW, int RG>
inline int corr_gramk_launch(CorrGramKParams p, hipStream_t stream) {
  p.strips = cdiv(p.W, 8);
  p.brows = cdiv(p.H, 2);
  p.groups = (D + 1) / RG;
  const long nblk = (long)p.N * p.brows * p.strips * p.groups;
  if (nblk <= 0) return 0;
  const size_t lds = NW > 1 ? (size_t)RG * NW * 256 * sizeof(float) : 0;
  return launch("corr_gramk", corr_gramk_kernel<D, NW, RG>, dim3((unsigned)nblk), dim3(NW * 64), lds, stream, p);
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

}  // namespace mfn
