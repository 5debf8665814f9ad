// conv.h -- Convolution / Deconvolution forward for gfx950 as an implicit GEMM on fp32 MFMA.
//
// the pyramid, decoder and context convolutions of MaskFlownet_S
// (/root/reference/network/MaskFlownet.py:79-163: nn.Conv2D 3x3 with stride 1|2 and dilation 1..16 + LeakyReLU(0.1),
// nn.Conv2DTranspose 4x4 stride 2 pad 1 for upfeat*).  Semantics: MXNet Convolution / Deconvolution
// (oracle/mfn_ref_body.inc conv2d_fwd / conv2d_transpose_fwd).
//
//   out[o, p] = bias[o] + sum_{c,t} W[o, c, t] * x[c, src_t(p)]       (zero outside the image)
//
// Same GEMM skeleton as the deformable convolution (deform_conv.h): a wave owns 32 output pixels (a 4x8 patch) x
// 32*MT filters, K = (channel pair, tap); v_mfma_f32_32x32x2_f32 takes B[k][j] from lane j + 32k, so lane (j, half)
// loads the T taps of channel 2*cp + half for pixel j -- plain dword loads at per-lane offsets computed once, the
// next pair's loads in flight while the current pair's T*MT MFMAs issue -- and A streams from the packed weights
// (dc_pack_weights_kernel's [M-group][pair][tap][half][filter] layout) through a double-buffered LDS-DMA stage.
// PT = 4: the block's four waves are four pixel tiles sharing the weight stage; PT = 1: four K slices of one
// pixel tile, added through LDS (coarse levels: few pixels, many channels).
// conv_generic_kernel covers every other parameter set (groups, other kernel sizes, narrow images).
#pragma once
#include "../mfn_rt.h"
#include "deform_conv.h"

namespace mfn {

// words of a channel pair's weights of one 32-filter tile in the bf16 x 3 form of conv_mfma_kernel<.., MMA = 1>:
// [term hi|mid|lo][channel of the pair][filter][taps 0..7] bf16 + [channel][filter] fp32 for tap 8 (832; 576 in fp32)
constexpr int DC_PAIR_W_BF16 = 3 * 2 * 32 * 8 / 2 + 2 * 32;


struct ConvParams {
  const float *x;
  const float *w;     // original layout [generic kernel]: (Cout, Cin/g, kh, kw), transposed: (Cin, Cout/g, kh, kw)
  const float *wt;    // packed [mgroup][ncp_pad][T][2][32*MT]
  const float *bias;
  float *out;
  int N, Cin, H, W, Cout, Ho, Wo;
  int kh, kw, sh, sw, ph, pw, dh, dw, groups;
  int transposed;
  int tiles_x, tiles_y, ntiles;
  float inv_tpi, inv_tiles_x;
  int ncp_pad, cps_per_slice, mgroups;
  size_t x_nstride;     // elements between consecutive images of `x` (Cin*H*W when dense; larger = the channel suffix
                        // buf[:, c0:] of a concat buffer, the input of the next densely connected layer)
  size_t out_nstride;   // elements between consecutive images of `out` (Cout*Ho*Wo when dense; larger = a channel slice
                        // of a concat buffer, x = concat(conv(x), x) MaskFlownet.py:219)
  int shuffle2;         // 1: the launch is a 4x4 / stride-2 / pad-1 transposed convolution run as a 3x3 convolution on the
                        // INPUT grid with 4*Cout pseudo-filters o' = 4*o + 2*py + px (one per output parity; five of a
                        // parity's nine taps carry zero weights): filter o' of grid pixel (y, x) is out[o][2y+py][2x+px].
                        // Cout / Ho / Wo / bias describe the pseudo problem: Cout = 4 * real filters, Ho x Wo = H x W.
  int leaky;            // fused LeakyReLU(0.1) (the reference's conv() = Conv2D + activate)
  int st_policy, xcd;
};

// source coordinate of tap (i, jj) for output pixel (ho, wo) along one axis
template <bool TRANS>
__device__ __forceinline__ bool conv_src(int o, int tap, int stride, int pad, int dil, int dim, int &src) {
  if (!TRANS) {
    src = o * stride - pad + tap * dil;
    return src >= 0 && src < dim;
  }
  const int num = o + pad - tap * dil;   // o = src*stride - pad + tap*dil
  if (num < 0) { src = 0; return false; }
  src = num / stride;
  return (num - src * stride == 0) && src < dim;
}

// channel pairs per weight chunk: the two stage buffers of a block (KS K-slices x KC pairs x T taps x 2 x 32*MT floats
// each) stay within 72 KB; 0 = this (MT, PT) combination does not fit
constexpr int conv_kc(int mt, int pt, int t) {
  const int unit = (4 / pt) * t * 32 * mt;
  return 4 * unit <= 4608 ? 4 : (2 * unit <= 4608 ? 2 : 0);
}

// MMA = 1 (conv.mma, 3x3 only): the bf16 x 3 operand split of deform_conv.h (DcGeom<.., MMA = 1>) -- a pair's weights are
// [filter tile][term][channel of the pair][filter][taps 0..7] bf16 + [channel][filter] fp32 for tap 8 (832 words per tile)
constexpr int conv_kc_mma(int mt, int pt) {
  const int stage = (4 / pt) * DC_PAIR_W_BF16 * mt;   // words of one pair for all K slices of the block
  return 4 * stage <= 9216 && mt <= 2 ? 4 : (2 * stage <= 9216 ? 2 : 0);
}

// ROW3 (3x3, column dilation 1, pad_w <= 1): the three taps of a kernel row are three adjacent floats -- one
// dword-aligned global_load_dwordx3 per row instead of three dword gathers.  With one or two filter tiles per wave the
// kernel is bound by what the texture addresser takes per gather instruction, not by the MFMAs (conv4_2: 46 TFLOP/s).
template <int MT, int PT, int KH, int KWD, bool TRANS, bool ROW3 = false, int MMA = 0>
__global__ __launch_bounds__(256, MT <= 2 ? 3 : 2) void conv_mfma_kernel(ConvParams p) {
  static_assert(!ROW3 || (KH == 3 && KWD == 3 && !TRANS), "ROW3 is the 3x3 convolution's row gather");
  static_assert(MMA == 0 || (KH == 3 && KWD == 3 && !TRANS), "the bf16 x 3 form is built for nine taps");
  constexpr int T = KH * KWD;
  constexpr int NW = 4, NTH = 256;
  constexpr int KS = NW / PT;              // K slices inside the block
  constexpr int RL = 32 * MT;
  constexpr int KC = MMA ? conv_kc_mma(MT, PT) : conv_kc(MT, PT, T);
  static_assert(KC >= 2, "weight stage does not fit the LDS budget");
  constexpr int PAIR_W = MMA ? MT * DC_PAIR_W_BF16 : T * 2 * RL;   // words of one channel pair's weights
  constexpr int CHUNK_F = KC * PAIR_W;
  constexpr int CH4 = CHUNK_F / 4;
  constexpr int NI = (KS * CH4 + NTH - 1) / NTH;
  constexpr int STAGE_F = NI * NTH * 4;
  MFN_DYN_SHARED(float, lds);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = MFN_UNIFORM(tid >> 6);
  const int half = lane >> 5, j = lane & 31;
  const int pt = wave / KS, ks = wave % KS;
  const int bx = p.xcd ? (int)mfn_xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int tile = bx * PT + pt;
  const int mg = blockIdx.z;
  const int m0 = mg * RL;

  const int H = p.H, W = p.W, Ho = p.Ho, Wo = p.Wo;
  const int plane = H * W;
  const size_t oplane = (size_t)Ho * Wo;

  // tile -> (image, tile row, tile column); 4x8-pixel tiles never span two images
  int n, ty, tx;
  {
    const int tpi = p.tiles_y * p.tiles_x;
    const int tl = min(tile, p.ntiles - 1);
    auto divmod = [](int a, int b, float inv_b, int &q, int &r) {
      q = (int)((float)a * inv_b);
      r = a - q * b;
      if (r < 0) { --q; r += b; }
      if (r >= b) { ++q; r -= b; }
    };
    int rt;
    divmod(tl, tpi, p.inv_tpi, n, rt);
    divmod(rt, p.tiles_x, p.inv_tiles_x, ty, tx);
  }
  n = MFN_UNIFORM(n);
  const int tile_ho0 = ty * 4, tile_wo0 = tx * 8;
  const int ho_ = tile_ho0 + (j >> 3), wo_ = tile_wo0 + (j & 7);
  const bool px_valid = tile < p.ntiles && ho_ < Ho && wo_ < Wo;
  const int ho = min(ho_, Ho - 1), wo = min(wo_, Wo - 1);

  // ---- weight staging plan (as dc_lds_kernel) -------------------------------------------------------------
  const size_t mg_floats = (size_t)p.ncp_pad * PAIR_W;
  const mfn_rsrc_t wrsrc = mfn_make_rsrc(p.wt + (size_t)mg * mg_floats, (unsigned)(mg_floats * 4));
  unsigned voff[NI];
  MFN_UNROLL
  for (int i = 0; i < NI; ++i) {
    const int it = (i * NW + wave) * 64 + lane;
    const int k = it / CH4, idx = it - k * CH4;
    voff[i] = k < KS ? (unsigned)(((size_t)k * p.cps_per_slice * PAIR_W + (size_t)idx * 4) * 4) : 0xFFFFFF00u;
  }
  auto issue = [&](int ch) {
    float *buf = lds + (ch & 1) * STAGE_F;
    const unsigned soff = (unsigned)((size_t)ch * CHUNK_F * 4);
    MFN_UNROLL
    for (int i = 0; i < NI; ++i) mfn_dma16_so(wrsrc, buf + (i * NW + wave) * 256, voff[i], soff);
  };
  issue(0);

  // ---- tap geometry: per-lane element offset of every tap, relative to channel 2*cp of image n, + validity ----------
  unsigned off[ROW3 ? 3 : T];   // ROW3: one offset per kernel row (its leftmost loaded column)
  bool val[T];   // loop-invariant lane masks (SGPR pairs): one v_cndmask per tap
  bool mL = false, mR = false;  // ROW3: the row's three loaded floats start one column right / left of tap 0
  if (ROW3) {
    const int sx0 = wo * p.sw - p.pw;                 // column of tap 0
    const int cs = min(max(sx0, 0), W - 3);           // first loaded column
    mL = sx0 - cs == -1;                               // left image edge: loaded [0,1,2] are taps 1, 2 and one beyond
    mR = sx0 - cs == 1;                                // right edge: loaded [W-3..W-1] are one before, taps 0 and 1
    MFN_UNROLL
    for (int i = 0; i < 3; ++i) {
      int sy;
      const bool vy = conv_src<false>(ho, i, p.sh, p.ph, p.dh, H, sy);
      off[i] = (unsigned)((vy ? sy : 0) * W + cs + half * plane);
      MFN_UNROLL
      for (int q = 0; q < 3; ++q) val[i * 3 + q] = vy && px_valid && sx0 + q >= 0 && sx0 + q < W;
    }
  } else {
    MFN_UNROLL
    for (int i = 0; i < KH; ++i) {
      int sy;
      const bool vy = conv_src<TRANS>(ho, i, p.sh, p.ph, p.dh, H, sy);
      MFN_UNROLL
      for (int q = 0; q < KWD; ++q) {
        int sx;
        const bool vx = conv_src<TRANS>(wo, q, p.sw, p.pw, p.dw, W, sx);
        const bool v = vy && vx && px_valid;
        off[(i * KWD + q) % (ROW3 ? 3 : T)] = (unsigned)((v ? sy * W + sx : 0) + half * plane);
        val[i * KWD + q] = v;
      }
    }
  }

  f32x16 acc[MT];
  MFN_UNROLL
  for (int mt = 0; mt < MT; ++mt)
    MFN_UNROLL
    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

  const float *xn = p.x + (size_t)n * p.x_nstride;            // uniform
  const int cp_base = ks * p.cps_per_slice;
  const int nchunks = p.cps_per_slice / KC;
  const int ncp = (p.Cin + 1) / 2;
  const int npairs = MFN_UNIFORM(max(0, min(p.cps_per_slice, ncp - cp_base)));   // pairs of this slice that exist

  // loads of one channel pair: lane (j, half) reads channel 2*cp + half through a uniform base + its 32-bit offset.
  // The RAW values go to the operand buffer; validity is applied where they are used (tap_value), pairs later: a select
  // next to the load makes hipcc wait for the load right there, before the MFMAs it is meant to overlap with.
  // An odd Cin's last pair has no second channel: those lanes read the first one and the value is dropped at use (their
  // packed weights are zero as well, but the value could be non-finite).
  auto load_pair = [&](int cp, float (&v)[T]) {
    const float *base = xn + (size_t)(2 * cp) * plane;      // uniform
    const unsigned adj = (2 * cp + 1 < p.Cin) ? 0u : (unsigned)(half * plane);   // uniform choice
    if (ROW3) {
      MFN_UNROLL
      for (int i = 0; i < 3; ++i) {
        // the row's three taps are picked from the three loaded floats HERE, on the load's result: picked at use
        // (`mR ? v[3i+1] : v[3i]`) hipcc turned the selects of array elements into dynamically indexed loads and kept the
        // operand buffers in scratch (80-160 bytes per lane, profiles/r03_kernel_resources.txt)
        const f3u r = mfn_load3u(base + (off[i] - adj));
        v[i * 3 + 0] = mR ? r.y : r.x;
        v[i * 3 + 1] = mL ? r.x : (mR ? r.z : r.y);
        v[i * 3 + 2] = mL ? r.y : r.z;
      }
    } else {
      MFN_UNROLL
      for (int t = 0; t < T; ++t) v[t] = base[off[t] - adj];
    }
  };
  // tap t of the pair in operand buffer v; lone: the pair is an odd Cin's last one (upper half has no channel)
  auto tap_value = [&](const float (&v)[T], int t, bool lone) -> float {
    return (val[t] && !(lone && half)) ? v[t] : 0.f;
  };

  // Operand pipeline: the loads of pair k + PD are issued before the MFMAs of pair k.  With one or two filter tiles per
  // wave a pair's MFMAs last 0.25 - 0.5 us, less than a global load under load: three pairs ahead there (PD = 3),
  // one pair ahead with three or four filter tiles.  Every iteration issues exactly one pair's loads (clamped to the
  // slice's last pair), so the counted waits below are exact.
  // MMA: a pair's matrix work is 2-3x shorter, three pairs ahead for every tiling (with two-pair chunks the operand buffer of
  // pair k is then (two chunks' worth) k % 4: the chunk loop below runs two chunks per trip)
  constexpr int PD = MMA ? 3 : ((MT <= 2 && KC == 4) ? 3 : 1);
  constexpr int NB = PD + 1;
  constexpr int LOADS = ROW3 ? 3 : T;
  static_assert(KC % NB == 0 || NB % KC == 0, "pair k lives in operand buffer k % NB across chunk boundaries");
  float vb[NB][T];
  const int cp_last = min(cp_base + max(npairs - 1, 0), max(ncp - 1, 0));
  MFN_UNROLL
  for (int d = 0; d < PD; ++d) load_pair(min(cp_base + d, cp_last), vb[d]);
  // A operands: the packed weights hold a lane's MT filter values of a (tap, half) next to each other
  // ([pair][tap][half][lane j][mt]): one 16- / 8-byte LDS read per tap.  The reads of pair k+1 are issued before the MFMAs
  // of pair k (second register set) -- except across a chunk boundary, where the other stage buffer is only valid after
  // the barrier.
  float areg[2][T][MT];
  auto load_a = [&](const float *ap, float (&a)[T][MT]) {
    MFN_UNROLL
    for (int t = 0; t < T; ++t) {
      const float *q = ap + (size_t)t * 2 * RL;
      if (MT == 4) { const float4 v = *reinterpret_cast<const float4 *>(q); a[t][0] = v.x; a[t][1] = v.y; a[t][2] = v.z; a[t][3 % MT] = v.w; }
      else if (MT == 2) { const float2 v = *reinterpret_cast<const float2 *>(q); a[t][0] = v.x; a[t][1 % MT] = v.y; }
      else { MFN_UNROLL for (int mt = 0; mt < MT; ++mt) a[t][mt] = q[mt]; }
    }
  };
  auto run_chunk = [&](int ch, auto base_c) {
    constexpr int BASE = decltype(base_c)::value;   // operand buffer of the chunk's first pair
    // chunk ch's weights have landed for this wave (they are older than the PD pairs of operand loads in flight) ...
    MFN_WAIT_VM(PD * LOADS);
    MFN_WAIT_LGKM0();
    MFN_RAW_BARRIER();   // ... and for every wave; nobody reads the other buffer any more
    if (ch + 1 < nchunks) issue(ch + 1);
    const float *abuf = lds + (ch & 1) * STAGE_F + ks * CHUNK_F + (half * 32 + j) * MT;
    if (!MMA) load_a(abuf, areg[0]);
    MFN_UNROLL
    for (int kk = 0; kk < KC; ++kk) {
      const int k = ch * KC + kk;
      if (!MMA && kk + 1 < KC) load_a(abuf + (size_t)(kk + 1) * T * 2 * RL, areg[(kk + 1) & 1]);
      load_pair(min(cp_base + k + PD, cp_last), vb[(BASE + kk + PD) % NB]);
      MFN_SCHED_BARRIER();
      if (MMA) {
        if (k < npairs) {   // uniform
          const bool lone = 2 * (cp_base + k) + 1 >= p.Cin;   // uniform
          const float *pw = lds + (ch & 1) * STAGE_F + ks * CHUNK_F + (size_t)kk * PAIR_W;   // this pair's weights (uniform)
          // operand reads of every filter tile first (their latency runs under the split), then the split, then the products
          mfn_bf16x8 ah[MT], am[MT], al[MT];
          float a8[MT];
          MFN_UNROLL
          for (int mt = 0; mt < MT; ++mt) {
            const float *pm = pw + mt * DC_PAIR_W_BF16;
            ah[mt] = mfn_read_bf16x8(pm + ((0 * 2 + half) * 32 + j) * 4);
            am[mt] = mfn_read_bf16x8(pm + ((1 * 2 + half) * 32 + j) * 4);
            al[mt] = mfn_read_bf16x8(pm + ((2 * 2 + half) * 32 + j) * 4);
            a8[mt] = pm[768 + half * 32 + j];
          }
          float x8[8];
          MFN_UNROLL
          for (int t = 0; t < 8; ++t) x8[t] = tap_value(vb[(BASE + kk) % NB], t, lone);
          const float b8 = tap_value(vb[(BASE + kk) % NB], 8, lone);
          mfn_bf16x8 bh, bm, bl;
          mfn_split3x8(x8, bh, bm, bl);
          // six matrix products + tap 8 in fp32 per filter tile, the tiles interleaved: consecutive instructions are independent
          MFN_UNROLL
          for (int mt = 0; mt < MT; ++mt) acc[mt] = MFN_MFMA_32x32x16_BF16(al[mt], bh, acc[mt]);
          MFN_UNROLL
          for (int mt = 0; mt < MT; ++mt) acc[mt] = MFN_MFMA_32x32x16_BF16(ah[mt], bl, acc[mt]);
          MFN_UNROLL
          for (int mt = 0; mt < MT; ++mt) acc[mt] = MFN_MFMA_32x32x16_BF16(am[mt], bm, acc[mt]);
          MFN_UNROLL
          for (int mt = 0; mt < MT; ++mt) acc[mt] = MFN_MFMA_32x32x16_BF16(am[mt], bh, acc[mt]);
          MFN_UNROLL
          for (int mt = 0; mt < MT; ++mt) acc[mt] = MFN_MFMA_32x32x16_BF16(ah[mt], bm, acc[mt]);
          MFN_UNROLL
          for (int mt = 0; mt < MT; ++mt) acc[mt] = MFN_MFMA_32x32x16_BF16(ah[mt], bh, acc[mt]);
          MFN_UNROLL
          for (int mt = 0; mt < MT; ++mt) acc[mt] = MFN_MFMA_32x32x2(a8[mt], b8, acc[mt]);
          MFN_SCHED_GROUP(0x100, 4 * MT); MFN_SCHED_GROUP(0x002, 60); MFN_SCHED_GROUP(0x008, 7 * MT);
        }
      } else if (k < npairs) {   // uniform
        const bool lone = 2 * (cp_base + k) + 1 >= p.Cin;   // uniform
        MFN_UNROLL
        for (int t = 0; t < T; ++t) {
          const float bv = tap_value(vb[(BASE + kk) % NB], t, lone);
          MFN_UNROLL
          for (int mt = 0; mt < MT; ++mt) acc[mt] = MFN_MFMA_32x32x2(areg[kk & 1][t][mt], bv, acc[mt]);
        }
      }
      MFN_SCHED_BARRIER();
    }
  };
  if (KC % NB == 0) {
    for (int ch = 0; ch < nchunks; ++ch) run_chunk(ch, DcInt<0>{});
  } else {   // NB / KC chunks per trip (two): the operand buffer index stays a compile-time constant
    for (int ch = 0; ch < nchunks; ch += 2) {
      run_chunk(ch, DcInt<0>{});
      if (ch + 1 < nchunks) run_chunk(ch + 1, DcInt<KC % NB>{});
    }
  }

  // ---- in-block K-slice reduction through LDS ------------------------------------------------------------------
  if (KS > 1) {
    MFN_WAIT_LGKM0();
    MFN_RAW_BARRIER();
    float *red = lds;   // [ks-1][MT*16][64]
    if (ks != 0) {
      MFN_UNROLL
      for (int mt = 0; mt < MT; ++mt)
        MFN_UNROLL
        for (int r = 0; r < 16; ++r) red[(size_t)(((ks - 1) * MT + mt) * 16 + r) * 64 + lane] = acc[mt][r];
    }
    __syncthreads();
    if (ks != 0) return;
    for (int k = 1; k < KS; ++k) {
      MFN_UNROLL
      for (int mt = 0; mt < MT; ++mt)
        MFN_UNROLL
        for (int r = 0; r < 16; ++r) acc[mt][r] += red[(size_t)(((k - 1) * MT + mt) * 16 + r) * 64 + lane];
    }
    MFN_WAIT_LGKM0();  // (only this wave is left) the transposition buffer below reuses this memory
  } else {
    MFN_WAIT_LGKM0();
    MFN_RAW_BARRIER();  // weight stages are dead: their memory becomes the transposition buffers
  }

  // ---- epilogue: bias, LeakyReLU(0.1), stores.  D reg r of lane (j, half): filter (r&3)+8*(r>>2)+4*half, pixel j ----
  float *obase = p.out + (size_t)n * p.out_nstride;
  const bool vec = !p.shuffle2 && (Wo % 4 == 0) && ((((size_t)p.out) & 15) == 0) && (p.out_nstride % 4 == 0);
  if (p.shuffle2) {
    if (px_valid) {
      const size_t rplane = 4 * oplane;   // real output plane: (2 Ho) x (2 Wo)
      MFN_UNROLL
      for (int mt = 0; mt < MT; ++mt) {
        // the 16 pseudo-filters of this lane are 4 real filters (r >> 2): their bias requested up front
        float b4[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
          MFN_UNROLL
          for (int g = 0; g < 4; ++g) b4[g] = p.bias[min((m0 + mt * 32 + 8 * g + 4 * half) >> 2, (p.Cout >> 2) - 1)];
        }
        MFN_UNROLL
        for (int r = 0; r < 16; ++r) {
          const int op = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;   // pseudo-filter
          if (op < p.Cout) {
            const int o = op >> 2, py = (op >> 1) & 1, pxx = op & 1;
            float v = acc[mt][r] + b4[r >> 2];
            if (p.leaky) v = fmaxf(v, 0.1f * v);
            obase[(size_t)o * rplane + (size_t)(2 * ho + py) * (2 * Wo) + 2 * wo + pxx] = v;
          }
        }
      }
    }
  } else if (vec) {
    // transpose the 32x32 tile through LDS so that a lane holds 4 adjacent pixels of one filter: 16-byte stores
    constexpr int TS = 40;
    float *tr = lds + (KS > 1 ? 0 : pt) * (32 * TS);
    const int quad = lane & 7, orow = lane >> 3;
    const int px0 = quad * 4;
    const int oy = tile_ho0 + (px0 >> 3), ox = tile_wo0 + (px0 & 7);
    const bool tile_ok = tile < p.ntiles;
    // this lane's 4 MT bias values, requested ahead of the transposition (a conditional load inside the store loop sits in its
    // own branch with a full wait behind it; deform_conv.h: dc_lds_kernel's epilogue)
    float bq[MT][4];
    MFN_UNROLL
    for (int mt = 0; mt < MT; ++mt)
      MFN_UNROLL
      for (int i = 0; i < 4; ++i) bq[mt][i] = 0.f;
    if (p.bias) {
      MFN_UNROLL
      for (int mt = 0; mt < MT; ++mt)
        MFN_UNROLL
        for (int i = 0; i < 4; ++i) bq[mt][i] = p.bias[min(m0 + mt * 32 + i * 8 + orow, p.Cout - 1)];
    }
    MFN_UNROLL
    for (int mt = 0; mt < MT; ++mt) {
      MFN_WAIT_LGKM0();
      MFN_UNROLL
      for (int r = 0; r < 16; ++r) tr[((r & 3) + 8 * (r >> 2) + 4 * half) * TS + j] = acc[mt][r];
      MFN_WAIT_LGKM0();
      MFN_WAVE_SYNC_EMU();
      MFN_UNROLL
      for (int i = 0; i < 4; ++i) {
        const int ol = i * 8 + orow;
        const int o = m0 + mt * 32 + ol;
        const float4 v = *reinterpret_cast<const float4 *>(tr + ol * TS + px0);
        if (tile_ok && o < p.Cout && oy < Ho && ox < Wo) {
          const float b = bq[mt][i];
          float e[4] = {v.x + b, v.y + b, v.z + b, v.w + b};
          if (p.leaky) {
            MFN_UNROLL
            for (int q = 0; q < 4; ++q) e[q] = fmaxf(e[q], 0.1f * e[q]);
          }
          mfn_store4_stream(obase + (size_t)o * oplane + (size_t)oy * Wo + ox, e[0], e[1], e[2], e[3], p.st_policy);
        }
      }
      MFN_WAVE_SYNC_EMU();
    }
  } else if (px_valid) {
    float *on = obase + (size_t)ho * Wo + wo;
    MFN_UNROLL
    for (int mt = 0; mt < MT; ++mt)
      MFN_UNROLL
      for (int r = 0; r < 16; ++r) {
        const int o = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (o < p.Cout) {
          float v = acc[mt][r] + (p.bias ? p.bias[o] : 0.f);
          if (p.leaky) v = fmaxf(v, 0.1f * v);
          on[(size_t)o * oplane] = v;
        }
      }
  }
}

template <int MT, int PT, int KH, int KWD, int MMA = 0>
inline size_t conv_lds_bytes() {
  constexpr int T = KH * KWD, RL = 32 * MT, KS = 4 / PT;
  constexpr int KC = MMA ? conv_kc_mma(MT, PT) : conv_kc(MT, PT, T);
  constexpr int CH4 = KC * (MMA ? MT * DC_PAIR_W_BF16 : T * 2 * RL) / 4;
  constexpr int NI = (KS * CH4 + 255) / 256;
  const size_t stage = (size_t)2 * NI * 256 * 16;
  const size_t red = KS > 1 ? (size_t)(KS - 1) * MT * 16 * 64 * 4 : 0;
  const size_t tr = (size_t)4 * 32 * 40 * 4;
  size_t m = stage > red ? stage : red;
  return m > tr ? m : tr;
}

template <int MT, int PT, int KH, int KWD, bool TRANS, bool ROW3 = false, int MMA = 0>
inline int conv_mfma_launch(const ConvParams &p, hipStream_t stream, const char *name) {
  const int bx = cdiv(p.ntiles, PT);
  if (bx <= 0) return 0;
  return launch(name, conv_mfma_kernel<MT, PT, KH, KWD, TRANS, ROW3, MMA>, dim3(bx, 1, p.mgroups), dim3(256),
                conv_lds_bytes<MT, PT, KH, KWD, MMA>(), stream, p);
}

// weights -> wt[mg][cp][t][half][lane j][mt] (filter mg*RL + mt*32 + j); regular: w (Cout, Cin, T), transposed: w (Cin, Cout, T)
struct ConvPackParams { const float *w; float *wt; int Cin, Cout, RL, mgroups, ncp_pad, T, transposed; };
// the weight the kernels multiply channel c's tap t with for (pseudo-)filter o; zero outside the tensors
__device__ __forceinline__ float conv_weight_at(const ConvPackParams &p, int c, int o, int t) {
  if (c >= p.Cin || o >= p.Cout) return 0.f;
  if (p.transposed == 2) {   // 4x4 / stride 2 / pad 1 transposed conv as 3x3 conv with pseudo-filters o = 4*o_real + 2*py + px:
    // output row 2y+py takes input rows y-1, y, y+1 (3x3 tap r) through kernel rows {3, 1, -} (py = 0) / {-, 2, 0} (py = 1)
    const int orl = o >> 2, py = (o >> 1) & 1, px = o & 1, r = t / 3, sx = t - 3 * r;
    const int iy = py ? (r == 1 ? 2 : (r == 2 ? 0 : -1)) : (r == 0 ? 3 : (r == 1 ? 1 : -1));
    const int ix = px ? (sx == 1 ? 2 : (sx == 2 ? 0 : -1)) : (sx == 0 ? 3 : (sx == 1 ? 1 : -1));
    return (iy >= 0 && ix >= 0) ? p.w[(((size_t)c * (p.Cout >> 2) + orl) * 4 + iy) * 4 + ix] : 0.f;
  }
  if (p.transposed == 3) {   // 3x3 / stride 2 / pad 1 / adj 1 transposed conv, same pseudo-filters: output row 2y+py takes input row y
    // through kernel row 1 (py = 0) or 2 (py = 1) and input row y+1 through kernel row 0 (py = 1)
    const int orl = o >> 2, py = (o >> 1) & 1, px = o & 1, r = t / 3, sx = t - 3 * r;
    const int iy = py ? (r == 1 ? 2 : (r == 2 ? 0 : -1)) : (r == 1 ? 1 : -1);
    const int ix = px ? (sx == 1 ? 2 : (sx == 2 ? 0 : -1)) : (sx == 1 ? 1 : -1);
    return (iy >= 0 && ix >= 0) ? p.w[(((size_t)c * (p.Cout >> 2) + orl) * 3 + iy) * 3 + ix] : 0.f;
  }
  return p.transposed ? p.w[((size_t)c * p.Cout + o) * p.T + t] : p.w[((size_t)o * p.Cin + c) * p.T + t];
}
__global__ __launch_bounds__(256) void conv_pack_weights_kernel(ConvPackParams p) {
  const size_t total = (size_t)p.mgroups * p.ncp_pad * p.T * 2 * p.RL;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int r = (int)(idx % p.RL);           // position inside a (tap, half) row: lane j's MT filter values are adjacent
  const int half = (int)((idx / p.RL) & 1);
  const int t = (int)((idx / ((size_t)2 * p.RL)) % p.T);
  const int cp = (int)((idx / ((size_t)2 * p.RL * p.T)) % p.ncp_pad);
  const int mg = (int)(idx / ((size_t)2 * p.RL * p.T * p.ncp_pad));
  const int mtn = p.RL / 32, jl = r / mtn, mt = r - jl * mtn;
  const int c = 2 * cp + half, o = mg * p.RL + mt * 32 + jl;
  p.wt[idx] = conv_weight_at(p, c, o, t);
}
// the same filters for the bf16 x 3 split (conv_mfma_kernel<.., MMA = 1>): one thread per (filter group, pair, filter tile, channel, filter)
__global__ __launch_bounds__(256) void conv_pack_weights_bf16_kernel(ConvPackParams p) {
  const int mtn = p.RL / 32;
  const size_t total = (size_t)p.mgroups * p.ncp_pad * mtn * 2 * 32;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int m = (int)(idx & 31), kb = (int)((idx >> 5) & 1);
  const int mt = (int)((idx >> 6) % mtn);
  const int cp = (int)(((idx >> 6) / mtn) % p.ncp_pad), mg = (int)(((idx >> 6) / mtn) / p.ncp_pad);
  const int c = 2 * cp + kb, o = mg * p.RL + mt * 32 + m;
  float x[8];
  MFN_UNROLL
  for (int e = 0; e < 8; ++e) x[e] = conv_weight_at(p, c, o, e);
  mfn_bf16x8 h, mm, l;
  mfn_split3x8(x, h, mm, l);
  float *base = p.wt + (((size_t)mg * p.ncp_pad + cp) * mtn + mt) * DC_PAIR_W_BF16;
  mfn_write_bf16x8(base + ((0 * 2 + kb) * 32 + m) * 4, h);
  mfn_write_bf16x8(base + ((1 * 2 + kb) * 32 + m) * 4, mm);
  mfn_write_bf16x8(base + ((2 * 2 + kb) * 32 + m) * 4, l);
  base[768 + kb * 32 + m] = conv_weight_at(p, c, o, 8);
}
inline int conv_pack_bf16_launch(ConvPackParams pp, hipStream_t stream) {
  const size_t total = (size_t)pp.mgroups * pp.ncp_pad * (pp.RL / 32) * 2 * 32;
  return launch("conv_pack_weights_bf16", conv_pack_weights_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, pp);
}
inline int conv_pack_launch(ConvPackParams pp, hipStream_t stream) {
  const size_t total = (size_t)pp.mgroups * pp.ncp_pad * pp.T * 2 * pp.RL;
  return launch("conv_pack_weights", conv_pack_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, pp);
}

// ---- few-filter 3x3 convolution: the prediction heads (MaskFlownet.py:131-163: pred_flow / pred_mask, 2 + 1 filters over up to
// 579 channels) --------------------------------------------------------------------------------------------------------------
// On the 32-filter MFMA tile a 2-filter head runs at 6 TFLOP/s (pred_flow2: 322 us for 228 MB of input); the layer is a
// channel reduction that is bound by reading x once.  Here a lane owns a 4 x 2 block of output pixels and ALL CO <= 4 filters:
// per channel four rows, ONE 16-byte load each -- the left / right neighbour columns come from the adjacent lanes by a DPP
// wave shift (lanes are consecutive 4-pixel groups of a row and 64 is a multiple of the groups per row or the other way
// round, so lane 0 always starts a row) -- feed 72 * CO FMAs; the filter taps are scalar loads.  The block's NW waves split
// the channels (c = wave, wave + NW, ...) and add their partial sums through LDS in wave order (deterministic).
// Levels with few lane tiles (heads3 .. heads6: 48 .. 1 tiles of 64 lanes) also split the channels over blockIdx.y; those
// blocks write raw partial sums and conv_few_reduce_kernel adds them in block order (+ bias, activation).
// stride 1, pad 1, dilation 1, W % 4 == 0, 16-byte aligned rows (the plan checks).
struct ConvFewParams {
  const float *x, *w, *bias;
  float *out;
  int N, Cin, H, W, Cout;
  size_t x_nstride, out_nstride;
  int leaky;
  int gw, gh, G;   // 4-pixel groups per row, row pairs per image, lanes in total
  int kb, cpk;     // channel blocks over blockIdx.y (coarse levels: a level's few lane tiles leave CUs idle) and channels per block;
  float *partial;  // kb > 1: raw partial sums [kb][N][CO][H][W], summed in block order by conv_few_reduce_kernel
};
template <int CO, int NW>
__global__ __launch_bounds__(NW * 64) void conv_few_kernel(ConvFewParams p) {
  MFN_DYN_SHARED(float, red);   // [NW-1][CO*8][64]
  const int lane = threadIdx.x & 63;
  const int wave = MFN_UNIFORM(threadIdx.x >> 6);
  const int gid = blockIdx.x * 64 + lane;
  const bool valid = gid < p.G;
  const int gc = valid ? gid : 0;
  const int H = p.H, W = p.W;
  const int rowi = gc / p.gw, xq = gc - rowi * p.gw;
  const int n = rowi / p.gh, y0 = 2 * (rowi - n * p.gh);
  const int x0 = 4 * xq;
  const size_t plane = (size_t)H * W;
  // the four source rows y0-1 .. y0+2: clamped index + a 0 / 1 factor (the zero padding of the convolution)
  int ro[4];
  float rk[4];
  MFN_UNROLL
  for (int r = 0; r < 4; ++r) {
    const int yy = y0 - 1 + r;
    rk[r] = (yy >= 0 && yy < H && valid) ? 1.f : 0.f;
    ro[r] = min(max(yy, 0), H - 1) * W + x0;
  }
  const float lk = xq > 0 ? 1.f : 0.f, rrk = xq + 1 < p.gw ? 1.f : 0.f;   // the row's first / last group: padding columns
  float acc[CO][2][4];
  MFN_UNROLL
  for (int o = 0; o < CO; ++o)
    MFN_UNROLL
    for (int i = 0; i < 8; ++i) acc[o][i >> 2][i & 3] = 0.f;
  const float *xn = p.x + (size_t)n * p.x_nstride;
  // two channels per iteration, the loads (rows and filter taps) of both requested before either is used: on the coarse levels
  // a block is alone on its CU and every iteration is one memory round trip (one channel per iteration: 59 us for 33 channels
  // per wave at level 6)
  auto rows_of = [&](int c, float4 (&m)[4]) {
    const float *pc = xn + (size_t)(c < p.Cin ? c : 0) * plane;
    MFN_UNROLL
    for (int r = 0; r < 4; ++r) m[r] = *reinterpret_cast<const float4 *>(pc + ro[r]);
  };
  auto accumulate = [&](int c, int c_hi, const float4 (&m)[4]) {
    const float live = c < c_hi ? 1.f : 0.f;    // the odd channel out of a pair contributes nothing
    const float *wc = p.w + (size_t)(c < p.Cin ? c : 0) * 9;   // filter o: + o * Cin * 9 (uniform: scalar loads)
    float v[4][6];
    MFN_UNROLL
    for (int r = 0; r < 4; ++r) {
      const float k = rk[r] * live;
      v[r][1] = m[r].x * k; v[r][2] = m[r].y * k; v[r][3] = m[r].z * k; v[r][4] = m[r].w * k;
      v[r][0] = mfn_dpp_wave_shr1(0.f, v[r][4]) * lk;    // the previous lane's last column
      v[r][5] = mfn_dpp_wave_shl1(0.f, v[r][1]) * rrk;   // the next lane's first column
    }
    MFN_UNROLL
    for (int o = 0; o < CO; ++o)
      MFN_UNROLL
      for (int r = 0; r < 3; ++r) {
        const float w0 = wc[(size_t)o * p.Cin * 9 + r * 3], w1 = wc[(size_t)o * p.Cin * 9 + r * 3 + 1], w2 = wc[(size_t)o * p.Cin * 9 + r * 3 + 2];
        MFN_UNROLL
        for (int yy = 0; yy < 2; ++yy)
          MFN_UNROLL
          for (int i = 0; i < 4; ++i)
            acc[o][yy][i] = fmaf(w2, v[r + yy][i + 2], fmaf(w1, v[r + yy][i + 1], fmaf(w0, v[r + yy][i], acc[o][yy][i])));
      }
  };
  const int c_end = min(p.Cin, ((int)blockIdx.y + 1) * p.cpk);
  MFN_NOUNROLL
  for (int c = (int)blockIdx.y * p.cpk + wave; c < c_end; c += 2 * NW) {
    float4 ma[4], mb[4];
    rows_of(c, ma);
    rows_of(c + NW, mb);
    accumulate(c, c_end, ma);
    accumulate(c + NW, c_end, mb);
  }
  // K slices -> wave 0, in wave order
  if (NW > 1) {
    if (wave > 0) {
      float *dst = red + ((size_t)(wave - 1) * CO * 8) * 64 + lane;
      MFN_UNROLL
      for (int o = 0; o < CO; ++o)
        MFN_UNROLL
        for (int i = 0; i < 8; ++i) dst[(o * 8 + i) * 64] = acc[o][i >> 2][i & 3];
    }
    __syncthreads();
    if (wave > 0) return;
    MFN_NOUNROLL
    for (int k = 1; k < NW; ++k) {
      const float *src = red + ((size_t)(k - 1) * CO * 8) * 64 + lane;
      MFN_UNROLL
      for (int o = 0; o < CO; ++o)
        MFN_UNROLL
        for (int i = 0; i < 8; ++i) acc[o][i >> 2][i & 3] += src[(o * 8 + i) * 64];
    }
  }
  if (!valid) return;
  if (p.kb > 1) {   // raw partial sums of this channel block
    float *pn = p.partial + (((size_t)blockIdx.y * p.N + n) * CO) * plane + (size_t)y0 * W + x0;
    MFN_UNROLL
    for (int o = 0; o < CO; ++o)
      MFN_UNROLL
      for (int yy = 0; yy < 2; ++yy)
        if (y0 + yy < H)
          *reinterpret_cast<float4 *>(pn + (size_t)o * plane + (size_t)yy * W) = make_float4(acc[o][yy][0], acc[o][yy][1], acc[o][yy][2], acc[o][yy][3]);
    return;
  }
  float *on = p.out + (size_t)n * p.out_nstride + (size_t)y0 * W + x0;
  MFN_UNROLL
  for (int o = 0; o < CO; ++o) {
    const float b = p.bias ? p.bias[o] : 0.f;
    MFN_UNROLL
    for (int yy = 0; yy < 2; ++yy) {
      if (y0 + yy >= H) continue;
      float v[4];
      MFN_UNROLL
      for (int i = 0; i < 4; ++i) {
        const float s = acc[o][yy][i] + b;
        v[i] = p.leaky ? fmaxf(s, 0.1f * s) : s;
      }
      *reinterpret_cast<float4 *>(on + (size_t)o * plane + (size_t)yy * W) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}
// A row's 4-pixel groups must lie inside ONE wave (64 % (W/4) == 0, i.e. W <= 256): the halo columns travel by DPP wave shifts, and
// lane 0 / 63 of a wave has no neighbour to receive from -- a row that spans several waves (W = 512, 768, ...) would lose the
// column at every wave seam (wider images take conv_mfma_kernel)
inline bool conv_few_shape_ok(int W) {
  const int gw = W / 4;
  return W % 4 == 0 && gw >= 1 && gw <= 64 && 64 % gw == 0;
}
// channel blocks of a level: until the launch has ~256 blocks, at most 8, at least 32 channels each
inline int conv_few_kb(int N, int Cin, int H, int W) {
  const int nblk = cdiv(N * ((H + 1) / 2) * (W / 4), 64);
  int kb = 1;
  while (kb < 8 && nblk * kb < 192 && Cin / (kb * 2) >= 32) kb *= 2;
  return kb;
}
struct ConvFewReduceParams { const float *partial; const float *bias; float *out; int N, CO, kb, leaky; size_t plane4, out_nstride4; };
__global__ __launch_bounds__(256) void conv_few_reduce_kernel(ConvFewReduceParams p) {
  const size_t n4 = (size_t)p.N * p.CO * p.plane4;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const float4 *src = reinterpret_cast<const float4 *>(p.partial);
  float4 sum = src[i];
  for (int k = 1; k < p.kb; ++k) {
    const float4 v = src[(size_t)k * n4 + i];
    sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
  }
  const size_t img = i / ((size_t)p.CO * p.plane4), rem = i - img * (size_t)p.CO * p.plane4;
  const int o = (int)(rem / p.plane4);
  const float b = p.bias ? p.bias[o] : 0.f;
  float r[4] = {sum.x + b, sum.y + b, sum.z + b, sum.w + b};
  if (p.leaky) { MFN_UNROLL for (int q = 0; q < 4; ++q) r[q] = fmaxf(r[q], 0.1f * r[q]); }
  reinterpret_cast<float4 *>(p.out)[img * p.out_nstride4 + rem] = make_float4(r[0], r[1], r[2], r[3]);
}
template <int CO>
inline int conv_few_launch_co(ConvFewParams p, hipStream_t s) {
  const int nblk = cdiv(p.G, 64);
  // channels over 8 waves where the level has >= 128 wave tiles (level 2: 192 blocks = 1536 waves), over 16 below
  int rc;
  if (nblk >= 128) rc = launch("conv3x3_few", conv_few_kernel<CO, 8>, dim3(nblk, p.kb), dim3(512), (size_t)7 * CO * 8 * 64 * sizeof(float), s, p);
  else rc = launch("conv3x3_few", conv_few_kernel<CO, 16>, dim3(nblk, p.kb), dim3(1024), (size_t)15 * CO * 8 * 64 * sizeof(float), s, p);
  if (rc || p.kb <= 1) return rc;
  const size_t plane4 = (size_t)p.H * p.W / 4;
  ConvFewReduceParams rp{p.partial, p.bias, p.out, p.N, CO, p.kb, p.leaky, plane4, p.out_nstride / 4};
  const size_t n4 = (size_t)p.N * CO * plane4;
  return launch("conv3x3_few_reduce", conv_few_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, rp);
}
// workspace: the channel blocks' partial sums (0 when the level needs one block)
inline size_t conv_few_workspace_bytes(int N, int Cin, int H, int W, int Cout) {
  const int kb = conv_few_kb(N, Cin, H, W);
  return kb > 1 ? (size_t)kb * N * Cout * H * W * sizeof(float) : 0;
}
inline int conv_few_launch(const ConvParams &c, void *workspace, size_t ws_bytes, hipStream_t s) {
  int kb = conv_few_kb(c.N, c.Cin, c.H, c.W);
  if (kb > 1 && (!workspace || ws_bytes < conv_few_workspace_bytes(c.N, c.Cin, c.H, c.W, c.Cout) || ((uintptr_t)workspace & 15))) kb = 1;   // still correct
  const int cpk = kb > 1 ? ((cdiv(c.Cin, kb) + 1) & ~1) : c.Cin;
  ConvFewParams p{c.x, c.w, c.bias, c.out, c.N, c.Cin, c.H, c.W, c.Cout, c.x_nstride, c.out_nstride, c.leaky, c.W / 4, (c.H + 1) / 2,
                  c.N * ((c.H + 1) / 2) * (c.W / 4), kb, cpk, (float *)workspace};
  switch (c.Cout) {
    case 1: return conv_few_launch_co<1>(p, s);
    case 2: return conv_few_launch_co<2>(p, s);
    case 3: return conv_few_launch_co<3>(p, s);
    default: return conv_few_launch_co<4>(p, s);
  }
}

// ---- generic fallback: one thread per output element ------------------------------------------------------
__global__ __launch_bounds__(256) void conv_generic_kernel(ConvParams p) {
  const size_t oplane = (size_t)p.Ho * p.Wo;
  const size_t total = (size_t)p.N * p.Cout * oplane;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int wo = (int)(idx % p.Wo), ho = (int)((idx / p.Wo) % p.Ho);
  const int o = (int)((idx / oplane) % p.Cout);
  const int n = (int)(idx / (oplane * p.Cout));
  const int cpg = p.Cin / p.groups, opg = p.Cout / p.groups;
  const int g = o / opg, ol = o - g * opg;
  const size_t plane = (size_t)p.H * p.W;
  float s = 0.f;
  for (int cl = 0; cl < cpg; ++cl) {
    const int c = g * cpg + cl;
    const float *pl = p.x + (size_t)n * p.x_nstride + (size_t)c * plane;
    for (int i = 0; i < p.kh; ++i)
      for (int q = 0; q < p.kw; ++q) {
        int sy, sx;
        bool v;
        if (p.transposed) v = conv_src<true>(ho, i, p.sh, p.ph, p.dh, p.H, sy) & conv_src<true>(wo, q, p.sw, p.pw, p.dw, p.W, sx);
        else v = conv_src<false>(ho, i, p.sh, p.ph, p.dh, p.H, sy) & conv_src<false>(wo, q, p.sw, p.pw, p.dw, p.W, sx);
        if (!v) continue;
        const float wv = p.transposed ? p.w[(((size_t)c * opg + ol) * p.kh + i) * p.kw + q]
                                      : p.w[(((size_t)o * cpg + cl) * p.kh + i) * p.kw + q];
        s = fmaf(wv, pl[(size_t)sy * p.W + sx], s);
      }
  }
  s += p.bias ? p.bias[o] : 0.f;
  if (p.leaky) s = fmaxf(s, 0.1f * s);
  p.out[(size_t)n * p.out_nstride + (size_t)o * oplane + (size_t)ho * p.Wo + wo] = s;
}
inline int conv_generic_launch(const ConvParams &p, hipStream_t stream) {
  const size_t total = (size_t)p.N * p.Cout * p.Ho * p.Wo;
  if (!total) return 0;
  return launch("conv_generic", conv_generic_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, p);
}

}  // namespace mfn
