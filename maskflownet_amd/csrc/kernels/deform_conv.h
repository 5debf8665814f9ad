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

// measurement builds only (tools/ablate.py): 1 no MFMA, 2 no LDS gather, 4 no window DMA, 8 no stores
#ifndef MFN_DC_ABLATE
#define MFN_DC_ABLATE 0
#endif

namespace mfn {

struct DeformParams {
  const float *x;
  const float *offset;  // (N, 2*kh*kw*dg, Ho, Wo), or NULL in shared-flow mode
  const float *flow;    // shared-flow mode: (N,2,Ho,Wo); offset of every tap = flow*scale/stride
  float flow_scale, flow_stride;
  const float *w;   // original (Cout, Cin/groups, kh, kw)   [generic kernel]
  const float *wt;  // packed [mgroup][ncp_pad][9][2][32*MT]      [MFMA kernel]
  const float *bias;
  float *out;
  int N, Cin, H, W, Cout, CoutP, Ho, Wo;
  int kh, kw, sh, sw, ph, pw, dh, dw, groups, dg;
  int P;  // N*Ho*Wo
  int allow_fast;  // tuning: 0 forces the per-tap path
  unsigned long long *timeline;  // measurement only (mfn_debug_set_timeline)
  int stage_window;              // tuning: 0 disables the LDS source-window staging
  int xcd;                       // 1: blockIdx.x -> tile range remapped per XCD (1-D grids only)
  int st_policy;                 // cache policy of the output stores (mfn_store4_stream)
  int vec_store;                 // out / partial are 16-byte aligned and Wo % 4 == 0: 16-byte epilogue stores
  // fused epilogue of the matching module (MaskFlownet.py:232-233): out = act(out * sigmoid(mask) + add)
  const float *ep_mask;          // (N,1,Ho,Wo) or NULL
  const float *ep_add;           // (N,Cout,Ho,Wo) or NULL
  int ep_leaky;                  // LeakyReLU(0.1)
  int ncp_pad, cps_per_slice, ksb, mgroups;  // packed-weight rows per M-group, K-slice length, cross-block K split
  float inv_tpi, inv_tiles_x;                // 1 / (tiles_y * tiles_x), 1 / tiles_x
  int tile_w, tiles_x, tiles_y, ntiles;      // 32-pixel tiles: (32/tile_w) x tile_w output pixels (tile_w 16 or 8), or
                                             // tile_w == 0: 32 consecutive pixels of the flattened (n,ho,wo) index
  float *partial;                             // ksb > 1: raw partial sums [ksb][N][Cout][Ho][Wo]
  int dcm_groups, dcm_gps;                    // dc_mma_kernel (deform_conv_mma.h): 16-channel groups of the call / per K slice
  size_t x_nstride, out_nstride;              // dc_mma_kernel: elements between consecutive images of x / out (0: dense) -- channel slices
                                              // of a concat buffer (the plain-convolution form, mfn_conv2d_fwd)
};

// weights (Cout, Cin, 9) -> wt[mg][cp][t][half][RL]: filter o = mg*RL + r, channel c = 2*cp + half; zero padded
// in c (odd Cin, rows up to ncp_pad) and o (Cout not a multiple of RL).  One M-group is one linear
// array, so a K-chunk of it is one contiguous LDS-DMA transfer.
struct PackParams { const float *w; float *wt; int Cin, Cout, RL, mgroups, ncp_pad, T; };
__global__ __launch_bounds__(256) void dc_pack_weights_kernel(PackParams p) {
  const size_t total = (size_t)p.mgroups * p.ncp_pad * p.T * 2 * p.RL;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int r = (int)(idx % p.RL);
  const int half = (int)((idx / p.RL) & 1);
  const int t = (int)((idx / ((size_t)2 * p.RL)) % p.T);
  const int cp = (int)((idx / ((size_t)2 * p.RL * p.T)) % p.ncp_pad);
  const int mg = (int)(idx / ((size_t)2 * p.RL * p.T * p.ncp_pad));
  const int c = 2 * cp + half, o = mg * p.RL + r;
  p.wt[idx] = (c < p.Cin && o < p.Cout) ? p.w[((size_t)o * p.Cin + c) * p.T + t] : 0.f;
}

// the matching module's epilogue on one output value (bias already added): v * sigmoid(mask) + add, LeakyReLU(0.1)
__device__ __forceinline__ float dc_epilogue(float v, const float *mask, const float *add, int leaky, size_t mask_idx,
                                             size_t out_idx) {
  if (mask) v = v * (1.f / (1.f + expf(-mask[mask_idx])));
  if (add) v = v + add[out_idx];
  return leaky ? fmaxf(v, 0.1f * v) : v;
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
struct DcAxis3 { float a[3], b[3]; int idx[4]; };  // idx: clamped row / column numbers of the 4 neighbourhood lines

// ---- geometry shared by host and device ------------------------------------------------------------
// chunk = KC channel pairs of one M-group's packed weights = KC*18*RL floats, RL = 32*MT filters
// KC is sized so that the two weight stage buffers take <= 24 KB: with the 24 KB of x windows a block then
// needs <= 48 KB of LDS and three blocks fit a CU (which is also what the VGPR budget allows).
constexpr int dc_kc(int mt, int kw) {
  const int q = 4 / (mt * kw);
  return q >= 4 ? 4 : (q >= 2 ? 2 : 1);
}
template <int V> struct DcInt { static constexpr int value = V; };
template <int MT, int KW> struct DcGeom {
  static constexpr int RL = 32 * MT;
  static constexpr int KC = dc_kc(MT, KW);
  static constexpr int PAIR_W = 18 * RL;   // words of one channel pair's weights
  static constexpr int CHUNK_F = KC * PAIR_W;   // floats
  static constexpr int CH4 = CHUNK_F / 4;       // float4 items
};

// waves per SIMD the register allocator is held to (three for one or two filter tiles per wave: LDS allows
// three blocks per CU, and a level-2 or level-3 launch is then resident in one round)
constexpr int dc_min_waves(int mt, int pt, int nw = 4) {
  return nw == 8 ? 2 : (mt <= 2 ? 3 : (mt == 3 ? 2 : (pt >= 2 ? 2 : 1)));
}

// NW = waves per block (4, or 8 for the coarsest level: twice the in-block K slices, half the channel-pair chain per wave)
template <int MT, int PT, int NW = 4>
__global__ __launch_bounds__(NW * 64, dc_min_waves(MT, PT, NW)) void dc_lds_kernel(DeformParams p) {
  constexpr int T = 9;
  constexpr int NTH = NW * 64;
  constexpr int KW = NW / PT;  // K-slices handled inside the block (one wave each per pixel tile)
  using G = DcGeom<MT, KW>;
  constexpr int RL = G::RL, KC = G::KC, CH4 = G::CH4;
  constexpr int NI = (KW * CH4 + NTH - 1) / NTH;  // DMA instructions per thread per stage
  constexpr int STAGE_F = NI * NTH * 4;   // floats per stage buffer
  MFN_DYN_SHARED(float, lds);                 // 2 weight stage buffers + x windows (all reused for the K-slice reduction)
  // staged source window per wave and channel: 10 rows x 24 floats under a 2x16 pixel tile, 12 rows x 20 floats
  // under a 4x8 tile -- 60 float4 slots per channel, one channel pair = 2 wave DMA instructions (120 of 128 lanes)
  constexpr int XW_NI = 2;
  constexpr int XW_F = XW_NI * 256;           // floats per channel-pair buffer
  const int XW_ROWS = p.tile_w == 16 ? 10 : 12, XW_C4 = p.tile_w == 16 ? 6 : 5, XW_COLS = 4 * XW_C4;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = MFN_UNIFORM(tid >> 6);
  const int half = lane >> 5, j = lane & 31;
  MFN_STAMP(p.timeline, 0);
#ifdef MFN_DC_STAGGER  // measurement build: the launch's thirds (= the three resident blocks of a CU) start MFN_DC_STAGGER x 64 x k cycles apart
  for (int i = 0, nsl = (int)((blockIdx.x * 3u) / gridDim.x); i < nsl; ++i) __builtin_amdgcn_s_sleep(MFN_DC_STAGGER);
#endif
  const int pt = wave / KW, kw = wave % KW;
  // neighbouring tiles stage overlapping source windows: keep them on one XCD's L2
  const int bx = p.xcd ? (int)mfn_xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int tile = bx * PT + pt;
  const int gs = blockIdx.y * KW + kw;        // global K-slice of this wave
  const int mg = blockIdx.z;                  // M-group: filters [mg*RL, mg*RL + RL)
  const int m0 = mg * RL;

  const int H = p.H, W = p.W, Ho = p.Ho, Wo = p.Wo;
  const size_t plane = (size_t)H * W;
  const size_t oplane = (size_t)Ho * Wo;
  // 2-D pixel tiles keep the source window of a wave small (4x8 px: 7 rows x 11 columns plus the flow's
  // variation, against 4 rows x 35 columns for 32 px of one row) and never span two images.
  int n, ho, wo;
  int tile_ho0 = 0, tile_wo0 = 0;  // origin of a 2-D tile (uniform)
  bool px_valid;
  if (p.tile_w) {
    const int tpi = p.tiles_y * p.tiles_x;
    const int tl = min(tile, p.ntiles - 1);
    // tile -> (image, tile row, tile column) without integer division: tile counts are far below 2^24, so the
    // float quotient is off by at most one and a single correction step makes it exact
    auto divmod = [](int a, int b, float inv_b, int &q, int &r) {
      q = (int)((float)a * inv_b);
      r = a - q * b;
      if (r < 0) { --q; r += b; }
      if (r >= b) { ++q; r -= b; }
    };
    int rt, ty, tx;
    divmod(tl, tpi, p.inv_tpi, n, rt);
    divmod(rt, p.tiles_x, p.inv_tiles_x, ty, tx);
    const int sh16 = p.tile_w == 16 ? 4 : 3;
    tile_ho0 = ty * (32 >> sh16);
    tile_wo0 = tx * p.tile_w;
    ho = tile_ho0 + (j >> sh16);
    wo = tile_wo0 + (j & (p.tile_w - 1));
    px_valid = tile < p.ntiles && ho < Ho && wo < Wo;
    ho = min(ho, Ho - 1);
    wo = min(wo, Wo - 1);
  } else {
    const int plin = tile * 32 + j;
    px_valid = plin < p.P;
    const int pc = px_valid ? plin : 0;
    n = pc / (Ho * Wo);
    const int rem = pc - n * (Ho * Wo);
    ho = rem / Wo;
    wo = rem - ho * Wo;
  }
  const int h_in = ho * p.sh - p.ph, w_in = wo * p.sw - p.pw;

  // ---- weight staging plan: item -> byte offset inside this M-group's packed array -----------------
  const size_t mg_floats = (size_t)p.ncp_pad * G::PAIR_W;
  const mfn_rsrc_t wrsrc = mfn_make_rsrc(p.wt + (size_t)mg * mg_floats, (unsigned)(mg_floats * 4));
  unsigned voff[NI];
  MFN_UNROLL
  for (int i = 0; i < NI; ++i) {
    const int it = (i * NW + wave) * 64 + lane;
    const int k = it / CH4, idx = it - k * CH4;
    voff[i] = k < KW ? (unsigned)(((size_t)(blockIdx.y * KW + k) * p.cps_per_slice * G::PAIR_W + (size_t)idx * 4) * 4)
                     : 0xFFFFFF00u;
  }
  auto issue = [&](int ch) {
    float *buf = lds + (ch & 1) * STAGE_F;
    const unsigned soff = (unsigned)((size_t)ch * G::CHUNK_F * 4);
    MFN_UNROLL
    for (int i = 0; i < NI; ++i) {
      float *dst = buf + (i * NW + wave) * 256;
      mfn_dma16_so(wrsrc, dst, voff[i], soff);
    }
  };

  issue(0);  // the first weight chunk lands while the tap geometry below is computed

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
    // all eighteen requested before the first is compared: behind a short-circuit `&&` hipcc sinks each pair of loads into
    // its own branch -- nine dependent round trips at the head of every block
    MFN_COMPILER_FENCE();
    int same = 1;
    MFN_UNROLL
    for (int t = 1; t < T; ++t) same &= (int)(offh[t] == offh[0]) & (int)(offw[t] == offw[0]);
    shared = same != 0;
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
  int nb_c0 = 0;  // unclamped first column of the 4x4 neighbourhood (may lie outside the image at its left / right edge)
  {
    int lo0 = 0;
    MFN_UNROLL
    for (int i = 0; i < 3; ++i) {
      bool v; int lo, hi; float l;
      dc_axis(offh[0], h_in, i, H, v, lo, hi, l);
      v = v && px_valid;
      const int ulo = (int)fminf(fmaxf(floorf((float)i + offh[0]), -1.0e6f), 1.0e6f);  // unclamped floor
      // (i + off can round UP to an integer in fp32 -- off a hair below one: the oracle then reads line i + 1 with weight 1, which
      // is the same pair of lines with weights (0, 1): still the shared path, see deform_conv_mma.h)
      const bool up = i > 0 && ulo == lo0 + i + 1 && l == 0.f;
      if (i == 0) lo0 = ulo; else regular = regular && (ulo == lo0 + i || up);
      ay.a[i] = v ? (up ? 0.f : 1.f - l) : 0.f;
      ay.b[i] = v ? (up ? 1.f : l) : 0.f;
    }
    MFN_UNROLL
    for (int m = 0; m < 4; ++m) ay.idx[m] = min(max(h_in + lo0 + m, 0), H - 1);
    MFN_UNROLL
    for (int i = 0; i < 3; ++i) {
      bool v; int lo, hi; float l;
      dc_axis(offw[0], w_in, i, W, v, lo, hi, l);
      v = v && px_valid;
      const int ulo = (int)fminf(fmaxf(floorf((float)i + offw[0]), -1.0e6f), 1.0e6f);
      const bool up = i > 0 && ulo == lo0 + i + 1 && l == 0.f;
      if (i == 0) lo0 = ulo; else regular = regular && (ulo == lo0 + i || up);
      ax.a[i] = v ? (up ? 0.f : 1.f - l) : 0.f;
      ax.b[i] = v ? (up ? 1.f : l) : 0.f;
    }
    MFN_UNROLL
    for (int m = 0; m < 4; ++m) ax.idx[m] = min(max(w_in + lo0 + m, 0), W - 1);
    nb_c0 = w_in + lo0;
  }
  const bool fast = __all(regular || !px_valid) != 0;  // wave-uniform

  // ---- source-window staging (fast path): with a smooth flow the 4x4 neighbourhoods of a wave's 32
  // pixels overlap almost completely.  Gathering them lane by lane costs ~41 L1 accesses per wave
  // instruction and keeps the texture addresser ~65% busy (profiles/r01_deform_pmc.md); instead the
  // wave DMAs the bounding box of its neighbourhoods (full rows, <= 8 x 48 floats per channel) into a
  // private LDS window and gathers from there with ds_read_b32.  Windows that do not fit (large or
  // discontinuous flow, tiles spanning two images, W % 4 != 0) take the lean per-tap path instead.
  int wr0, wc0;
  bool staged;
  {
    const int big = 1 << 28;
    int rlo = px_valid ? ay.idx[0] : big, rhi = px_valid ? ay.idx[3] : -big;
    int clo = px_valid ? ax.idx[0] : big, chi = px_valid ? ax.idx[3] : -big;
    const int nlo = 0, nhi = 0;  // windows are only staged under 2-D tiles, which never span two images
    if (p.tile_w && fast) {
      rlo = mfn_wave_min_i32(rlo); rhi = mfn_wave_max_i32(rhi);
      clo = mfn_wave_min_i32(clo); chi = mfn_wave_max_i32(chi);
    }
    wr0 = rlo;
    wc0 = clo & ~3;  // 16-byte aligned window origin
    staged = MFN_UNIFORM((int)(fast && p.stage_window && p.tile_w && (W % 4 == 0) && nlo == nhi && (rhi - wr0 < XW_ROWS) &&
                               (chi - wc0 < XW_COLS))) != 0;
    wr0 = MFN_UNIFORM(wr0);
    wc0 = MFN_UNIFORM(wc0);
#ifdef MFN_EMU_DEBUG
    if (lane == 0) printf("tile %d wave %d: rows %d..%d cols %d..%d n %d..%d staged %d fast %d\n", tile, wave, rlo, rhi, clo, chi, nlo, nhi, (int)staged, (int)fast);
#endif
  }

  f32x16 acc[MT];
  MFN_UNROLL
  for (int mt = 0; mt < MT; ++mt)
    MFN_UNROLL
    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

  const float *xn = p.x + (size_t)n * p.Cin * plane;
  const int cp_base = gs * p.cps_per_slice;
  const int nchunks = p.cps_per_slice / KC;  // cps_per_slice is a multiple of KC

  // this lane's weight of tap t inside a channel pair's block `pw` (wave-uniform pointer): the filter (j) x channel (half)
  // element the fp32 MFMA wants as A
  auto a_of = [&](const float *pw, int t, int mt) -> float { return pw[(t * 2 + half) * RL + mt * 32 + j]; };
  const int full_pairs = p.Cin / 2;  // pairs whose two channels both exist; an odd Cin adds one half pair
  // one channel pair on the fast path: separable bilinear interpolation of the 4x4 neighbourhood into
  // the 9 column values, which ARE the B operands of the 9 k-steps
  auto fast_pair = [&](const float (&v)[4][4], const float *pw) {
    float tr[4][3];
    MFN_UNROLL
    for (int m = 0; m < 4; ++m)
      MFN_UNROLL
      for (int q = 0; q < 3; ++q) tr[m][q] = ax.a[q] * v[m][q] + ax.b[q] * v[m][q + 1];
    MFN_UNROLL
    for (int i = 0; i < 3; ++i)
      MFN_UNROLL
      for (int q = 0; q < 3; ++q) {
        const float cv = ay.a[i] * tr[i][q] + ay.b[i] * tr[i + 1][q];
        MFN_UNROLL
        for (int mt = 0; mt < MT; ++mt)
          acc[mt] = MFN_MFMA_32x32x2(a_of(pw, i * 3 + q, mt), cv, acc[mt]);
      }
  };
  // second tier (window does not fit: a rough or discontinuous flow, outliers): every row of the 4x4 neighbourhood is ONE
  // dword-aligned 16-byte load straight from global memory, at the neighbourhood's first column clamped into [0, W - 4];
  // the loads of pair k + 2 are in flight while pair k + 1 is interpolated and pair k multiplied (software pipeline
  // below).  Lanes at the image's left / right edge see their columns shifted by `cshift` inside the loaded quad: their
  // x pass runs on a general 3 x 4 weight matrix (zero where a column lies outside -- those carry weight 0 anyway), chosen
  // per wave; everybody else keeps the banded two-term form.  Third tier (images narrower than four columns): 16 dword
  // gathers at clamped columns.
  const bool small = (size_t)p.N * p.Cin * plane < ((size_t)1 << 30);
  const bool gtier = !staged && fast && small && W >= 4;
  const bool dwgather = !staged && fast && small && !gtier;
  const int cbase = min(max(nb_c0, 0), max(W - 4, 0));
  const int cshift = nb_c0 - cbase;
  const bool gbanded = __all(cshift == 0 || !px_valid) != 0;  // wave-uniform
  unsigned rowoff[4];
  MFN_UNROLL
  for (int m = 0; m < 4; ++m)
    rowoff[m] = (unsigned)(n * p.Cin * (int)plane + half * (int)plane + ay.idx[m] * W + ax.idx[0]);
  auto dwgather_pair = [&](int cp, const float *pw) {
    const float *base = p.x + (size_t)(2 * cp) * plane;  // uniform
    float v[4][4];
    MFN_UNROLL
    for (int m = 0; m < 4; ++m)
      MFN_UNROLL
      for (int q = 0; q < 4; ++q) v[m][q] = base[rowoff[m] + (unsigned)(ax.idx[q] - ax.idx[0])];
    fast_pair(v, pw);
  };
  // per-tap path (arbitrary offsets, and the half pair of an odd Cin; never hot in the reference
  // network).  Deliberately lean in registers, not fast: a rolled tap loop that re-reads its offset,
  // rebuilds the tap geometry and feeds the MFMA at once, so the fast path sets the VGPR budget.
  auto slow_pair = [&](int cp, const float *pw) {
    const int c = 2 * cp + half;
    const bool cvalid = c < p.Cin;
    const float *pl = xn + (size_t)(cvalid ? c : 0) * plane;
    MFN_NOUNROLL
    for (int t = 0; t < T; ++t) {
      float oh, ow;
      if (p.offset) {
        const float *op = p.offset + (size_t)n * 2 * T * oplane + (size_t)ho * Wo + wo;
        oh = op[(size_t)(2 * t) * oplane];
        ow = op[(size_t)(2 * t + 1) * oplane];
      } else {
        oh = offh[0];
        ow = offw[0];
      }
      const int ti = t / 3, tj = t - 3 * ti;
      const DcTap tp = dc_make_tap(oh, ow, h_in, w_in, ti * p.dh, tj * p.dw, H, W, px_valid && cvalid);
      const int bb = tp.base & 0x3FFFFFFF, dwi = (tp.base >> 30) & 1;
      const float v1 = pl[bb], v2 = pl[bb + dwi], v3 = pl[bb + tp.dhW], v4 = pl[bb + tp.dhW + dwi];
      const float cv = tp.w1 * v1 + tp.w2 * v2 + tp.w3 * v3 + tp.w4 * v4;
      MFN_UNROLL
      for (int mt = 0; mt < MT; ++mt) acc[mt] = MFN_MFMA_32x32x2(a_of(pw, t, mt), cv, acc[mt]);
    }
  };

  // ---- staged-window plumbing ---------------------------------------------------------------------------
  float *xwin = lds + 2 * STAGE_F + wave * (3 * XW_F);  // this wave's three pair buffers (ring)
  const mfn_rsrc_t xrsrc = mfn_make_rsrc(p.x, (unsigned)((size_t)p.N * p.Cin * plane * 4));
  unsigned xvoff[XW_NI];  // byte offset of this lane's window slots, relative to channel 2*cp of image 0
  int loff[4][4];      // LDS float offset of the 16 neighbourhood values inside a pair buffer
  if (staged) {
    const int nimg = MFN_UNIFORM(n);
    MFN_UNROLL
    for (int i = 0; i < XW_NI; ++i) {
      const int slot = i * 64 + lane;               // float4 slots: [channel 0/1][XW_ROWS][XW_C4 float4]
      const int chs = slot / 60, rem = slot - chs * 60;  // XW_ROWS * XW_C4 == 60 for both tile shapes
      const int row = p.tile_w == 16 ? rem / 6 : rem / 5, c4 = rem - row * XW_C4;
      const int r = wr0 + row, c = wc0 + 4 * c4;
      xvoff[i] = (chs < 2 && r <= H - 1 && c <= W - 4)
                     ? (unsigned)(((size_t)nimg * p.Cin * plane + (size_t)chs * plane + (size_t)r * W + c) * 4)
                     : 0xFFFFFF00u;                  // outside the image: never read, the DMA writes zeros
    }
    MFN_UNROLL
    for (int m = 0; m < 4; ++m)
      MFN_UNROLL
      for (int q = 0; q < 4; ++q)
        loff[m][q] = px_valid ? half * (XW_ROWS * XW_COLS) + (ay.idx[m] - wr0) * XW_COLS + (ax.idx[q] - wc0) : 0;  // lanes past
        // the image read slot 0 (their weights are zero): their own neighbourhood may lie anywhere relative to the window
  }
  auto issue_x = [&](int cp, int buf) {
    const unsigned soff = (unsigned)((size_t)(2 * cp) * plane * 4);
    if (MFN_DC_ABLATE & 4) return;
    MFN_UNROLL
    for (int i = 0; i < XW_NI; ++i) mfn_dma16_so(xrsrc, xwin + buf * XW_F + i * 256, xvoff[i], soff);
  };
  // ---- staged waves: one flat software pipeline over the channel pairs of this K-slice ---------------------
  // iteration k issues the MFMAs of pair k interleaved with the interpolation of pair k+1 (whose 16 window
  // values were requested from LDS just before), while the windows of pairs k+2 and k+3 are in flight as LDS-DMA
  // and the next weight chunk streams into the other stage buffer.  DMA completion is in issue order, so every
  // wait below is a count of the newer transfers that may still be outstanding.
  int k_done = 0;  // pairs of this slice already accumulated
  if (staged) {
    const int nf = MFN_UNIFORM(max(0, min(nchunks * KC, full_pairs - cp_base)));
    auto gather = [&](int buf, float (&v)[4][4]) {
      const float *xb = xwin + buf * XW_F;
      MFN_UNROLL
      for (int m = 0; m < 4; ++m)
        MFN_UNROLL
        for (int q = 0; q < 4; ++q) v[m][q] = (MFN_DC_ABLATE & 2) ? (float)(m + q + buf) : xb[loff[m][q]];
    };
    auto interp_row = [&](const float (&v)[4][4], float (&tr)[4][3], int m) {
      MFN_UNROLL
      for (int q = 0; q < 3; ++q) tr[m][q] = ax.a[q] * v[m][q] + ax.b[q] * v[m][q + 1];
    };
    // the same for two neighbourhood rows at once: v_pk_mul_f32 + v_pk_fma_f32, one issue slot for two values (the mul is
    // rounded, then the fma: bit-identical to the scalar form).  fp32 MFMA shares the VALU's issue slots, so the 12 saved
    // of a pair's 42 are worth 48 of its 745 cycles.
    // vp[h][q] = rows 2h and 2h + 1 of neighbourhood column q in ONE 64-bit register pair (the two ds_read_b32 write its
    // halves): no moves to build the packed operands
    auto gather2 = [&](int buf, f32x2 (&vp)[2][4]) {
      const float *xb = xwin + buf * XW_F;
      MFN_UNROLL
      for (int h = 0; h < 2; ++h)
        MFN_UNROLL
        for (int q = 0; q < 4; ++q) {
          vp[h][q].x = (MFN_DC_ABLATE & 2) ? (float)(2 * h + q + buf) : xb[loff[2 * h][q]];
          vp[h][q].y = (MFN_DC_ABLATE & 2) ? (float)(2 * h + 1 + q + buf) : xb[loff[2 * h + 1][q]];
        }
    };
    auto interp_rows2 = [&](const f32x2 (&vp)[2][4], f32x2 (&trp)[2][3], int h) {
      MFN_UNROLL
      for (int q = 0; q < 3; ++q)
        trp[h][q] = mfn_fma2(mfn_f2(ax.b[q], ax.b[q]), vp[h][q + 1], mfn_mul2(mfn_f2(ax.a[q], ax.a[q]), vp[h][q]));
    };
    auto interp_col2 = [&](const f32x2 (&trp)[2][3], float (&cv)[9], int i) {
      MFN_UNROLL
      for (int q = 0; q < 3; ++q) {
        const float lo = i == 0 ? trp[0][q].x : (i == 1 ? trp[0][q].y : trp[1][q].x);
        const float hi = i == 0 ? trp[0][q].y : (i == 1 ? trp[1][q].x : trp[1][q].y);
        cv[i * 3 + q] = ay.a[i] * lo + ay.b[i] * hi;
      }
    };
    auto interp_col = [&](const float (&tr)[4][3], float (&cv)[9], int i) {
      MFN_UNROLL
      for (int q = 0; q < 3; ++q) cv[i * 3 + q] = ay.a[i] * tr[i][q] + ay.b[i] * tr[i + 1][q];
    };
    auto mfma_tap = [&](const float *ap, int t, float b) {
      if (MFN_DC_ABLATE & 1) { acc[0][t] += b; return; }
      MFN_UNROLL
      for (int mt = 0; mt < MT; ++mt) acc[mt] = MFN_MFMA_32x32x2(ap[(t * 2) * RL + mt * 32], b, acc[mt]);
    };
    float v[4][4], tr[4][3], cv[9], cvn[9];
    f32x2 vp[2][4], trp[2][3];
    if (nf > 0) issue_x(cp_base, 0);
    if (nf > 1) issue_x(cp_base + 1, 1);
    if (nf > 2) issue_x(cp_base + 2, 2);
    if (nf > 0) {
      if (nf > 2) MFN_WAIT_VM(2 * XW_NI); else MFN_WAIT_VM(0);  // window 0 (and the first weight chunk) landed
      gather(0, v);
      MFN_UNROLL
      for (int m = 0; m < 4; ++m) interp_row(v, tr, m);
      MFN_UNROLL
      for (int i = 0; i < 3; ++i) interp_col(tr, cv, i);
    }
    bool w_in_flight = false;  // a weight chunk was issued in the previous iteration
    // One pipeline step for pair k.  Pair p lives in window buffer p % 3; the buffer numbers are compile-time
    // (the loop below is unrolled over the ring) so that the 16 gather addresses are loop-invariant registers
    // and the buffer is an instruction offset -- fp32 MFMA shares the VALU's ALUs, every saved VALU counts.
    // cur / nxt: the nine column values of pair k (consumed here) and of pair k + 1 (formed here): the two sets swap roles
    // from step to step and the loop below is unrolled over six steps (window ring x value sets), so that no value is
    // copied at the loop's back edge -- with a copy hipcc spent ~14 v_mov per step on the ALUs the MFMAs need
    auto step = [&](int k, auto bn_c, float (&cur)[9], float (&nxt)[9]) {
      constexpr int BN = decltype(bn_c)::value;  // buffer of pair k+1 (read now)
      constexpr int BF = (BN + 2) % 3;           // buffer of pair k = where pair k+3 is streamed to
      const int ch = k / KC, kk = k - ch * KC;
      const bool more3 = k + 3 < nf;
      bool w_now = false;
      if (kk == 0) {
        // weights of chunk ch and window k+1 landed (window k+2 may still fly); every wave is past chunk ch-1
        if (k + 2 < nf) MFN_WAIT_VM(XW_NI); else MFN_WAIT_VM(0);
        MFN_WAIT_LGKM0();
        MFN_RAW_BARRIER();
        if (k == 0) MFN_STAMP(p.timeline, 1);
        if (ch + 1 < nchunks) { issue(ch + 1); w_now = true; }
        if (more3) issue_x(cp_base + k + 3, BF);
      } else {
        if (more3) {
          issue_x(cp_base + k + 3, BF);
          // outstanding, oldest first: window k+1, [weights issued last iteration], window k+2, window k+3
          if (w_in_flight) MFN_WAIT_VM(NI + 2 * XW_NI); else MFN_WAIT_VM(2 * XW_NI);
        } else {
          MFN_WAIT_VM(0);
        }
      }
      w_in_flight = w_now;
      const float *pw = lds + (ch & 1) * STAGE_F + kw * G::CHUNK_F + (size_t)kk * G::PAIR_W;   // this pair's weights (uniform)
      const float *ap = pw + half * RL + j;
      // The interpolation of pair k + 1 runs in the last step too (on the previous step's window values, results unused):
      // with the MFMAs in two branches the accumulators lived in two register sets and hipcc copied all sixteen (after
      // draining the MFMA pipe) at the end of EVERY step.  The LDS reads themselves must NOT run there: left in flight when
      // the loop ends, they land in registers that are dead from the compiler's point of view -- it had handed them to the
      // epilogue without a wait, and one pass in two came back with a wrong 4x8 tile somewhere (tools/r03_det.py; every
      // comparison with the oracle had passed).
      if (k + 1 < nf) gather2(BN, vp);
        mfma_tap(ap, 0, cur[0]); interp_rows2(vp, trp, 0);
        mfma_tap(ap, 1, cur[1]);
        mfma_tap(ap, 2, cur[2]); interp_rows2(vp, trp, 1);
        mfma_tap(ap, 3, cur[3]);
        mfma_tap(ap, 4, cur[4]); interp_col2(trp, nxt, 0);
        mfma_tap(ap, 5, cur[5]); interp_col2(trp, nxt, 1);
        mfma_tap(ap, 6, cur[6]); interp_col2(trp, nxt, 2);
        mfma_tap(ap, 7, cur[7]);
        mfma_tap(ap, 8, cur[8]);
      MFN_SCHED_BARRIER();
    };
    for (int k = 0; k < nf;) {
      step(k, DcInt<1>{}, cv, cvn);
      if (++k >= nf) break;
      step(k, DcInt<2>{}, cvn, cv);
      if (++k >= nf) break;
      step(k, DcInt<0>{}, cv, cvn);
      if (++k >= nf) break;
      step(k, DcInt<1>{}, cvn, cv);
      if (++k >= nf) break;
      step(k, DcInt<2>{}, cv, cvn);
      if (++k >= nf) break;
      step(k, DcInt<0>{}, cvn, cv);
      ++k;
    }
    k_done = nf;
  }

  // ---- global-gather waves: the same flat pipeline, the 4x4 neighbourhood rows arriving in registers ----------------
  // Pair p's four row quads live in register slot p % 2; step k multiplies pair k, interpolates pair k + 1 (requested one
  // step ago) and requests pair k + 2.  The loads are counted by hand like the LDS-DMA transfers (mfn_gload4_async).  The
  // waits rely on in-order completion only AMONG the register loads and AMONG the LDS-DMA transfers, never between the two
  // kinds: with "the weight chunk is older than the four quads still allowed in flight" as the argument for vmcnt(4) at a
  // chunk boundary, one pass in a dozen came back with a block of wrong values at level 3 (tools/r03_det.py) -- a
  // register load can complete before an older LDS-DMA.  So: a chunk boundary drains everything (vmcnt(0)), and "pair
  // k + 1 landed" is vmcnt(4): pair k + 1 outstanding would need all four quads of pair k + 2 outstanding as well.
  if (gtier) {
    const int nf = MFN_UNIFORM(max(0, min(nchunks * KC, full_pairs - cp_base)));
    unsigned boff[4];  // byte offset of the lane's four row quads inside channel 2*cp of image 0
    MFN_UNROLL
    for (int m = 0; m < 4; ++m)
      boff[m] = (unsigned)(n * p.Cin * (int)plane + half * (int)plane + ay.idx[m] * W + cbase) * 4u;
    f32x4 R[2][4];
    auto fetch = [&](int cp, auto slot_c) {
      constexpr int S = decltype(slot_c)::value;
      const float *base = p.x + (size_t)(2 * cp) * plane;  // uniform
      MFN_UNROLL
      for (int m = 0; m < 4; ++m) mfn_gload4_async(R[S][m], base, boff[m]);
    };
    auto mfma_tap = [&](const float *ap, int t, float b) {
      MFN_UNROLL
      for (int mt = 0; mt < MT; ++mt) acc[mt] = MFN_MFMA_32x32x2(ap[(t * 2) * RL + mt * 32], b, acc[mt]);
    };
    auto pipeline = [&](auto gen_c) {
      constexpr bool GEN = decltype(gen_c)::value;
      // x-pass weights: banded (two terms per tap column) or, for waves with edge lanes, the 3 x 4 matrix
      float X[3][4];
      if (GEN) {
        MFN_UNROLL
        for (int q = 0; q < 3; ++q)
          MFN_UNROLL
          for (int u = 0; u < 4; ++u)
            X[q][u] = (u == q + cshift ? ax.a[q] : 0.f) + (u == q + cshift + 1 ? ax.b[q] : 0.f);
      }
      f32x2 trp[4];  // x pass of a neighbourhood row: columns 0 and 1 as a register pair (packed y pass), column 2 beside it
      float tr2[4];
      auto xrow = [&](const f32x4 &r, int m) {
        float t[3];
        MFN_UNROLL
        for (int q = 0; q < 3; ++q) {
          if (GEN) {
            float a = X[q][0] * r[0];
            a = fmaf(X[q][1], r[1], a);
            a = fmaf(X[q][2], r[2], a);
            t[q] = fmaf(X[q][3], r[3], a);
          } else {
            t[q] = fmaf(ax.b[q], r[q + 1], ax.a[q] * r[q]);
          }
        }
        trp[m] = mfn_f2(t[0], t[1]);
        tr2[m] = t[2];
      };
      auto ycol = [&](float (&cv)[9], int i) {
        const f32x2 c01 = mfn_fma2(mfn_f2(ay.b[i], ay.b[i]), trp[i + 1], mfn_mul2(mfn_f2(ay.a[i], ay.a[i]), trp[i]));
        cv[i * 3 + 0] = c01.x;
        cv[i * 3 + 1] = c01.y;
        cv[i * 3 + 2] = fmaf(ay.b[i], tr2[i + 1], ay.a[i] * tr2[i]);
      };
      float cv[9], cvn[9];
      // every request is unconditional (past the last pair the last one is requested again and never used): a request
      // inside a branch would make the slot registers phi values, see MFN_REGFENCE4
      const int cp_last = cp_base + nf - 1;
      if (nf > 0) {
        fetch(cp_base, DcInt<0>{});
        fetch(min(cp_base + 1, cp_last), DcInt<1>{});
        MFN_LANDED4(R[0][0], R[0][1], R[0][2], R[0][3], 4);  // the first weight chunk is older than both
        MFN_UNROLL
        for (int m = 0; m < 4; ++m) xrow(R[0][m], m);
        MFN_UNROLL
        for (int i = 0; i < 3; ++i) ycol(cv, i);
        MFN_REGFENCE9(cv);
      }
      auto step = [&](int k, auto sn_c, float (&cur)[9], float (&nxt)[9]) {
        constexpr int SN = decltype(sn_c)::value;  // slot of pair k + 1 (read now); pair k + 2 goes to the other one
        constexpr int SF = 1 - SN;
        const int ch = k / KC, kk = k - ch * KC;
        if (kk == 0) {
          MFN_WAIT_VM(0);  // this wave's part of weight chunk ch landed (and the quads of pair k + 1 with it)
          MFN_WAIT_LGKM0();
          MFN_RAW_BARRIER();
          if (k == 0) MFN_STAMP(p.timeline, 1);
        }
        fetch(min(cp_base + k + 2, cp_last), DcInt<SF>{});
        if (kk == 0 && ch + 1 < nchunks) issue(ch + 1);
        MFN_WAIT_VM(4);  // pair k + 1 landed (see above; a weight chunk requested in this or the last step is waited for too)
        MFN_REGFENCE4(R[SN][0], R[SN][1], R[SN][2], R[SN][3]);
        const float *pw = lds + (ch & 1) * STAGE_F + kw * G::CHUNK_F + (size_t)kk * G::PAIR_W;
        const float *ap = pw + half * RL + j;
        // the last step interpolates a repeated pair (results unused): one MFMA stream, no accumulator copies
          mfma_tap(ap, 0, cur[0]); xrow(R[SN][0], 0);
          mfma_tap(ap, 1, cur[1]); xrow(R[SN][1], 1);
          mfma_tap(ap, 2, cur[2]); xrow(R[SN][2], 2);
          mfma_tap(ap, 3, cur[3]); xrow(R[SN][3], 3);
          mfma_tap(ap, 4, cur[4]); ycol(nxt, 0);
          mfma_tap(ap, 5, cur[5]); ycol(nxt, 1);
          mfma_tap(ap, 6, cur[6]); ycol(nxt, 2);
          mfma_tap(ap, 7, cur[7]);
          mfma_tap(ap, 8, cur[8]);
        MFN_REGFENCE9(nxt);  // slot SN is read out: the next step's request may overwrite it
        MFN_SCHED_BARRIER();
      };
      for (int k = 0; k < nf;) {
        step(k, DcInt<1>{}, cv, cvn);
        if (++k >= nf) break;
        step(k, DcInt<0>{}, cvn, cv);
        ++k;
      }
      // nothing may still be on its way into the slot registers when they are given to somebody else
      if (nf > 0) {
        MFN_WAIT_VM(0);
        MFN_REGFENCE4(R[0][0], R[0][1], R[0][2], R[0][3]);
        MFN_REGFENCE4(R[1][0], R[1][1], R[1][2], R[1][3]);
      }
    };
    if (gbanded) pipeline(std::integral_constant<bool, false>{}); else pipeline(std::integral_constant<bool, true>{});
    k_done = nf;
  }

  // ---- everything else: the odd half pair, padding, per-tap offsets, images narrower than four columns ----------
  for (int ch = k_done / KC; ch < nchunks; ++ch) {
    const bool resumed = ch * KC < k_done;  // chunk already opened (barrier passed, next chunk issued) above
    if (!resumed) {
      MFN_WAIT_VM(0);      // chunk ch's DMA (issued one chunk of MFMA work ago) has landed for this wave
      MFN_WAIT_LGKM0();
      MFN_RAW_BARRIER();   // ... and for every wave; everyone is done reading the other buffer
      if (ch == 0) MFN_STAMP(p.timeline, 1);
      if (ch + 1 < nchunks) issue(ch + 1);
    }
    const float *abuf = lds + (ch & 1) * STAGE_F + kw * G::CHUNK_F;   // the chunk's weights (uniform; a_of picks the lane's)
    const int cp0 = cp_base + ch * KC;
    int k = resumed ? k_done - ch * KC : 0;
    if (dwgather) {
      const int nrow = max(0, min(KC, full_pairs - cp0));
      MFN_NOUNROLL
      for (; k < nrow; ++k) dwgather_pair(cp0 + k, abuf + (size_t)k * G::PAIR_W);
    }  // otherwise (arbitrary per-tap offsets) every pair of this chunk takes the per-tap path below
    MFN_NOUNROLL
    for (; k < KC; ++k)
      if (2 * (cp0 + k) < p.Cin) slow_pair(cp0 + k, abuf + (size_t)k * G::PAIR_W);  // padded pairs: zero weights, skip
  }

  MFN_STAMP(p.timeline, 2);
  MFN_STAMP_INFO(p.timeline, (staged ? 1 : 0) | (gtier ? 2 : 0) | ((gtier && !gbanded) || dwgather ? 4 : 0) | (fast ? 8 : 0));
  // ---- in-block K-slice reduction through LDS ----------------------------------------------------------
  if (KW > 1) {
    MFN_WAIT_LGKM0();
    MFN_RAW_BARRIER();  // staging buffers are dead from here on
    float *red = lds;   // [pt][kw-1][MT*16][64]
    if (kw != 0) {
      MFN_UNROLL
      for (int mt = 0; mt < MT; ++mt)
        MFN_UNROLL
        for (int r = 0; r < 16; ++r) red[(size_t)(((pt * (KW - 1) + (kw - 1)) * MT + mt) * 16 + r) * 64 + lane] = acc[mt][r];
    }
    __syncthreads();
    if (kw != 0) return;
    for (int k = 1; k < KW; ++k) {
      MFN_UNROLL
      for (int mt = 0; mt < MT; ++mt)
        MFN_UNROLL
        for (int r = 0; r < 16; ++r) acc[mt][r] += red[(size_t)(((pt * (KW - 1) + (k - 1)) * MT + mt) * 16 + r) * 64 + lane];
    }
  }

  // ---- epilogue.  D reg r of lane (j,half): filter row (r&3)+8*(r>>2)+4*half, pixel j ------------------
  const bool raw = p.ksb > 1;  // cross-block K split: raw partial sums, bias added by dc_reduce_kernel
  const bool ep = p.ep_mask || p.ep_add || p.ep_leaky;
  float *obase = (raw ? p.partial + (size_t)blockIdx.y * p.N * p.Cout * oplane : p.out) + (size_t)n * p.Cout * oplane;
  // 2-D tiles: transpose the 32x32 tile through this wave's (now idle) window ring so that a lane holds 4
  // adjacent pixels of one filter and writes them with one 16-byte store -- 4 stores per lane instead of 16
  // dword stores that each touch eight 32-byte segments (the store tail is issue-bound, MI355X_MICROARCH.md).
  constexpr bool kRingFree = (KW > 1 ? (size_t)PT * (KW - 1) * MT * 16 * 64 * 4 : 0) <= (size_t)2 * STAGE_F * 4;
  if (kRingFree && p.tile_w && p.vec_store && !(MFN_DC_ABLATE & 8)) {
    constexpr int TS = 40;  // row stride in floats: 16-byte aligned rows, the two lane halves on disjoint banks
    float *tr = lds + 2 * STAGE_F + wave * (3 * XW_F);
    const int quad = lane & 7, orow = lane >> 3;
    const int px0 = quad * 4;
    const int prow = p.tile_w == 16 ? px0 >> 4 : px0 >> 3, pcol = px0 & (p.tile_w - 1);
    const int oy = tile_ho0 + prow, ox = tile_wo0 + pcol;
    const bool tile_ok = tile < p.ntiles;
    const bool st_ok = tile_ok && oy < Ho;
    // What the stores add to the sums is requested HERE, ahead of the transposition, with clamped addresses and no per-lane
    // condition: written as `cond ? p.bias[o] : 0` inside the store loop every load sat in its own branch with a full wait
    // behind it (4 MT round trips per lane at the end of every block; with the matching epilogue 16 MT more, and a sigmoid
    // with its division per stored VALUE although the mask term is the same for all filters of a pixel).
    const bool epi = ep && !raw;
    float bq[MT][4];
    MFN_UNROLL
    for (int mt = 0; mt < MT; ++mt)
      MFN_UNROLL
      for (int i = 0; i < 4; ++i) bq[mt][i] = 0.f;
    if (p.bias && !raw) {
      MFN_UNROLL
      for (int mt = 0; mt < MT; ++mt)
        MFN_UNROLL
        for (int i = 0; i < 4; ++i) bq[mt][i] = p.bias[min(m0 + mt * 32 + i * 8 + orow, p.Cout - 1)];
    }
    const size_t pix0 = st_ok ? (size_t)n * oplane + (size_t)oy * Wo : 0;  // this lane's output row (clamped: loads only)
    int oxq[4];
    MFN_UNROLL
    for (int q = 0; q < 4; ++q) oxq[q] = st_ok ? min(ox + q, Wo - 1) : 0;
    float sg[4] = {1.f, 1.f, 1.f, 1.f};
    if (epi && p.ep_mask) {
      float mv[4];
      MFN_UNROLL
      for (int q = 0; q < 4; ++q) mv[q] = p.ep_mask[pix0 + oxq[q]];
      MFN_UNROLL
      for (int q = 0; q < 4; ++q) sg[q] = 1.f / (1.f + expf(-mv[q]));
    }
    const size_t obatch = st_ok ? (size_t)n * p.Cout * oplane + (size_t)oy * Wo : 0;
    MFN_UNROLL
    for (int mt = 0; mt < MT; ++mt) {
      float addv[4][4];
      if (epi && p.ep_add) {  // this filter tile's add terms travel while the tile is transposed
        MFN_UNROLL
        for (int i = 0; i < 4; ++i) {
          const size_t orow_off = obatch + (size_t)min(m0 + mt * 32 + i * 8 + orow, p.Cout - 1) * oplane;
          MFN_UNROLL
          for (int q = 0; q < 4; ++q) addv[i][q] = p.ep_add[orow_off + oxq[q]];
        }
      }
      MFN_WAIT_LGKM0();  // the previous tile's reads are done (wave-private buffer: no barrier needed)
      MFN_UNROLL
      for (int r = 0; r < 16; ++r) tr[((r & 3) + 8 * (r >> 2) + 4 * half) * TS + j] = acc[mt][r];
      MFN_WAIT_LGKM0();
      MFN_UNROLL
      for (int i = 0; i < 4; ++i) {
        const int ol = i * 8 + orow;
        const int o = m0 + mt * 32 + ol;
        float4 v = *reinterpret_cast<const float4 *>(tr + ol * TS + px0);
        float e[4] = {v.x + bq[mt][i], v.y + bq[mt][i], v.z + bq[mt][i], v.w + bq[mt][i]};
        if (epi) {  // dc_epilogue, value by value
          MFN_UNROLL
          for (int q = 0; q < 4; ++q) {
            if (p.ep_mask) e[q] = e[q] * sg[q];
            if (p.ep_add) e[q] = e[q] + addv[i][q];
            if (p.ep_leaky) e[q] = fmaxf(e[q], 0.1f * e[q]);
          }
        }
        if (st_ok && o < p.Cout) {
          float *dst = obase + (size_t)o * oplane + (size_t)oy * Wo + ox;
          if (ox + 3 < Wo) {
            mfn_store4_stream(dst, e[0], e[1], e[2], e[3], raw ? 0 : p.st_policy);  // partial sums are re-read: plain
          } else {
            MFN_UNROLL
            for (int q = 0; q < 4; ++q)
              if (ox + q < Wo) dst[q] = e[q];
          }
        }
      }
    }
  } else if (px_valid) {
    float *on = obase + (size_t)ho * Wo + wo;
    MFN_UNROLL
    for (int mt = 0; mt < MT; ++mt)
      MFN_UNROLL
      for (int r = 0; r < 16; ++r) {
        const int o = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (o < p.Cout && !((MFN_DC_ABLATE & 8) && acc[mt][r] != 123.f)) {
          float v = acc[mt][r] + ((p.bias && !raw) ? p.bias[o] : 0.f);
          if (ep && !raw)
            v = dc_epilogue(v, p.ep_mask, p.ep_add, p.ep_leaky, (size_t)n * oplane + (size_t)ho * Wo + wo,
                            ((size_t)n * p.Cout + o) * oplane + (size_t)ho * Wo + wo);
          on[(size_t)o * oplane] = v;
        }
      }
  }
  MFN_STAMP(p.timeline, 3);
}

template <int MT, int PT, int NW = 4>
inline size_t dc_lds_bytes() {
  constexpr int KW = NW / PT;
  constexpr int NI = (KW * DcGeom<MT, KW>::CH4 + NW * 64 - 1) / (NW * 64);
  const size_t stage = (size_t)2 * NI * NW * 64 * 16;
  const size_t red = KW > 1 ? (size_t)PT * (KW - 1) * MT * 16 * 64 * 4 : 0;
  const size_t xwin = (size_t)NW * 3 * (2 * 256) * 4;  // NW waves x 3 buffers x one channel-pair window (XW_F floats)
  return stage + xwin > red ? stage + xwin : red;
}

template <int MT, int PT, int NW = 4>
inline int dc_lds_launch(const DeformParams &p, hipStream_t stream, const char *name) {
  const int tiles = p.tile_w ? p.ntiles : cdiv(p.P, 32);
  const int bx = cdiv(tiles, PT);
  if (bx <= 0) return 0;
  return launch(name, dc_lds_kernel<MT, PT, NW>, dim3(bx, p.ksb, p.mgroups), dim3(NW * 64), dc_lds_bytes<MT, PT, NW>(),
                stream, p);
}

// cross-block K-split reduction: out = bias + sum_s partial[s], slices in index order (deterministic)
struct DcReduceParams {
  const float *partial; const float *bias; float *out; size_t total; int ksb, Cout; size_t oplane;
  const float *ep_mask, *ep_add; int ep_leaky;
};
__global__ __launch_bounds__(256) void dc_reduce_kernel(DcReduceParams p) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= p.total) return;
  float s = p.partial[i];
  for (int k = 1; k < p.ksb; ++k) s += p.partial[(size_t)k * p.total + i];
  const int o = (int)((i / p.oplane) % p.Cout);
  const size_t n_ = i / (p.oplane * p.Cout);
  p.out[i] = dc_epilogue(s + (p.bias ? p.bias[o] : 0.f), p.ep_mask, p.ep_add, p.ep_leaky, n_ * p.oplane + i % p.oplane, i);
}
inline int dc_reduce_launch(DcReduceParams rp, hipStream_t stream) {
  if (!rp.total) return 0;
  return launch("dc_reduce", dc_reduce_kernel, dim3((unsigned)((rp.total + 255) / 256)), dim3(256), 0, stream, rp);
}

inline int dc_pack_launch(PackParams pp, hipStream_t stream) {
  const size_t total = (size_t)pp.mgroups * pp.ncp_pad * pp.T * 2 * pp.RL;
  return launch("dc_pack_weights", dc_pack_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                stream, pp);
}

struct DcCopyParams { const float *src; float *dst; size_t n; };
__global__ __launch_bounds__(256) void dc_copy_kernel(DcCopyParams c) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < c.n) c.dst[i] = c.src[i];
}
inline int dc_copy_launch(const float *src, float *dst, size_t n, hipStream_t stream) {
  if (!n) return 0;
  return launch("dc_copy_weights", dc_copy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                DcCopyParams{src, dst, n});
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
  p.out[idx] = dc_epilogue(s + (p.bias ? p.bias[o] : 0.f), p.ep_mask, p.ep_add, p.ep_leaky,
                           (size_t)n * oplane + (size_t)ho * p.Wo + wo, idx);
}

inline int dc_generic_launch(const DeformParams &p, hipStream_t stream) {
  const size_t total = (size_t)p.N * p.Cout * p.Ho * p.Wo;
  if (!total) return 0;
  return launch("dc_generic", dc_generic_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, p);
}

// ---- offset builder (MaskFlownet.py:230) ----------------------------------------------------------------
struct OffsetsParams { const float *flow; float *offset; int N, H, W, taps; float scale, stride; int st_policy; };
__global__ __launch_bounds__(256) void offsets_from_flow_kernel(OffsetsParams p) {
  const size_t plane = (size_t)p.H * p.W;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;  // over N*2*plane
  if (idx >= (size_t)p.N * 2 * plane) return;
  const size_t n = idx / (2 * plane), r = idx - n * 2 * plane;
  const size_t t = r / plane, pix = r - t * plane;
  const float v = p.flow[idx] * p.scale / p.stride;
  float *o = p.offset + n * 2 * p.taps * plane + t * plane + pix;
  for (int k = 0; k < p.taps; ++k) mfn_store1_stream(o + (size_t)2 * k * plane, v, p.st_policy);
}
// the same with four pixels per lane (plane % 4 == 0, 16-byte aligned tensors): 9 x 16 bytes stored per lane
__global__ __launch_bounds__(256) void offsets_from_flow_v4_kernel(OffsetsParams p) {
  const size_t plane4 = (size_t)p.H * p.W / 4;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;  // over N*2*plane/4
  if (idx >= (size_t)p.N * 2 * plane4) return;
  const size_t n = idx / (2 * plane4), r = idx - n * 2 * plane4;
  const size_t t = r / plane4, q = r - t * plane4;
  const float4 f = reinterpret_cast<const float4 *>(p.flow)[idx];
  const float a = f.x * p.scale / p.stride, b = f.y * p.scale / p.stride, c = f.z * p.scale / p.stride, d = f.w * p.scale / p.stride;
  float *o = p.offset + (n * 2 * p.taps * plane4 + t * plane4 + q) * 4;
  for (int k = 0; k < p.taps; ++k) mfn_store4_stream(o + (size_t)2 * k * plane4 * 4, a, b, c, d, p.st_policy);
}
// gradient of the above: gflow[n][dir][pixel] (+)= scale / stride * sum over the taps of goffset[n][2 tap + dir][pixel]
struct OffsetsBwdParams { const float *goffset; float *gflow; int N, H, W, taps; float scale, stride; int req; };
__global__ __launch_bounds__(256) void offsets_from_flow_bwd_kernel(OffsetsBwdParams p) {
  const size_t plane = (size_t)p.H * p.W;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;  // over N*2*plane
  if (idx >= (size_t)p.N * 2 * plane) return;
  const size_t n = idx / (2 * plane), r = idx - n * 2 * plane;
  const size_t t = r / plane, pix = r - t * plane;
  const float *g = p.goffset + n * 2 * p.taps * plane + t * plane + pix;
  float sum = 0.f;
  for (int k = 0; k < p.taps; ++k) sum += g[(size_t)2 * k * plane];
  const float v = sum * (p.scale / p.stride);
  p.gflow[idx] = p.req == 3 ? p.gflow[idx] + v : v;
}
inline int offsets_from_flow_launch(OffsetsParams p, hipStream_t stream) {
  const size_t total = (size_t)p.N * 2 * p.H * p.W;
  if (!total) return 0;
  if (((size_t)p.H * p.W) % 4 == 0 && (((uintptr_t)p.flow | (uintptr_t)p.offset) & 15) == 0)
    return launch("offsets_from_flow_v4", offsets_from_flow_v4_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0,
                  stream, p);
  return launch("offsets_from_flow", offsets_from_flow_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                stream, p);
}

}  // namespace mfn
