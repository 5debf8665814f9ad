// dc_backward.h -- input + offset gradient of the 3x3 deformable convolution in the FORWARD's orientation (r02).
//
// The reference feeds one (dy, dx) to all nine taps (MaskFlownet.py:230), so a pixel's nine samples lie in one 4x4
// neighbourhood with separable weights (deform_conv.h, "regular" fast path).  dc_bwd_input_shared_kernel (backward.h)
// used that with lane = CHANNEL: geometry records broadcast through LDS, 18 cross-lane reductions per pixel for the offset
// gradient, strided 16-byte x loads from 32 channel planes per instruction, one wave per SIMD, ~4.5 atomic instructions
// per channel and 64 pixels -- 114 k cycles per 64 pixels x 32 channels of which 14 k are MFMA
// (profiles/r01f_bwd_phases.txt).  Here a lane is a PIXEL, as in the forward kernel; a block is an 8x16-pixel region
// (four waves, one 4x8 tile each) times 32 input channels:
//   * column gradients D_t[channel][pixel] = sum_o W[o][channel][t] * gout[o][pixel] on v_mfma_f32_32x32x2_f32, nine
//     accumulator tiles (one per tap); A = the weights in their NATURAL layout, streamed through LDS by DMA (a filter's
//     32-channel x 9-tap segment is 288 consecutive floats; lane (channel i, half) reads word 288*o + 9*i + t: the two
//     half-waves' bank sets {9i} and {9i + 32} are disjoint and complementary, no packing pass), B = gout, coalesced;
//   * the tap geometry of a lane's pixel lives in its registers (no LDS records, no broadcasts);
//   * phase A, offset gradient: lane (pixel j, half) holds 16 of the 32 channels; per channel pair the 12 x 20 source
//     window of the tile arrives by LDS-DMA (the forward's 3-deep ring), 16 ds_read_b32 give the 4x4 neighbourhood, the
//     18 coordinate-gradient terms are summed over the lane's channels IN THE LANE -- one cross-half add per tile
//     instead of 18 DPP reductions per pixel;
//   * phase B, input gradient: the block keeps ONE 14 x 24 LDS plane per channel for its whole region.  A wave adds its
//     pixels' contributions with plain read-add-write, neighbourhood cell by neighbourhood cell: for a fixed cell (u, v)
//     the 32 lanes of a half-wave are 32 different pixels and hit distinct plane cells whenever the pixel -> cell map of
//     the tile is injective (checked once per tile through LDS; otherwise four turns by pixel parity), the LDS pipe keeps
//     a wave's accesses in order, and the two half-waves work on different channels.  LDS float atomics would cost ~170
//     cycles per wave instruction against ~6 for read + write (tools/ubench/atomic_patterns.hip).  The four waves never
//     meet in a plane: wave w's MFMA rows are the channels rotated by 8w, so at any step the waves hold different
//     channels (a block barrier every four steps keeps them within the rotation).  Four channels' chains are interleaved
//     to cover the LDS round trips.
//   * the merged planes are flushed to gx once per block with fp32 atomics: ~4 wave instructions per channel and 128
//     pixels -- global atomics cost ~0.35 ns per wave INSTRUCTION chip-wide however few lanes are active (same
//     microbenchmark), so what matters is how few there are.
// Tiles that do not qualify (per-tap offsets, irregular floors, windows that do not fit, W % 4 != 0) keep flag 0 and
// are done pixel by pixel by dc_bwd_input_tile_kernel (backward.h), which skips the pixels of flagged tiles.
#pragma once
#include "backward.h"
#include "deform_conv.h"

namespace mfn {

constexpr int DCP_KO = 8;                        // filters per weight chunk = 4 k-steps of 9 MFMAs
constexpr int DCP_ROWF = 288;                    // floats of one filter's segment: 32 channels x 9 taps
constexpr int DCP_WNI = 3;                       // weight DMA instructions per thread and chunk: 8 x 72 items in 768 slots
constexpr int DCP_STAGE_F = DCP_WNI * 256 * 4;   // floats per weight stage buffer
constexpr int DCP_XW_F = 512;                    // floats of a channel-pair source window: 2 x 12 x 20 = 480 used
constexpr int DCP_ROWS = 12, DCP_COLS = 20;      // source window of a 4x8 tile
constexpr int DCP_PR = 20, DCP_PC = 32;          // gx plane of the 8x16 region (bench flows, level 2: 99.9 % of the regions fit; 14 x 24: 60 %)
constexpr int DCP_PLANE = DCP_PR * DCP_PC + 8;   // plane stride: planes 4 apart (the two half-waves) sit 32 banks apart
constexpr int DCP_EXCH = 32;                     // ints of the setup exchange
constexpr int DCP_STASH = 19;                    // words a lane parks in LDS over phase A (what only phase B / the end needs)
// K loop and phase A: two weight stages + four x-window rings (+ exchange); phase B reuses the same memory for the planes
constexpr size_t dc_bwd_pix_lds_bytes() {
  const size_t a = (size_t)2 * DCP_STAGE_F + 4 * 3 * DCP_XW_F, b = (size_t)32 * DCP_PLANE;
  return ((a > b ? a : b) + DCP_EXCH + 4 * DCP_STASH * 64) * sizeof(float);
}

struct DcBwdPParams {
  const float *gout, *x, *offset, *w;
  float *gx, *goffset;
  int *flags;                    // per 4x8 pixel tile [n][cdiv(H,4)][cdiv(W,8)]: 1 = done here, 0 = left to dc_bwd_input_tile_kernel
  int N, Cin, H, W, Cout, ph, pw;
  int rx, ry;                    // 8x16 regions per row / column of an image
  float inv_rpi, inv_rx;
  int req_x, req_offset;
  int xcd;
  unsigned long long *timeline;  // measurement only: per block {setup, MFMA, phase A, phase B + flush} shader cycles of wave 0
};

__global__ __launch_bounds__(256, 1) void dc_bwd_input_pix_kernel(DcBwdPParams p) {
  constexpr int T = 9, KO = DCP_KO, NI = DCP_WNI, ROWS = DCP_ROWS, COLS = DCP_COLS, XW_NI = 2;
  constexpr int PR = DCP_PR, PC = DCP_PC, PL = DCP_PLANE;
  MFN_DYN_SHARED(float, lds);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = MFN_UNIFORM(tid >> 6);
  const int half = lane >> 5, j = lane & 31;
  constexpr int LDS_A = 2 * DCP_STAGE_F + 4 * 3 * DCP_XW_F, LDS_B = 32 * DCP_PLANE;
  int *exch = reinterpret_cast<int *>(lds + (LDS_A > LDS_B ? LDS_A : LDS_B));  // [4 waves][8]
  float *xwin = lds + 2 * DCP_STAGE_F + wave * (3 * DCP_XW_F);                  // this wave's three pair windows (ring)
  // Register diet (256 per wave at two waves per SIMD, 144 of them accumulators; a spill would also break the counted
  // vmcnt waits of phase A): what only phase B or the final store needs waits in LDS, word k of lane l at [k][l]
  float *stash = reinterpret_cast<float *>(exch) + DCP_EXCH + wave * (DCP_STASH * 64) + lane;
  const unsigned long long tk0 = MFN_CYCLES();

  const int H = p.H, W = p.W;
  const size_t plane = (size_t)H * W;
  const int bx = p.xcd ? (int)mfn_xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int cb = blockIdx.y * 32;
  // the rotation: MFMA row i of wave w is channel cb + ((i + 8w) & 31)
  const int rot = 8 * wave;

  // ---- weights of this channel block: chunk ch = filters [ch*KO, ch*KO + KO), their 288-float segments as they are
  const unsigned rowbytes = (unsigned)p.Cin * 36u;
  const mfn_rsrc_t wrsrc = mfn_make_rsrc(p.w + (size_t)cb * T, (unsigned)(((size_t)p.Cout * p.Cin - cb) * T * 4));
  unsigned voff[NI];
  MFN_UNROLL
  for (int i = 0; i < NI; ++i) {
    const int it = (i * 4 + wave) * 64 + lane;
    const int row = it / 72, seg = it - row * 72;
    voff[i] = row < KO ? (unsigned)row * rowbytes + (unsigned)seg * 16u : 0xFFFFFF00u;
  }
  auto issue_w = [&](int ch) {
    float *buf = lds + (ch & 1) * DCP_STAGE_F;
    const unsigned soff = (unsigned)ch * (unsigned)KO * rowbytes;
    MFN_UNROLL
    for (int i = 0; i < NI; ++i) mfn_dma16_so(wrsrc, buf + (i * 4 + wave) * 256, voff[i], soff);
  };
  const int nchunks = (p.Cout + KO - 1) / KO;
  issue_w(0);

  // ---- this lane's pixel: region -> (image, region row, region column); wave w = tile (w >> 1, w & 1) of the region
  int n, ho, wo, tyi, txi;
  {
    const int rpi = p.ry * p.rx;
    auto divmod = [](int a, int b, float inv_b, int &q, int &r) {
      q = (int)((float)a * inv_b);
      r = a - q * b;
      if (r < 0) { --q; r += b; }
      if (r >= b) { ++q; r -= b; }
    };
    int rt, ry_, rx_;
    divmod(bx, rpi, p.inv_rpi, n, rt);
    divmod(rt, p.rx, p.inv_rx, ry_, rx_);
    tyi = 2 * ry_ + (wave >> 1);
    txi = 2 * rx_ + (wave & 1);
    ho = tyi * 4 + (j >> 3);
    wo = txi * 8 + (j & 7);
  }
  const int ty4 = (H + 3) >> 2, tx8 = (W + 7) >> 3;
  const bool tile_ok = tyi < ty4 && txi < tx8;
  const bool px_valid = tile_ok && ho < H && wo < W;
  ho = min(ho, H - 1);
  wo = min(wo, W - 1);
  const size_t pix = (size_t)ho * W + wo;
  const int h_in = ho - p.ph, w_in = wo - p.pw;

  // gout of the first chunk travels while the geometry is computed
  const float *gptr = p.gout + (size_t)n * p.Cout * plane + pix;
  float gb0[KO / 2], gb1[KO / 2];  // gout values of the current / next chunk (static ring: the chunk loop is unrolled by two)
  auto load_g = [&](int ch, float (&g)[KO / 2]) {
    MFN_UNROLL
    for (int kk = 0; kk < KO / 2; ++kk) {
      const int o = ch * KO + 2 * kk + half;
      g[kk] = gptr[(size_t)min(o, p.Cout - 1) * plane];
    }
  };
  load_g(0, gb0);

  // ---- tap geometry (deform_conv.h: dc_axis; backward.h: dcs_axis) ---------------------------------------------------
  float geo[DCS_GS];
  float vyf[3], vxf[3];
  bool ok = true;
  int lo0y, lo0x;
  {
    const float *op = p.offset + (size_t)n * 2 * T * plane + pix;
    const float oh = op[0], ow = op[plane];
    MFN_UNROLL
    for (int t = 1; t < T; ++t) ok = ok && (op[(size_t)(2 * t) * plane] == oh) && (op[(size_t)(2 * t + 1) * plane] == ow);
    lo0y = (int)fminf(fmaxf(floorf(oh), -1.0e6f), 1.0e6f);
    lo0x = (int)fminf(fmaxf(floorf(ow), -1.0e6f), 1.0e6f);
    dcs_axis(oh, h_in, H, lo0y, geo, DCS_AY, DCS_BY, DCS_FH0, DCS_FH1, vyf, ok);
    dcs_axis(ow, w_in, W, lo0x, geo, DCS_AX, DCS_BX, DCS_FW0, DCS_FW1, vxf, ok);
  }
  const int ly0 = h_in + lo0y, lx0 = w_in + lo0x;  // first line / column of the 4x4 neighbourhood (may lie outside)
  // clamped lines of the neighbourhood (where x is read; clamped duplicates carry zero weight) and the bounding boxes.
  // Lanes outside the image / past the last tile duplicate a valid pixel: they take part in the boxes.
  int iy[4], ix[4];
  MFN_UNROLL
  for (int m = 0; m < 4; ++m) { iy[m] = min(max(ly0 + m, 0), H - 1); ix[m] = min(max(lx0 + m, 0), W - 1); }
  const int wr0 = mfn_wave_min_i32(iy[0]), rhi = mfn_wave_max_i32(iy[3]);
  const int wc0 = mfn_wave_min_i32(ix[0]) & ~3, chi = mfn_wave_max_i32(ix[3]);
  const int gylo = mfn_wave_min_i32(ly0), gyhi = mfn_wave_max_i32(ly0) + 3;
  const int gxlo = mfn_wave_min_i32(lx0), gxhi = mfn_wave_max_i32(lx0) + 3;
  bool cand = tile_ok && (__all(ok || !px_valid) != 0) && (W % 4 == 0) && (rhi - wr0 < ROWS) && (chi - wc0 < COLS) &&
              (gyhi - gylo < ROWS) && (gxhi - gxlo < COLS);
  // pixel -> cell map of the tile: injective (one turn per neighbourhood cell), injective within the four pixel-parity
  // classes (four turns), or neither (left to the tile kernel).  The map lives in this wave's (still idle) window ring.
  int mode = 1;
  const int cls = (j & 1) | (((j >> 3) & 1) << 1);
  if (cand && p.req_x) {
    int *map = reinterpret_cast<int *>(xwin);
    const int key = (ly0 - gylo) * COLS + (lx0 - gxlo);
    const bool mine = half == 0 && px_valid;
    if (mine) map[key] = j;
    MFN_WAIT_LGKM0();
    const bool inj = __all(!mine || map[key] == j) != 0;
    MFN_WAIT_LGKM0();
    if (!inj) {
      bool okc = true;
      MFN_UNROLL
      for (int k = 0; k < 4; ++k) {
        if (mine && cls == k) map[key] = j;
        MFN_WAIT_LGKM0();
        okc = okc && (!mine || cls != k || map[key] == j);
        MFN_WAIT_LGKM0();
      }
      mode = 4;
      cand = __all(okc) != 0;
    }
  }
  mode = MFN_UNIFORM(mode);
  // ---- the region's plane window: bounding box over the candidate waves ------------------------------------------------
  if (lane == 0) {
    int *e = exch + wave * 8;
    e[0] = cand ? 1 : 0; e[1] = gylo; e[2] = gyhi; e[3] = gxlo; e[4] = gxhi;
  }
  MFN_LDS_BARRIER();
  int py0 = 1 << 28, py1 = -(1 << 28), px0 = 1 << 28, px1 = -(1 << 28);
  MFN_UNROLL
  for (int w2 = 0; w2 < 4; ++w2) {
    const int *e = exch + w2 * 8;
    if (e[0]) { py0 = min(py0, e[1]); py1 = max(py1, e[2]); px0 = min(px0, e[3]); px1 = max(px1, e[4]); }
  }
  py0 = MFN_UNIFORM(py0); py1 = MFN_UNIFORM(py1); px0 = MFN_UNIFORM(px0); px1 = MFN_UNIFORM(px1);
  const bool region_fits = (py1 - py0 < PR) && (px1 - px0 < PC);
  const bool fast = MFN_UNIFORM((int)(cand && region_fits)) != 0;
  if (tile_ok && lane == 0) p.flags[((size_t)n * ty4 + tyi) * tx8 + txi] = fast ? 1 : 0;  // every channel block: same value

  // ---- source-window plumbing (as the forward's): slot -> (channel of the pair, row, float4 column) ---------------------
  const mfn_rsrc_t xrsrc = mfn_make_rsrc(p.x, (unsigned)((size_t)p.N * p.Cin * plane * 4));
  unsigned xvoff[XW_NI];
  MFN_UNROLL
  for (int i = 0; i < XW_NI; ++i) {
    const int slot = i * 64 + lane;
    const int chs = slot / 60, rem = slot - chs * 60;
    const int row = rem / 5, c4 = rem - row * 5;
    const int r = wr0 + row, c = wc0 + 4 * c4;
    // the pair of step r: the channels of MFMA rows q and q + 4 (the two half-waves)
    xvoff[i] = (fast && chs < 2 && r <= H - 1 && c <= W - 4)
                   ? (unsigned)(((size_t)n * p.Cin * plane + (size_t)(4 * chs) * plane + (size_t)r * W + c) * 4)
                   : 0xFFFFFF00u;
  }
  int lrow[4], lcol[4];  // LDS float offset of neighbourhood value (m, q) inside a pair window: lrow[m] + lcol[q]
  MFN_UNROLL
  for (int m = 0; m < 4; ++m) { lrow[m] = half * (ROWS * COLS) + (iy[m] - wr0) * COLS; lcol[m] = ix[m] - wc0; }
  // D row r of the MFMA tile (lane half h): row (r&3) + 8*(r>>2) + 4h = channel cb + ((row + rot) & 31)
  auto row_q = [](int r) { return (r & 3) + 8 * (r >> 2); };
  auto issue_x = [&](int r, int buf) {
    const int c0 = cb + ((row_q(r) + rot) & 31);  // its partner is c0 + 4 (never wraps: row_q + rot is 0..3 mod 8)
    // channels past Cin (ragged last block) are masked in the sums: whatever lies there (the next image, or zeros past the
    // end of the buffer) is fetched and dropped
    const unsigned soff = (unsigned)((size_t)min(c0, p.Cin - 1) * plane * 4);
    MFN_UNROLL
    for (int i = 0; i < XW_NI; ++i) mfn_dma16_so(xrsrc, xwin + buf * DCP_XW_F + i * 256, xvoff[i], soff);
  };
  if (fast && p.req_offset) { issue_x(0, 0); issue_x(1, 1); issue_x(2, 2); }
  {
    // forward weights with the taps' validity folded in (an invalid tap row / column contributes nothing), the lane's
    // cell of neighbourhood corner (0, 0) in a plane, and the validity factors of the offset gradient
    MFN_UNROLL
    for (int i = 0; i < 3; ++i) {
      const float vy = px_valid ? vyf[i] : 0.f, vx = px_valid ? vxf[i] : 0.f;
      stash[(0 + i) * 64] = vy * geo[DCS_AY + i];
      stash[(3 + i) * 64] = vy * geo[DCS_BY + i];
      stash[(6 + i) * 64] = vx * geo[DCS_AX + i];
      stash[(9 + i) * 64] = vx * geo[DCS_BX + i];
      stash[(13 + i) * 64] = vy;
      stash[(16 + i) * 64] = vx;
    }
    reinterpret_cast<int *>(stash)[12 * 64] = (ly0 - py0) * PC + (lx0 - px0);
  }
  const unsigned long long tk1 = MFN_CYCLES();

  // ---- column gradients of all nine taps: K = filters -------------------------------------------------------------------
  f32x16 acc[T];
  MFN_UNROLL
  for (int t = 0; t < T; ++t)
    MFN_UNROLL
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  auto chunk = [&](int ch, auto buf_c, const float (&cur)[KO / 2], float (&nxt)[KO / 2]) {
    constexpr int BUF = decltype(buf_c)::value;
    MFN_WAIT_VM(0);       // chunk ch (and this lane's gout values for it) landed ...
    MFN_LDS_BARRIER();    // ... for every wave; everyone is past chunk ch - 1
    if (ch + 1 < nchunks) {
      issue_w(ch + 1);
      load_g(ch + 1, nxt);
    }
    if (fast) {
      const float *ap = lds + BUF * DCP_STAGE_F + half * DCP_ROWF + ((j + rot) & 31) * T;
      MFN_UNROLL
      for (int kk = 0; kk < KO / 2; ++kk) {
        const bool o_ok = ch * KO + 2 * kk + half < p.Cout;
        const float bv = (o_ok && px_valid) ? cur[kk] : 0.f;
        MFN_UNROLL
        for (int t = 0; t < T; ++t) acc[t] = MFN_MFMA_32x32x2(ap[kk * 2 * DCP_ROWF + t], bv, acc[t]);
      }
    }
  };
  for (int ch = 0; ch < nchunks; ch += 2) {
    chunk(ch, DcInt<0>{}, gb0, gb1);
    if (ch + 1 < nchunks) chunk(ch + 1, DcInt<1>{}, gb1, gb0);
  }
  unsigned long long tk2 = 0, tk3 = 0;
  if (p.timeline) { MFN_OPAQUE(acc[0][0]); MFN_OPAQUE(acc[8][15]); tk2 = MFN_CYCLES(); }

  // ---- phase A: offset gradient ------------------------------------------------------------------------------------------
  if (fast && p.req_offset) {
    float sh[T], sw[T];
    MFN_UNROLL
    for (int t = 0; t < T; ++t) sh[t] = sw[t] = 0.f;
    auto step_a = [&](auto r_c) {
      constexpr int r = decltype(r_c)::value;
      constexpr int BX = r % 3;
      constexpr int newer = (15 - r) < 2 ? (15 - r) : 2;   // windows issued after window r that may still fly
      MFN_WAIT_VM(newer * XW_NI);
      float X[4][4];
      const float *xb = xwin + BX * DCP_XW_F;
      MFN_UNROLL
      for (int m = 0; m < 4; ++m)
        MFN_UNROLL
        for (int q = 0; q < 4; ++q) X[m][q] = xb[lrow[m] + lcol[q]];
      MFN_WAIT_LGKM0();
      if (r + 3 < 16) issue_x(r + 3, BX);
      const bool c_ok = cb + ((row_q(r) + rot) & 31) + 4 * half < p.Cin;
      float DV[3][4], DH[4][3];
      MFN_UNROLL
      for (int i = 0; i < 3; ++i)
        MFN_UNROLL
        for (int q = 0; q < 4; ++q) DV[i][q] = X[i + 1][q] - X[i][q];
      MFN_UNROLL
      for (int m = 0; m < 4; ++m)
        MFN_UNROLL
        for (int q = 0; q < 3; ++q) DH[m][q] = X[m][q + 1] - X[m][q];
      MFN_UNROLL
      for (int i = 0; i < 3; ++i)
        MFN_UNROLL
        for (int q = 0; q < 3; ++q) {
          const int t = 3 * i + q;
          const float cg = c_ok ? acc[t][r] : 0.f;
          // d/dh: fw0*(v21-v11) + fw1*(v22-v12);  d/dw: fh0*(v12-v11) + fh1*(v22-v21)   (deformable_col2im_coord)
          const float th = geo[DCS_FW0 + q] * DV[i][q] + geo[DCS_FW1 + q] * DV[i][q + 1];
          const float tw = geo[DCS_FH0 + i] * DH[i][q] + geo[DCS_FH1 + i] * DH[i + 1][q];
          sh[t] = fmaf(th, cg, sh[t]);
          sw[t] = fmaf(tw, cg, sw[t]);
        }
    };
    step_a(DcInt<0>{}); step_a(DcInt<1>{}); step_a(DcInt<2>{}); step_a(DcInt<3>{});
    step_a(DcInt<4>{}); step_a(DcInt<5>{}); step_a(DcInt<6>{}); step_a(DcInt<7>{});
    step_a(DcInt<8>{}); step_a(DcInt<9>{}); step_a(DcInt<10>{}); step_a(DcInt<11>{});
    step_a(DcInt<12>{}); step_a(DcInt<13>{}); step_a(DcInt<14>{}); step_a(DcInt<15>{});
    // the two half-waves hold the other 16 channels of the same pixel; half 0 writes d/dh, half 1 d/dw
    MFN_UNROLL
    for (int t = 0; t < T; ++t) {
      const float h2 = sh[t] + __shfl_xor(sh[t], 32), w2 = sw[t] + __shfl_xor(sw[t], 32);
      const float m9 = stash[(13 + t / 3) * 64] * stash[(16 + t % 3) * 64];
      const float v = (half ? w2 : h2) * m9;
      float *dst = p.goffset + ((size_t)n * 2 * T + 2 * t + half) * plane + pix;
      if (px_valid) {
        if (gridDim.y == 1) *dst += v;       // zero-filled (write) or the caller's values (add); nobody else writes this pixel
        else if (v != 0.f) atomicAdd(dst, v);
      }
    }
  }
  if (p.timeline) tk3 = MFN_CYCLES();
  if (!p.req_x) return;   // uniform

  // ---- phase B: input gradient -----------------------------------------------------------------------------------------
  MFN_WAIT_VM(0);
  MFN_LDS_BARRIER();      // the stage buffers and window rings are dead: the planes take their place
  for (int e = tid; e < 32 * PL / 4; e += 256) reinterpret_cast<float4 *>(lds)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
  MFN_LDS_BARRIER();
  {
    float ay[3], by[3], ax[3], bxw[3];
    MFN_UNROLL
    for (int i = 0; i < 3; ++i) {
      ay[i] = stash[(0 + i) * 64]; by[i] = stash[(3 + i) * 64]; ax[i] = stash[(6 + i) * 64]; bxw[i] = stash[(9 + i) * 64];
    }
    const int cell0 = reinterpret_cast<const int *>(stash)[12 * 64];  // neighbourhood corner (0, 0) in a plane
    auto group_b = [&](auto g_c) {
      constexpr int g = decltype(g_c)::value;   // steps 4g .. 4g + 3: MFMA rows 8g + c + 4 half, c = 0..3
      float *pl[4];
      MFN_UNROLL
      for (int c = 0; c < 4; ++c) pl[c] = lds + (size_t)((8 * g + c + 4 * half + rot) & 31) * PL + cell0;
      auto rounds = [&]() {
        float R[4][2][4];  // per chain: the x-folded rows u - 1 and u
        MFN_UNROLL
        for (int u = 0; u < 4; ++u) {
          if (u < 3) {
            MFN_UNROLL
            for (int c = 0; c < 4; ++c) {
              const float c0 = acc[3 * u][4 * g + c], c1 = acc[3 * u + 1][4 * g + c], c2 = acc[3 * u + 2][4 * g + c];
              float *Ru = R[c][u & 1];
              Ru[0] = c0 * ax[0];
              Ru[1] = fmaf(c0, bxw[0], c1 * ax[1]);
              Ru[2] = fmaf(c1, bxw[1], c2 * ax[2]);
              Ru[3] = c2 * bxw[2];
            }
          }
          MFN_UNROLL
          for (int v = 0; v < 4; ++v) {
            float o[4];
            MFN_UNROLL
            for (int c = 0; c < 4; ++c) o[c] = pl[c][u * PC + v];
            MFN_UNROLL
            for (int c = 0; c < 4; ++c) {
              float G;
              if (u == 0) G = ay[0] * R[c][0][v];
              else if (u == 3) G = by[2] * R[c][0][v];   // row 2 sits in slot 2 & 1 = 0
              else G = fmaf(by[u - 1], R[c][(u - 1) & 1][v], ay[u] * R[c][u & 1][v]);
              pl[c][u * PC + v] = o[c] + G;
            }
            MFN_COMPILER_FENCE();  // the next cell's reads are ISSUED after this cell's writes: another lane's cell (u, v)
                                   // is this lane's cell (u', v'); the in-order LDS pipe then orders them
          }
        }
      };
      if (fast) {
        // (emulation: free-running lanes take the whole sequence one lane at a time, mfn_rt.h)
        if (mode == 1) {
          if (px_valid) { MFN_EMU_LOCK(); rounds(); MFN_EMU_UNLOCK(); }
        } else {
          MFN_UNROLL
          for (int k = 0; k < 4; ++k) {
            int ck = cls;
            MFN_OPAQUE(ck);  // the four turns stay four ordered passes
            if (px_valid && ck == k) { MFN_EMU_LOCK(); rounds(); MFN_EMU_UNLOCK(); }
          }
        }
      }
      MFN_LDS_BARRIER();  // the waves stay within one group of each other: their rotated channel sets never meet
    };
    group_b(DcInt<0>{}); group_b(DcInt<1>{}); group_b(DcInt<2>{}); group_b(DcInt<3>{});
  }
  // ---- flush: wave w takes planes w, w + 4, ...; 64 consecutive cells of the touched rows per instruction
  if (py1 >= py0) {
    const int nrows = py1 - py0 + 1;
    constexpr int FS = (PR * PC + 63) / 64;  // 64-cell slices of a plane
    int goff[FS];
    MFN_UNROLL
    for (int s = 0; s < FS; ++s) {
      const int e = s * 64 + lane;
      const int row = e / PC, col = e - row * PC;
      const int yy = py0 + row, xx = px0 + col;
      goff[s] = (row < nrows && yy >= 0 && yy < H && xx >= 0 && xx < W) ? yy * W + xx : -1;
    }
    for (int pi = wave; pi < 32; pi += 4) {
      if (cb + pi >= p.Cin) break;  // uniform
      const float *pp = lds + (size_t)pi * PL;
      float *gim = p.gx + ((size_t)n * p.Cin + cb + pi) * plane;
      float fv[FS];
      MFN_UNROLL
      for (int s = 0; s < FS; ++s) fv[s] = pp[min(s * 64 + lane, PR * PC - 1)];
      MFN_UNROLL
      for (int s = 0; s < FS; ++s)
        if (s * 64 < nrows * PC && goff[s] >= 0 && fv[s] != 0.f) atomicAdd(gim + goff[s], fv[s]);
    }
  }
  if (p.timeline && tid == 0) {
    unsigned long long *b_ = p.timeline + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4;
    const unsigned long long tk4 = MFN_CYCLES();
    b_[0] = tk1 - tk0; b_[1] = tk2 - tk1; b_[2] = tk3 - tk2; b_[3] = tk4 - tk3;
  }
}

}  // namespace mfn
