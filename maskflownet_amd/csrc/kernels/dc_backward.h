// dc_backward.h -- input + offset gradient of the 3x3 deformable convolution in the FORWARD's orientation (r02).
//
// The reference feeds one (dy, dx) to all nine taps (MaskFlownet.py:230), so a pixel's nine samples lie in one 4x4
// neighbourhood with separable weights (deform_conv.h, "regular" fast path).  dc_bwd_input_shared_kernel (backward.h)
// used that with lane = CHANNEL: geometry records broadcast through LDS, 18 cross-lane reductions per pixel for the offset
// gradient, strided 16-byte x loads from 32 channel planes per instruction, ~4.5 atomic instructions per channel and 64
// pixels -- 114 k cycles per 64 pixels x 32 channels of which 14 k are MFMA (profiles/r01f_bwd_phases.txt).  Here a lane
// is a PIXEL, as in the forward kernel; a block is an 8x16-pixel region (four waves, one 4x8 tile each) times 16 input
// channels (32 in round 2: see the note at DCP_LDS_A), two blocks per CU:
//   * column gradients D_t[channel][pixel] = sum_o W[o][channel][t] * gout[o][pixel] on v_mfma_f32_32x32x2_f32, five
//     accumulator tiles: the MFMA's 32 rows are (16 channels) x (tap parity), tile tp holds taps 2 tp and 2 tp + 1;
//     A = the weights in their NATURAL layout, streamed through LDS by DMA (a filter's 16-channel x 9-tap segment is 144
//     consecutive floats; lane (row i, half) reads word 144*o + 9*channel(i) + 2 tp + (i >> 4), no packing pass),
//     B = gout, coalesced;
//   * the tap geometry of a lane's pixel lives in its registers (no LDS records, no broadcasts);
//   * phase A, offset gradient: lane (pixel j, half) holds 8 of the 16 channels; per channel pair a 16 x 24 source
//     window of the tile arrives by LDS-DMA (3-deep ring, as in the forward kernel), 16 ds_read_b32 give the 4x4
//     neighbourhood, the 18 coordinate-gradient terms are summed over the lane's channels IN THE LANE -- one cross-half
//     add per tile instead of 18 DPP reductions per pixel;
//   * phase B, input gradient (round 6 form): every WAVE keeps planes of 13 x 24 cells for its tile, a cell = the four channels
//     of an accumulator quad (16 bytes), placed around the tile's own neighbourhoods (or, where those spread too far, around what
//     the tile's centre pixel predicts).  A wave adds its pixels' contributions with plain read-add-write, neighbourhood cell by
//     neighbourhood cell, one ds_read_b128 + ds_write_b128 per cell and four channels: for a fixed cell (u, v) two lanes of a
//     half-wave (two pixels) hit the same plane cell only if their neighbourhoods start at the same cell, which a flow that
//     compresses the image produces (bench flows, level 2: 72 % of the tiles have such a pair); every pixel therefore gets a
//     TURN, its rank among the pixels of its tile that share its cell (found once per tile through LDS), and the cells are
//     walked once per turn (the second pixel of a pair hands its values to the first by lane shuffles instead).  The LDS pipe
//     keeps a wave's accesses in order and the two half-waves work on different channel quads; LDS float atomics would cost
//     ~170 cycles per wave instruction against ~6 for a read or write (tools/ubench/atomic_patterns.hip).  No barriers inside
//     the phase.  Neighbourhoods that leave the planes (a flow that tears) go to gx directly, 16 atomics per pixel and channel.
//     (Rounds 2-5: ONE 22 x 32 plane per channel for the block's whole region, a dword per instruction, the waves' channel sets
//     kept apart by a permutation of the accumulator rows and a block barrier per group of two channels per lane -- 24.3 k of the
//     kernel's 58 k cycles at level 2 against 21 k now; 107.6 / 76.4 / 47.1 / 35.7 us at levels 2..5 against 102.2 / 73.5 /
//     43.5 / 30.8, profiles/r06_dc_bwd_input_planes.txt.  The rows' permutation is still there: phase A's channel pairs use it.)
//   * the flush adds the four waves' planes cell by cell on its way to gx, once per block, with fp32 atomics over the union of the
//     waves' touched boxes only: global atomics cost ~0.35 ns per wave INSTRUCTION chip-wide however few lanes are active (same
//     microbenchmark), so what matters is how few instructions there are (64 cells of one channel each).  A union box of more
//     than DCP_WFS slices (a flow that tears the region apart) is flushed by every wave for itself.
// Tiles with per-tap offsets or irregular floors (never in the reference network) are done tap by tap and pixel by pixel
// by the same block, straight to memory, from the same column gradients (round 2: a second launch behind a flag buffer).
#pragma once
#include "backward.h"
#include "deform_conv.h"

namespace mfn {

constexpr int DCP_CB = 16;                       // input channels of a block
constexpr int DCP_ROWF = DCP_CB * 9;             // floats of one filter's segment: 16 channels x 9 taps
constexpr int DCP_KO = 16, DCP_KS = DCP_KO / 2;  // filters per weight chunk; its k-steps (two filters per fp32 MFMA)
// a chunk is 16 x 36 sixteen-byte items = 9 wave DMA instructions; every wave issues three (uniform wait counts), the
// three that carry nothing write zeros into a dump
constexpr int DCP_WNI = 3;
constexpr int DCP_STAGE_F = 9 * 256, DCP_DUMP_F = 3 * 256;
constexpr int DCP_ROWS = 16, DCP_COLS = 24;      // source window of a 4x8 tile
constexpr int DCP_XW_NI = 3;                     // 2 channels x 16 rows x 6 float4 = 192 slots = 3 wave DMA instructions
constexpr int DCP_XW_F = DCP_XW_NI * 256;        // floats of a channel-pair source window
constexpr int DCP_RD = 3;                        // source windows in flight (the wave's own ring)
constexpr int DCP_EXCH = 32;                     // ints of the block's touched-box exchange
// Wave-private gx planes (phase B), four channels to a cell: 13 x 24 cells around the 7 x 11 box of a tile whose flow is a shift.
// The planes take the block's WHOLE LDS (4 waves x 4 quads x 312 cells x 16 bytes = 79 872 bytes): what phase B needs from the
// stash and the exchange area is in registers by then.
#ifndef MFN_DCP_ABL   // measurement builds only (wrong results): 1 drop the neighbourhoods outside the planes, 2 no flush, 4 no walk
#define MFN_DCP_ABL 0
#endif
#ifndef MFN_DCP_PERM
#define MFN_DCP_PERM 1
#endif
#ifndef MFN_DCP_WFS
#define MFN_DCP_WFS 8
#endif
constexpr int DCP_WR = 13, DCP_WC = 24, DCP_WS = 24;   // rows, columns, row stride in cells
constexpr int DCP_WPL = DCP_WR * DCP_WS;
constexpr int DCP_WFS = MFN_DCP_WFS;                       // slices of the merged flush: a regular region's union box is 11 x 19 = 209 cells
constexpr int DCP_STASH = 21;                    // words a lane parks in LDS (what only phase B / the end needs)
// K loop and phase A: two weight stages + dump + four x-window rings; phase B reuses the same memory for the 16 planes.
// 80 000 bytes: TWO blocks per CU (round 3).  Round 2's block was 32 channels -- nine 32 x 32 accumulator tiles, 466
// registers, 119 KB of LDS, one wave per SIMD, and every phase of this kernel waits on its own LDS / memory round trips
// with nobody to fill them.  With 16 channels the MFMA's 32 rows are (channel, tap parity): five accumulator tiles
// (80 registers) hold the nine taps (the tenth slot is dropped), the block fits twice, and the two blocks of a CU are in
// different phases.
constexpr int DCP_LDS_A = 2 * DCP_STAGE_F + DCP_DUMP_F + 4 * DCP_RD * DCP_XW_F, DCP_LDS_B = 4 * DCP_CB * DCP_WPL;
constexpr int DCP_LDS_MAIN = DCP_LDS_A;
constexpr size_t dc_bwd_input_pix_lds_bytes() { return ((size_t)DCP_LDS_MAIN + DCP_EXCH + 4 * DCP_STASH * 64) * sizeof(float); }
static_assert(2 * dc_bwd_input_pix_lds_bytes() <= 160 * 1024, "two blocks per CU");
static_assert((size_t)DCP_LDS_B * sizeof(float) <= dc_bwd_input_pix_lds_bytes(), "the wave-private planes take the whole block's LDS");
static_assert(DCP_WFS % 2 == 0, "the merged flush takes its slices two at a time");
static_assert(DCP_WPL <= DCP_RD * DCP_XW_F, "the turn map lives in a wave's window ring");

struct DcBwdPParams {
  const float *gout, *x, *offset, *w;
  float *gx, *goffset;
  int N, Cin, H, W, Cout, ph, pw;
  int rx, ry;                    // 8x16 regions per row / column of an image
  float inv_rpi, inv_rx;
  int req_x, req_offset;
  int xcd;
  unsigned long long *timeline;  // measurement only: per block {setup | MFMA << 32, phase A, phase B, flush} shader cycles of wave 0
  int tl_detail;                 // ... or, instead of the last three, phase B: {before the groups, first group's fold + walk, its barrier wait}
  // flow mode (mfn_deform_conv_shared_bwd): every tap's offset IS flow[n][dir][pixel] * flow_scale / flow_stride -- no offset
  // tensor is read (offset == NULL), and d/dflow[n][dir][pixel] (req_offset: its request) takes the place of goffset
  const float *flow;
  float *gflow;
  float flow_scale, flow_stride;
};

__global__ __launch_bounds__(256, 2) void dc_bwd_input_pix_kernel(DcBwdPParams p) {
  constexpr int T = 9, TP = 5, KO = DCP_KO, KS = DCP_KS, NI = DCP_WNI, ROWS = DCP_ROWS, COLS = DCP_COLS, XW_NI = DCP_XW_NI;
  constexpr int RD = DCP_RD, CB = DCP_CB;
  constexpr int NST = 8;        // channel pairs of a lane: phase A's steps
  constexpr int CPG = 4;        // channels a lane walks at a time in phase B: an accumulator quad (two groups)
  constexpr int PRe = DCP_WR, PCe = DCP_WC, PSe = DCP_WS, PLe = DCP_WPL;   // a wave's gx planes: rows, columns, row stride, cells
  constexpr int WFS = DCP_WFS;
  MFN_DYN_SHARED(float, lds);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = MFN_UNIFORM(tid >> 6);
  const int half = lane >> 5, j = lane & 31;
  int *exch = reinterpret_cast<int *>(lds + DCP_LDS_MAIN);  // [4 waves][8]
  float *wdump = lds + 2 * DCP_STAGE_F;
  float *xwin = lds + 2 * DCP_STAGE_F + DCP_DUMP_F + wave * (RD * DCP_XW_F);  // this wave's three pair windows (ring)
  // What only phase B or the final store needs waits in LDS, word k of lane l at [k][l]
  float *stash = reinterpret_cast<float *>(exch) + DCP_EXCH + wave * (DCP_STASH * 64) + lane;
  const unsigned long long tk0 = MFN_CYCLES();

  const int H = p.H, W = p.W;
  const size_t plane = (size_t)H * W;
  const int bx = p.xcd ? (int)mfn_xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int cb = blockIdx.y * CB;
  // The MFMA's 32 rows: row i = (channel slot i & 15, tap parity i >> 4).  Slot sigma of wave w is channel
  // cb + (sigma ^ xw): the waves' slots are the channels permuted (bits 1 and 3 flipped; bit 2 -- the two half-waves -- stays: a
  // lane pair is channels c and c + 4).  Rounds 2-5 needed that to keep the waves apart in phase B's shared planes; with a wave's
  // own planes it only decides which channel pair a wave's phase-A windows carry at a step (no measurable effect either way:
  // MFN_DCP_PERM=0 builds time the same at every level, profiles/r06_dc_bwd_input_planes.txt).
  auto xw_of = [](int w) { return MFN_DCP_PERM ? ((w & 1) << 1) | ((w & 2) << 2) : 0; };
  const int xw = MFN_UNIFORM(xw_of(wave));
  // accumulator register 8 s + q of a lane (tap parity s, q = 0..7): slot (q & 3) + 8 (q >> 2) + 4 half
  auto slot_of = [&](int q, int h) { return (q & 3) + 8 * (q >> 2) + 4 * h; };
  auto chan_of = [&](int q, int h) { return slot_of(q, h) ^ xw; };  // channel inside the block

  // ---- weights of this channel block: chunk ch = filters [ch*KO, ch*KO + KO), their 144-float segments as they are
  const unsigned rowbytes = (unsigned)p.Cin * 36u;
  const mfn_rsrc_t wrsrc = mfn_make_rsrc(p.w + (size_t)cb * T, (unsigned)(((size_t)p.Cout * p.Cin - cb) * T * 4));
  unsigned voff[NI];
  MFN_UNROLL
  for (int i = 0; i < NI; ++i) {
    const int q = i * 4 + wave;                         // wave instruction 0..11; 0..8 carry the chunk
    const int it = q * 64 + lane;
    const int row = it / 36, seg = it - row * 36;
    voff[i] = q < 9 ? (unsigned)row * rowbytes + (unsigned)seg * 16u : 0xFFFFFF00u;
  }
  // blockIdx.z: a slice of the filters (coarse levels have fewer regions than the chip has block slots; everything after the
  // K loop is linear in the column gradients, so every slice adds its share of both gradients)
  const int nchunks_all = (p.Cout + KO - 1) / KO;
  const int cpz = (nchunks_all + (int)gridDim.z - 1) / (int)gridDim.z;
  const int ch_lo = (int)blockIdx.z * cpz;
  const int nchunks = max(0, min(nchunks_all, ch_lo + cpz) - ch_lo);
  auto issue_w = [&](int ch, int buf) {
    float *dst = lds + buf * DCP_STAGE_F;
    const unsigned soff = (unsigned)(ch_lo + ch) * (unsigned)KO * rowbytes;
    MFN_UNROLL
    for (int i = 0; i < NI; ++i) {
      const int q = i * 4 + wave;
      mfn_dma16_so(wrsrc, q < 9 ? dst + q * 256 : wdump + (q - 9) * 256, voff[i], soff);
    }
  };

  // ---- this lane's pixel: region -> (image, region row, region column); wave w = tile (w >> 1, w & 1) of the region
  int n, ho, wo, tyi, txi, ry0, rx0;
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
    ry0 = ry_ * 8; rx0 = rx_ * 16;
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

  // gout and weights of the first chunk travel while the geometry is computed
  const float *gptr = p.gout + (size_t)n * p.Cout * plane + pix;
  float gb0[KS], gb1[KS];  // gout values of two chunks (static ring: the chunk loop is unrolled by two)
  auto load_g = [&](int ch, float (&g)[KS]) {
    MFN_UNROLL
    for (int kk = 0; kk < KS; ++kk) {
      const int o = (ch_lo + ch) * KO + 2 * kk + half;
      g[kk] = gptr[(size_t)min(o, p.Cout - 1) * plane];
    }
  };
  if (nchunks > 0) { load_g(0, gb0); issue_w(0, 0); }

  // ---- tap geometry (deform_conv.h: dc_axis; backward.h: dcs_axis) ---------------------------------------------------
  float geo[DCS_GS];
  float vyf[3], vxf[3];
  bool ok = true;
  int lo0y, lo0x;
  const bool fm = p.flow != nullptr;  // uniform: the offsets come from the flow field (the forward's arithmetic, kernels/deform_conv.h)
  const float *offn = fm ? p.flow + (size_t)n * 2 * plane : p.offset + (size_t)n * 2 * T * plane;
  // offset of the tile's centre pixel (uniform): where its source window is placed -- and this pixel's own.
  // ALL requested before the first is used: written as `ok = ok && load == ...` hipcc made every load conditional on the one
  // before, twenty round trips one after the other (11 k of the 17 k cycles of this kernel's setup at level 2).
  float ctile[2];
  auto scaled = [&](float v) { return fm ? v * p.flow_scale / p.flow_stride : v; };
  auto centre_floor = [&](const float (&c)[2], int &fh, int &fw) {
    fh = MFN_UNIFORM((int)fminf(fmaxf(floorf(c[0]), -1.0e6f), 1.0e6f));
    fw = MFN_UNIFORM((int)fminf(fmaxf(floorf(c[1]), -1.0e6f), 1.0e6f));
  };
  {
    const float *o1 = offn + (size_t)min(tyi * 4 + 2, H - 1) * W + min(txi * 8 + 4, W - 1);
    const float *op = offn + pix;
    float cv[2] = {o1[0], o1[plane]};
    float ov[2 * T];
    ov[0] = op[0]; ov[1] = op[plane];
    if (!fm) {
      MFN_UNROLL
      for (int t = 2; t < 2 * T; ++t) ov[t] = op[(size_t)t * plane];
    }
    MFN_COMPILER_FENCE();
    ctile[0] = scaled(cv[0]); ctile[1] = scaled(cv[1]);
    const float oh = scaled(ov[0]), ow = scaled(ov[1]);
    if (!fm) {  // one offset for all nine taps?
      int same = 1;
      MFN_UNROLL
      for (int t = 1; t < T; ++t) same &= (int)(ov[2 * t] == oh) & (int)(ov[2 * t + 1] == ow);
      ok = same != 0;
    }
    lo0y = (int)fminf(fmaxf(floorf(oh), -1.0e6f), 1.0e6f);
    lo0x = (int)fminf(fmaxf(floorf(ow), -1.0e6f), 1.0e6f);
    dcs_axis(oh, h_in, H, lo0y, geo, DCS_AY, DCS_BY, DCS_FH0, DCS_FH1, vyf, ok);
    dcs_axis(ow, w_in, W, lo0x, geo, DCS_AX, DCS_BX, DCS_FW0, DCS_FW1, vxf, ok);
  }
  const int ly0 = h_in + lo0y, lx0 = w_in + lo0x;  // first line / column of the 4x4 neighbourhood (may lie outside)
  const bool fast = MFN_UNIFORM((int)(tile_ok && (__all(ok || !px_valid) != 0))) != 0;

  // ---- source window of the tile (phase A): 16 x 24 around the 7 x 11 box the centre pixel's offset predicts -----------
  int iy[4], ix[4];  // clamped lines of the neighbourhood (where x is read; clamped duplicates carry zero weight)
  MFN_UNROLL
  for (int m = 0; m < 4; ++m) { iy[m] = min(max(ly0 + m, 0), H - 1); ix[m] = min(max(lx0 + m, 0), W - 1); }
  int wr0, wc0;
  {
    int fh, fw;
    centre_floor(ctile, fh, fw);
    wr0 = tyi * 4 - p.ph + fh - (ROWS - 7) / 2;
    wc0 = (txi * 8 - p.pw + fw - (COLS - 11) / 2 + 2) & ~3;  // 16-byte aligned columns
  }
  const bool xin = iy[0] >= wr0 && iy[3] < wr0 + ROWS && ix[0] >= wc0 && ix[3] < wc0 + COLS;
  // all of the tile's neighbourhoods inside the window (else, rarely, this wave reads x from global memory)
  const bool xfit = MFN_UNIFORM(__all(xin || !px_valid)) != 0;
  const mfn_rsrc_t xrsrc = mfn_make_rsrc(p.x, (unsigned)((size_t)p.N * p.Cin * plane * 4));
  unsigned xvoff[XW_NI];
  MFN_UNROLL
  for (int i = 0; i < XW_NI; ++i) {
    const int slot = i * 64 + lane;
    const int chs = slot / 96, rem = slot - chs * 96;   // float4 slots: [channel 0/1][16 rows][6 float4]
    const int row = rem / 6, c4 = rem - row * 6;
    const int r = wr0 + row, c = wc0 + 4 * c4;
    // the pair of a step: the channels of the two half-waves, c and c + 4
    xvoff[i] = (r >= 0 && r <= H - 1 && c >= 0 && c <= W - 4)
                   ? (unsigned)(((size_t)n * p.Cin * plane + (size_t)(4 * chs) * plane + (size_t)r * W + c) * 4)
                   : 0xFFFFFF00u;  // outside the image: never read, the DMA writes zeros
  }
  int lrow[4], lcol[4];  // LDS float offset of neighbourhood value (m, q) inside a pair window: lrow[m] + lcol[q]
  MFN_UNROLL
  for (int m = 0; m < 4; ++m) {
    lrow[m] = half * (ROWS * COLS) + min(max(iy[m] - wr0, 0), ROWS - 1) * COLS;
    lcol[m] = min(max(ix[m] - wc0, 0), COLS - 1);
  }
  auto issue_xb = [&](int r) {
    // half 0's channel of step r; its partner is + 4 (bit 2 of the slot, which xw leaves alone).  Channels past Cin (ragged
    // last block) are masked in the sums: whatever lies there (the next image, or zeros past the end of the buffer) is
    // fetched and dropped
    const int c0 = cb + chan_of(r, 0);
    const unsigned soff = (unsigned)((size_t)min(c0, p.Cin - 1) * plane * 4);
    float *dst = xwin + (r % RD) * DCP_XW_F;
    MFN_UNROLL
    for (int i = 0; i < XW_NI; ++i) mfn_dma16_so(xrsrc, dst + i * 256, xvoff[i], soff);
  };

  // ---- gx plane of the region (phase B): 22 x 32 around the 11 x 19 box the centre pixel's offset predicts ------------
  int py0, px0;
  {
    int fh, fw;
    {   // this wave's planes: around the tile's neighbourhoods where they fit, else around what its centre pixel predicts
      centre_floor(ctile, fh, fw);
      py0 = tyi * 4 - p.ph + fh - (PRe - 7) / 2;
      px0 = txi * 8 - p.pw + fw - (PCe - 11) / 2;
      const int big = 1 << 28;
      const int ymin = mfn_wave_min_i32(px_valid ? ly0 : big), ymax = mfn_wave_max_i32(px_valid ? ly0 : -big);
      const int xmin = mfn_wave_min_i32(px_valid ? lx0 : big), xmax = mfn_wave_max_i32(px_valid ? lx0 : -big);
      if (ymax >= ymin && ymax - ymin + 4 <= PRe) py0 = ymin - (PRe - (ymax - ymin + 4)) / 2;   // uniform
      if (xmax >= xmin && xmax - xmin + 4 <= PCe) px0 = xmin - (PCe - (xmax - xmin + 4)) / 2;
      py0 = MFN_UNIFORM(py0); px0 = MFN_UNIFORM(px0);
    }
  }
  const int cry = ly0 - py0, crx = lx0 - px0;                 // the neighbourhood's corner (0, 0) in plane coordinates
  const bool inplane = cry >= 0 && cry + 3 < PRe && crx >= 0 && crx + 3 < PCe;
  // turn of this pixel: its rank among the tile's pixels whose neighbourhoods start at the same plane cell; -1: none
  // (outside the image / the plane).  The map lives in this wave's (still idle) window ring.
  int turn = -1, nturns = 0;
  int partner = -1, merged = 0;  // lane j' of the pixel whose contributions this lane adds to its own (or -1); any pair in the tile
  if (fast && p.req_x) {
    int *map = reinterpret_cast<int *>(xwin);
    const int key = cry * PSe + crx;
    bool pending = px_valid && inplane;
    while (__any(pending)) {
      if (pending) map[key] = j;   // both half-waves hold the same pixels: lane j and j + 32 write the same value
      MFN_WAIT_LGKM0();
      MFN_WAVE_SYNC_EMU();
      const bool won = pending && map[key] == j;
      MFN_WAIT_LGKM0();
      MFN_WAVE_SYNC_EMU();
      if (won) { turn = nturns; pending = false; }
      ++nturns;
    }
    nturns = MFN_UNIFORM(nturns);
    // The second pixel of a cell hands its contributions to the first (a lane shuffle per value, no ordering between
    // them) instead of walking the 16 cells in a turn of its own: pairs are the common case, the walk is the long pole.
    if (nturns >= 2) {
      if (turn == 0) map[key] = -1;
      MFN_WAIT_LGKM0();
      MFN_WAVE_SYNC_EMU();
      if (turn == 1) map[key] = j;
      MFN_WAIT_LGKM0();
      MFN_WAVE_SYNC_EMU();
      if (turn == 0) partner = map[key];
      MFN_WAIT_LGKM0();
      if (turn == 1) turn = -3; else if (turn >= 2) turn -= 1;
      nturns -= 1;
      merged = 1;
    }
    // touched box of the plane (flush), per wave; the block's is their union
    const bool mine = px_valid && inplane;
    const int big = 1 << 28;
    const int y0 = mfn_wave_min_i32(mine ? cry : big), y1 = mfn_wave_max_i32(mine ? cry + 3 : -big);
    const int x0 = mfn_wave_min_i32(mine ? crx : big), x1 = mfn_wave_max_i32(mine ? crx + 3 : -big);
    // (wave-private planes: the box in IMAGE coordinates, and where the wave's planes lie)
    if (lane == 0) {
      int *e = exch + wave * 8;
      const bool anyb = y1 >= y0;
      e[0] = anyb ? y0 + py0 : y0; e[1] = anyb ? y1 + py0 : y1;
      e[2] = anyb ? x0 + px0 : x0; e[3] = anyb ? x1 + px0 : x1;
      e[4] = py0; e[5] = px0;
    }
  } else if (lane == 0) {
    int *e = exch + wave * 8;
    e[0] = 1 << 28; e[1] = -(1 << 28); e[2] = 1 << 28; e[3] = -(1 << 28);
    e[4] = 0; e[5] = 0;
  }
  MFN_WAIT_LGKM0();
  // the first three source windows are requested behind the first weight chunk's barrier (K loop): requested here, the
  // chunk's vmcnt(0) would wait for them before the first MFMA
  const bool xdma = fast && p.req_offset && xfit;  // uniform
  {
    // forward weights with the taps' validity folded in (an invalid tap row / column contributes nothing), the lane's
    // cell of neighbourhood corner (0, 0) in a plane, its turn, and the validity factors of the offset gradient
    MFN_UNROLL
    for (int i = 0; i < 3; ++i) {
      const float vy = px_valid ? vyf[i] : 0.f, vx = px_valid ? vxf[i] : 0.f;
      stash[(0 + i) * 64] = vy;
      stash[(3 + i) * 64] = vx;
      stash[(6 + i) * 64] = vy * geo[DCS_AY + i];
      stash[(9 + i) * 64] = vy * geo[DCS_BY + i];
      stash[(12 + i) * 64] = vx * geo[DCS_AX + i];
      stash[(15 + i) * 64] = vx * geo[DCS_BX + i];
    }
    int *si = reinterpret_cast<int *>(stash);
    si[18 * 64] = cry * PSe + crx;
    si[19 * 64] = (px_valid && !inplane) ? -2 : turn;   // -2: the neighbourhood leaves the plane -> straight to gx
    si[20 * 64] = partner;
  }

  const unsigned long long tks = MFN_CYCLES();
  // ---- column gradients of all nine taps: K = filters.  Tile tp holds taps 2 tp (rows 0..15) and 2 tp + 1 (rows 16..31;
  // for tp = 4 that is "tap 9", the next channel's tap 0: never read from the accumulators) -----------------------------
  f32x16 acc[TP];
  MFN_UNROLL
  for (int t = 0; t < TP; ++t)
    MFN_UNROLL
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  // tap t of the lane's q-th channel
#define DCP_ACC(t, q) acc[(t) >> 1][(((t) & 1) << 3) | (q)]
  const int arow = half * DCP_ROWF + ((j & 15) ^ xw) * T + (j >> 4);  // this lane's row of A inside a k-step (two filters)
  auto chunk = [&](int ch, auto buf_c, const float (&cur)[KS], float (&nn)[KS]) {
    constexpr int BUF = decltype(buf_c)::value;
    // chunk ch and this lane's gout values for it have landed ...
    MFN_WAIT_VM(0);
    MFN_LDS_BARRIER();    // ... for every wave; everyone is past chunk ch - 1, whose buffer takes chunk ch + 1
    if (ch + 1 < nchunks) {
      load_g(ch + 1, nn);
      issue_w(ch + 1, BUF ^ 1);
    }
    if (BUF == 0 && ch == 0 && xdma) { issue_xb(0); issue_xb(1); issue_xb(2); }
    {   // (every tile: the tiles that do not qualify need the same column gradients)
      const float *ap = lds + BUF * DCP_STAGE_F + arow;
      float a[2][TP];
      MFN_UNROLL
      for (int t = 0; t < TP; ++t) a[0][t] = ap[2 * t];
      MFN_UNROLL
      for (int kk = 0; kk < KS; ++kk) {
        // the next k-step's weights are requested before this k-step's MFMAs issue
        if (kk + 1 < KS) {
          MFN_UNROLL
          for (int t = 0; t < TP; ++t) a[(kk + 1) & 1][t] = ap[(kk + 1) * 2 * DCP_ROWF + 2 * t];
        }
        const bool o_ok = (ch_lo + ch) * KO + 2 * kk + half < p.Cout;
        const float bv = (o_ok && px_valid) ? cur[kk] : 0.f;
        MFN_SCHED_BARRIER();
        MFN_UNROLL
        for (int t = 0; t < TP; ++t) acc[t] = MFN_MFMA_32x32x2(a[kk & 1][t], bv, acc[t]);
        MFN_SCHED_BARRIER();
      }
    }
  };
  for (int ch = 0; ch < nchunks; ch += 2) {
    chunk(ch, DcInt<0>{}, gb0, gb1);
    if (ch + 1 < nchunks) chunk(ch + 1, DcInt<1>{}, gb1, gb0);
  }
  if (nchunks == 0 && xdma) { issue_xb(0); issue_xb(1); issue_xb(2); }
  unsigned long long tk1 = 0, tk2 = 0, tk3 = 0;
  if (p.timeline) { MFN_OPAQUE(acc[0][0]); MFN_OPAQUE(acc[TP - 1][7]); tk1 = MFN_CYCLES(); }

  // ---- tiles that do not qualify (per-tap offsets, irregular floors: never in the reference network) ----------------------
  // Tap by tap and pixel by pixel, straight to memory (deformable_col2im / deformable_col2im_coord as the tap-by-tap kernel
  // of backward.h states them): the column gradients are the ones above.  Round 2 left these tiles to a second launch of
  // dc_bwd_input_tile_kernel behind a flag buffer -- 4 us per call for a kernel whose blocks all found nothing to do.
  if (tile_ok && !fast) {
    const float *op = p.offset + (size_t)n * 2 * T * plane + pix;
    float fsum_h = 0.f, fsum_w = 0.f;  // flow mode: the taps share the offset, their gradients add up
    auto tap = [&](auto t_c) {
      constexpr int t = decltype(t_c)::value, ti = t / 3, tj = t % 3;
      const float oh = fm ? scaled(offn[pix]) : op[(size_t)(2 * t) * plane];
      const float ow = fm ? scaled(offn[plane + pix]) : op[(size_t)(2 * t + 1) * plane];
      bool vh, vw;
      int hl, hh, wl, wh;
      float lh, lw;
      dc_axis(oh, h_in, ti, H, vh, hl, hh, lh);
      dc_axis(ow, w_in, tj, W, vw, wl, wh, lw);
      const bool valid = vh && vw && px_valid;
      const int gb = valid ? (h_in + hl) * W + (w_in + wl) : 0, dyg = valid ? (hh - hl) * W : 0, dxg = valid ? wh - wl : 0;
      const float g0 = (1.f - lh) * (1.f - lw), g1 = (1.f - lh) * lw, g2 = lh * (1.f - lw), g3 = lh * lw;
      // deformable_col2im_coord: 4 samples with MXNet's clamping, weights of d/dh and d/dw
      float ah = valid ? (float)(h_in + ti) + oh : 0.f, aw = valid ? (float)(w_in + tj) + ow : 0.f;
      int chl = (int)ah, cwl = (int)aw, chh, cwh;
      if (chl >= H - 1) { chh = chl = H - 1; ah = (float)chl; } else chh = chl + 1;
      if (cwl >= W - 1) { cwh = cwl = W - 1; aw = (float)cwl; } else cwh = cwl + 1;
      const int i11 = chl * W + cwl, sdh = (chh - chl) * W, sdw = cwh - cwl;
      const float fw0 = (float)(cwl + 1) - aw, fw1 = aw - (float)cwl, fh0 = (float)(chl + 1) - ah, fh1 = ah - (float)chl;
      float sh_ = 0.f, sw_ = 0.f;
      MFN_UNROLL
      for (int q = 0; q < 8; ++q) {
        const int ch = cb + chan_of(q, half);
        const float cg = (ch < p.Cin && valid) ? DCP_ACC(t, q) : 0.f;
        const size_t cofs = ((size_t)n * p.Cin + min(ch, p.Cin - 1)) * plane;
        if (p.req_x) {
          float *gim = p.gx + cofs + gb;
          const float c1 = g0 * cg, c2 = g1 * cg, c3 = g2 * cg, c4 = g3 * cg;
          if (c1 != 0.f) atomicAdd(gim, c1);
          if (c2 != 0.f) atomicAdd(gim + dxg, c2);
          if (c3 != 0.f) atomicAdd(gim + dyg, c3);
          if (c4 != 0.f) atomicAdd(gim + dyg + dxg, c4);
        }
        if (p.req_offset) {
          const float *im = p.x + cofs + i11;
          const float v11 = im[0], v12 = im[sdw], v21 = im[sdh], v22 = im[sdh + sdw];
          sh_ = fmaf(-fw0 * v11 - fw1 * v12 + fw0 * v21 + fw1 * v22, cg, sh_);
          sw_ = fmaf(-fh0 * v11 + fh0 * v12 - fh1 * v21 + fh1 * v22, cg, sw_);
        }
      }
      if (p.req_offset && px_valid) {
        if (fm) { fsum_h += sh_; fsum_w += sw_; }
        else {
          float *gof = p.goffset + ((size_t)n * 2 * T + 2 * t) * plane + pix;
          if (sh_ != 0.f) atomicAdd(gof, sh_);
          if (sw_ != 0.f) atomicAdd(gof + plane, sw_);
        }
      }
    };
    tap(DcInt<0>{}); tap(DcInt<1>{}); tap(DcInt<2>{}); tap(DcInt<3>{}); tap(DcInt<4>{});
    tap(DcInt<5>{}); tap(DcInt<6>{}); tap(DcInt<7>{}); tap(DcInt<8>{});
    if (fm && p.req_offset && px_valid) {
      const float ratio = p.flow_scale / p.flow_stride;
      float *gf = p.gflow + (size_t)n * 2 * plane + pix;
      if (fsum_h != 0.f) atomicAdd(gf, fsum_h * ratio);
      if (fsum_w != 0.f) atomicAdd(gf + plane, fsum_w * ratio);
    }
  }

  // ---- phase A: offset gradient ------------------------------------------------------------------------------------------
  float vt[T], vsum = 0.f;   // this lane's offset gradients (half 0: d/dh, half 1: d/dw)
  MFN_UNROLL
  for (int t = 0; t < T; ++t) vt[t] = 0.f;
  if (fast && p.req_offset) {
    float sh[T], sw[T];
    MFN_UNROLL
    for (int t = 0; t < T; ++t) sh[t] = sw[t] = 0.f;
    const unsigned xn0 = (unsigned)n * (unsigned)p.Cin * (unsigned)plane;  // (the rare wave whose window does not hold its neighbourhoods)
    auto gather = [&](auto dma_c, int r, const float *xb, float (&X)[4][4]) {
      if (decltype(dma_c)::value) {
        MFN_UNROLL
        for (int m = 0; m < 4; ++m)
          MFN_UNROLL
          for (int q = 0; q < 4; ++q) X[m][q] = xb[lrow[m] + lcol[q]];
      } else {
        // 32-bit element offsets from the (uniform) tensor base: the host admits N * Cin * H * W < 2^30 here; sixteen
        // 64-bit lane addresses were this kernel's largest register consumer
        const int c = min(cb + chan_of(r, half), p.Cin - 1);
        const unsigned cbase = xn0 + (unsigned)c * (unsigned)plane;
        MFN_UNROLL
        for (int m = 0; m < 4; ++m)
          MFN_UNROLL
          for (int q = 0; q < 4; ++q) X[m][q] = xb[cbase + (unsigned)(iy[m] * W + ix[q])];
      }
    };
    auto sums = [&](auto r_c, const float (&X)[4][4]) {
      constexpr int r = decltype(r_c)::value;
      const bool c_ok = cb + chan_of(r, half) < p.Cin;
      MFN_UNROLL
      for (int i = 0; i < 3; ++i)
        MFN_UNROLL
        for (int q = 0; q < 3; ++q) {
          const int t = 3 * i + q;
          const float cg = c_ok ? DCP_ACC(t, r) : 0.f;
          // d/dh: fw0*(v21-v11) + fw1*(v22-v12);  d/dw: fh0*(v12-v11) + fh1*(v22-v21)   (deformable_col2im_coord)
          const float th = geo[DCS_FW0 + q] * (X[i + 1][q] - X[i][q]) + geo[DCS_FW1 + q] * (X[i + 1][q + 1] - X[i][q + 1]);
          const float tw = geo[DCS_FH0 + i] * (X[i][q + 1] - X[i][q]) + geo[DCS_FH1 + i] * (X[i + 1][q + 1] - X[i + 1][q]);
          sh[t] = fmaf(th, cg, sh[t]);
          sw[t] = fmaf(tw, cg, sw[t]);
        }
    };
    // software pipeline: step r gathers the neighbourhood of pair r + 1, sums pair r from registers, then hands the
    // buffer it has just read to window r + 1 + RD.  DMA completion is in issue order: the waits count the newer windows
    // that may still fly.  (The K loop ended with every DMA of this wave landed except, without chunks, the three windows.)
    auto phase_a = [&](auto dma_c) {
      constexpr bool DMA = decltype(dma_c)::value;
      float Xa[4][4], Xb[4][4];
      if (DMA) MFN_WAIT_VM((RD - 1) * XW_NI);
      gather(dma_c, 0, DMA ? xwin : p.x, Xa);
      MFN_WAIT_LGKM0();
      if (DMA) issue_xb(RD);
      auto step_a = [&](auto r_c, float (&Xc)[4][4], float (&Xn)[4][4]) {
        constexpr int r = decltype(r_c)::value;
        if (r + 1 < NST) {
          constexpr int newer = (NST - 2 - r) < (RD - 1) ? (NST - 2 - r) : (RD - 1);   // windows r + 2 .. min(r + RD, NST - 1)
          if (DMA) MFN_WAIT_VM(newer * XW_NI);
          gather(dma_c, r + 1, DMA ? xwin + ((r + 1) % RD) * DCP_XW_F : p.x, Xn);
        }
        sums(r_c, Xc);
        if (r + 1 + RD < NST) {
          MFN_WAIT_LGKM0();
          if (DMA) issue_xb(r + 1 + RD);
        }
      };
      step_a(DcInt<0>{}, Xa, Xb); step_a(DcInt<1>{}, Xb, Xa); step_a(DcInt<2>{}, Xa, Xb); step_a(DcInt<3>{}, Xb, Xa);
      step_a(DcInt<4>{}, Xa, Xb); step_a(DcInt<5>{}, Xb, Xa); step_a(DcInt<6>{}, Xa, Xb); step_a(DcInt<7>{}, Xb, Xa);
    };
    // two copies on purpose: with the global loads of the rare path in the same code, hipcc waits vmcnt(0) before every
    // gather of the DMA path (a register with a load pending on the OTHER path) and the window ring degenerates
    if (xfit) phase_a(DcInt<1>{}); else phase_a(DcInt<0>{});
    // the two half-waves hold the other 8 channels of the same pixel; half 0 writes d/dh, half 1 d/dw
    {
      // half 0 needs the other half's d/dh sums, half 1 its d/dw sums: ONE exchange per tap (each half sends what the other
      // needs), all nine requested before the first is used
      float mine[T], theirs[T];
      MFN_UNROLL
      for (int t = 0; t < T; ++t) { mine[t] = half ? sw[t] : sh[t]; theirs[t] = __shfl_xor(half ? sh[t] : sw[t], 32); }
      MFN_COMPILER_FENCE();
      MFN_UNROLL
      for (int t = 0; t < T; ++t) {
        const float m9 = stash[(0 + t / 3) * 64] * stash[(3 + t % 3) * 64];
        vt[t] = (mine[t] + theirs[t]) * m9;
        vsum += vt[t];
      }
    }
  }
  // With several channel blocks (and filter slices) a pixel's values add up through atomics (the buffer is zero-filled in write
  // mode) -- nine per lane with an offset tensor, queued here.  Measured on one box, levels 5..2: here 37.8 / 47.1 / 75.0 / 109.3 us;
  // three taps behind each of phase B's group barriers 36.6 / 47.6 / 77.4 / 113.8; after the flush 36.5 / 47.5 / 82.6 / 116.5
  // (the wave is 1-5 k cycles late for phase B's first barrier this way, but later the atomics pile up behind the flush's).
  // One writer per value (Cin <= 16, no filter slices): plain stores, "add" reads its old values together.
  if (fast && p.req_offset && px_valid) {
    const bool single = gridDim.y == 1 && gridDim.z == 1;
    if (fm) {  // d/dflow: the nine taps share the offset, so their gradients add up (MaskFlownet.py:230)
      const float v = vsum * (p.flow_scale / p.flow_stride);
      float *dst = p.gflow + ((size_t)n * 2 + half) * plane + pix;
      if (single) *dst = p.req_offset == 3 ? *dst + v : v;
      else if (v != 0.f) atomicAdd(dst, v);
    } else {
      float *dst = p.goffset + ((size_t)n * 2 * T + half) * plane + pix;
      if (single) {
        if (p.req_offset == 3) {
          float old[T];
          MFN_UNROLL
          for (int t = 0; t < T; ++t) old[t] = dst[(size_t)(2 * t) * plane];
          MFN_COMPILER_FENCE();
          MFN_UNROLL
          for (int t = 0; t < T; ++t) vt[t] += old[t];
        }
        MFN_UNROLL
        for (int t = 0; t < T; ++t) dst[(size_t)(2 * t) * plane] = vt[t];
      } else {
        MFN_UNROLL
        for (int t = 0; t < T; ++t)
          if (vt[t] != 0.f) atomicAdd(dst + (size_t)(2 * t) * plane, vt[t]);
      }
    }
  }
  if (p.timeline) tk2 = MFN_CYCLES();
  if (!p.req_x) return;   // uniform

  // ---- phase B: input gradient -----------------------------------------------------------------------------------------
  // Zero the 16 planes, four groups of CPG channels (chains) per lane, flush.
  // Every wave has waited for its own DMAs by now (the K loop's last chunk: vmcnt(0); phase A: its last gather), so the
  // barrier alone makes the stage buffers and window rings dead: the planes take their place.  No vmcnt(0) here -- it would
  // wait for the offset gradient's atomics, which may drain under phase B.
  MFN_LDS_BARRIER();
  float ay[3], by[3], ax[3], bxw[3];
  MFN_UNROLL
  for (int i = 0; i < 3; ++i) {
    ay[i] = stash[(6 + i) * 64]; by[i] = stash[(9 + i) * 64]; ax[i] = stash[(12 + i) * 64]; bxw[i] = stash[(15 + i) * 64];
  }
  const int cell0 = reinterpret_cast<const int *>(stash)[18 * 64];  // neighbourhood corner (0, 0) in a plane
  const int myturn = reinterpret_cast<const int *>(stash)[19 * 64];
  const int mypartner = reinterpret_cast<const int *>(stash)[20 * 64];
  const int psrc = mypartner >= 0 ? mypartner + 32 * half : lane;  // the lane whose values are added to this lane's
  const float pmask = mypartner >= 0 ? 1.f : 0.f;
  const bool any_merged = MFN_UNIFORM(merged) != 0;
  // the flush plan: 64 consecutive cells of the planes' touched box per atomic instruction
  int fy0 = 1 << 28, fy1 = -(1 << 28), fx0 = 1 << 28, fx1 = -(1 << 28);
  int woy[4], wox[4];   // wave-private planes: where every wave's planes lie (a wave without cells: far away, nothing is inside)
  MFN_UNROLL
  for (int w2 = 0; w2 < 4; ++w2) {
    const int *e = exch + w2 * 8;
    fy0 = min(fy0, e[0]); fy1 = max(fy1, e[1]); fx0 = min(fx0, e[2]); fx1 = max(fx1, e[3]);
    woy[w2] = MFN_UNIFORM(e[1] >= e[0] ? e[4] : (1 << 28)); wox[w2] = MFN_UNIFORM(e[5]);
  }
  fy0 = MFN_UNIFORM(fy0); fy1 = MFN_UNIFORM(fy1); fx0 = MFN_UNIFORM(fx0); fx1 = MFN_UNIFORM(fx1);
  const bool any_cells = fy1 >= fy0;
  // (the waves' planes may lie hundreds of thousands of pixels apart -- a flow that sends tiles far away: the cell count in 64 bits)
  const int ncols = any_cells ? fx1 - fx0 + 1 : 1;
  const long long ncells64 = any_cells ? (long long)(fy1 - fy0 + 1) * (long long)ncols : 0;
  const int ncells = ncells64 > (long long)(1 << 30) ? (1 << 30) : (int)ncells64;
  // cell e of the union box (image coordinates) in every wave's planes (its float offset inside a plane, or
  // a zero mask where the cell lies outside them: untouched cells of a plane are zero)
  constexpr int NWF = WFS;
  int wfo[NWF][4], wgo[NWF];
  unsigned wfm[NWF];
  const bool merged_flush = ncells <= WFS * 64;   // uniform
  {
    const float inv_ncols = 1.f / (float)ncols;
    MFN_UNROLL
    for (int sl = 0; sl < NWF; ++sl) {
      wgo[sl] = -1; wfm[sl] = 0;
      MFN_UNROLL
      for (int w2 = 0; w2 < 4; ++w2) wfo[sl][w2] = 0;
      if (!merged_flush || sl * 64 >= ncells) continue;   // uniform
      const int e = sl * 64 + lane;
      int row = (int)((float)e * inv_ncols), col = e - row * ncols;  // cell counts are far below 2^24: off by one at most
      if (col < 0) { --row; col += ncols; }
      if (col >= ncols) { ++row; col -= ncols; }
      const int yy = fy0 + row, xx = fx0 + col;
      wgo[sl] = (e < ncells && yy >= 0 && yy < H && xx >= 0 && xx < W) ? yy * W + xx : -1;
      MFN_UNROLL
      for (int w2 = 0; w2 < 4; ++w2) {
        const int oy = yy - woy[w2], ox = xx - wox[w2];
        const bool in = oy >= 0 && oy < PRe && ox >= 0 && ox < PCe;
        wfo[sl][w2] = in ? oy * PSe + ox : 0;
        wfm[sl] |= in ? 1u << w2 : 0u;
      }
    }
  }
  unsigned long long td0 = 0, td1 = 0, td2 = 0, td3 = 0;

  float *const myplanes = lds + wave * (CB * PLe);
  int oby0 = 0, oby1 = -1, obx0 = 0, obx1 = -1;   // this wave's own box (the flush of a torn region)
  {
    // the planes lie over the exchange area and the stashes: every wave has what it needs from them (above) behind this barrier.
    // A wave zeroes its own planes: its LDS accesses stay in order, no barrier after
    const int own0 = MFN_UNIFORM(exch[wave * 8 + 0]), own1 = MFN_UNIFORM(exch[wave * 8 + 1]), own2 = MFN_UNIFORM(exch[wave * 8 + 2]), own3 = MFN_UNIFORM(exch[wave * 8 + 3]);
    oby0 = own0; oby1 = own1; obx0 = own2; obx1 = own3;
    MFN_LDS_BARRIER();
    for (int e = lane; e < CB * PLe / 4; e += 64) reinterpret_cast<float4 *>(myplanes)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    MFN_WAVE_SYNC_EMU();
  }
  auto group_b = [&](auto g_c) {
    constexpr int g = decltype(g_c)::value;   // chains c = 0 .. CPG - 1: the lane's channels q = CPG g + c
    if (g == 0 && p.timeline) td0 = MFN_CYCLES();
    if (fast) {
      // the nine taps folded onto the 4x4 neighbourhood, along x first, then along y
      float G[CPG][4][4];
      MFN_UNROLL
      for (int c = 0; c < CPG; ++c) {
        float R[3][4];
        MFN_UNROLL
        for (int i = 0; i < 3; ++i) {
          const float c0 = DCP_ACC(3 * i, CPG * g + c), c1 = DCP_ACC(3 * i + 1, CPG * g + c), c2 = DCP_ACC(3 * i + 2, CPG * g + c);
          R[i][0] = c0 * ax[0];
          R[i][1] = fmaf(c0, bxw[0], c1 * ax[1]);
          R[i][2] = fmaf(c1, bxw[1], c2 * ax[2]);
          R[i][3] = c2 * bxw[2];
        }
        MFN_UNROLL
        for (int v = 0; v < 4; ++v) {
          G[c][0][v] = ay[0] * R[0][v];
          G[c][1][v] = fmaf(by[0], R[0][v], ay[1] * R[1][v]);
          G[c][2][v] = fmaf(by[1], R[1][v], ay[2] * R[2][v]);
          G[c][3][v] = by[2] * R[2][v];
        }
      }
      if (any_merged) {  // uniform: every lane takes part in the shuffles
        MFN_UNROLL
        for (int c = 0; c < CPG; ++c)
          MFN_UNROLL
          for (int u = 0; u < 4; ++u)
            MFN_UNROLL
            for (int v = 0; v < 4; ++v) G[c][u][v] = fmaf(__shfl(G[c][u][v], psrc), pmask, G[c][u][v]);
      }
      if (g == 0 && p.timeline) { MFN_OPAQUE(G[CPG - 1][3][3]); td1 = MFN_CYCLES(); }
      // the lane's four channels of this group are accumulator slots c + 8 g + 4 half = quad 2 g + half of the
      // wave's planes (kept in SLOT order: the flush undoes the waves' channel permutations), component c
      float4 *pq = reinterpret_cast<float4 *>(myplanes + (size_t)(2 * g + half) * (4 * PLe)) + cell0;
      // (emulation: free-running lanes take a whole walk one lane at a time, mfn_rt.h)
      for (int t = 0; t < ((MFN_DCP_ABL & 4) ? 0 : nturns); ++t) {
        if (myturn == t) {
          MFN_EMU_LOCK();
          MFN_UNROLL
          for (int u = 0; u < 4; ++u)
            MFN_UNROLL
            for (int v = 0; v < 4; ++v) {
              float4 o4 = pq[u * PSe + v];
              o4.x += G[0][u][v]; o4.y += G[1][u][v]; o4.z += G[2][u][v]; o4.w += G[3][u][v];
              pq[u * PSe + v] = o4;
              // the next cell's read is ISSUED after this cell's writes: another lane's cell (u, v) is this lane's
              // cell (u', v'); the in-order LDS pipe then orders them
              MFN_COMPILER_FENCE();
            }
          MFN_EMU_UNLOCK();
        }
      }
      if (!(MFN_DCP_ABL & 1) && __any(myturn == -2)) {  // neighbourhoods outside the plane: straight to gx
        if (myturn == -2) {
          MFN_UNROLL
          for (int c = 0; c < CPG; ++c) {
            const int ch = cb + chan_of(CPG * g + c, half);
            float *gim = p.gx + ((size_t)n * p.Cin + min(ch, p.Cin - 1)) * plane;
            MFN_UNROLL
            for (int u = 0; u < 4; ++u)
              MFN_UNROLL
              for (int v = 0; v < 4; ++v) {
                const int yy = ly0 + u, xx = lx0 + v;
                if (G[c][u][v] != 0.f && ch < p.Cin && yy >= 0 && yy < H && xx >= 0 && xx < W)
                  atomicAdd(gim + (size_t)yy * W + xx, G[c][u][v]);
              }
          }
        }
      }
    }
    if (g == 0 && p.timeline) { MFN_WAIT_LGKM0(); td2 = MFN_CYCLES(); td3 = td2; }   // (no barrier behind a group any more)
  };
  group_b(DcInt<0>{}); group_b(DcInt<1>{});
  MFN_LDS_BARRIER();   // the flush reads every wave's planes
#undef DCP_ACC
  // ---- flush ------------------------------------------------------------------------------------------------------------
  if (p.timeline) tk3 = MFN_CYCLES();
  // wave w takes the channels 4 w .. 4 w + 3 and adds the four waves' planes cell by cell on the way out (a
  // cell outside a wave's planes: masked).  Wave w2 keeps slot sigma = channel ^ xw(w2) at quad sigma >> 2, component sigma & 3:
  // the block's channel quad w is its quad w ^ (xw(w2) >> 2), components swapped in pairs where xw(w2) has bit 1.
  auto flush_merged = [&](auto nsl_c) {
    constexpr int NSL = decltype(nsl_c)::value;
    if (cb + 4 * wave >= p.Cin) return;  // uniform
    float *gim = p.gx + ((size_t)n * p.Cin + cb + 4 * wave) * plane;
    MFN_UNROLL
    for (int sl0 = 0; sl0 < NSL; sl0 += 2) {   // two slices' reads in flight (32 registers)
    if (sl0 * 64 >= ncells) break;   // uniform
    float4 fr[2][4];
    MFN_UNROLL
    for (int sl = sl0; sl < sl0 + 2; ++sl)
      MFN_UNROLL
      for (int w2 = 0; w2 < 4; ++w2)
        fr[sl - sl0][w2] = reinterpret_cast<const float4 *>(lds + (size_t)w2 * (CB * PLe) + (size_t)(wave ^ (xw_of(w2) >> 2)) * (4 * PLe))[wfo[sl][w2]];
    MFN_UNROLL
    for (int sl = sl0; sl < sl0 + 2; ++sl) {
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      MFN_UNROLL
      for (int w2 = 0; w2 < 4; ++w2) {
        const bool in = ((wfm[sl] >> w2) & 1u) != 0;
        const float4 f = fr[sl - sl0][w2];
        if (xw_of(w2) & 2) { v[0] += in ? f.z : 0.f; v[1] += in ? f.w : 0.f; v[2] += in ? f.x : 0.f; v[3] += in ? f.y : 0.f; }
        else { v[0] += in ? f.x : 0.f; v[1] += in ? f.y : 0.f; v[2] += in ? f.z : 0.f; v[3] += in ? f.w : 0.f; }
      }
      if (sl * 64 < ncells && wgo[sl] >= 0) {
        MFN_UNROLL
        for (int k = 0; k < 4; ++k)
          if (cb + 4 * wave + k < p.Cin && v[k] != 0.f) atomicAdd(gim + (size_t)k * plane + wgo[sl], v[k]);
      }
    }
    }
  };
  // ... or, the union box being too large for that (a flow that tears the region apart): every wave its own planes, its own box
  auto flush_own = [&]() {
    if (oby1 < oby0) return;   // uniform
    const int nc = obx1 - obx0 + 1, ncl = (oby1 - oby0 + 1) * nc;
    for (int sl = 0; sl * 64 < ncl; ++sl) {
      const int e1 = sl * 64 + lane;
      const int row = e1 / nc, col = e1 - row * nc;
      const int yy = oby0 + row, xx = obx0 + col;
      const bool okc = e1 < ncl && yy >= 0 && yy < H && xx >= 0 && xx < W;
      const int lo = min(max((yy - py0) * PSe + (xx - px0), 0), PLe - 1);
      for (int sq = 0; sq < 4; ++sq) {
        const float4 f = reinterpret_cast<const float4 *>(myplanes + (size_t)sq * (4 * PLe))[lo];
        const float fv4[4] = {f.x, f.y, f.z, f.w};
        MFN_UNROLL
        for (int k = 0; k < 4; ++k) {
          const int ch = cb + ((4 * sq + k) ^ xw);
          if (okc && ch < p.Cin && fv4[k] != 0.f) atomicAdd(p.gx + ((size_t)n * p.Cin + ch) * plane + (size_t)yy * W + xx, fv4[k]);
        }
      }
    }
  };
  if (any_cells && !(MFN_DCP_ABL & 2)) {
    if (!merged_flush) flush_own();
    else flush_merged(DcInt<WFS>{});
  }
  if (p.timeline && tid == 0) {
    unsigned long long *b_ = p.timeline + (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4;
    const unsigned long long tk4 = MFN_CYCLES();
    b_[0] = ((tks - tk0) & 0xffffffffull) | ((tk1 - tks) << 32);
    if (p.tl_detail) { b_[1] = td0 - tk2; b_[2] = td2 - td0; b_[3] = td3 - td2; }  // before the groups | first group: fold + walk | its barrier
    else { b_[1] = tk2 - tk1; b_[2] = tk3 - tk2; b_[3] = tk4 - tk3; }
  }
}

// ---- weight (+ bias) gradient in the forward's orientation -----------------------------------------------------------------
// gw[o][c][t] = sum over pixels of gout[o][pixel] * col[(c, t)][pixel].  dc_bwd_weight_mfma_kernel (backward.h) forms every
// column value from a per-tap geometry table and four global gathers (64 loads per lane and 32-pixel tile); here the
// columns are produced as the forward kernel produces them -- lane = pixel, one 12 x 20 source window per channel pair by
// LDS-DMA, 16 ds_read_b32, the separable interpolation (42 FMAs) gives the nine tap values of a channel -- and handed to
// the MFMA through LDS with the PIXELS as the reduction dimension: D[filter][channel] per tap, A = a gout tile, B = the
// column tile, both read with 16-byte LDS reads (k-step s of half h is pixel 16 h + s, the same for both operands).
// A block walks 4x8-pixel tiles for one 32-channel block: each wave prepares the geometry of one of the next four tiles
// (record in LDS: 24 words per pixel), then for every tile the four waves produce eight channels each, and split the
// 9 x (Cout / 32) accumulator tiles between them.  Everything a tile needs from memory -- its gout values, its source
// windows, every fourth tile the next tiles' offsets -- is requested one tile ahead.  The sums leave through LDS as contiguous (c, t) rows: coalesced atomics.
// Tiles whose offsets differ per tap, whose floors are irregular or whose window does not fit (rough flows) take a slow
// producer (per-tap geometry, four global loads per value) in the same place.
constexpr int DCW_GWD = 24;                      // words of a pixel's geometry record (word-major: [word][32 pixels])
constexpr int DCW_SLOT = DCW_GWD * 32 + 8;       // + header: window origin, image, flags
constexpr int DCW_RS = 36;                       // row stride of the column / gout tiles: 16-byte rows, conflict-free b128 reads
constexpr int DCW_XW_F = 512;                    // pair window: 2 x 12 x 20 = 480 used
constexpr int DCW_STG = 292;                     // row stride of the flush staging [32 filters][288 (c, t)]
constexpr int DCW_COL_F = 9 * 32 * DCW_RS;
struct DcBwdWPParams {
  const float *gout, *x, *offset;
  float *gw, *gbias;             // gbias NULL: not requested (the channel-block-0 blocks add it)
  int N, Cin, H, W, Cout, ph, pw;
  int tiles_x, tiles_y, ntiles;  // 4x8 pixel tiles
  float inv_tpi, inv_tiles_x;
  int tiles_per_block;
  float *slabs;                  // [channel block][block][Cout / 32 tiles][32][288] partial sums, or NULL: add to gw with atomics
  float *bias_slabs;             // [block][Cout / 32 tiles * 32] of the channel-block-0 blocks, or NULL: atomics
  unsigned long long *timeline;  // measurement only: per block, wave 0: {wait + gout store | loads << 32, produce | barrier << 32, consume | barrier << 32, total}
  // flow mode (mfn_deform_conv_shared_bwd; DcBwdPParams): offsets from flow[n][dir][pixel]; dc_bwd_weight_pc_kernel only
  const float *flow;
  float flow_scale, flow_stride;
  // write-mode gradients of the call's other kernels (gx, goffset / gflow) that this launch clears on its way in (it then runs
  // first): every thread of the grid zeroes a strided share with 16-byte stores; NULL / 0 = nothing to clear
  float *zero_p[2];
  size_t zero_n[2];
};

// ---- producer and consumer waves ------------------------------------------------------------------------------------------
// With four waves that all produce, then all multiply (round 2's first form, removed in round 3): per tile 2.6 k cycles of producing (LDS / memory round
// trips, little arithmetic) + 3.5 k of MFMA + two barriers.  Here a block is EIGHT waves: waves 0-3 produce the columns of
// tile i + 1 into one half of a double buffer while waves 4-7 multiply tile i out of the other half; a SIMD holds one wave
// of each kind, so the producer's round trips run under the consumer's MFMAs.  One barrier per tile.  With the LDS taken by
// the two column buffers the producers read the 4x4 neighbourhoods straight from global memory (one dword-aligned 16-byte
// load per neighbourhood row, as the forward's second tier; tiles with clamped columns: 16 dword loads), one tile ahead.
constexpr size_t dc_bwd_weight_pc_lds_bytes(int mtot) {
  return ((size_t)8 * DCW_SLOT + 2 * DCW_COL_F + (size_t)2 * mtot * 32 * DCW_RS) * sizeof(float);
}
template <int MTOT>
__global__ __launch_bounds__(512, 1) void dc_bwd_weight_pc_kernel(DcBwdWPParams p) {
  constexpr int T = 9, UN = 9 * MTOT, UMAX = (UN + 3) / 4, RS = DCW_RS, GOUT_F = MTOT * 32 * DCW_RS;
  MFN_DYN_SHARED(float, lds);
  float *geom = lds;                                   // [2 sets][4 slots][DCW_SLOT]
  float *colB = lds + 8 * DCW_SLOT;                    // [2][9 taps][32 channels][RS]
  float *goutB = colB + 2 * DCW_COL_F;                 // [2][MTOT * 32 filters][RS]
  const int tid = threadIdx.x & 255, lane = tid & 63;
  const int wave = MFN_UNIFORM((int)threadIdx.x >> 6);
  const bool producer = wave < 4;                      // uniform per wave
  const int pw = wave & 3;
  const int half = lane >> 5, j = lane & 31;
  const int H = p.H, W = p.W;
  const size_t plane = (size_t)H * W;
  const int cb = blockIdx.y * 32;
  const int t0 = blockIdx.x * p.tiles_per_block, t1 = min(t0 + p.tiles_per_block, p.ntiles);
  const int ntile = t1 - t0;

  MFN_UNROLL
  for (int k = 0; k < 2; ++k) {
    float *z = p.zero_p[k];
    if (!z) continue;  // uniform
    const size_t n = p.zero_n[k];
    const size_t gthreads = (size_t)gridDim.x * gridDim.y * 512, gtid = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 512 + threadIdx.x;
    if ((reinterpret_cast<size_t>(z) & 15) == 0) {
      for (size_t q = gtid; q * 4 + 3 < n; q += gthreads) *reinterpret_cast<float4 *>(z + q * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
      for (size_t i = (n & ~(size_t)3) + gtid; i < n; i += gthreads) z[i] = 0.f;
    } else {
      for (size_t i = gtid; i < n; i += gthreads) z[i] = 0.f;
    }
  }

  float bsum[MTOT * 4];  // producer thread (filter row tid >> 5, pixel tid & 31): filters (tid >> 5) + 8 i
  MFN_UNROLL
  for (int i = 0; i < MTOT * 4; ++i) bsum[i] = 0.f;

  // ---- geometry of one tile per producer wave, parked in LDS (as dc_bwd_weight_pix_kernel; the neighbourhood's lines are
  // kept as offsets inside a channel plane)
  struct GeoIn { float off[2 * T]; int n, ho, wo; bool px_valid; };
  auto geo_load = [&](int tile, GeoIn &q) {
    const bool tile_ok = tile < t1;
    const int tpi = p.tiles_y * p.tiles_x;
    const int tl = min(tile, p.ntiles - 1);
    auto divmod = [](int a, int b, float inv_b, int &qq, int &r) {
      qq = (int)((float)a * inv_b);
      r = a - qq * b;
      if (r < 0) { --qq; r += b; }
      if (r >= b) { ++qq; r -= b; }
    };
    int rt, ty, tx;
    divmod(tl, tpi, p.inv_tpi, q.n, rt);
    divmod(rt, p.tiles_x, p.inv_tiles_x, ty, tx);
    q.ho = ty * 4 + (j >> 3);
    q.wo = tx * 8 + (j & 7);
    q.px_valid = tile_ok && q.ho < H && q.wo < W;
    q.ho = min(q.ho, H - 1);
    q.wo = min(q.wo, W - 1);
    if (p.flow) {  // flow mode: one offset for all taps, in the forward's arithmetic
      const float *fp = p.flow + (size_t)q.n * 2 * plane + (size_t)q.ho * W + q.wo;
      const float oh = fp[0] * p.flow_scale / p.flow_stride, ow = fp[plane] * p.flow_scale / p.flow_stride;
      MFN_UNROLL
      for (int t = 0; t < T; ++t) { q.off[2 * t] = oh; q.off[2 * t + 1] = ow; }
    } else {
      const float *op = p.offset + (size_t)q.n * 2 * T * plane + (size_t)q.ho * W + q.wo;
      MFN_UNROLL
      for (int t = 0; t < 2 * T; ++t) q.off[t] = op[(size_t)t * plane];
    }
  };
  auto geo_store = [&](const GeoIn &q, int set) {
    const int h_in = q.ho - p.ph, w_in = q.wo - p.pw;
    const float oh = q.off[0], ow = q.off[1];
    bool regular = true;
    MFN_UNROLL
    for (int t = 1; t < T; ++t) regular = regular && (q.off[2 * t] == oh) && (q.off[2 * t + 1] == ow);
    float a_y[3], b_y[3], a_x[3], b_x[3];
    int iy[4], ix[4], c0x = 0;  // c0x: the neighbourhood's first column before clamping
    {
      int lo0 = 0;
      MFN_UNROLL
      for (int i = 0; i < 3; ++i) {
        bool v; int lo, hi; float l;
        dc_axis(oh, h_in, i, H, v, lo, hi, l);
        v = v && q.px_valid;
        const int ulo = (int)fminf(fmaxf(floorf((float)i + oh), -1.0e6f), 1.0e6f);  // unclamped floor
        if (i == 0) lo0 = ulo; else regular = regular && (ulo == lo0 + i);
        a_y[i] = v ? 1.f - l : 0.f;
        b_y[i] = v ? l : 0.f;
      }
      MFN_UNROLL
      for (int m = 0; m < 4; ++m) iy[m] = min(max(h_in + lo0 + m, 0), H - 1);
      MFN_UNROLL
      for (int i = 0; i < 3; ++i) {
        bool v; int lo, hi; float l;
        dc_axis(ow, w_in, i, W, v, lo, hi, l);
        v = v && q.px_valid;
        const int ulo = (int)fminf(fmaxf(floorf((float)i + ow), -1.0e6f), 1.0e6f);
        if (i == 0) lo0 = ulo; else regular = regular && (ulo == lo0 + i);
        a_x[i] = v ? 1.f - l : 0.f;
        b_x[i] = v ? l : 0.f;
      }
      MFN_UNROLL
      for (int m = 0; m < 4; ++m) ix[m] = min(max(w_in + lo0 + m, 0), W - 1);
      c0x = w_in + lo0;
    }
    const bool fast = __all(regular || !q.px_valid) != 0;
    // every lane's four columns consecutive (not clamped at the image's left / right edge): one 16-byte load per row
    const bool consec = __all(ix[3] - ix[0] == 3) != 0;
    float *g = geom + (set * 4 + pw) * DCW_SLOT;
    if (half == 0) {
      MFN_UNROLL
      for (int i = 0; i < 3; ++i) {
        g[(0 + i) * 32 + j] = a_y[i]; g[(3 + i) * 32 + j] = b_y[i];
        g[(6 + i) * 32 + j] = a_x[i]; g[(9 + i) * 32 + j] = b_x[i];
      }
      int *gi = reinterpret_cast<int *>(g);
      MFN_UNROLL
      for (int m = 0; m < 4; ++m) gi[(12 + m) * 32 + j] = iy[m] * W;
      // one 16-byte load per row at the base column (inside the row); where the image's edge clamps the columns, produce()
      // picks column ix[m] - base out of the four loaded values (two bits per m)
      const int cbase = min(max(c0x, 0), W - 4);
      int shifts = 0;
      MFN_UNROLL
      for (int m = 0; m < 4; ++m) shifts |= (ix[m] - cbase) << (2 * m);
      gi[16 * 32 + j] = cbase;
      gi[17 * 32 + j] = shifts;
      gi[20 * 32 + j] = q.ho * W + q.wo;
      gi[21 * 32 + j] = q.px_valid ? 1 : 0;
      gi[22 * 32 + j] = q.ho;
      gi[23 * 32 + j] = q.wo;
    }
    if (lane == 0) {
      int *hd = reinterpret_cast<int *>(g + DCW_GWD * 32);
      hd[0] = fast ? (consec ? 2 : 1) : 0; hd[2] = q.n;
    }
  };
  // ---- producer: what tile i needs from memory goes into registers one tile ahead: its gout values and the 4x4
  // neighbourhoods of this wave's eight channels (four pairs; lane = pixel, half = channel of the pair)
  float gvn[MTOT * 4];
  f4u xv[4][4];
  auto tile_loads = [&](int i) {
    const float *g = geom + (((i >> 2) & 1) * 4 + (i & 3)) * DCW_SLOT;
    const int *gi = reinterpret_cast<const int *>(g);
    const int *hd = reinterpret_cast<const int *>(g + DCW_GWD * 32);
    const int mode = MFN_UNIFORM(hd[0]), n = MFN_UNIFORM(hd[2]);
    {
      const int px = tid & 31, osub = tid >> 5;
      const float *gp = p.gout + (size_t)n * p.Cout * plane + gi[20 * 32 + px];
      MFN_UNROLL
      for (int k = 0; k < MTOT * 4; ++k) gvn[k] = gp[(size_t)min(osub + 8 * k, p.Cout - 1) * plane];
    }
    {
      // ONE unconditional path (tiles with consecutive columns, tiles the image's edge clamps, and -- their values unused --
      // the tiles of the per-tap path): a load in a branch becomes "old or new value" at the join, hipcc copies the loaded
      // registers there and waits for every load in flight to do so -- the requests were never ahead of anything
      (void)mode;
      int row[4];
      MFN_UNROLL
      for (int m = 0; m < 4; ++m) row[m] = gi[(12 + m) * 32 + j];
      const int cbase = gi[16 * 32 + j];
      MFN_UNROLL
      for (int k = 0; k < 4; ++k) {
        // channels past Cin (ragged last block) are written as zeros: any valid plane is read in their place
        const int c = min(cb + 2 * (4 * pw + k) + half, p.Cin - 1);
        const float *pl = p.x + ((size_t)n * p.Cin + c) * plane + cbase;
        MFN_UNROLL
        for (int m = 0; m < 4; ++m) xv[k][m] = mfn_load4u(pl + row[m]);
      }
    }
  };
  auto produce = [&](int i) {   // gout tile and columns of tile i (values loaded by tile_loads(i)) -> buffer i & 1
    const float *g = geom + (((i >> 2) & 1) * 4 + (i & 3)) * DCW_SLOT;
    const int *gi = reinterpret_cast<const int *>(g);
    const int *hd = reinterpret_cast<const int *>(g + DCW_GWD * 32);
    const int mode = MFN_UNIFORM(hd[0]), n = MFN_UNIFORM(hd[2]);
    float *colT = colB + (i & 1) * DCW_COL_F, *goutT = goutB + (i & 1) * GOUT_F;
    {
      const int px = tid & 31, osub = tid >> 5;
      const bool pvx = gi[21 * 32 + px] != 0;
      MFN_UNROLL
      for (int k = 0; k < MTOT * 4; ++k) {
        const float v = (pvx && osub + 8 * k < p.Cout) ? gvn[k] : 0.f;
        goutT[(osub + 8 * k) * RS + px] = v;
        bsum[k] += v;
      }
    }
    if (mode != 0) {
      float a_y[3], b_y[3], a_x[3], b_x[3];
      MFN_UNROLL
      for (int k = 0; k < 3; ++k) {
        a_y[k] = g[(0 + k) * 32 + j]; b_y[k] = g[(3 + k) * 32 + j]; a_x[k] = g[(6 + k) * 32 + j]; b_x[k] = g[(9 + k) * 32 + j];
      }
      // columns clamped at the image's edge (mode 1): column m of the neighbourhood is loaded value (shifts >> 2m) & 3
      const int shifts = mode == 1 ? gi[17 * 32 + j] : 0xE4;   // 0xE4: the identity (3, 2, 1, 0)
      MFN_UNROLL
      for (int k = 0; k < 4; ++k) {
        const int cl = 2 * (4 * pw + k) + half;   // channel of this lane inside the block
        const bool c_ok = cb + cl < p.Cin;
        float tr[4][3];
        MFN_UNROLL
        for (int m = 0; m < 4; ++m) {
          f4u v = xv[k][m];
          if (mode == 1) {  // uniform
            const f4u u = v;
            auto pick = [&](int sft) { return sft == 0 ? u.x : (sft == 1 ? u.y : (sft == 2 ? u.z : u.w)); };
            v.x = pick(shifts & 3); v.y = pick((shifts >> 2) & 3); v.z = pick((shifts >> 4) & 3); v.w = pick((shifts >> 6) & 3);
          }
          tr[m][0] = a_x[0] * v.x + b_x[0] * v.y;
          tr[m][1] = a_x[1] * v.y + b_x[1] * v.z;
          tr[m][2] = a_x[2] * v.z + b_x[2] * v.w;
        }
        MFN_UNROLL
        for (int ii = 0; ii < 3; ++ii)
          MFN_UNROLL
          for (int q = 0; q < 3; ++q) {
            const float cv = a_y[ii] * tr[ii][q] + b_y[ii] * tr[ii + 1][q];
            colT[((ii * 3 + q) * 32 + cl) * RS + j] = c_ok ? cv : 0.f;
          }
      }
    } else {
      // per-tap geometry (dc_make_tap) and four global loads per value: arbitrary offsets, irregular floors
      const bool pv = gi[21 * 32 + j] != 0;
      const int ho = gi[22 * 32 + j], wo = gi[23 * 32 + j];
      const float *op = p.flow ? p.flow + (size_t)n * 2 * plane + (size_t)ho * W + wo
                               : p.offset + (size_t)n * 2 * T * plane + (size_t)ho * W + wo;
      const float fsc = p.flow ? p.flow_scale : 1.f, fst = p.flow ? p.flow_stride : 1.f;  // (v * 1 / 1 == v)
      MFN_NOUNROLL
      for (int k = 0; k < 4; ++k) {
        const int cl = 2 * (4 * pw + k) + half;
        const bool c_ok = cb + cl < p.Cin;
        const float *pl = p.x + ((size_t)n * p.Cin + (c_ok ? cb + cl : 0)) * plane;
        MFN_NOUNROLL
        for (int t = 0; t < T; ++t) {
          const int ti = t / 3, tj = t - 3 * ti;
          const int tq = p.flow ? 0 : t;
          const DcTap tp = dc_make_tap(op[(size_t)(2 * tq) * plane] * fsc / fst, op[(size_t)(2 * tq + 1) * plane] * fsc / fst,
                                       ho - p.ph, wo - p.pw, ti, tj, H, W, pv && c_ok);
          const int bb = tp.base & 0x3FFFFFFF, dwi = (tp.base >> 30) & 1;
          const float cv = tp.w1 * pl[bb] + tp.w2 * pl[bb + dwi] + tp.w3 * pl[bb + tp.dhW] + tp.w4 * pl[bb + tp.dhW + dwi];
          colT[(t * 32 + cl) * RS + j] = cv;
        }
      }
    }
  };
  // one producer step: tile i into its buffer, then the requests for tile i + 1 (and, around every fourth tile, the
  // geometry of the next four)
  GeoIn gin;
  // measurement only (p.timeline): shader cycles of wave 0 in produce / geometry / load issue / barrier, of wave 4 in MFMA / barrier
  unsigned long long tq[6] = {0, 0, 0, 0, 0, 0};
  auto producer_step = [&](int i) {
    const bool next_group = ((i >> 2) + 1) * 4 < ntile;   // uniform: a group of tiles after this one
    const unsigned long long c0 = MFN_CYCLES();
    produce(i);
    const unsigned long long c1 = MFN_CYCLES();
    if ((i & 3) == 2 && next_group) geo_store(gin, ((i >> 2) + 1) & 1);   // visible after this step's barrier
    const unsigned long long c2 = MFN_CYCLES();
    tile_loads(max(min(i + 1, ntile - 1), 0));   // unconditional (past the last tile: the last one again, unused); a new group's
                                         // geometry was parked one step ago
    if ((i & 3) == 1 && next_group) geo_load(t0 + ((i >> 2) + 1) * 4 + pw, gin);
    const unsigned long long c3 = MFN_CYCLES();
    tq[0] += c1 - c0; tq[1] += c2 - c1; tq[2] += c3 - c2;
  };

  // the slab of filter tile f: both kinds of waves store their half of it once the consumers have staged it
  float *stg = colB;  // [32 filters][DCW_STG]
  auto store_slab = [&](int f) {
    if (p.slabs) {
      float *slab = p.slabs + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * MTOT + f) * (32 * 288);
      for (int e = threadIdx.x; e < 32 * 288; e += 512) {
        const int ol = e / 288, col = e - ol * 288;
        slab[e] = stg[ol * DCW_STG + col];
      }
    } else {
      for (int e = threadIdx.x; e < 32 * 288; e += 512) {
        const int ol = e / 288, col = e - ol * 288;
        const int o = f * 32 + ol, c = cb + col / 9;
        const float v = stg[ol * DCW_STG + col];
        if (o < p.Cout && c < p.Cin && v != 0.f) atomicAdd(p.gw + ((size_t)o * p.Cin + cb) * 9 + col, v);
      }
    }
  };
  // Two instruction streams with the same sequence of block barriers (the hardware counts arrivals, it does not match
  // program counters): in one stream the consumers' accumulators would stay live through the producers' code (spills).
  if (producer) {
    geo_load(t0 + pw, gin);
    geo_store(gin, 0);
    MFN_LDS_BARRIER();
    tile_loads(0);
    producer_step(0);
    MFN_LDS_BARRIER();
    for (int i = 0; i + 1 < ntile; ++i) {   // the consumers multiply tile i meanwhile
      producer_step(i + 1);
      const unsigned long long cb0 = MFN_CYCLES();
      MFN_LDS_BARRIER();  // tile i + 1 is complete, tile i's buffer is free
      tq[3] += MFN_CYCLES() - cb0;
    }
    if (ntile > 0) MFN_LDS_BARRIER();    // the consumers' last tile
    if (p.timeline && threadIdx.x == 0) {
      unsigned long long *b_ = p.timeline + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8;
      b_[0] = tq[0]; b_[1] = tq[1]; b_[2] = tq[2]; b_[3] = tq[3];
    }
    // bias gradient: row sums of gout over this block's pixels (channel block 0)
    if (p.gbias && blockIdx.y == 0) {
      MFN_UNROLL
      for (int i = 0; i < MTOT * 4; ++i) {
        float v = bsum[i];
        for (int sft = 16; sft >= 1; sft >>= 1) v += __shfl_xor(v, sft, 32);
        const int o = (tid >> 5) + 8 * i;
        if ((tid & 31) == 0 && o < p.Cout) {
          if (p.bias_slabs) p.bias_slabs[(size_t)blockIdx.x * (MTOT * 32) + o] = v;
          else if (v != 0.f) atomicAdd(p.gbias + o, v);
        }
      }
    }
    for (int f = 0; f < MTOT; ++f) {
      MFN_LDS_BARRIER();
      store_slab(f);
      MFN_LDS_BARRIER();
    }
  } else {
    f32x16 acc[UMAX];
    MFN_UNROLL
    for (int u = 0; u < UMAX; ++u)
      MFN_UNROLL
      for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
    MFN_LDS_BARRIER();
    MFN_LDS_BARRIER();
    for (int i = 0; i < ntile; ++i) {
      // ---- D[filter][channel] of tap t, filter tile f: accumulator tile u = t + 9 f, every fourth one is this wave's --------
      const float *colT = colB + (i & 1) * DCW_COL_F, *goutT = goutB + (i & 1) * GOUT_F;
      const unsigned long long cm0 = MFN_CYCLES();
      MFN_UNROLL
      for (int ul = 0; ul < UMAX; ++ul) {
        const int u = pw + 4 * ul;
        if (u < UN) {  // uniform
          const int t = u % 9, f = u / 9;
          const float4 *bp = reinterpret_cast<const float4 *>(colT + (t * 32 + j) * RS + half * 16);
          const float4 *ap = reinterpret_cast<const float4 *>(goutT + (f * 32 + j) * RS + half * 16);
          float4 a4[4], b4[4];
          MFN_UNROLL
          for (int q = 0; q < 4; ++q) { a4[q] = ap[q]; b4[q] = bp[q]; }
          MFN_UNROLL
          for (int q = 0; q < 4; ++q) {
            acc[ul] = MFN_MFMA_32x32x2(a4[q].x, b4[q].x, acc[ul]);
            acc[ul] = MFN_MFMA_32x32x2(a4[q].y, b4[q].y, acc[ul]);
            acc[ul] = MFN_MFMA_32x32x2(a4[q].z, b4[q].z, acc[ul]);
            acc[ul] = MFN_MFMA_32x32x2(a4[q].w, b4[q].w, acc[ul]);
          }
        }
      }
      const unsigned long long cm1 = MFN_CYCLES();
      MFN_LDS_BARRIER();
      tq[5] += MFN_CYCLES() - cm1; tq[4] += cm1 - cm0;
    }
    if (p.timeline && threadIdx.x == 256) {
      unsigned long long *b_ = p.timeline + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8;
      b_[4] = tq[4]; b_[5] = tq[5]; b_[6] = (unsigned long long)ntile; b_[7] = 1;
    }
    // the sums leave as contiguous (c, t) rows: D reg r of lane (j, half) = filter (r&3)+8*(r>>2)+4*half, channel j
    for (int f = 0; f < MTOT; ++f) {
      MFN_UNROLL
      for (int ul = 0; ul < UMAX; ++ul) {
        const int u = pw + 4 * ul;
        if (u < UN && u / 9 == f) {  // uniform
          const int t = u % 9;
          MFN_UNROLL
          for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * half) * DCW_STG + j * 9 + t] = acc[ul][r];
        }
      }
      MFN_LDS_BARRIER();
      store_slab(f);
      MFN_LDS_BARRIER();
    }
  }
}

// gw[o][c][t] (+)= sum over the blocks of a channel block of their slabs, in a fixed order (deterministic).  A block of 256
// threads = 16 consecutive gw elements x 16 groups of slabs, added through LDS (64 x 4: 20 us at level 2, too few loads in
// flight).
struct DcBwdWRParams { const float *slabs, *bias_slabs; float *gw, *gbias; int Cin, Cout, mtot, nblk, req_add, bias_add, wblocks; };
__global__ __launch_bounds__(256) void dc_bwd_weight_reduce_kernel(DcBwdWRParams p) {
  MFN_DYN_SHARED(float, part);  // [16 groups][16 elements]
  const int el = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const bool bias = (int)blockIdx.x >= p.wblocks;   // the blocks after the weights': gbias
  const size_t idx = (size_t)(bias ? blockIdx.x - p.wblocks : blockIdx.x) * 16 + el;
  const size_t total = bias ? (size_t)p.Cout : (size_t)p.Cout * p.Cin * 9;
  float sum = 0.f;
  // slabs grp, grp + 16, ... added in that order, eight requests in flight (one load, one wait, one add at a time was 16 round
  // trips per thread: most of this kernel's 8.7 us)
  auto slab_sum = [&](const float *base, size_t stride) {
    float acc = 0.f;
    for (int b = grp; b < p.nblk; b += 16 * 8) {
      float v[8];
      MFN_UNROLL
      for (int u = 0; u < 8; ++u) {
        const int bb = b + 16 * u;
        const float x = base[(size_t)min(bb, p.nblk - 1) * stride];
        v[u] = bb < p.nblk ? x : 0.f;
      }
      MFN_UNROLL
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    return acc;
  };
  if (idx < total) {
    if (bias) {
      sum = slab_sum(p.bias_slabs + idx, (size_t)p.mtot * 32);
    } else {
      const int o = (int)(idx / ((size_t)p.Cin * 9)), rem = (int)(idx - (size_t)o * p.Cin * 9);
      const int c = rem / 9, t = rem - c * 9;
      const int cbk = c >> 5, col = (c & 31) * 9 + t, f = o >> 5, ol = o & 31;
      const float *sp = p.slabs + (((size_t)cbk * p.nblk) * p.mtot + f) * (32 * 288) + ol * 288 + col;
      sum = slab_sum(sp, (size_t)p.mtot * (32 * 288));
    }
  }
  part[grp * 16 + el] = sum;
  __syncthreads();
  if (grp == 0 && idx < total) {
    float v = 0.f;
    for (int g2 = 0; g2 < 16; ++g2) v += part[g2 * 16 + el];
    float *dst = bias ? p.gbias : p.gw;
    dst[idx] = (bias ? p.bias_add : p.req_add) ? dst[idx] + v : v;
  }
}

}  // namespace mfn
