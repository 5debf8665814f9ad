// deform_conv_mma.h -- DeformableConvolution forward on the bf16 matrix cores of gfx950 (round 5).
//
// Replaces MXNet contrib.DeformableConvolution at /root/reference/network/layer.py:117-124 for the shapes the network
// uses it with (3x3, stride 1, pad 1, dilation 1, one group: MaskFlownet.py:155-158); semantics as
// oracle/mfn_ref_body.inc deform_conv_fwd.  Everything else stays on deform_conv.h (dc_lds_kernel / dc_generic_kernel).
//
//   out[o, p] = bias[o] + sum_{c, tap} W[o, c, tap] * col[c, tap, p],   col = bilinear gather of x at p + tap + offset(p)
//
// fp32-EQUIVALENT arithmetic on v_mfma_f32_32x32x16_bf16 (16x the rate of the fp32 MFMA, and beside the VALU instead of on
// its ALUs): both operands are written as three bf16 terms (hi + mid + lo = 24 significant bits, the split is exact) and SIX
// of the nine partial products -- those of weight >= 2^-16 -- are accumulated in fp32; the dropped ones are <= 2^-24 of a
// product each (~1 ulp per product, what an fp32 FMA chain loses per step anyway).  The acceptance rule is the one of the
// Gram cost volume: error against the fp64 oracle within 1.25 x the exact-fp32 kernel's (observed 0.7-1.1 x; tests/test_gpu_parity.py
// test_deform_mma_error_vs_fp64).  Non-finite inputs differ: an inf column value splits into inf + NaN.
//
// What changed against dc_lds_kernel<.., MMA = 1> (round 3/4, removed), from its timelines (profiles/r05_dc_probe_before.txt:
// level 2 = 0.7 us start + 4.4 set-up + 12.0 loop + 1.5 epilogue + 4 us of late blocks; 16.2 us with every matrix
// instruction, gather and DMA compiled out):
//   * the weights are split ONCE, at pack time (dcm_pack_weights_kernel); a column value is split once per (pixel, channel)
//     and its B operand is reused by ALL MT filter tiles of the wave and all six products (before: one 32-filter tile per
//     wave, the gather / interpolation / split repeated per filter group);
//   * no fp32 MFMA: K is ordered in GROUPS of 16 channels = 8 pair steps (lane (pixel j, half kb) owns channel 2 s + kb:
//     its taps 0..7 are the eight K elements of its k-block) + 1 left-over step whose K elements are tap 8 of the lane's
//     eight channels of the group -- 9 K = 16 steps for 144 (channel, tap) pairs, nothing padded (before: tap 8 on
//     v_mfma_f32_32x32x2_f32, 64 cycles per step on the VALU's ALUs);
//   * a wave's MFMAs never wait for each other: MT >= 2 interleaves the filter tiles, MT = 1 alternates two accumulators;
//   * B(t + 1) is formed completely (gather, interpolation, split) under the MFMAs of step t (double-buffered operands);
//   * set-up for the shared-offset case only (the reference feeds one offset to all nine taps, MaskFlownet.py:230): one
//     floor per axis, the 4x4 neighbourhood at UNCLAMPED consecutive rows / columns of a zero-filled window (clamped taps
//     carry weight 0 on the row / column behind the image, so the clamp never has to move an address): ONE LDS address
//     register and 16 immediates instead of 16 address registers;
//   * every DMA is issued unconditionally (prefetches past the end of a K slice read zeros or the next slice and are never
//     used): the issue sequence is periodic, every s_waitcnt count is a compile-time constant (dcm_wait_count);
//   * lanes whose neighbourhood falls outside the wave's 12 x 20 window (2 % of the tiles of the bench flows; before: the
//     whole tile fell to the slower global-gather tier and its block finished ~4 us after the others) fetch their 16 values
//     from global memory under an exec mask, everybody else keeps the window;
//   * the K-slice reduction and the epilogue run on the waves in parallel (tile mt is summed and stored by slice mt % KW).
// Per-tap offsets (MXNet's general semantics, never hot in the reference) take a lean per-tap column path inside the same kernel.
#pragma once
#include "deform_conv.h"
#include "msplit.h"

namespace mfn {

// The wave's source window of one channel pair: [channel 0/1][ROWS][COLS] floats, transferred as NI 1 KB wave instructions into a
// slot of the wave's ring (RING * 512 floats).  Two shapes: 12 x 20 (three slots: the window of pair k + 3 is requested in step k) for
// the tiles whose neighbourhoods fit it (>= 98 % under the bench's flows); 16 x 24 (two slots of 768 floats in the same ring: one
// step of cover instead of two) for the rest -- before, such a tile fell to per-lane global loads and its block finished ~6 us
// after the others, which is what the whole launch then takes.
constexpr int DCM_XW_F = 512;      // floats of a ring slot of the small window
template <bool BIG, int RING> struct DcmWin {
  static constexpr int ROWS = BIG ? 16 : 12, COLS = BIG ? 24 : 20, C4 = COLS / 4, CH = ROWS * COLS;
  // floats between the slot's two channels.  Small window: 268 = 12 mod 64 -- a wave's 32 pixels (4 rows x 8 columns, 20 floats per
  // row) read banks {0..7, 20..27, 40..47, 60..3} for the lanes of channel 0; channel 1 at + 240 (= 48 mod 64) lands on
  // {48..55, 4..11, 24..31, 44..51}, half of it on banks of channel 0 (profiles/r05_dc_pmc.md: 38 % of the LDS cycles were bank
  // conflicts); at + 268 it reads {12..19, 32..39, 52..59, 8..15}: nothing shared with channel 0.  The 28 floats between are 7 of the
  // transfer's 128 float4 slots (2 x 60 + 7 <= 128: still two instructions).  (Launch times did not move: LDS waits were 6 % of the
  // wave cycles before.)
  static constexpr int CHS = BIG ? CH : 268;
  static constexpr bool IS_BIG = BIG;
  static constexpr int NI = BIG ? 3 : 2, SLOT_F = BIG ? 768 : 512;
  static constexpr int DEPTH = BIG ? (RING * DCM_XW_F) / 768 : RING;   // slots = how many pairs ahead a window is requested
  static_assert(CHS % 4 == 0 && CHS >= CH && CHS + CH <= SLOT_F && CHS + CH <= NI * 256 && DEPTH >= 2, "the pair's window fits its slot and its transfers");
};

// Plain 3x3 / stride 1 / pad 1 convolution (dc_mma_kernel<.., CONV = true>, mfn_conv2d_fwd): every tap sits ON a pixel, so the window is
// the tile's 4 x 8 pixels + a one-pixel border -- 6 rows x 16 columns from the 16-byte aligned column 4 left of the tile -- and a pair
// is ONE transfer instruction (2 x 24 float4 slots of its 64).  The two channels lie 104 floats apart: channel 0's lanes then read
// banks {3..10, 19..26, 35..42, 51..58} and channel 1's + 8 of that: no bank shared.  Six slots fit the wave's ring: six pairs ahead.
template <int RING> struct DcmWinConv {
  static constexpr int ROWS = 6, COLS = 16, C4 = 4, CH = ROWS * COLS, CHS = 104;
  static constexpr int NI = 1, SLOT_F = 256;
  static constexpr int DEPTH = (RING * DCM_XW_F) / SLOT_F;
  static_assert(CHS % 4 == 0 && CHS >= CH && CHS + CH <= SLOT_F && DEPTH >= 2, "the pair's window fits one transfer");
};

// K steps per weight chunk (= block barrier period).  One M-group's step is 3 * MT KB per K slice.
constexpr int dcm_kc(int mt, int kw) { return mt * kw == 1 ? 3 : 1; }
// stage buffers of the weight ring: chunk c + NSTAGE - 1 is requested when chunk c opens (KC = 1: two steps of latency cover,
// where three stages + the window rings + the tap-8 slots still fit the CU's 160 KB)
constexpr int dcm_nstage(int mt, int pt, int kw, int ring) {
  if (dcm_kc(mt, kw) != 1) return 2;
  const int words3 = 3 * kw * 3 * mt * 256 + pt * kw * (ring * DCM_XW_F + 512) + 256;
  const int budget = pt * kw <= 4 ? 80 * 1024 : 160 * 1024;   // blocks of four waves are meant to run (at least) two per CU
  return words3 * 4 <= budget ? 3 : 2;
}
// waves per SIMD the register allocator is held to: what one block needs to fit a CU at all (nw / 4), and three / two / one for
// one / two / more filter tiles per wave (level 2: three 4-wave blocks per CU = the whole launch in one residency round)
constexpr int dcm_min_waves(int mt, int nw) {
  const int fit = (nw + 3) / 4, want = mt == 1 ? 3 : (mt == 2 ? 2 : 1);
  return fit > want ? fit : want;
}

// measurement builds only (tools/dcm_ablate_build.py): 1 no matrix instructions, 2 no LDS gather reads, 4 no window transfers,
// 8 no operand split, 16 no interpolation, 32 no weight transfers, 64 no weight reads, 128 no block barrier in the loop
#ifndef MFN_DCM_ABLATE
#define MFN_DCM_ABLATE 0
#endif
// The column values' three bf16 terms by msplit.h (12 v_cvt_pk_bf16_f32 + 4 v_mfma_f32_4x4x4_16B_bf16 per K step, lane-local, the same
// terms bit for bit) instead of the 46 VALU instructions of mfn_split3x8_scalar, where it pays: one or two filter tiles per wave (round 6,
// same box: level 2 20.50 -> 19.57 us, level 3 17.53 -> 17.25, level 5 11.29 -> 10.97; three tiles per wave -- level 4, one wave per
// SIMD -- 13.47 -> 13.82: stays on the VALU).  1 / 0 force it on / off (measurement builds).
#ifndef MFN_DCM_MSPLIT
#define MFN_DCM_MSPLIT (MT <= 2)
#endif
#ifndef MFN_DCM_MINW   // measurement builds override the register budget
#define MFN_DCM_MINW(mt, nw) dcm_min_waves(mt, nw)
#endif

template <int MT, int PT, int KW, int RING> struct DcmGeom {
  static constexpr int NW = PT * KW, NTH = NW * 64;
  static constexpr int KC = dcm_kc(MT, KW);
  static constexpr int NSTAGE = dcm_nstage(MT, PT, KW, RING);
  static constexpr int STEP_W = 3 * MT * 256;          // words of one K step's A operands (one M-group): [term][ft][kb][m][8 bf16]
  static constexpr int WI_TOTAL = KW * KC * 3 * MT;    // 1 KB wave transfers per weight chunk (all K slices of the block)
  static constexpr int NI = (WI_TOTAL + NW - 1) / NW;  // ... per wave at most; a wave whose last one would be past WI_TOTAL issues one fewer
  static constexpr int NI_MIN = WI_TOTAL / NW;         // ... at least: the waits count with this (a wave that issued one more waits for it too)
  static constexpr int STAGE_W = WI_TOTAL * 256;       // words per stage buffer
  // K-slice reduction through the (then idle) stage buffers: every slice's MT tiles at once where they fit -- one barrier --, else
  // tile by tile (one 32 x 32 tile per non-owner slice, two barriers per tile)
  static constexpr bool RED_ALL = KW > 1 && PT * KW * MT * 1024 <= NSTAGE * STAGE_W;
  static constexpr int RED_W = KW > 1 ? (RED_ALL ? PT * KW * MT * 1024 : PT * (KW - 1) * 1024) : 0;
  static constexpr int XW_OFF = NSTAGE * STAGE_W > RED_W ? NSTAGE * STAGE_W : RED_W;
  static constexpr int T8_OFF = XW_OFF + NW * RING * DCM_XW_F;      // tap 8 of a group's eight pairs: [wave][pair][lane]
  static constexpr int DUMP_OFF = T8_OFF + NW * 512;
  static constexpr int LDS_W = DUMP_OFF + 256;
  static_assert(LDS_W * 4 <= 160 * 1024, "a block's LDS must fit the CU");
};

// s_waitcnt vmcnt count at the head of step s (0..8) of a group, from a simulation of the periodic issue sequence: a step
// issues, after its wait, [NI weight transfers of chunk c + NSTAGE - 1 if it opens chunk c (s % KC == 0)] then [XW_NI window
// transfers of pair k + RING if s < 8].  Needed at the wait: the window of the next pair step (gathered in this step; none in
// step 7) and, at a chunk boundary, this step's weight chunk.  Transfers complete in issue order, so the count is the number
// issued after the youngest needed one.  1000 = nothing needed.
// REQ: requests of the lanes outside the window (the tier that has them), issued behind the step's transfers by every step that
// prepares a pair (all but step 7) -- they join the same in-order queue.
template <int KC, int NSTAGE, int RING, int NI, int XW_NI, int REQ = 0> constexpr int dcm_wait_count(int s) {
  int issued = 0, result = 1000;
  int stampW[64] = {}, stampX[80] = {};
  for (int u = 0; u < 45; ++u) {
    const int ss = u % 9, grp = u / 9;
    if (u == 36 + s) {
      int need = -1;
      if (ss % KC == 0 && stampW[u / KC] > need) need = stampW[u / KC];
      if (ss != 7) {
        const int np = ss == 8 ? 8 * (grp + 1) : 8 * grp + ss + 1;
        if (stampX[np] > need) need = stampX[np];
      }
      result = need < 0 ? 1000 : issued - need;
    }
    if (ss % KC == 0) { issued += NI; stampW[u / KC + NSTAGE - 1] = issued; }
    if (ss < 8) { issued += XW_NI; stampX[8 * grp + ss + RING] = issued; }
    if (ss != 7) issued += REQ;
  }
  return result;
}
// ... and the count with which a preparing step (s != 7) waits for the requests it consumes -- those of the PREVIOUS preparing
// step --, at the point behind its own transfers and requests: everything issued since
template <int KC, int NSTAGE, int RING, int NI, int XW_NI, int REQ> constexpr int dcm_landed_count(int s) {
  int issued = 0, last_req = 0, result = 0;
  for (int u = 0; u < 45; ++u) {
    const int ss = u % 9;
    if (ss % KC == 0) issued += NI;
    if (ss < 8) issued += XW_NI;
    if (ss != 7) {
      issued += REQ;
      if (u == 36 + s) result = issued - last_req;
      last_req = issued;
    }
  }
  return result;
}

// the steps that wait with the same count as step s, as a bit mask -- for the first such step only (0 for the others)
template <int KC, int NSTAGE, int RING, int NI, int XW_NI> constexpr unsigned dcm_wait_mask(int s) {
  const int v = dcm_wait_count<KC, NSTAGE, RING, NI, XW_NI>(s);
  unsigned m = 0;
  for (int q = 0; q < 9; ++q) {
    if (dcm_wait_count<KC, NSTAGE, RING, NI, XW_NI>(q) != v) continue;
    if (q < s) return 0;
    m |= 1u << q;
  }
  return m;
}

// transfers the prologue issues after window 0: what may still be outstanding when pair 0 is gathered
template <int KC, int NSTAGE, int RING, int NI, int XW_NI> constexpr int dcm_prologue_after_x0() {
  int issued = 0, at_x0 = -1;
  for (int u = -36; u < 0; ++u) {
    const int ss = ((u % 9) + 9) % 9;
    const int kl = (u - ss) / 9 * 8 + ss;
    if (ss % KC == 0 && u / KC + NSTAGE - 1 >= 0) issued += NI;
    if (ss < 8 && kl + RING >= 0) { issued += XW_NI; if (kl + RING == 0) at_x0 = issued; }
  }
  return issued - at_x0;
}

// weights (Cout, Cin, 9) -> packed[mg][grp][s 0..8][term h|m|l][ft][kb][m 0..31][e 0..7] bf16; filter o = (mg * MT + ft) * 32 + m;
// s < 8: channel 16 grp + 2 s + kb, tap e; s == 8: channel 16 grp + 2 e + kb, tap 8.  Zero padded in o and c.  One thread per
// (mg, grp, s, ft, kb, m); one M-group is one linear array (+ one chunk of padding: the prefetch past the last step).
struct DcmPackParams { const float *w; float *wt; int Cin, Cout, MT, mgroups, ngroups; size_t mg_words; };
__global__ __launch_bounds__(256) void dcm_pack_weights_kernel(DcmPackParams p) {
  const size_t total = (size_t)p.mgroups * p.ngroups * 9 * p.MT * 64;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int m = (int)(idx & 31), kb = (int)((idx >> 5) & 1);
  size_t r = idx >> 6;
  const int ft = (int)(r % p.MT); r /= p.MT;
  const int s = (int)(r % 9); r /= 9;
  const int grp = (int)(r % p.ngroups);
  const int mg = (int)(r / p.ngroups);
  const int o = (mg * p.MT + ft) * 32 + m;
  float x[8];
  MFN_UNROLL
  for (int e = 0; e < 8; ++e) {
    const int c = s < 8 ? 16 * grp + 2 * s + kb : 16 * grp + 2 * e + kb;
    const int tap = s < 8 ? e : 8;
    x[e] = (o < p.Cout && c < p.Cin) ? p.w[((size_t)o * p.Cin + c) * 9 + tap] : 0.f;
  }
  mfn_bf16x8 h, mm, l;
  mfn_split3x8(x, h, mm, l);
  float *base = p.wt + (size_t)mg * p.mg_words + ((size_t)grp * 9 + s) * (3 * p.MT * 256) + (size_t)(kb * 32 + m) * 4;
  mfn_write_bf16x8(base + (0 * p.MT + ft) * 256, h);
  mfn_write_bf16x8(base + (1 * p.MT + ft) * 256, mm);
  mfn_write_bf16x8(base + (2 * p.MT + ft) * 256, l);
}
inline int dcm_pack_launch(DcmPackParams pp, hipStream_t stream) {
  const size_t total = (size_t)pp.mgroups * pp.ngroups * 9 * pp.MT * 64;
  return launch("dcm_pack_weights", dcm_pack_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, pp);
}

struct DcmB { mfn_bf16x8 h, m, l; };

// CONV: the plain 3x3 / stride 1 / pad 1 convolution on the same machinery (no offsets: the nine taps ARE the 3 x 3 pixels around the
// output pixel, read straight from the window; Cin need not be a multiple of 16 -- the packed weights of the missing channels are
// zero and their window transfers are suppressed; x and out may be channel slices of concat buffers).
template <int MT, int PT, int KW, int RING, bool CONV = false>
__global__ __launch_bounds__(PT * KW * 64, MFN_DCM_MINW(MT, PT * KW)) void dc_mma_kernel(DeformParams p) {
  using G = DcmGeom<MT, PT, KW, RING>;
  constexpr int NW = G::NW, KC = G::KC, NI = G::NI, NIW = G::NI_MIN;
  constexpr int NACC = MT == 1 ? 2 : 1;   // MT = 1: two accumulators take the products alternately (no dependent MFMA pair)
  // One pixel tile per block: every wave is a K slice of its own and nobody shares its weights -- each wave then transfers exactly
  // its own slice's blocks into its part of the stage buffers and the K loop needs NO block barrier (a wave's own program order
  // says when a stage is free).  With several pixel tiles the waves of a K slice share one copy, staged cooperatively.
  constexpr bool PRIVATE_W = PT == 1;
  static_assert(!PRIVATE_W || G::NI * NW == G::WI_TOTAL, "one pixel tile: the slices' blocks divide evenly over the waves");
  static_assert(9 % KC == 0, "chunk boundaries at fixed steps of a group");
  static_assert(RING >= 3 && RING * DCM_XW_F >= 32 * 40, "the epilogue transposes a tile through the wave's window ring");
  MFN_DYN_SHARED(float, lds);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = MFN_UNIFORM(tid >> 6);
  const int kb = lane >> 5, j = lane & 31;
  const int pt = wave / KW, kw = wave % KW;
  MFN_STAMP(p.timeline, 0);
  const int bx = p.xcd ? (int)mfn_xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int tile = bx * PT + pt;
  const int mg = blockIdx.z;
  const int m0 = mg * MT * 32;
  const int H = p.H, W = p.W;
  const size_t plane = (size_t)H * W;
  const size_t xns = p.x_nstride ? p.x_nstride : (size_t)p.Cin * plane;   // elements between the images of x

  // ---- weight staging plan ------------------------------------------------------------------------------------------
  const int gps = p.dcm_gps;   // groups per K slice
  const size_t mg_words = (size_t)(p.dcm_groups * 9 + KC) * G::STEP_W;
  const mfn_rsrc_t wrsrc = mfn_make_rsrc(p.wt + (size_t)mg * mg_words, (unsigned)(mg_words * 4));
  unsigned voffW[NI];
  MFN_UNROLL
  for (int i = 0; i < NI; ++i) {
    const int ii = PRIVATE_W ? wave * (KC * 3 * MT) + i : i * NW + wave;   // PRIVATE_W: the wave's own slice (ii / (KC 3 MT) == kw)
    const int k = ii / (KC * 3 * MT), rem = ii - k * (KC * 3 * MT);
    voffW[i] = ii < G::WI_TOTAL ? (unsigned)((((size_t)k * gps * 9) * G::STEP_W + (size_t)rem * 256) * 4) + (unsigned)lane * 16u : 0xFFFFFF00u;
  }
  auto issue_w = [&](int ch) {   // chunk ch of every K slice of the block -> stage ch % NSTAGE
    float *buf = lds + (ch % G::NSTAGE) * G::STAGE_W;
    // a chunk past the slice's last step (the unconditional prefetch) is requested out of range: zero fill, no memory access
    const unsigned soff = ch * KC < gps * 9 ? (unsigned)((size_t)ch * KC * G::STEP_W * 4) : 0x7FFFFF00u;
    if (MFN_DCM_ABLATE & 32) return;
    MFN_UNROLL
    for (int i = 0; i < NI; ++i) {
      const int ii = PRIVATE_W ? wave * (KC * 3 * MT) + i : i * NW + wave;
      if (ii < G::WI_TOTAL) mfn_dma16_so(wrsrc, buf + ii * 256, voffW[i], soff);   // wave-uniform
    }
  };

  // ---- the wave's pixel tile: 4 rows x 8 columns of one image ---------------------------------------------------------
  int n, ty, tx;
  {
    auto divmod = [](int a, int b, float inv_b, int &q, int &r) {   // a / b for a, b < 2^24 without an integer division
      q = (int)((float)a * inv_b);
      r = a - q * b;
      if (r < 0) { --q; r += b; }
      if (r >= b) { ++q; r -= b; }
    };
    const int tl = min(tile, p.ntiles - 1);
    int rt;
    divmod(tl, p.tiles_y * p.tiles_x, p.inv_tpi, n, rt);
    divmod(rt, p.tiles_x, p.inv_tiles_x, ty, tx);
  }
  const int tile_ho0 = ty * 4, tile_wo0 = tx * 8;
  int ho = tile_ho0 + (j >> 3), wo = tile_wo0 + (j & 7);
  const bool px_valid = tile < p.ntiles && ho < H && wo < W;
  ho = min(ho, H - 1);
  wo = min(wo, W - 1);
  const int h_in = ho - 1, w_in = wo - 1;   // stride 1, pad 1

  // ---- offsets: one (dy, dx) per pixel (flow mode), or the eighteen of the operator's tensor, which qualify when equal --------
  float off_h = 0.f, off_w = 0.f;
  bool shared = true;
  if (CONV) {
  } else if (p.offset) {
    const float *op = p.offset + (size_t)n * 18 * plane + (size_t)ho * W + wo;
    float oh[9], ow[9];
    MFN_UNROLL
    for (int t = 0; t < 9; ++t) {
      oh[t] = op[(size_t)(2 * t) * plane];
      ow[t] = op[(size_t)(2 * t + 1) * plane];
    }
    MFN_COMPILER_FENCE();   // all eighteen requested before the first is compared (deform_conv.h)
    int same = 1;
    MFN_UNROLL
    for (int t = 1; t < 9; ++t) same &= (int)(oh[t] == oh[0]) & (int)(ow[t] == ow[0]);
    shared = same != 0;
    off_h = oh[0];
    off_w = ow[0];
  } else {
    const float *fp = p.flow + (size_t)n * 2 * plane + (size_t)ho * W + wo;
    off_h = fp[0] * p.flow_scale / p.flow_stride;       // MaskFlownet.py:230
    off_w = fp[plane] * p.flow_scale / p.flow_stride;
  }

  MFN_STAMP2(p.timeline, 0);   // offsets requested and compared
  // ---- geometry of the shared-offset path: per tap row / column the pair of weights on neighbourhood lines i, i + 1 ----------
  float ya[3], yb[3], xa[3], xb[3];
  int row0, col0;                  // unclamped first row / column of the 4x4 neighbourhood
  bool regular = shared && p.allow_fast != 0;
  {
    // tap i sits on neighbourhood lines (i, i + 1) with weights (1 - l, l) when floor(i + off) = floor(off) + i.  In fp32 the sum
    // i + off can round UP to the next integer (off a hair below one: floor(0 + off) = -1, 1 + off == 1.0f) -- the oracle then
    // reads line i + 1 with weight 1 (l = 0), which is the same pair of lines with weights (0, 1): still the shared path
    // (ONE such pixel in a level-2 batch used to send its whole tile to the per-tap path, 75 us instead of 25)
    auto axis = [&](float off, int in0, int dim, float (&wa)[3], float (&wb)[3], int &first) {
      int lo0 = 0;
      MFN_UNROLL
      for (int i = 0; i < 3; ++i) {
        bool v; int lo, hi; float l;
        dc_axis(off, in0, i, dim, v, lo, hi, l);
        v = v && px_valid;
        const int ulo = (int)fminf(fmaxf(floorf((float)i + off), -1.0e6f), 1.0e6f);
        if (i == 0) lo0 = ulo;
        const bool up = ulo == lo0 + i + 1 && l == 0.f;   // rounded up to an integer
        regular = regular && (ulo == lo0 + i || up);
        wa[i] = v ? (up ? 0.f : 1.f - l) : 0.f;
        wb[i] = v ? (up ? 1.f : l) : 0.f;
      }
      first = in0 + lo0;
    };
    if (CONV) {   // the nine taps are the pixels (h_in + i, w_in + j): nothing to interpolate, rows / columns outside the image read
      row0 = h_in;  // the window's zeros
      col0 = w_in;
      MFN_UNROLL
      for (int i = 0; i < 3; ++i) { ya[i] = xa[i] = 1.f; yb[i] = xb[i] = 0.f; }
    } else {
      axis(off_h, h_in, H, ya, yb, row0);
      axis(off_w, w_in, W, xa, xb, col0);
    }
  }
  const bool fast = CONV || __all(regular || !px_valid) != 0;   // wave-uniform; false: the per-tap column path

  MFN_STAMP2(p.timeline, 1);   // geometry
  // ---- the wave's source window: the box of its lanes' neighbourhoods in the small shape where that fits, else the big one, and
  // the lanes outside even that fetch from global memory ---------------------------------------------------------------------------
  int wr0 = 0, wc0 = 0;
  bool inwin = true;
  unsigned xvoff[3] = {0xFFFFFF00u, 0xFFFFFF00u, 0xFFFFFF00u};
  int gofs = 0;   // index into lds[] of this lane's first neighbourhood value inside slot 0: ONE address register, the 16 reads
                  // are its immediates
  auto setup_window = [&](auto win_c) {
    using WN = decltype(win_c);
    const int big = 1 << 28;
    const bool use = px_valid && fast;
    wr0 = mfn_wave_min_i32(use ? row0 : big);
    wc0 = mfn_wave_min_i32(use ? col0 : big);
    // one lane with a wild offset must not drag the window away from everybody else: never further up / left of the tile's
    // centre lane than the window can reach back from it (rows / columns outside the image are zero-filled, wherever they are)
    const int rc = mfn_readlane_i32(row0, 12), cc = mfn_readlane_i32(col0, 12);
    wr0 = wr0 == big ? 0 : max(wr0, rc - (WN::ROWS - 4));
    wc0 = (wc0 == big ? 0 : max(wc0, cc - (WN::COLS - 4))) & ~3;   // 16-byte aligned origin (two's complement: also negative)
    inwin = !px_valid || (row0 >= wr0 && row0 - wr0 <= WN::ROWS - 4 && col0 >= wc0 && col0 - wc0 <= WN::COLS - 4);
    if (WN::IS_BIG && fast && !__all(inwin)) {   // (wave-uniform)
      // The neighbourhoods do not fit even the big window: lanes will be left outside, and which ones is the choice of the origin.
      // The box's corner (above) serves the lanes at the low end -- with ONE wild lane up or left of the tile that is the clamp's
      // [centre - 12, centre]: every lane below / right of the centre lane is outside (SURVEY 8(d)'s flows: 2 % wild lanes, ~30 of a
      // wave's 64 lanes outside, each costing 16 requests per step).  Centre the window on the tile instead: the tile's first
      // row / column as the median of three lanes' estimates (one wild lane among them does not move it), the window's slack
      // split evenly round the tile's 4 x 8 pixels.
      auto med3 = [](int a, int b, int c) { return max(min(a, b), min(max(a, b), c)); };
      const int br = med3(mfn_readlane_i32(row0, 12) - 1, mfn_readlane_i32(row0, 19) - 2, mfn_readlane_i32(row0, 9) - 1);
      const int bc = med3(mfn_readlane_i32(col0, 12) - 4, mfn_readlane_i32(col0, 19) - 3, mfn_readlane_i32(col0, 9) - 1);
      // (refining the estimate by the clamped mean deviation of all 32 pixels changed nothing measurable: 49.4 k -> 50.2 k pairs/s on a
      // box whose headline was 3 % higher)
      wr0 = br - (WN::ROWS - 4 - 3) / 2;
      wc0 = (bc - ((WN::COLS - 4 - 7) / 2 - 1)) & ~3;   // (the 16-byte alignment takes 0..3 off: 5..8 columns left of the tile, 5..8 right)
      inwin = !px_valid || (row0 >= wr0 && row0 - wr0 <= WN::ROWS - 4 && col0 >= wc0 && col0 - wc0 <= WN::COLS - 4);
    }
    MFN_UNROLL
    for (int i = 0; i < WN::NI; ++i) {
      const int slot = i * 64 + lane;                       // float4 slots: [channel 0/1][ROWS][C4 float4]
      const int chs = slot / (WN::CHS / 4), rem = slot - chs * (WN::CHS / 4);   // (rem past the channel's rows: the padding, or the slot's end)
      const int row = rem / WN::C4, c4 = rem - row * WN::C4;
      const int r = wr0 + row, c = wc0 + 4 * c4;
      xvoff[i] = (fast && chs < 2 && row < WN::ROWS && r >= 0 && r <= H - 1 && c >= 0 && c <= W - 4)
                     ? (unsigned)(((size_t)n * xns + (size_t)chs * plane + (size_t)r * W + c) * 4)
                     : 0xFFFFFF00u;                         // outside the image: never read, the transfer writes zeros
    }
    // lanes that are not in the window read the slot's first value (finite)
    gofs = (px_valid && inwin && fast) ? kb * WN::CHS + (row0 - wr0) * WN::COLS + (col0 - wc0) : 0;
    gofs += G::XW_OFF + wave * (RING * DCM_XW_F);
    MFN_OPAQUE(gofs);   // as an opaque sum: hipcc otherwise keeps XW_OFF apart and forms eight addresses per step
  };
  unsigned xvoff_c0 = 0xFFFFFF00u;   // CONV: the transfer's lanes of channel 0 only (the last pair of an odd channel count)
  bool all_in = true, bigwin = false;
  if (CONV) {
    using WN = DcmWinConv<RING>;
    wr0 = tile_ho0 - 1;
    wc0 = tile_wo0 - 4;
    const int chs = lane / (WN::CHS / 4), rem = lane - chs * (WN::CHS / 4);
    const int row = rem / WN::C4, c4 = rem - row * WN::C4;
    const int r = wr0 + row, c = wc0 + 4 * c4;
    const bool ok = chs < 2 && row < WN::ROWS && r >= 0 && r <= H - 1 && c >= 0 && c <= W - 4;
    xvoff[0] = ok ? (unsigned)(((size_t)n * xns + (size_t)chs * plane + (size_t)r * W + c) * 4) : 0xFFFFFF00u;
    xvoff_c0 = chs == 0 ? xvoff[0] : 0xFFFFFF00u;
    gofs = kb * WN::CHS + (row0 - wr0) * WN::COLS + (col0 - wc0);
    gofs += G::XW_OFF + wave * (RING * DCM_XW_F);
    MFN_OPAQUE(gofs);
  } else {
    setup_window(DcmWin<false, RING>{});
    all_in = __all(inwin) != 0;   // wave-uniform
    bigwin = fast && !all_in;
    if (bigwin) {
      setup_window(DcmWin<true, RING>{});
      all_in = __all(inwin) != 0;
    }
  }
  float *xwin = lds + G::XW_OFF + wave * (RING * DCM_XW_F);
  const mfn_rsrc_t xrsrc = mfn_make_rsrc(p.x, (unsigned)((size_t)p.N * xns * 4));
  const int cp_base = kw * gps * 8;    // first channel pair of this K slice
  int buf_issue = 0;                   // ring slot of the next window transfer (uniform)
  auto issue_x = [&](int kl, auto win_c) {   // window of the slice's pair kl -> ring slot kl % DEPTH
    using WN = decltype(win_c);
    const int c0 = 2 * (cp_base + kl);   // CONV: channels past Cin (the zero-padded tail of the last 16-channel group) are not read --
                                         // behind a concat slice lies the next image's memory, written or not
    const unsigned soff = (kl < gps * 8 && (!CONV || c0 < p.Cin)) ? (unsigned)((size_t)c0 * plane * 4) : 0x7FFFFF00u;   // past the slice: out of range
    float *dst = xwin + buf_issue * WN::SLOT_F;
    if (!(MFN_DCM_ABLATE & 4)) {
      if (CONV) mfn_dma16_so(xrsrc, dst, c0 + 1 < p.Cin ? xvoff[0] : xvoff_c0, soff);
      else {
        MFN_UNROLL
        for (int i = 0; i < WN::NI; ++i) mfn_dma16_so(xrsrc, dst + i * 256, xvoff[i], soff);
      }
    }
    buf_issue = buf_issue + 1 == WN::DEPTH ? 0 : buf_issue + 1;
  };

  // ---- prologue: what the steps before the first would have issued, in their order (chunks 0 .. NSTAGE - 2, windows 0 .. DEPTH - 1) ----
  auto prologue = [&](auto win_c) {
    using WN = decltype(win_c);
    constexpr int U0 = -9 * 4;   // far enough back for every prefetch distance
    MFN_UNROLL
    for (int u = U0; u < 0; ++u) {
      const int ss = ((u % 9) + 9) % 9;
      const int kl = (u - ss) / 9 * 8 + ss;                                  // pair of step u (ss < 8)
      if (ss % KC == 0 && (u - ss % KC) / KC + G::NSTAGE - 1 >= 0) issue_w((u - ss % KC) / KC + G::NSTAGE - 1);
      if (ss < 8 && kl + WN::DEPTH >= 0) issue_x(kl + WN::DEPTH, win_c);
    }
  };
  MFN_STAMP2(p.timeline, 2);   // window box
  if (CONV) prologue(DcmWinConv<RING>{}); else if (bigwin) prologue(DcmWin<true, RING>{}); else prologue(DcmWin<false, RING>{});
  MFN_STAMP2(p.timeline, 3);   // prologue transfers issued

  f32x16 acc[MT][NACC];
  MFN_UNROLL
  for (int mt = 0; mt < MT; ++mt)
    MFN_UNROLL
    for (int a = 0; a < NACC; ++a)
      MFN_UNROLL
      for (int r = 0; r < 16; ++r) acc[mt][a][r] = 0.f;

  const float *xn = p.x + (size_t)n * xns;

  // ---- column values of one (pixel, channel): the tiers ------------------------------------------------------------------------------
  // window (small or big shape): 16 LDS reads at one address register + immediates; the big window with the lanes that are outside
  // even that reading global memory at clamped rows / columns (their weights on lines outside the image are zero); per-tap offsets:
  // pertap_pair below.
  f32x2 vp[4][2];   // the 4x4 neighbourhood of the pair in preparation: vp[m][h] = columns 2h, 2h + 1 of row m (one ds_read2_b32)
  int buf_read = 0;  // ring slot of the next pair to gather (uniform)
  auto cols_gather = [&](int kn, auto win_c) {
    using WN = decltype(win_c);
    const float *xb_ = lds + (buf_read * WN::SLOT_F + gofs);
    if (CONV) {   // the 3 x 3 pixels: columns 0, 1 as a pair, column 2 alone
      MFN_UNROLL
      for (int m = 0; m < 3; ++m) {
        vp[m][0].x = xb_[m * WN::COLS];
        vp[m][0].y = xb_[m * WN::COLS + 1];
        vp[m][1].x = xb_[m * WN::COLS + 2];
      }
      buf_read = buf_read + 1 == WN::DEPTH ? 0 : buf_read + 1;
      return;
    }
    MFN_UNROLL
    for (int m = 0; m < 4; ++m)
      MFN_UNROLL
      for (int h = 0; h < 2; ++h) {
        vp[m][h].x = (MFN_DCM_ABLATE & 2) ? (float)(m + h + kn) : xb_[m * WN::COLS + 2 * h];
        vp[m][h].y = (MFN_DCM_ABLATE & 2) ? (float)(m - h + kn) : xb_[m * WN::COLS + 2 * h + 1];
      }
    buf_read = buf_read + 1 == WN::DEPTH ? 0 : buf_read + 1;
  };
  // Lanes whose neighbourhood lies outside the (big) window read it from global memory: rows clamped to the image, the four
  // columns as they are (immediate offsets of ONE address per row) -- every tap that touches a row or column outside the image has
  // weight zero (dc_axis: a valid tap's two lines are inside), so what is read there only has to be finite, and past either end of
  // the tensor the descriptor's range check returns 0.  The requests are statements hipcc does not count (one it knows about makes
  // it wait with a vmcnt that also drains the window / weight transfers of later steps, at every step), issued behind the step's
  // transfers under an exec mask of the outside lanes (mfn_bload1x4_async: the addresser's time goes with the ACTIVE lanes), TWO
  // preparations ahead: two register sets, vq[e & 1] for the e-th preparation of a group (eight per group: steps 0..6 and 8).
  // The waits count them (dcm_wait_count's REQ, dcm_landed_count).  Per step: 16 requests + 16 selects.
  // History on SURVEY 8(d)'s i.i.d. flow (sigma 2 px at EVERY level + 2 % wild: ~4 of a wave's 64 lanes outside), level-4 loop of a
  // block with such lanes against 7.1 us without: 10.8 (16 clamped addresses per step inside a divergent branch, known loads)
  // -> 11.3 (unknown loads, all lanes) -> 10.5 (two sets ahead) -> 9.2 us (exec mask); whole pass 42.3k -> 43.1k pairs/s;
  // the big window centred on the tile instead of cornered on its lowest lane (setup_window): 8.8 us, 49.4k pairs/s = 0.84 of `value`.
  float vq[2][16];
  unsigned rbo[4] = {0u, 0u, 0u, 0u};   // set by the tier that uses them
  unsigned long long lanes_out = 0ull;
  auto lanes_setup = [&]() {
    lanes_out = __ballot(!inwin);
    MFN_UNROLL
    for (int m = 0; m < 4; ++m)
      rbo[m] = inwin ? 0xFFFFF000u
                     : (unsigned)((size_t)n * xns + ((size_t)kb * H + min(max(row0 + m, 0), H - 1)) * W + col0) * 4u;   // (wraps below the tensor: out of range)
  };
  auto lanes_request = [&](int kn, auto set_c) {
    constexpr int SET = decltype(set_c)::value;
    const unsigned soff = kn < gps * 8 ? (unsigned)((size_t)(2 * (cp_base + kn)) * plane * 4) : 0x7FFFFF00u;   // past the slice: out of range
    MFN_UNROLL
    for (int m = 0; m < 4; ++m)
      mfn_bload1x4_async(vq[SET][4 * m], vq[SET][4 * m + 1], vq[SET][4 * m + 2], vq[SET][4 * m + 3], xrsrc, rbo[m], soff, lanes_out);
  };
  auto lanes_landed = [&](auto set_c, auto nafter_c) {
    constexpr int SET = decltype(set_c)::value;
    constexpr int NAFTER = decltype(nafter_c)::value < 63 ? decltype(nafter_c)::value : 63;
    MFN_LANDED4(vq[SET][0], vq[SET][1], vq[SET][2], vq[SET][3], NAFTER);
    MFN_REGFENCE4(vq[SET][4], vq[SET][5], vq[SET][6], vq[SET][7]);
    MFN_REGFENCE4(vq[SET][8], vq[SET][9], vq[SET][10], vq[SET][11]);
    MFN_REGFENCE4(vq[SET][12], vq[SET][13], vq[SET][14], vq[SET][15]);
    if (!inwin) {
      MFN_UNROLL
      for (int m = 0; m < 4; ++m)
        MFN_UNROLL
        for (int h = 0; h < 2; ++h) {
          vp[m][h].x = vq[SET][4 * m + 2 * h];
          vp[m][h].y = vq[SET][4 * m + 2 * h + 1];
        }
    }
  };
  // separable interpolation, y pass then x pass, in SCALAR fp32 instructions: taps 0..7 into x8 (the K elements of the lane's
  // k-block), tap 8 beside them.  (Packed v_pk_mul_f32 / v_pk_fma_f32 on the register pairs the LDS reads deliver would be 12
  // instructions fewer -- and measured 586 cycles per step for these 30 instructions against 171 for the 38 of the split: packed
  // fp32 issued behind matrix instructions that are still in flight stalls, MI355X_MICROARCH.md "anti-lever beside MFMAs".)
  auto cols_finish = [&](float (&x8)[8], float &c8) {
    if (CONV) {
      x8[0] = vp[0][0].x; x8[1] = vp[0][0].y; x8[2] = vp[0][1].x;
      x8[3] = vp[1][0].x; x8[4] = vp[1][0].y; x8[5] = vp[1][1].x;
      x8[6] = vp[2][0].x; x8[7] = vp[2][0].y; c8 = vp[2][1].x;
      return;
    }
    f32x2 ty[3][2];
    if (MFN_DCM_ABLATE & 16) {
      MFN_UNROLL
      for (int e = 0; e < 8; ++e) x8[e] = e < 4 ? vp[e][0].x : vp[e - 4][1].y;
      c8 = vp[0][0].y;
      return;
    }
    MFN_UNROLL
    for (int i = 0; i < 3; ++i)
      MFN_UNROLL
      for (int h = 0; h < 2; ++h) {
#ifdef MFN_DCM_INTERP_PACKED   // measurement builds
        ty[i][h] = mfn_fma2(mfn_f2(yb[i], yb[i]), vp[i + 1][h], mfn_mul2(mfn_f2(ya[i], ya[i]), vp[i][h]));
#else
        ty[i][h].x = fmaf(yb[i], vp[i + 1][h].x, ya[i] * vp[i][h].x);
        ty[i][h].y = fmaf(yb[i], vp[i + 1][h].y, ya[i] * vp[i][h].y);
#endif
      }
    MFN_UNROLL
    for (int i = 0; i < 3; ++i) {
      const float t0 = ty[i][0].x, t1 = ty[i][0].y, t2 = ty[i][1].x, t3 = ty[i][1].y;
      const float c0 = fmaf(xb[0], t1, xa[0] * t0), c1 = fmaf(xb[1], t2, xa[1] * t1), c2 = fmaf(xb[2], t3, xa[2] * t2);
      if (i < 2) { x8[3 * i] = c0; x8[3 * i + 1] = c1; x8[3 * i + 2] = c2; }
      else { x8[6] = c0; x8[7] = c1; c8 = c2; }
    }
  };
  DcmB B;
  GramSel sel;
  if (MFN_DCM_MSPLIT) sel = gram_make_sel(lane);
  // the six products of one K step against the MT filter tiles.  a_read: this step's [term][ft] blocks from the stage buffer --
  // requested BEFORE the next pair's gather, so that the first matrix instruction waits for its own operand only (LDS returns in
  // order: behind the gather it would wait for all sixteen neighbourhood values as well)
  mfn_bf16x8 ah[MT], am[MT], al[MT];
  auto a_read = [&](const float *a) {
    MFN_UNROLL
    for (int ft = 0; ft < MT; ++ft) {
      if (MFN_DCM_ABLATE & 64) { ah[ft] = am[ft] = al[ft] = B.h; continue; }
      al[ft] = mfn_read_bf16x8(a + (2 * MT + ft) * 256);
      ah[ft] = mfn_read_bf16x8(a + (0 * MT + ft) * 256);
      am[ft] = mfn_read_bf16x8(a + (1 * MT + ft) * 256);
    }
  };
  auto mma_issue = [&](const DcmB &b) {
    if (MFN_DCM_ABLATE & 1) {
      MFN_UNROLL
      for (int ft = 0; ft < MT; ++ft) acc[ft][0][0] += mfn_bf16_at(reinterpret_cast<const float *>(&ah[ft]), 0) + mfn_bf16_at(reinterpret_cast<const float *>(&al[ft]), 1) + mfn_bf16_at(reinterpret_cast<const float *>(&am[ft]), 2) + mfn_bf16_at(reinterpret_cast<const float *>(&b.l), 0);
      return;
    }
#define MFN_DCM_(A_, B_, P_)                                                                      \
  MFN_UNROLL                                                                                      \
  for (int ft = 0; ft < MT; ++ft) acc[ft][(P_) % NACC] = MFN_MFMA_32x32x16_BF16(A_[ft], B_, acc[ft][(P_) % NACC]);
    MFN_DCM_(al, b.h, 0)   // smallest products first: 2^-16 class, then 2^-8, then the leading one
    MFN_DCM_(ah, b.l, 1)
    MFN_DCM_(am, b.m, 0)
    MFN_DCM_(am, b.h, 1)
    MFN_DCM_(ah, b.m, 0)
    MFN_DCM_(ah, b.h, 1)
#undef MFN_DCM_
  };

  // Per-tap offsets (MXNet's general operator; the reference never calls it so): the pair of THIS step tap by tap -- geometry
  // rebuilt per tap and per step (lean in registers, not fast), taps 0..7 against the fp32 sum of the weight's three terms on
  // v_mfma_f32_32x32x2_f32 (exact), tap 8 into the left-over step's operand like everybody else's.
  auto pertap_pair = [&](int kp, const float *a, float *t8dst) {
    const int c = 2 * (cp_base + kp) + kb;
    const float *pl = xn + (size_t)c * plane;
    MFN_NOUNROLL
    for (int t = 0; t < 9; ++t) {
      float oh = off_h, ow = off_w;
      if (p.offset) {
        const float *op = p.offset + (size_t)n * 18 * plane + (size_t)ho * W + wo;
        oh = op[(size_t)(2 * t) * plane];
        ow = op[(size_t)(2 * t + 1) * plane];
      }
      const int ti = (t * 11) >> 5, tj = t - 3 * ti;
      const DcTap tp = dc_make_tap(oh, ow, h_in, w_in, ti, tj, H, W, px_valid);
      const int bb = tp.base & 0x3FFFFFFF, dwi = (tp.base >> 30) & 1;
      const float v1 = pl[bb], v2 = pl[bb + dwi], v3 = pl[bb + tp.dhW], v4 = pl[bb + tp.dhW + dwi];
      const float cvt = tp.w1 * v1 + tp.w2 * v2 + tp.w3 * v3 + tp.w4 * v4;
      if (t < 8) {
        MFN_UNROLL
        for (int ft = 0; ft < MT; ++ft) {
          const float wv = mfn_bf16_at(a + (0 * MT + ft) * 256, t) + mfn_bf16_at(a + (1 * MT + ft) * 256, t) + mfn_bf16_at(a + (2 * MT + ft) * 256, t);
          acc[ft][0] = MFN_MFMA_32x32x2(wv, cvt, acc[ft][0]);
        }
      } else {
        *t8dst = cvt;
      }
    }
  };

  // ---- B(0) ---------------------------------------------------------------------------------------------------------------------
  float *xt8 = lds + G::T8_OFF + wave * 512 + lane;   // tap 8 of the group's eight pairs, [pair][lane]: the left-over step's K elements
  auto first_operand = [&](auto win_c, auto out_c) {
    using WN = decltype(win_c);
    MFN_WAIT_VM((dcm_prologue_after_x0<KC, G::NSTAGE, WN::DEPTH, NIW, WN::NI>()));   // window 0 landed
    float x8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, c8 = 0.f;
    if (fast) {
      if (decltype(out_c)::value) lanes_request(0, std::integral_constant<int, 1>{});
      cols_gather(0, win_c);
      if (decltype(out_c)::value) {
        lanes_landed(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
        // the pair the loop's first step prepares.  (The wait above was for EVERYTHING the prologue issued: the loop's counts
        // assume the steady sequence, requests included, which the prologue does not replay.)
        lanes_request(1, std::integral_constant<int, 0>{});
      }
      cols_finish(x8, c8);
    }
    xt8[0] = c8;
    if (MFN_DCM_MSPLIT) gram_msplit8(x8, sel, B.h, B.m, B.l);
    else mfn_split3x8(x8, B.h, B.m, B.l);
    MFN_STAMP2(p.timeline, 4);   // first operand formed (window 0 landed)
  };

  // ---- the K loop: 9 steps per 16-channel group.  Step t: the six products of B(t) against the MT filter tiles; under them
  // B(t + 1) is gathered and interpolated, then split into the same registers (the matrix instructions read their operands at
  // issue).  The tiers are INSTANCES of the loop, not branches inside one: a load hipcc knows about makes it wait -- at the join, for
  // every wave -- with a vmcnt that also drains the window / weight transfers it does not know about; and with one site of matrix
  // instructions per branch hipcc keeps the accumulators in two register sets and copies all of them at every join.
  // Window tiers: the nine steps of a group UNROLLED (a rolled loop over the groups): the step number is a compile-time constant,
  // so the waits, the chunk boundaries and the left-over step cost no scalar compare chain, and a step is straight-line code in
  // which hipcc places the interpolation between the matrix instructions (rolled with uniform branches: 16.6 us of loop at level 2
  // against 12.9).
  auto run_groups = [&](auto win_c, auto out_c) {
    using WN = decltype(win_c);
    constexpr int CPG = 9 / KC;   // chunks per group
    if (decltype(out_c)::value) lanes_setup();
    first_operand(win_c, out_c);
    MFN_NOUNROLL
    for (int g = 0; g < gps; ++g) {
      mfn_static_for<9>([&](auto s_c) {
        constexpr int S = decltype(s_c)::value;
        constexpr bool OUT = decltype(out_c)::value;
        constexpr int NWAIT = dcm_wait_count<KC, G::NSTAGE, WN::DEPTH, NIW, WN::NI, OUT ? 16 : 0>(S);
        if (NWAIT < 1000) MFN_WAIT_VM(NWAIT < 63 ? NWAIT : 63);
        const int ch = g * CPG + S / KC;
        constexpr int kk = S % KC;
        const int kn = S == 8 ? (g + 1) * 8 : g * 8 + S + 1;   // the pair in preparation under this step's products
        constexpr int E = S == 8 ? 7 : S;   // the group's E-th preparation (steps 0..6 and 8)
        if (kk == 0) {   // this chunk's weights landed for every wave; everybody is done with the stage that is refilled now
          MFN_WAIT_LGKM0();
          if (!PRIVATE_W && !(MFN_DCM_ABLATE & 128)) MFN_RAW_BARRIER();
          if (S == 0 && g == 0) MFN_STAMP(p.timeline, 1);
          issue_w(ch + G::NSTAGE - 1);
        }
        if (S < 8) issue_x(g * 8 + S + WN::DEPTH, win_c);
        // the lanes outside the window: preparation E asks for the pair of preparation E + 1, into the other register set
        if (OUT && S != 7) lanes_request(S == 8 ? g * 8 + 9 : g * 8 + S + 2, std::integral_constant<int, (E + 1) & 1>{});
        const float *a = lds + (ch % G::NSTAGE) * G::STAGE_W + (size_t)(kw * KC + kk) * G::STEP_W + (kb * 32 + j) * 4;
        float x8[8];
        a_read(a);
        if (S != 7) cols_gather(kn, win_c);
        mma_issue(B);
        if (OUT && S != 7)
          lanes_landed(std::integral_constant<int, E & 1>{},
                       std::integral_constant<int, dcm_landed_count<KC, G::NSTAGE, WN::DEPTH, NIW, WN::NI, 16>(S)>{});
        if (S == 7) {   // the left-over step's operand: tap 8 of the eight pairs
          MFN_UNROLL
          for (int e = 0; e < 8; ++e) x8[e] = xt8[e * 64];
        } else {
          float c8;
          cols_finish(x8, c8);
          xt8[(S == 8 ? 0 : S + 1) * 64] = c8;   // the slot of the pair just formed inside ITS group
        }
#ifdef MFN_DCM_SPLIT_PACKED
        if (!(MFN_DCM_ABLATE & 8)) mfn_split3x8(x8, B.h, B.m, B.l);
#else
        if (!(MFN_DCM_ABLATE & 8)) {
          if (MFN_DCM_MSPLIT) gram_msplit8(x8, sel, B.h, B.m, B.l);
          else mfn_split3x8_scalar(x8, B.h, B.m, B.l);
        }
#endif
        MFN_SCHED_BARRIER();
      });
    }
  };
  // Per-tap tier: one rolled loop, the step inside its group (s) a wave-uniform counter.
  auto run_pertap = [&]() {
    using WN = DcmWin<false, RING>;
    first_operand(WN{}, std::false_type{});
    const int nsteps = gps * 9;
    int s = 0, kl = 0;   // step inside the group; pair of the step
    MFN_NOUNROLL
    for (int t = 0; t < nsteps; ++t) {
      mfn_static_for<9>([&](auto s_c) {   // one test per distinct count: the steps that share it as a bit mask
        constexpr int S = decltype(s_c)::value;
        constexpr int NWAIT = dcm_wait_count<KC, G::NSTAGE, WN::DEPTH, NIW, WN::NI>(S);
        constexpr unsigned MASK = dcm_wait_mask<KC, G::NSTAGE, WN::DEPTH, NIW, WN::NI>(S);   // 0 unless S is the first step with this count
        if (NWAIT < 1000 && MASK != 0 && ((MASK >> s) & 1u)) MFN_WAIT_VM(NWAIT < 63 ? NWAIT : 63);
      });
      const int ch = t / KC, kk = t - ch * KC;
      if (kk == 0) {
        MFN_WAIT_LGKM0();
        if (!PRIVATE_W) MFN_RAW_BARRIER();
        if (t == 0) MFN_STAMP(p.timeline, 1);
        issue_w(ch + G::NSTAGE - 1);
      }
      if (s < 8) issue_x(kl + WN::DEPTH, WN{});   // (zero-filled: keeps the transfer sequence, hence the counts, those of the window tiers)
      const float *a = lds + (ch % G::NSTAGE) * G::STAGE_W + (size_t)(kw * KC + kk) * G::STEP_W + (kb * 32 + j) * 4;
      if (s < 8) pertap_pair(kl, a, xt8 + (kl & 7) * 64);
      else { a_read(a); mma_issue(B); }
      if (s == 7) {
        float x8[8];
        MFN_UNROLL
        for (int e = 0; e < 8; ++e) x8[e] = xt8[e * 64];
        mfn_split3x8(x8, B.h, B.m, B.l);
      }
      if (s < 8) ++kl;
      s = s == 8 ? 0 : s + 1;
    }
  };
  if constexpr (CONV) run_groups(DcmWinConv<RING>{}, std::false_type{});
  else {
    if (!fast) run_pertap();
    else if (!bigwin) run_groups(DcmWin<false, RING>{}, std::false_type{});
    else if (all_in) run_groups(DcmWin<true, RING>{}, std::false_type{});
    else run_groups(DcmWin<true, RING>{}, std::true_type{});
  }
  MFN_STAMP(p.timeline, 2);
  MFN_STAMP_INFO(p.timeline, !fast ? 3 : (!bigwin ? 0 : (all_in ? 1 : 2)));   // the first wave's tier

  MFN_STAMP2(p.timeline, 5);
  // ---- K-slice reduction: tile mt is summed (slice order 0..KW-1, deterministic) and stored by slice mt % KW ------------------------
  MFN_WAIT_VM(0);       // the prefetches past the end have landed: nothing is on its way into LDS any more
  f32x16 fin[MT];
  MFN_UNROLL
  for (int mt = 0; mt < MT; ++mt)
    MFN_UNROLL
    for (int r = 0; r < 16; ++r) fin[mt][r] = NACC == 2 ? acc[mt][0][r] + acc[mt][NACC - 1][r] : acc[mt][0][r];
  if (KW > 1) {
    MFN_WAIT_LGKM0();
    MFN_RAW_BARRIER();  // every wave is out of the loop: the stage buffers are free
    if (G::RED_ALL) {   // every slice's tiles at once: [pt][kw][mt][16][64]
      float *mine = lds + (size_t)((pt * KW + kw) * MT) * 1024 + lane;
      MFN_UNROLL
      for (int mt = 0; mt < MT; ++mt)
        if (kw != mt % KW) {
          MFN_UNROLL
          for (int r = 0; r < 16; ++r) mine[mt * 1024 + r * 64] = fin[mt][r];
        }
      __syncthreads();
      MFN_UNROLL
      for (int mt = 0; mt < MT; ++mt) {
        const int owner = mt % KW;
        if (kw != owner) continue;
        f32x16 sum;
        MFN_UNROLL
        for (int k = 0; k < KW; ++k) {
          f32x16 part;
          if (k == owner) part = fin[mt];
          else {
            const float *src = lds + (size_t)((pt * KW + k) * MT + mt) * 1024 + lane;
            MFN_UNROLL
            for (int r = 0; r < 16; ++r) part[r] = src[r * 64];
          }
          MFN_UNROLL
          for (int r = 0; r < 16; ++r) sum[r] = k == 0 ? part[r] : sum[r] + part[r];
        }
        fin[mt] = sum;
      }
    } else {
      float *red = lds + (size_t)pt * (KW - 1) * 1024;
      MFN_UNROLL
      for (int mt = 0; mt < MT; ++mt) {
        const int owner = mt % KW;
        if (kw != owner) {
          float *dst = red + (size_t)(kw < owner ? kw : kw - 1) * 1024 + lane;
          MFN_UNROLL
          for (int r = 0; r < 16; ++r) dst[r * 64] = fin[mt][r];
        }
        __syncthreads();
        if (kw == owner) {
          f32x16 sum;
          MFN_UNROLL
          for (int k = 0; k < KW; ++k) {
            f32x16 part;
            if (k == owner) part = fin[mt];
            else {
              const float *src = red + (size_t)(k < owner ? k : k - 1) * 1024 + lane;
              MFN_UNROLL
              for (int r = 0; r < 16; ++r) part[r] = src[r * 64];
            }
            MFN_UNROLL
            for (int r = 0; r < 16; ++r) sum[r] = k == 0 ? part[r] : sum[r] + part[r];
          }
          fin[mt] = sum;
        }
        if (mt + 1 < MT) __syncthreads();
      }
    }
  }

  MFN_STAMP2(p.timeline, 6);   // K slices reduced
  // ---- epilogue.  D reg r of lane (j, kb): filter row (r&3) + 8*(r>>2) + 4*kb, pixel j.  The 32 x 32 tile is transposed through
  // the wave's idle window ring so that a lane holds four adjacent pixels of one filter: 4 x 16-byte stores instead of 16 dword stores
  constexpr int TS = 40;
  float *tr = xwin;
  const size_t oplane = plane;
  const int quad = lane & 7, orow = lane >> 3;
  const int px0 = quad * 4;
  const int oy = tile_ho0 + (px0 >> 3), ox = tile_wo0 + (px0 & 7);
  const bool st_ok = tile < p.ntiles && oy < H;
  const bool ep = p.ep_mask || p.ep_add || p.ep_leaky;
  const size_t pix0 = st_ok ? (size_t)n * oplane + (size_t)oy * W : 0;
  int oxq[4];
  MFN_UNROLL
  for (int q = 0; q < 4; ++q) oxq[q] = st_ok ? min(ox + q, W - 1) : 0;
  float sg[4] = {1.f, 1.f, 1.f, 1.f};
  if (ep && p.ep_mask) {
    float mv[4];
    MFN_UNROLL
    for (int q = 0; q < 4; ++q) mv[q] = p.ep_mask[pix0 + oxq[q]];
    MFN_UNROLL
    for (int q = 0; q < 4; ++q) sg[q] = 1.f / (1.f + expf(-mv[q]));
  }
  const size_t obatch = st_ok ? (size_t)n * p.Cout * oplane + (size_t)oy * W : 0;
  float *obase = p.out + (size_t)n * (p.out_nstride ? p.out_nstride : (size_t)p.Cout * oplane);
  MFN_UNROLL
  for (int mt = 0; mt < MT; ++mt) {
    if (KW > 1 && kw != mt % KW) continue;
    float bq[4], addv[4][4];
    MFN_UNROLL
    for (int i = 0; i < 4; ++i) bq[i] = p.bias ? p.bias[min(m0 + mt * 32 + i * 8 + orow, p.Cout - 1)] : 0.f;
    if (ep && p.ep_add) {
      MFN_UNROLL
      for (int i = 0; i < 4; ++i) {
        const size_t orow_off = obatch + (size_t)min(m0 + mt * 32 + i * 8 + orow, p.Cout - 1) * oplane;
        MFN_UNROLL
        for (int q = 0; q < 4; ++q) addv[i][q] = p.ep_add[orow_off + oxq[q]];
      }
    }
    MFN_WAIT_LGKM0();   // the previous tile's reads are done (wave-private buffer: no barrier needed)
    MFN_UNROLL
    for (int r = 0; r < 16; ++r) tr[((r & 3) + 8 * (r >> 2) + 4 * kb) * TS + j] = fin[mt][r];
    MFN_WAIT_LGKM0();
    MFN_UNROLL
    for (int i = 0; i < 4; ++i) {
      const int ol = i * 8 + orow;
      const int o = m0 + mt * 32 + ol;
      const float4 v = *reinterpret_cast<const float4 *>(tr + ol * TS + px0);
      float e[4] = {v.x + bq[i], v.y + bq[i], v.z + bq[i], v.w + bq[i]};
      if (ep) {
        MFN_UNROLL
        for (int q = 0; q < 4; ++q) {
          if (p.ep_mask) e[q] = e[q] * sg[q];
          if (p.ep_add) e[q] = e[q] + addv[i][q];
          if (p.ep_leaky) e[q] = fmaxf(e[q], 0.1f * e[q]);
        }
      }
      if (st_ok && o < p.Cout) {
        float *dst = obase + (size_t)o * oplane + (size_t)oy * W + ox;
        if (ox + 3 < W) {
          mfn_store4_stream(dst, e[0], e[1], e[2], e[3], p.st_policy);
        } else {
          MFN_UNROLL
          for (int q = 0; q < 4; ++q)
            if (ox + q < W) dst[q] = e[q];
        }
      }
    }
  }
  MFN_STAMP(p.timeline, 3);
}

template <int MT, int PT, int KW, int RING, bool CONV = false>
inline int dc_mma_launch(const DeformParams &p, hipStream_t stream, const char *name = "dc_mma") {
  using G = DcmGeom<MT, PT, KW, RING>;
  const int bx = cdiv(p.ntiles, PT);
  if (bx <= 0) return 0;
  return launch(name, dc_mma_kernel<MT, PT, KW, RING, CONV>, dim3(bx, 1, p.mgroups), dim3(G::NTH), (size_t)G::LDS_W * sizeof(float), stream, p);
}
// the plain-convolution form: 3x3 / stride 1 / pad 1 / dilation 1, one group, 16-byte rows; any channel count
inline bool dcm_conv_shape_ok(int N, size_t x_nstride, int H, int W, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int groups) {
  return kh == 3 && kw == 3 && sh == 1 && sw == 1 && ph == 1 && pw == 1 && dh == 1 && dw == 1 && groups == 1 && W % 4 == 0 && W >= 8 &&
         H >= 1 && (size_t)N * x_nstride < ((size_t)1 << 30);
}

// what the kernel needs of a call: the network's operator shape, 16-byte rows, whole groups of 16 channels
inline bool dcm_shape_ok(int N, int Cin, int H, int W, int Ho, int Wo, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                         int groups, int dg) {
  return kh == 3 && kw == 3 && sh == 1 && sw == 1 && ph == 1 && pw == 1 && dh == 1 && dw == 1 && groups == 1 && dg == 1 && Ho == H &&
         Wo == W && W % 4 == 0 && W >= 8 && H >= 1 && Cin % 16 == 0 && (size_t)N * Cin * H * W < ((size_t)1 << 30);
}

}  // namespace mfn
