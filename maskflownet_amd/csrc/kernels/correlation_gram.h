// correlation_gram.h -- the cost volume of 32-channel levels as a BANDED GRAM MATRIX on the bf16 matrix cores (gfx950).
//
// Replaces MXNet Correlation at /root/reference/network/MaskFlownet.py:193-195 (md=4, 81 ch) and :440-441 (md=2, 25 ch)
// for C == 32 (level 2, the launch BASELINE.json's north_star names) and, as a two-chunk K loop, C == 64 (level 3; round 6);
// semantics as oracle/mfn_ref_body.inc correlation_fwd.
//
// Why another formulation (profiles/r03_corr_pmc.md): corr_dma_kernel (correlation.h) spends 36 packed
// FMAs and 64 bytes of LDS operand reads per lane-channel; at level 2 that is 5.6 us of VALU issue + 3.2 us of LDS returns per
// CU which do not overlap, next to 7.1 us of HBM streaming.  Here the contraction over the 32 channels is ONE K = 32 step of
// v_mfma_f32_16x16x32_bf16:
//   * out[dy][dx][y][x] = sum_c f1[c][y][x] * f2[c][y+dy][x+dx] is the band |x'-x| <= md, |y'-y| <= md of the Gram matrix
//     G[(y,x)][(y',x')] = sum_c f1[c,y,x] f2[c,y',x'].  A 16 x 16 tile of G with M = an 8 x 2 pixel block of f1 and
//     N = a 16 x 1 row segment of f2 starting 4 columns left of the block is 50.6 % useful for md = 4 (10 tiles per 16 pixels x 81
//     displacements) -- the densest (M, N) shape pair for the 16 x 16 tile.
//   * fp32-EQUIVALENT arithmetic (not the bit pattern of an fp32 FMA chain): every operand is split hi + mid + lo into three
//     bf16 terms (24 significant bits, the split itself is exact) and SIX of the nine partial products -- those of weight
//     >= 2^-16: hh, hm, mh, mm, hl, lh -- are accumulated in fp32 by the matrix core; the three dropped ones (ml, lm, ll) are
//     <= 2^-24 of the product each, i.e. together up to ~1 ulp PER PRODUCT (not per sum), which is what an fp32 FMA chain loses per
//     accumulation step anyway.  Held to it by tests/test_gpu_parity.py::test_correlation_gram_error_vs_fp64 (error against the
//     fp64 oracle <= 1.5 x the FMA kernel's on the same input, plus a per-element relative check).  Non-finite inputs differ:
//     an inf feature splits into inf + NaN (inf - bf16(inf)), so the affected outputs are NaN where the FMA chain gives inf
//     (include/mfn_hip.h, "Arithmetic").  mfn_rt.h mfn_split3x8.
//   * one WAVE = one work item: an 8-pixel wide column strip x `rows` output rows of one image.  The wave walks down the f2 rows
//     ys-md .. ys+rows-1+md ("steps"); each f2 row segment (16 px x 32 ch) is converted ONCE into a B operand (12 VGPRs) and
//     multiplied with the up to md+1 resident A operands (8 x 2 pixel blocks of f1, converted once, 12 VGPRs each) whose rows
//     lie within +-md: per step <= 5 chains of 6 MFMAs, each chain a complete (all 32 channels) 16 x 16 result that is
//     stored at once -- no accumulators live across steps, no LDS operand re-reads (24 ds_read_b32 per 60 MFMAs).
//   * the matrix core's D layout (lane = x', register i = pixel x) holds the band skewed: dx = x' - x depends on i.  Three
//     v_mov_b32_dpp row_shl:i bring the four pixels of one displacement into one lane; a lane then owns (dx, 4 adjacent x) and
//     stores 16 bytes straight into plane dy*D+dx: 9 of 16 lanes per row, 36 x 16 bytes per buffer_store_dwordx4, the band
//     mask is the descriptor's range check (out-of-band lanes carry an out-of-range offset).
//   * raw fp32 operand tiles arrive through WAVE-PRIVATE LDS-DMA rings (buffer_load_dwordx4 ... lds, 2 x 1 KB per tile, laid
//     out [channel][16 px] so that the lane's 8 channels {4j + lane/16} are 8 conflict-free ds_read_b32 at stride 256 B):
//     no block barrier anywhere, every wave is its own pipeline.  Loads run A steps ahead; the waits are COUNTED with the
//     stores in the count (vector memory operations complete in issue order on gfx9): mfn_wait_vm_dyn.
//   * zero padding = the descriptor's range check (rows outside the image: num_records 0; columns: per-lane offsets).
// Cost per 16 pixels x 81 displacements: 60 MFMAs (~1000 matrix-pipe cycles per SIMD), ~(2.3 x 2 + 1) x 38 conversion VALU + 30 DPP
// moves, 24 + 8 LDS reads, 10 stores -- against 1440 VALU lane-ops and 288 16-byte LDS reads per pixel before.
#pragma once
#include "../mfn_rt.h"
#include "msplit.h"

// measurement builds only (tools/gram_ablate_build.py): bit 1 no matrix instructions / de-skew / stores (DMA, LDS reads and
// conversions remain), 2 no stores, 4 no conversions (operands are the raw bits), 8 no DMA (the ring is never filled)
#ifndef MFN_GRAM_ABLATE
#define MFN_GRAM_ABLATE 0
#endif
// 1: the conversions of the next step's operands are pinned between this step's matrix instructions; 0: hipcc's own order
#ifndef MFN_GRAM_SCHED
#define MFN_GRAM_SCHED 1
#endif
// (measured and removed: the residuals of the operand split by v_dot2c_f32_bf16 d, {-1, 0}, packed -- 7 instead of 9 VALU per pair,
// exact in isolation (tools/ubench/dot2c_check.hip), no faster in the kernel (11.5 against 11.4 us) and it needs hand-placed wait states)

namespace mfn {

struct CorrGramParams {
  const float *f1;
  const float *f2;
  float *out;
  int N, H, W;
  int C;                  // 32, or 64 (two channel chunks; form 48 only)
  int rows;               // output rows per work item (even)
  int strips, segs;       // ceil(W / 8), ceil(H / rows)
  int bx_per_row;         // blocks along x: ceil(strips / waves per block)
  unsigned bx_magic, segs_magic;   // mfn_make_magic of bx_per_row / segs: block id -> (strip group, segment, image) by two
                                   // s_mul_hi_u32 instead of two runtime divisions (~40 scalar instructions in front of the first DMA)
  size_t out_nstride;     // elements between images of `out` (a channel slice of a concat buffer when > D*D*H*W)
  int store_policy;       // mfn_bstore4
  int leaky;              // fused LeakyReLU(0.1)
  int xcd_swizzle;        // 0, or the number of blocks of the launch (mfn_xcd_remap's range: no second, dependent kernarg load for gridDim)
  float inv_c;            // 1/32, folded into the f1 operand before the split (a power of two: exact)
  unsigned long long *timeline;  // measurement builds only (-DMFN_TIMELINE=1): 4 stamps per block, or NULL
};

struct GramOp { mfn_bf16x8 h, m, l; };
// TERMS == 1: the raw fp32 values themselves -- value j is channel 4j + lane/16 of the lane's pixel, which is exactly the A / B
// operand of K step j of v_mfma_f32_16x16x4_f32 (lane = pixel + 16 k): no conversion at all, eight fp32 matrix instructions per
// chain (an fmaf chain over the channels in order, bitwise) instead of six bf16 ones.
struct GramOpF { float r[8]; };
template <int TERMS> struct GramOpSel { typedef GramOp type; };
template <> struct GramOpSel<1> { typedef GramOpF type; };

// One PAIR of fp32 values -> word q of the three bf16 terms (mfn_split3x8 / mfn_split2x8 of mfn_rt.h, a quarter at a time, so that
// the kernel can place the nine VALU instructions of a pair behind one matrix instruction).
struct GramWords { unsigned h[4], m[4], l[4]; };
template <int TERMS>
__device__ __forceinline__ void gram_split_pair(float x0, float x1, GramWords &w, int q) {
#if defined(MFN_EMU)
  unsigned short h0, m0, l0, h1, m1, l1;
  if (TERMS == 3) { mfn_split3(x0, h0, m0, l0); mfn_split3(x1, h1, m1, l1); }
  else {
    h0 = hipemu_f32_to_bf16(x0); l0 = hipemu_f32_to_bf16(x0 - hipemu_bf16_to_f32(h0)); m0 = l0;
    h1 = hipemu_f32_to_bf16(x1); l1 = hipemu_f32_to_bf16(x1 - hipemu_bf16_to_f32(h1)); m1 = l1;
  }
  w.h[q] = (unsigned)h0 | ((unsigned)h1 << 16); w.m[q] = (unsigned)m0 | ((unsigned)m1 << 16); w.l[q] = (unsigned)l0 | ((unsigned)l1 << 16);
#else
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  const f32x2 v = {x0, x1};
  const unsigned hp = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2));
  const f32x2 hf = {__builtin_bit_cast(float, hp << 16), __builtin_bit_cast(float, hp & 0xffff0000u)};
  const f32x2 r1 = v - hf;
  const unsigned mp = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf2));
  w.h[q] = hp;
  if (TERMS == 3) {
    const f32x2 mf = {__builtin_bit_cast(float, mp << 16), __builtin_bit_cast(float, mp & 0xffff0000u)};
    const f32x2 r2 = r1 - mf;
    w.m[q] = mp; w.l[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf2));
  } else { w.m[q] = mp; w.l[q] = mp; }
#endif
}
__device__ __forceinline__ void gram_words_to_op(const GramWords &w, GramOp &o) {
#if defined(MFN_EMU)
  memcpy(&o.h, w.h, 16); memcpy(&o.m, w.m, 16); memcpy(&o.l, w.l, 16);
#else
  typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
  const u32x4_ h = {w.h[0], w.h[1], w.h[2], w.h[3]}, m = {w.m[0], w.m[1], w.m[2], w.m[3]}, l = {w.l[0], w.l[1], w.l[2], w.l[3]};
  o.h = __builtin_bit_cast(mfn_bf16x8, h); o.m = __builtin_bit_cast(mfn_bf16x8, m); o.l = __builtin_bit_cast(mfn_bf16x8, l);
#endif
}

// The static schedule of a wave.  D = 2*md+1; T = f1 blocks (8 x 2 px) per work item: the item's rows are 2T.  Step
// s = 0 .. 2T+2MD-1 brings f2 row s of the item's window (image row ys-MD+s); block t (rows ys+2t, +1) meets it when
// e = s - 2t is in [0, 2MD+1]: block row 0 then has displacement row e, block row 1 has e-1.  SP waves share an item: wave
// PAR < SP takes the steps s = SP*j + PAR (its own steps j = 0 .. J-1) -- every wave converts all T blocks, but each f2 row
// is converted by one wave only, so SP = 2 doubles the waves in flight at 18 % more conversions (6-row items).
// Tile sequence of a wave in consumption order: per own step j [block t if this is the first own step with s >= 2t], f2 row s(j).
template <int D, int T, int SP, int PAR>
struct GramSched {
  static constexpr int MD = (D - 1) / 2;
  static constexpr int S = 2 * T + 2 * MD;
  static constexpr int J = (S - PAR + SP - 1) / SP;           // own steps
  static constexpr int TT = J + T;                            // tiles
  static constexpr int step(int j) { return SP * j + PAR; }
  static constexpr int blocks_by(int j) { return (step(j) / 2 + 1) < T ? (step(j) / 2 + 1) : T; }   // blocks due at or before own step j
  static constexpr int block_at(int j) { return (j == 0 ? blocks_by(0) : blocks_by(j) - blocks_by(j - 1)) > 0 ? blocks_by(j) - 1 : -1; }
  static constexpr int seqN(int j) { return j + blocks_by(j); }                                     // position of own step j's f2 row
  static constexpr int kind(int k) {   // tile k: block t (>= 0) or -(j+1) for the f2 row of own step j
    int pos = 0;
    for (int j = 0; j < J; ++j) {
      if (block_at(j) >= 0) { if (pos == k) return block_at(j); ++pos; }
      if (pos == k) return -(j + 1);
      ++pos;
    }
    return -1;
  }
  // at most one block becomes due per own step (2 steps per block, SP <= 2)
  static_assert(SP == 1 || SP == 2, "one or two waves per item");
};

// The body of a wave: straight-line code (mfn_static_for), every counted wait an immediate.  NSLOT = 2 KB ring slots (= tiles
// in flight).  Stores start in the first step: loads, matrix work and stores overlap over the whole life of an item (the first form
// of this kernel kept the f2 rows in a register window and walked the blocks; its first MD iterations only filled the window and
// the launch was load phase, then store phase -- profiles/r04_corr_gram_experiments.md).
// COOP: the NWV = 4 waves of a block (adjacent strips, 32 px = one 128-byte line per output row) hand their de-skewed results
// to each other through LDS and store FULL lines, written through: a wave alone owns 32-byte runs -- one L2 write request per
// run (995 k per level-2 launch against 498 k for the FMA kernel), 36 of 64 lanes per store instruction, and written through
// they are partial-line writes (14.2 us for the volume alone), so the non-cooperative form stores plain and leaves the lines
// to the L2 and to the flush at the end of the kernel.  One block barrier per step; `stg` = the block's two staging buffers.
template <int D, int T, int NSLOT, int TERMS, int POL, bool LEAKY, int SP, int PAR, bool COOP = false, int NC = 1>
__device__ __forceinline__ void corr_gram_wave(const CorrGramParams &p, float *ring, int lane, int n, int ys, int x0,
                                               float *stg = nullptr, int wave = 0, int xb0 = 0) {
  using SC = GramSched<D, T, SP, PAR>;
  constexpr int MD = SC::MD;
  constexpr int J = SC::J;
  constexpr int TT = SC::TT;
  constexpr int SLOT_F = 512;        // floats per raw tile: 32 channels x 16 px
  constexpr int XOFF = 4;            // the f2 segment starts XOFF columns left of the strip (16-byte aligned, >= MD)
  constexpr unsigned INVALID = 0xFFFFFF00u;
  static_assert(MD >= 1 && MD <= 4 && T >= 1 && NSLOT >= 3 && NSLOT <= TT, "band wider than the 16-px segment / ring deeper than the item");
  // TERMS: 1 fp32 matrix instructions on the raw values; 2 / 3 bf16 terms split on the VALU; 4 three terms split on the matrix
  // cores (MSPLIT); 5 the same with the results' way out pipelined one step behind the chains (PIPE, cooperative form only):
  // step j's matrix instructions cover the de-skew + staging of step j-1's accumulators and the read-back + stores of step
  // j-2's lines, so that a step is [operand reads] [matrix instructions with everything else between them] [DMA issue] [barrier]
  // instead of ending in a serial tail (row shifts that wait for the last matrix instruction, five masked 16-byte LDS writes,
  // barrier, read-back, stores) that nothing covered.
  constexpr bool MSPLIT = TERMS == 4 || TERMS == 5;
  constexpr bool PIPE = TERMS >= 5;          // 6 (measurement): the pipelined way out with the VALU split
  constexpr int ST = TERMS >= 3 ? 3 : TERMS;  // bf16 terms of the VALU split
  // NC: chunks of 32 channels (C = 32 NC; round 6: level 3's 64 channels as a two-chunk K loop).  A logical tile of the sequence is
  // NC raw tiles (one per chunk) in NC consecutive ring slots, an operand is NC register sets, a chain runs its six products once
  // per chunk into its own accumulator (summed where the results leave the registers: no dependent chain of 6 NC instructions).
  static_assert(NC == 1 || PIPE, "several channel chunks: the pipelined cooperative form only");
  static_assert(!PIPE || (COOP && MFN_GRAM_ABLATE == 0), "the pipelined way out is the cooperative form's");

  const int H = p.H, W = p.W;
  const int plane = H * W;
  const int R = min(2 * T, H - ys);      // output rows of this item (the last segment may be short)
  const float *f1n = p.f1 + (size_t)n * (32 * NC) * plane;
  const float *f2n = p.f2 + (size_t)n * (32 * NC) * plane;
  float *outn = p.out + (size_t)n * p.out_nstride;
  const unsigned img_bytes = (unsigned)(32 * NC * plane) * 4u;
  const unsigned chunk_bytes = (unsigned)(32 * plane) * 4u;

  // ---- per-lane constants --------------------------------------------------------------------------------------------
  // DMA of an f2 segment: instruction j covers channels 16j .. 16j+15, lane -> (channel lane/4, 16-byte quad lane%4)
  // DMA of an f1 block:   lane -> (channel lane/4, row (lane/2)%2, quad lane%2): LDS layout [channel][row*8 + x]
  unsigned voffN[2], voffM[2];
  {
    const int xq = x0 - XOFF + 4 * (lane & 3);
    const bool okN = xq >= 0 && xq < W;
    const int xm = x0 + 4 * (lane & 1);
    const bool okM = xm < W;
    MFN_UNROLL
    for (int j = 0; j < 2; ++j) {
      const int c = (lane >> 2) + 16 * j;
      voffN[j] = okN ? (unsigned)(c * plane + xq) * 4u : INVALID;
      voffM[j] = okM ? (unsigned)(c * plane + ((lane >> 1) & 1) * W + xm) * 4u : INVALID;
    }
  }
  // operand gather: lane (g = lane/16, idx = lane%16) reads channels 4j + g, j < 8, of pixel idx
  const int rdoff = (lane >> 4) * 16 + (lane & 15);
  // store: after the row shifts lane (g, n0) owns displacement dx = n0 - XOFF - 4h for the pixels x0+4h .. +3 of block row yy
  // (h = g&1, yy = g>>1).  Byte offset relative to the block's base, which points at plane -D (displacement row -1) of output
  // row ys+2t; the chain adds e*D*plane*4 as soffset: block row 0 lands in displacement row e, block row 1 in e-1.
  unsigned voffS, voffS_up, voffS_lo;   // both block rows / row 0 only (e = 0) / row 1 only (e = D)
  {
    const int g = lane >> 4, n0 = lane & 15, h = g & 1, yy = g >> 1;
    const int dxi = n0 - XOFF - 4 * h + MD;
    const bool ok = dxi >= 0 && dxi < D && x0 + 4 * h < W;
    voffS = ok ? (unsigned)(((1 - yy) * D + dxi) * plane + yy * W + x0 + 4 * h) * 4u : INVALID;
    voffS_up = yy == 0 ? voffS : INVALID;
    voffS_lo = yy == 1 ? voffS : INVALID;
  }
  const unsigned dplane4 = (unsigned)(D * plane) * 4u;   // bytes between displacement rows of the output

  // ---- the DMA pipeline: everything below is unrolled, so n_issued, the stamps and every wait count fold to constants ------
  unsigned n_issued = 0;            // vector memory instructions issued by this wave so far
  unsigned stamp[TT];               // n_issued right after tile k's DMA
  auto issue_tile = [&](auto k_c) __attribute__((always_inline)) {    // tile k of the sequence -> ring slot k % NSLOT
    constexpr int k = decltype(k_c)::value;
    constexpr int kd = SC::kind(k);
    if (MFN_GRAM_ABLATE & 8) { stamp[k] = n_issued; return; }
    MFN_UNROLL
    for (int cc = 0; cc < NC; ++cc) {
      float *slot = ring + ((k % NSLOT) * NC + cc) * SLOT_F;
      if (kd >= 0) {   // f1 block t: rows ys+2t, +1.  A second row that is row H (odd H) is NOT zero-filled except for the last channel: for
        // c < C - 1 the transfer reads row 0 of channel c + 1 (voff + soff is still inside the image's planes).  Those lanes' results
        // are never stored -- the row masks vo_mid / vo_last (r1 false) and, in the cooperative store, rowok_bits drop them; the
        // odd-H cases of tests/test_emu_parity.py::test_correlation_gram_band_on_matrix_cores ((2,32,13,20), (1,32,7,36)) pin that
        const bool in = ys + 2 * kd < H;
        const unsigned soff = (unsigned)((ys + 2 * kd) * W) * 4u + cc * chunk_bytes;
        mfn_dma16_row(f1n, img_bytes, soff, in, slot, voffM[0]);
        mfn_dma16_row(f1n, img_bytes, soff, in, slot + 256, voffM[1]);
      } else {         // f2 row ys-MD+s; rows outside the image (MXNet's pad_size border) read zeros
        const int row = ys - MD + SC::step(-kd - 1);
        const bool in = row >= 0 && row < H;
        const unsigned soff = (unsigned)(row * W) * 4u + cc * chunk_bytes;
        mfn_dma16_row(f2n, img_bytes, soff, in, slot, voffN[0]);
        mfn_dma16_row(f2n, img_bytes, soff, in, slot + 256, voffN[1]);
      }
    }
    n_issued += 2 * NC;
    stamp[k] = n_issued;
  };
  auto wait_tile = [&](int k) { mfn_wait_vm_dyn(n_issued - stamp[k]); };
  // prologue: the first NSLOT tiles go out BEFORE the store-side geometry below is computed (round 6: ~230 instructions, two of
  // them runtime divisions, used to stand between the kernel's entry and its first load); tiles 0 (block 0) and 1 (the f2 row
  // of own step 0) become the first operands
  mfn_static_for<NSLOT>([&](auto k_c) __attribute__((always_inline)) { issue_tile(k_c); });
  MFN_SCHED_BARRIER();
  float raw[8];
  auto read_raw = [&](int k, int cc = 0) {
    const float *su = ring + ((k % NSLOT) * NC + cc) * SLOT_F + rdoff;
    MFN_UNROLL
    for (int j = 0; j < 8; ++j) raw[j] = su[64 * j];
  };
  typedef typename GramOpSel<(TERMS == 1 ? 1 : 3)>::type Op;
  Op Mreg[T][NC];
  Op Ncur[NC];
  GramSel sel;
  constexpr bool WSEL = NC == 1;   // msplit.h: the wide selector at level 2 (faster inside the pass), the lane-local one with two chunks
  if constexpr (MSPLIT) sel = gram_make_sel<WSEL>(lane);
  auto convert = [&](Op &o, float scale) {
    if constexpr (MSPLIT) {
      f32x4 x0, x1;
      MFN_UNROLL
      for (int q = 0; q < 4; ++q) { x0[q] = raw[q] * scale; x1[q] = raw[4 + q] * scale; }
      mfn_bf16x8 term[3];
      MFN_UNROLL
      for (int st = 0; st < 5; ++st) gram_msplit_stage<WSEL>(st, sel, x0, x1, term);
      o.h = term[0]; o.m = term[1]; o.l = term[2];
    } else if constexpr (TERMS == 1) {
      MFN_UNROLL
      for (int q = 0; q < 8; ++q) o.r[q] = raw[q] * scale;
    } else {
      GramWords w;
      MFN_UNROLL
      for (int q = 0; q < 4; ++q) gram_split_pair<ST>(raw[2 * q] * scale, raw[2 * q + 1] * scale, w, q);
      gram_words_to_op(w, o);
    }
  };
  // per block: descriptor of its output rows and the store offsets with the rows a short last segment does not have masked
  mfn_rsrc_t rs[T];
  unsigned vo_mid[T], vo_first[T], vo_last[T];
  MFN_UNROLL
  for (int t = 0; t < T; ++t) {
    rs[t] = mfn_make_rsrc(outn + ((long long)(ys + 2 * t) * W - (long long)D * plane), 0x80000000u);
    const bool r0 = 2 * t < R, r1 = 2 * t + 1 < R;
    vo_first[t] = r0 ? voffS_up : INVALID;
    vo_mid[t] = r1 ? voffS : vo_first[t];
    vo_last[t] = r1 ? voffS_lo : INVALID;
  }

  // ---- cooperative stores: staging geometry (all per-lane values are step-invariant) --------------------------------------
  // A step's results are `nch` chains (f1 blocks, highest block first: ci = 0 ... ) x 2 block rows x D displacements = lines of
  // 128 bytes (32 px); line = (ci*2 + yy)*D + dxi.  Consecutive chains of a step are blocks t, t-1 with e, e+2: their output
  // addresses differ by the CONSTANT (2*D*plane - 2*W) floats, so a lane's global offset is a per-step uniform base (the
  // first chain's, in soffset) + a lane constant.  Staging is XOR-swizzled per line so that the 8 lanes of a ds_write_b128
  // group (8 displacements = 8 lines, same 16-byte column) hit 8 different bank groups; a store instruction reads 1 KB = 8
  // lines back, lane -> (line 8k + lane/8, 16-byte chunk lane%8).
  constexpr int MAXCH = T < MD + 1 ? T : MD + 1;          // chains per step at most
  constexpr int LINES = MAXCH * 2 * D;
  constexpr int NSJ_MAX = ((LINES + 7) / 8 + 3) / 4;      // store instructions per wave and step at most
  constexpr int STG_F = NSJ_MAX * 4 * 8 * 32;             // floats per staging buffer: whole 1 KB groups for every (wave, slot) -- the
                                                          // read-back of a group past LINES (its lanes' stores are masked) stays inside
  int stw[MAXCH];                                         // staging write: float offset of this lane's 16 bytes of chain ci (or -1)
  unsigned cvoff[NSJ_MAX > 0 ? NSJ_MAX : 1];              // store: lane constant of slot j (or INVALID)
  int cbit[NSJ_MAX > 0 ? NSJ_MAX : 1];                    // store: bit (ci*2 + yy) of the line slot j reads
  int strd_off = 0;                                       // staging read: float offset inside a 1 KB group
  mfn_rsrc_t rs_item = rs[0];
  unsigned rowok_bits = 0;                                // bit (t*2 + yy): output row ys + 2t + yy exists in this item
  if (COOP) {
    const int g = lane >> 4, n0 = lane & 15, h = g & 1, yy = g >> 1;
    const int dxi = n0 - XOFF - 4 * h + MD;
    MFN_UNROLL
    for (int ci = 0; ci < MAXCH; ++ci) {
      const int line = (ci * 2 + yy) * D + dxi;
      stw[ci] = (dxi >= 0 && dxi < D) ? line * 32 + (((wave * 2 + h) ^ (line & 7)) * 4) : -1;
    }
    strd_off = (lane >> 3) * 32 + (((lane & 7) ^ ((lane >> 3) & 7)) * 4);
    MFN_UNROLL
    for (int j = 0; j < NSJ_MAX; ++j) {
      const int line = 8 * (wave + 4 * j) + (lane >> 3);
      const int ci = line / (2 * D), yl = (line / D) & 1, dl = line % D;
      const int x = xb0 + 4 * (lane & 7);
      const bool ok = line < LINES && x < W;
      cbit[j] = 1 << (ci * 2 + yl);
      cvoff[j] = ok ? (unsigned)(ci * (2 * D * plane - 2 * W) + ((1 - yl) * D + dl) * plane + yl * W + 4 * (lane & 7)) * 4u : INVALID;
    }
    rs_item = mfn_make_rsrc(outn + ((long long)ys * W + xb0 - (long long)D * plane), 0x80000000u);
    MFN_UNROLL
    for (int t = 0; t < T; ++t) rowok_bits |= (2 * t < R ? 1u : 0u) << (2 * t) | (2 * t + 1 < R ? 1u : 0u) << (2 * t + 1);
  }

  wait_tile(1);
  MFN_STAMP(p.timeline, 1);
  MFN_UNROLL
  for (int cc = 0; cc < NC; ++cc) {
    read_raw(0, cc); convert(Mreg[0][cc], p.inv_c);
    read_raw(1, cc); convert(Ncur[cc], 1.0f);
  }
  MFN_WAIT_LGKM0();
  mfn_static_for<2>([&](auto i_c) __attribute__((always_inline)) {
    constexpr int k = NSLOT + decltype(i_c)::value;
    if constexpr (k < TT) issue_tile(std::integral_constant<int, k>{});
  });

  // chains of step s: blocks t_hi .. t_lo (e = s - 2t within [0, 2MD+1]); 1 KB groups of their lines, per wave
  struct Ch {
    static constexpr int hi(int s) { return (s / 2) < (T - 1) ? (s / 2) : (T - 1); }
    static constexpr int lo(int s) { return (s - (2 * MD + 1) + 1) / 2 > 0 ? (s - (2 * MD + 1) + 1) / 2 : 0; }
    static constexpr int per_wave(int s) { return (((hi(s) - lo(s) + 1) * 2 * D + 7) / 8 + 3) / 4; }
  };
  auto chains_hi = [](int s) constexpr { return Ch::hi(s); };
  auto chains_lo = [](int s) constexpr { return Ch::lo(s); };
  auto stores_per_wave = [](int s) constexpr { return Ch::per_wave(s); };
  unsigned valid_pending = 0;       // valid_bits of the step whose lines wait in the staging buffer
  // the staged lines of own step jp (already read back into v[]) -> global memory: the same number of store instructions in
  // every wave (the counted waits are static); first chain = block t_hi, e0 = s - 2 t_hi: its base is plane (e0-1)*D of row
  // ys + 2 t_hi, i.e. soffset (e0*D*plane + 2 t_hi W)*4 from rs_item (plane -D of row ys)
  auto store_lines = [&](auto jp_c, const f32x4 *v, unsigned valid) __attribute__((always_inline)) {
    constexpr int sp = SC::step(decltype(jp_c)::value);
    constexpr int th = Ch::hi(sp);
    constexpr int nsj = Ch::per_wave(sp);
    const unsigned soff = (unsigned)((sp - 2 * th) * D * plane + 2 * th * W) * 4u;
    MFN_UNROLL
    for (int jj = 0; jj < nsj; ++jj) {
      const unsigned vo = (valid & (unsigned)cbit[jj]) ? cvoff[jj] : INVALID;
      if (MFN_GRAM_ABLATE & 2) { if (v[jj][0] == 1.2345e30f) mfn_bstore4_so(rs_item, vo, soff, v[jj], POL); }
      else { mfn_bstore4_so(rs_item, vo, soff, v[jj], POL); n_issued += 1; }
    }
  };

  // de-skew + stage chain ci (block t_hi - ci) of step s out of a[] into sbuf: the cooperative form's way out of the registers
  auto stage_chain = [&](int s, int ci, const f32x4 (*a)[NC], float *sbuf, unsigned &valid_bits) __attribute__((always_inline)) {
    const int t = Ch::hi(s) - ci;
    const int e = s - 2 * t;
    f32x4 sum = a[t][0];
    MFN_UNROLL
    for (int cc = 1; cc < NC; ++cc) { sum[0] += a[t][cc][0]; sum[1] += a[t][cc][1]; sum[2] += a[t][cc][2]; sum[3] += a[t][cc][3]; }   // chunks in order
    f32x4 v;
    v[0] = sum[0];
    v[1] = mfn_dpp_row_shl<1>(sum[1], sum[1]);
    v[2] = mfn_dpp_row_shl<2>(sum[2], sum[2]);
    v[3] = mfn_dpp_row_shl<3>(sum[3], sum[3]);
    if (LEAKY) {
      MFN_UNROLL
      for (int i = 0; i < 4; ++i) v[i] = mfn_leaky01(v[i]);
    }
    if (e < D) valid_bits |= ((rowok_bits >> (2 * t)) & 1u) << (ci * 2);
    if (e >= 1) valid_bits |= ((rowok_bits >> (2 * t + 1)) & 1u) << (ci * 2 + 1);
    if (stw[ci] >= 0) *reinterpret_cast<f32x4 *>(sbuf + stw[ci]) = v;
  };
  f32x4 accP[T][NC];                // PIPE: the previous step's accumulators (one per channel chunk)
  unsigned valid_staged = 0;        // PIPE: valid_bits of the lines staged during the previous step

  mfn_static_for<J>([&](auto j_c) __attribute__((always_inline)) {
    constexpr int j = decltype(j_c)::value;
    constexpr int s = SC::step(j);
    // the tiles of own step j+1: its f2 row and, if one becomes due, a block (which precedes the row in the sequence)
    constexpr bool more = j + 1 < J;
    constexpr int tM = more ? SC::block_at(j + 1) : -1;
    constexpr bool moreM = tM >= 0;
    constexpr int kN = more ? SC::seqN(j + 1) : 0;
    if (more) wait_tile(kN);
    Op Nnext[NC];
    // raw tiles of the next step out of LDS first: their conversion (VALU) hides behind this step's matrix instructions
    float rawM[NC][8], rawN[NC][8];
    MFN_UNROLL
    for (int cc = 0; cc < NC; ++cc) {
      if (moreM) { read_raw(kN - 1, cc); MFN_UNROLL for (int i = 0; i < 8; ++i) rawM[cc][i] = raw[i] * p.inv_c; }
      if (more) { read_raw(kN, cc); MFN_UNROLL for (int i = 0; i < 8; ++i) rawN[cc][i] = raw[i]; }
    }
    GramWords wM, wN;
    // The chains of a step have independent accumulators: their matrix instructions are written interleaved (product k of
    // every active block, then product k+1 ...), so that a block's dependent accumulate never waits for its own predecessor,
    // and the conversion of the next step's operands is spread between them pair by pair (MFN_GRAM_SCHED pins that order with
    // scheduling fences; measured: 13.85 us with, 13.75 us without -- the wave is bound by its VALU / VMEM issue, not by
    // the order, profiles/r04_corr_gram_experiments.md).
    constexpr int NPROD = TERMS >= 3 ? 6 : (TERMS == 1 ? 8 : 3);
    constexpr int npairs = (TERMS == 1 || MSPLIT) ? 0 : (moreM ? 4 : 0) + (more ? 4 : 0);
    // MSPLIT: the next step's tiles as accumulator tiles; five split stages spread between this step's chains
    f32x4 xM0[NC], xM1[NC], xN0[NC], xN1[NC];
    mfn_bf16x8 cM[NC][3], cN[NC][3];
    int next_stage = 0;
    if constexpr (MSPLIT) {
      MFN_UNROLL
      for (int q = 0; q < 4; ++q) {
        MFN_UNROLL
        for (int cc = 0; cc < NC; ++cc) {
          if (moreM) { xM0[cc][q] = rawM[cc][q]; xM1[cc][q] = rawM[cc][4 + q]; }
          if (more) { xN0[cc][q] = rawN[cc][q]; xN1[cc][q] = rawN[cc][4 + q]; }
        }
      }
    }
    auto split_stage = [&]() {
      if constexpr (MSPLIT) {
        MFN_UNROLL
        for (int cc = 0; cc < NC; ++cc) {
          if (moreM) gram_msplit_stage<WSEL>(next_stage, sel, xM0[cc], xM1[cc], cM[cc]);
          if (more) gram_msplit_stage<WSEL>(next_stage, sel, xN0[cc], xN1[cc], cN[cc]);
        }
        ++next_stage;
      }
    };
    int done_pairs = 0, unit = 0, nunits = 0;
    MFN_UNROLL
    for (int t = 0; t < T; ++t) if (s - 2 * t >= 0 && s - 2 * t <= 2 * MD + 1) nunits += NPROD * NC;
    auto convert_next_pair = [&]() {
      if constexpr (TERMS != 1 && !MSPLIT) {
        if (done_pairs < npairs) {
          const int q = done_pairs & 3;
          if (moreM && done_pairs < 4) gram_split_pair<ST>(rawM[0][2 * q], rawM[0][2 * q + 1], wM, q);
          else gram_split_pair<ST>(rawN[0][2 * q], rawN[0][2 * q + 1], wN, q);
          ++done_pairs;
        }
      }
    };
    f32x4 acc[T][NC];
    MFN_UNROLL
    for (int t = 0; t < T; ++t)
      MFN_UNROLL
      for (int cc = 0; cc < NC; ++cc) { acc[t][cc][0] = 0.f; acc[t][cc][1] = 0.f; acc[t][cc][2] = 0.f; acc[t][cc][3] = 0.f; }
    // PIPE: what leaves the registers behind this step's matrix instructions -- the chains of step j-1 (accP) into staging buffer
    // (j-1) & 1 and, read back first, the lines of step j-2 out of buffer j & 1 (staged during step j-1, barrier at its end)
    constexpr int sP = j > 0 ? SC::step(j > 0 ? j - 1 : 0) : 0;
    constexpr int nchP = (PIPE && j > 0) ? Ch::hi(sP) - Ch::lo(sP) + 1 : 0;
    constexpr int nsjPP = (PIPE && j > 1) ? Ch::per_wave(SC::step(j > 1 ? j - 2 : 0)) : 0;
    f32x4 vpp[nsjPP > 0 ? nsjPP : 1];
    if constexpr (PIPE && j > 1) {
      const float *pbuf = stg + (j & 1) * STG_F;
      MFN_UNROLL
      for (int jj = 0; jj < nsjPP; ++jj) vpp[jj] = *reinterpret_cast<const f32x4 *>(pbuf + (wave + 4 * jj) * 256 + strd_off);
    }
    unsigned valid_bitsP = 0;
    int post_done = 0;
    auto post_chain = [&]() {
      stage_chain(sP, post_done, accP, stg + ((j + 1) & 1) * STG_F, valid_bitsP);
      ++post_done;
    };
    if (!(MFN_GRAM_ABLATE & 1)) {
      MFN_UNROLL
      for (int k = 0; k < NPROD; ++k) {
        MFN_UNROLL
        for (int t = 0; t < T; ++t) {
          const int e = s - 2 * t;
          if (e >= 0 && e <= 2 * MD + 1) {
            MFN_UNROLL
            for (int cc = 0; cc < NC; ++cc) {
              const Op &Mo = Mreg[t][cc];
              const Op &No = Ncur[cc];
              if constexpr (TERMS == 1) {
                acc[t][cc] = MFN_MFMA_16x16x4(Mo.r[k], No.r[k], acc[t][cc]);   // channels 4k .. 4k+3
              } else {
                // smallest terms first: l*h, h*l, m*m, m*h, h*m, h*h (two terms: l*h, h*l, h*h)
                const mfn_bf16x8 &a = TERMS >= 3 ? (k == 0 ? Mo.l : (k == 2 || k == 3 ? Mo.m : Mo.h)) : (k == 0 ? Mo.l : Mo.h);
                const mfn_bf16x8 &b = TERMS >= 3 ? (k == 1 ? No.l : (k == 2 || k == 4 ? No.m : No.h)) : (k == 1 ? No.l : No.h);
                acc[t][cc] = MFN_MFMA_16x16x32_BF16(a, b, acc[t][cc]);
              }
              if constexpr (MSPLIT || PIPE) {   // stage i behind unit floor(i * nunits / 5)
                if constexpr (MSPLIT) while (more && next_stage < 5 && next_stage * nunits <= unit * 5) split_stage();
                if constexpr (PIPE) while (post_done < nchP && post_done * nunits <= unit * nchP) post_chain();
                MFN_SCHED_BARRIER();
              }
              if (MFN_GRAM_SCHED) {   // pair i behind unit floor(i * nunits / npairs)
                while (done_pairs < npairs && done_pairs * nunits <= unit * npairs) convert_next_pair();
                MFN_SCHED_BARRIER();
              }
              ++unit;
            }
          }
        }
      }
    }
    while (done_pairs < npairs) convert_next_pair();
    if constexpr (MSPLIT) {
      while (more && next_stage < 5) split_stage();
      MFN_UNROLL
      for (int cc = 0; cc < NC; ++cc) {
        if (moreM) { Mreg[moreM ? tM : 0][cc].h = cM[cc][0]; Mreg[moreM ? tM : 0][cc].m = cM[cc][1]; Mreg[moreM ? tM : 0][cc].l = cM[cc][2]; }
        if (more) { Nnext[cc].h = cN[cc][0]; Nnext[cc].m = cN[cc][1]; Nnext[cc].l = cN[cc][2]; }
      }
    } else if constexpr (TERMS == 1) {
      if (moreM) { MFN_UNROLL for (int i = 0; i < 8; ++i) Mreg[moreM ? tM : 0][0].r[i] = rawM[0][i]; }
      if (more) { MFN_UNROLL for (int i = 0; i < 8; ++i) Nnext[0].r[i] = rawN[0][i]; }
    } else {
      if (moreM) gram_words_to_op(wM, Mreg[moreM ? tM : 0][0]);
      if (more) gram_words_to_op(wN, Nnext[0]);
    }
    MFN_WAIT_LGKM0();                      // the ring slots of the tiles just read are free: the next tiles of the sequence go there
    {
      constexpr int consumed = more ? kN + 1 : TT;       // tiles read so far
      constexpr int issued_before = moreM ? consumed - 2 + NSLOT : consumed - 1 + NSLOT;   // issued = NSLOT + tiles read before this step's reads
      mfn_static_for<2>([&](auto i_c) __attribute__((always_inline)) {
        constexpr int k = issued_before + decltype(i_c)::value;
        if constexpr (more && decltype(i_c)::value < (moreM ? 2 : 1) && k < TT) issue_tile(std::integral_constant<int, k>{});
      });
    }
    if constexpr (PIPE) {
      while (post_done < nchP) post_chain();
      if constexpr (j > 1) store_lines(std::integral_constant<int, (j > 1 ? j - 2 : 0)>{}, vpp, valid_staged);
      valid_staged = valid_bitsP;
      MFN_LDS_BARRIER();              // step j-1's lines are staged (read back in step j+1); buffer j & 1 is free for step j's
      MFN_UNROLL
      for (int t = 0; t < T; ++t)
        MFN_UNROLL
        for (int cc = 0; cc < NC; ++cc) accP[t][cc] = acc[t][cc];
    } else if (!(MFN_GRAM_ABLATE & 1)) {
      // active chains of this step, highest block first (ci = 0, 1, ...): t_hi = min(T-1, s/2) downwards while e = s - 2t <= 2MD+1
      constexpr int t_hi = chains_hi(s);
      constexpr int nch = t_hi - chains_lo(s) + 1;
      float *sbuf = stg + (j & 1) * STG_F;
      // Cooperative stores are one step late: the lines of step j-1 (staged before the barrier below) are read back here,
      // behind this step's matrix instructions, and stored after this step's own results went into the other staging buffer --
      // the barrier wait and the LDS round trip hide behind work instead of ending every step.
      constexpr int nsj_prev = COOP && j > 0 ? stores_per_wave(SC::step(j > 0 ? j - 1 : 0)) : 0;
      f32x4 vprev[nsj_prev > 0 ? nsj_prev : 1];
      if (COOP && j > 0) {
        MFN_LDS_BARRIER();                                   // every wave's 32 bytes of every line of step j-1 are staged
        const float *pbuf = stg + ((j - 1) & 1) * STG_F;
        MFN_UNROLL
        for (int jj = 0; jj < nsj_prev; ++jj) vprev[jj] = *reinterpret_cast<const f32x4 *>(pbuf + (wave + 4 * jj) * 256 + strd_off);
      }
      unsigned valid_bits = 0;                               // bit (ci*2 + yy): that line group is written this step and its row exists
      MFN_UNROLL
      for (int ci = 0; ci < nch; ++ci) {
        const int t = t_hi - ci;
        const int e = s - 2 * t;
        // de-skew: register i of lane n holds (x = 4h+i, dx = n-XOFF-4h-i); lane n0 collects dx0 = n0-XOFF-4h from lanes n0+i
        f32x4 v;
        v[0] = acc[t][0][0];
        v[1] = mfn_dpp_row_shl<1>(acc[t][0][1], acc[t][0][1]);
        v[2] = mfn_dpp_row_shl<2>(acc[t][0][2], acc[t][0][2]);
        v[3] = mfn_dpp_row_shl<3>(acc[t][0][3], acc[t][0][3]);
        if (LEAKY) {
          MFN_UNROLL
          for (int i = 0; i < 4; ++i) v[i] = mfn_leaky01(v[i]);
        }
        if (COOP) {
          // block row 0 is displacement row e (exists while e < D), block row 1 is e-1 (exists from e = 1)
          if (e < D) valid_bits |= ((rowok_bits >> (2 * t)) & 1u) << (ci * 2);
          if (e >= 1) valid_bits |= ((rowok_bits >> (2 * t + 1)) & 1u) << (ci * 2 + 1);
          if (stw[ci] >= 0) *reinterpret_cast<f32x4 *>(sbuf + stw[ci]) = v;   // (an unconditional write into a dummy slot instead of the exec mask: measured, no gain)
        } else {
          const unsigned vo = e == 0 ? vo_first[t] : (e == 2 * MD + 1 ? vo_last[t] : vo_mid[t]);
          if (MFN_GRAM_ABLATE & 2) { if (v[0] == 1.2345e30f) mfn_bstore4_so(rs[t], vo, (unsigned)e * dplane4, v, POL); }   // keeps the chain live
          else { mfn_bstore4_so(rs[t], vo, (unsigned)e * dplane4, v, POL); n_issued += 1; }
        }
      }
      if (COOP && j > 0) store_lines(std::integral_constant<int, (j > 0 ? j - 1 : 0)>{}, vprev, valid_pending);
      valid_pending = valid_bits;
    }
    if (more) { MFN_UNROLL for (int cc = 0; cc < NC; ++cc) Ncur[cc] = Nnext[cc]; }
    if (j == J / 2 - 1) MFN_STAMP(p.timeline, 2);
    MFN_SCHED_BARRIER();
  });
  if constexpr (PIPE) {   // what is still on its way: step J-1 in the registers, step J-2 in staging buffer J & 1
    constexpr int sL = SC::step(J - 1);
    constexpr int nchL = Ch::hi(sL) - Ch::lo(sL) + 1;
    unsigned valid_last = 0;
    MFN_UNROLL
    for (int ci = 0; ci < nchL; ++ci) stage_chain(sL, ci, accP, stg + ((J - 1) & 1) * STG_F, valid_last);
    {
      constexpr int nsj = Ch::per_wave(SC::step(J - 2));
      f32x4 v2[nsj];
      const float *pbuf = stg + (J & 1) * STG_F;
      MFN_UNROLL
      for (int jj = 0; jj < nsj; ++jj) v2[jj] = *reinterpret_cast<const f32x4 *>(pbuf + (wave + 4 * jj) * 256 + strd_off);
      store_lines(std::integral_constant<int, J - 2>{}, v2, valid_staged);
    }
    MFN_LDS_BARRIER();
    {
      constexpr int nsj = Ch::per_wave(sL);
      f32x4 v1[nsj];
      const float *pbuf = stg + ((J - 1) & 1) * STG_F;
      MFN_UNROLL
      for (int jj = 0; jj < nsj; ++jj) v1[jj] = *reinterpret_cast<const f32x4 *>(pbuf + (wave + 4 * jj) * 256 + strd_off);
      store_lines(std::integral_constant<int, J - 1>{}, v1, valid_last);
    }
  } else if (COOP && !(MFN_GRAM_ABLATE & 1)) {   // the last step's lines
    constexpr int nsj = Ch::per_wave(SC::step(J - 1));
    f32x4 vlast[nsj];
    MFN_LDS_BARRIER();
    const float *pbuf = stg + ((J - 1) & 1) * STG_F;
    MFN_UNROLL
    for (int jj = 0; jj < nsj; ++jj) vlast[jj] = *reinterpret_cast<const f32x4 *>(pbuf + (wave + 4 * jj) * 256 + strd_off);
    store_lines(std::integral_constant<int, J - 1>{}, vlast, valid_pending);
  }
  if (MFN_GRAM_ABLATE & 1) {   // keep the conversions live
    float sink = mfn_bf16_at(reinterpret_cast<const float *>(&Ncur[0]), 0) + mfn_bf16_at(reinterpret_cast<const float *>(&Mreg[T - 1][0]), 1);
    if (sink == 1.2345e30f) outn[lane] = sink;
  }
  MFN_STAMP(p.timeline, 3);
}

// One wave per (image, row segment, strip, step parity); NWV adjacent strips per block.  COOP (NWV = 4, SP = 1): the block's
// waves exchange their results through LDS (one barrier per step) and store full 128-byte lines.
template <int D, int T, int NSLOT, int NWV, int TERMS, int POL, bool LEAKY, int SP, bool COOP, int NC = 1>
__global__ __launch_bounds__(NWV * 64, 2) void corr_gram_kernel(CorrGramParams p) {
  static_assert(!COOP || (NWV == 4 && SP == 1), "cooperative stores: four strips = one 128-byte line, one wave per item");
  MFN_DYN_SHARED(float, lds_all);
  const int lane = threadIdx.x & 63;
  const int wave = MFN_UNIFORM(threadIdx.x >> 6);
  float *ring = lds_all + (size_t)wave * NSLOT * NC * 512;
  MFN_STAMP(p.timeline, 0);
  int bid = blockIdx.x;
  if (p.xcd_swizzle) bid = (int)mfn_xcd_remap((unsigned)bid, (unsigned)p.xcd_swizzle);
  // exact for bid < 2^32 / divisor (the launch checks it): q = floor(bid * ceil(2^32 / d) / 2^32)
  int rest = (int)mfn_div_magic((unsigned)bid, p.bx_magic);
  const int bxs = bid - rest * p.bx_per_row;
  const int par = rest % SP;
  rest /= SP;
  const int n = (int)mfn_div_magic((unsigned)rest, p.segs_magic);
  const int seg = rest - n * p.segs;
  const int sx = bxs * NWV + wave;
  if (!COOP && sx >= p.strips) return;   // COOP: a wave past the last strip keeps the block's barriers company (its lanes are all masked)
  const int x0 = sx * 8, ys = seg * (2 * T);
  if constexpr (COOP) corr_gram_wave<D, T, NSLOT, TERMS, POL, LEAKY, 1, 0, true, NC>(p, ring, lane, n, ys, x0, lds_all + (size_t)NWV * NSLOT * NC * 512, wave, bxs * NWV * 8);
  else {
    if (SP == 1 || par == 0) corr_gram_wave<D, T, NSLOT, TERMS, POL, LEAKY, SP, 0>(p, ring, lane, n, ys, x0);
    else corr_gram_wave<D, T, NSLOT, TERMS, POL, LEAKY, SP, SP - 1>(p, ring, lane, n, ys, x0);
  }
}

template <int D, int T, int NSLOT, int NWV, int TERMS, int POL, bool LEAKY, int SP = 1, bool COOP = false, int NC = 1>
inline int corr_gram_launch(CorrGramParams p, hipStream_t stream, const char *name) {
  constexpr int MD = (D - 1) / 2;
  constexpr int MAXCH = T < MD + 1 ? T : MD + 1;
  p.rows = 2 * T;
  p.strips = cdiv(p.W, 8);
  p.segs = cdiv(p.H, p.rows);
  p.bx_per_row = cdiv(p.strips, NWV);
  const long nblk = (long)p.N * p.segs * SP * p.bx_per_row;
  if (nblk <= 0) return 0;
  if (nblk * (long)(p.bx_per_row > p.segs ? p.bx_per_row : p.segs) >= (1L << 32)) return -1;   // the magic divisions' range
  if (p.xcd_swizzle) p.xcd_swizzle = (int)nblk;
  p.bx_magic = mfn_make_magic((unsigned)p.bx_per_row);
  p.segs_magic = mfn_make_magic((unsigned)p.segs);
  constexpr int NSJ_MAX = ((MAXCH * 2 * D + 7) / 8 + 3) / 4;
  const size_t lds = ((size_t)NWV * NSLOT * NC * 512 + (COOP ? 2 * NSJ_MAX * 4 * 8 * 32 : 0)) * sizeof(float);
  return launch(name, corr_gram_kernel<D, T, NSLOT, NWV, TERMS, POL, LEAKY, SP, COOP, NC>, dim3((unsigned)nblk), dim3(NWV * 64), lds, stream, p);
}

inline bool corr_variant_gram(int v) { return v == 46 || v == 48; }
// the launch's index range (corr_gram_launch: block id x divisor below 2^32 for the magic divisions): images of a million rows do not
// take this kernel -- the plan asks before it picks it, so that such a call runs on another kernel instead of failing
inline bool corr_gram_range_ok(int N, int H, int W, int rows) {
  const long segs = (H + rows - 1) / rows, bx = ((W + 7) / 8 + 3) / 4;
  const long nblk = (long)N * segs * bx;
  return nblk * (bx > segs ? bx : segs) < (1L << 32);
}
// Output rows per work item: 6 or 8 (T = 3 / 4 blocks; the schedule is compile-time).  One wave per item, eight resident waves
// per CU (two blocks of four): the level-2 launch of 384x512 at batch 8 is 2048 items of 6 rows = one residency round.  Fewer
// rows per item mean more halo (an item converts rows + 2*md f2 rows); 8 where 6 does not divide H and 8 does (448x1024:
// 112 rows).  corr.rows overrides.
inline int corr_gram_rows(int /*N*/, int H, int /*W*/, int override_rows, int C = 32) {
  if (C == 64) return 2;
  if (override_rows == 6 || override_rows == 8) return override_rows;
  return (H % 6 != 0 && H % 8 == 0) ? 8 : 6;
}
// corr.variant 40: three terms, six products, cooperative stores (full lines through LDS, written through unless the caller's
// policy says plain), a ring of 4 tiles.  The measured-and-lost forms of round 4 -- 41 (two terms, ~1e-5 relative), 42 (every
// wave stores its own 32-byte runs: 13.2 us against 10.45), 43 (two waves per item) -- are no longer instantiated in the shipped
// library; the template parameters that selected them (TERMS, SP, COOP) remain, profiles/r04_corr_gram_experiments.md
// holds their numbers.
template <int D>
inline int corr_gram_variant(const CorrGramParams &p, int variant, hipStream_t s) {
  const bool wt = (p.store_policy & 2) != 0;
#define MFN_GRAM_(TT_, TERMS_, NAME_) \
  (p.leaky ? (wt ? corr_gram_launch<D, TT_, 4, 4, TERMS_, 2, true, 1, true>(p, s, NAME_) : corr_gram_launch<D, TT_, 4, 4, TERMS_, 0, true, 1, true>(p, s, NAME_)) \
           : (wt ? corr_gram_launch<D, TT_, 4, 4, TERMS_, 2, false, 1, true>(p, s, NAME_) : corr_gram_launch<D, TT_, 4, 4, TERMS_, 0, false, 1, true>(p, s, NAME_)))
  if (p.C == 64) {   // two channel chunks (level 3): the plan's form only
#define MFN_GRAM2_(TT_) \
  (p.leaky ? (wt ? corr_gram_launch<D, TT_, 4, 4, 5, 2, true, 1, true, 2>(p, s, "corr_gram_v48c2") : corr_gram_launch<D, TT_, 4, 4, 5, 0, true, 1, true, 2>(p, s, "corr_gram_v48c2")) \
           : (wt ? corr_gram_launch<D, TT_, 4, 4, 5, 2, false, 1, true, 2>(p, s, "corr_gram_v48c2") : corr_gram_launch<D, TT_, 4, 4, 5, 0, false, 1, true, 2>(p, s, "corr_gram_v48c2")))
    return MFN_GRAM2_(1);   // 2-row items (4-row items tie, 6-row items lose: profiles/r06_corr_l3.txt)
#undef MFN_GRAM2_
  }
  if (variant == 46) return p.rows == 8 ? MFN_GRAM_(4, 1, "corr_gram_v46") : MFN_GRAM_(3, 1, "corr_gram_v46");
  return p.rows == 8 ? MFN_GRAM_(4, 5, "corr_gram_v48") : MFN_GRAM_(3, 5, "corr_gram_v48");
#undef MFN_GRAM_
}

}  // namespace mfn
