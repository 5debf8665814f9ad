// correlation_gram.h -- the cost volume of 32-channel levels as a BANDED GRAM MATRIX on the bf16 matrix cores (gfx950).
//
// Replaces MXNet Correlation at /root/reference/network/MaskFlownet.py:193-195 (md=4, 81 ch) and :440-441 (md=2, 25 ch)
// for C == 32 (level 2, the launch BASELINE.json's north_star names); semantics as oracle/mfn_ref_body.inc correlation_fwd.
//
// Why another formulation (profiles/r03_corr_pmc.md): corr_dma_kernel (correlation.h) spends 36 packed
// FMAs and 64 bytes of LDS operand reads per lane-channel; at level 2 that is 5.6 us of VALU issue + 3.2 us of LDS returns per
// CU which do not overlap, next to 7.1 us of HBM streaming.  Here the contraction over the 32 channels is ONE K = 32 step of
// v_mfma_f32_16x16x32_bf16:
//   * out[dy][dx][y][x] = sum_c f1[c][y][x] * f2[c][y+dy][x+dx] is the band |x'-x| <= md, |y'-y| <= md of the Gram matrix
//     G[(y,x)][(y',x')] = sum_c f1[c,y,x] f2[c,y',x'].  A 16 x 16 tile of G with M = an 8 x 2 pixel block of f1 and
//     N = a 16 x 1 row segment of f2 starting 4 columns left of the block is 50.6 % useful for md = 4 (10 tiles per 16 pixels x 81
//     displacements) -- the densest (M, N) shape pair for the 16 x 16 tile.
//   * exact fp32: every operand is split hi + mid + lo into three bf16 terms (24 significant bits) and the six products with
//     weight >= 2^-16 are accumulated in fp32 by the matrix core (the dropped ones are below one fp32 rounding): mfn_rt.h
//     mfn_split3x8.  TERMS = 2 is the measured two-term / three-product variant (2^-17 relative per product; not the default).
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

namespace mfn {

struct CorrGramParams {
  const float *f1;
  const float *f2;
  float *out;
  int N, H, W;            // C == 32
  int rows;               // output rows per work item (even)
  int strips, segs;       // ceil(W / 8), ceil(H / rows)
  int bx_per_row;         // blocks along x: ceil(strips / waves per block)
  size_t out_nstride;     // elements between images of `out` (a channel slice of a concat buffer when > D*D*H*W)
  int store_policy;       // mfn_bstore4
  int leaky;              // fused LeakyReLU(0.1)
  int xcd_swizzle;
  float inv_c;            // 1/32, folded into the f1 operand before the split (a power of two: exact)
};

struct GramOp { mfn_bf16x8 h, m, l; };

// D = 2*md+1; AU = units (f1 block + two f2 rows) the LDS-DMA runs ahead = ring slots per wave; NWV waves per block;
// TERMS = 3 exact, 2 measured variant.
//
// Iteration q of a wave (q = 0 .. T+MD-1, T = f1 blocks of the item): "unit" q = f2 rows 2q, 2q+1 of the item's window (image
// rows ys-MD+2q, +1) and f1 block q-MD.  The f2 rows live CONVERTED in a register window of 2*(MD+1) rows; block t = q-MD meets
// the rows 2t .. 2t+2MD+1 = window slots (2t+e) mod 2(MD+1), e = 0 .. 2MD+1: 2(MD+1) chains, every one of them active in
// every iteration (rows outside the image are zeros and produce the zeros MXNet's padding produces), so the body is ONE basic
// block -- hipcc interleaves the chains, the de-skew moves, the stores and the conversion of unit q+1.  The first MD
// iterations only fill the window.
// POL (store policy, mfn_bstore4) and LEAKY (fused LeakyReLU(0.1)) are compile-time: a uniform branch per chain would cut the
// body into basic blocks again.
template <int D, int AU, int NWV, int TERMS, int POL, bool LEAKY>
__global__ __launch_bounds__(NWV * 64, 2) void corr_gram_kernel(CorrGramParams p) {
  constexpr int MD = (D - 1) / 2;
  constexpr int NP = MD + 1;         // f2 row pairs under one f1 block
  constexpr int NROW = 2 * NP;       // rows of the register window = chains per block
  constexpr int SLOT_F = 512;        // floats per raw tile: 32 channels x 16 px
  constexpr int UNIT_F = 3 * SLOT_F; // f1 block, f2 row 2q, f2 row 2q+1
  constexpr int XOFF = 4;            // the f2 segment starts XOFF columns left of the strip (16-byte aligned, >= MD)
  constexpr unsigned INVALID = 0xFFFFFF00u;
  static_assert(MD >= 1 && MD <= 4 && AU >= 2 && AU <= 8, "band wider than the 16-px segment / stamp FIFO is 8 bytes");

  MFN_DYN_SHARED(float, lds_all);
  const int lane = threadIdx.x & 63;
  const int wave = MFN_UNIFORM(threadIdx.x >> 6);
  float *ring = lds_all + (size_t)wave * AU * UNIT_F;

  // ---- work item: (image, row segment, strip) ---------------------------------------------------------------------------
  int bid = blockIdx.x;
  if (p.xcd_swizzle) bid = (int)mfn_xcd_remap((unsigned)bid, gridDim.x);
  const int bxs = bid % p.bx_per_row;
  const int rest = bid / p.bx_per_row;
  const int seg = rest % p.segs;
  const int n = rest / p.segs;
  const int sx = bxs * NWV + wave;
  if (sx >= p.strips) return;   // no block-wide synchronisation in this kernel
  const int H = p.H, W = p.W;
  const int plane = H * W;
  const int x0 = sx * 8, ys = seg * p.rows;
  const int R = min(p.rows, H - ys);     // output rows of this item
  const int T = (R + 1) >> 1;            // 8 x 2 blocks of f1
  const int Q = T + MD;                  // iterations
  const float *f1n = p.f1 + (size_t)n * 32 * plane;
  const float *f2n = p.f2 + (size_t)n * 32 * plane;
  float *outn = p.out + (size_t)n * p.out_nstride;
  const unsigned img_bytes = (unsigned)(32 * plane) * 4u;

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
  // (h = g&1, yy = g>>1).  Byte offset relative to the chain's base, which points at plane (e-1)*D of output row ys+2t:
  // block row yy = 0 is displacement row e (exists while e < D), row yy = 1 is e-1 (exists from e = 1).
  unsigned voffS, voffS_up, voffS_lo;   // both block rows / row 0 only (e = 0) / row 1 only (e = D)
  {
    const int g = lane >> 4, n0 = lane & 15, h = g & 1, yy = g >> 1;
    const int dxi = n0 - XOFF - 4 * h + MD;
    const bool ok = dxi >= 0 && dxi < D && x0 + 4 * h < W;
    voffS = ok ? (unsigned)(((1 - yy) * D + dxi) * plane + yy * W + x0 + 4 * h) * 4u : INVALID;
    voffS_up = yy == 0 ? voffS : INVALID;
    voffS_lo = yy == 1 ? voffS : INVALID;
  }

  // ---- the DMA pipeline ------------------------------------------------------------------------------------------------
  // Every unit is six DMA instructions, also the ones that bring nothing (f1 blocks of the fill iterations, units past
  // the item's end: num_records 0 -> zeros, no memory traffic): the body has no branch around them.
  unsigned n_issued = 0;            // vector memory instructions issued by this wave so far (uniform)
  unsigned long long fifo = 0;      // n_issued right after each in-flight unit's DMA, oldest in the low byte
  auto issue_unit = [&](int u) {
    float *slot = ring + (u % AU) * UNIT_F;
    {   // f1 block u-MD: rows ys+2(u-MD), +1.  The row is folded into the descriptor's base and its range shrunk by as much:
        // the range check is exact for the image (a second row that is row H reads zeros, not the next image).
      const int t = u - MD;
      const bool in = t >= 0 && t < T;
      const int rw = in ? (ys + 2 * t) * W : 0;
      const mfn_rsrc_t r = mfn_make_rsrc(f1n + rw, in ? img_bytes - (unsigned)rw * 4u : 0u);
      mfn_dma16(r, slot, voffM[0]);
      mfn_dma16(r, slot + 256, voffM[1]);
    }
    MFN_UNROLL
    for (int k = 0; k < 2; ++k) {   // f2 rows ys-MD+2u+k; rows outside the image (MXNet's pad_size border) read zeros
      const int row = ys - MD + 2 * u + k;
      const bool in = row >= 0 && row < H && u < Q;
      const int rw = in ? row * W : 0;
      const mfn_rsrc_t r = mfn_make_rsrc(f2n + rw, in ? img_bytes - (unsigned)rw * 4u : 0u);
      mfn_dma16(r, slot + (1 + k) * SLOT_F, voffN[0]);
      mfn_dma16(r, slot + (1 + k) * SLOT_F + 256, voffN[1]);
    }
    n_issued += 6;
  };
  MFN_UNROLL
  for (int u = 0; u < AU; ++u) {
    issue_unit(u);
    fifo |= (unsigned long long)(n_issued & 0xffu) << (8 * u);
  }

  // Program order of an iteration (fenced with MFN_SCHED_BARRIER so that hipcc keeps ONE raw tile -- eight registers -- alive
  // at a time; left alone it hoists all 24 LDS reads and the three conversions to the top and spills):
  //   wait(unit q+1) | chain 0, read f2 row a | chain 1 || chain 2 + convert a -> window slot of e=0 | chain 3, read f2 row b ||
  //   chain 4 + convert b -> slot of e=1 | chain 5, read f1 block || chain 6, refill the ring slot | chain 7 + convert -> Mnext |
  //   chains 8, 9 (NROW = 10; for md = 2 the same stations at chains 0..5)
  float raw[8];
  int cslot = 0;                          // ring slot of the unit being consumed = (q+1) % AU
  auto wait_unit = [&]() {                // the unit in ring slot `cslot` has landed
    const unsigned stamp = (unsigned)(fifo & 0xffu);
    fifo >>= 8;
    mfn_wait_vm_dyn((n_issued - stamp) & 0xffu);
  };
  auto read_raw = [&](int part) {         // part 0: f1 block, 1 / 2: f2 rows
    const float *su = ring + cslot * UNIT_F + part * SLOT_F + rdoff;
    MFN_UNROLL
    for (int j = 0; j < 8; ++j) raw[j] = su[64 * j];
  };
  auto refill = [&](int u) {              // the slot just read is free: unit u + AU goes there
    MFN_WAIT_LGKM0();
    issue_unit(u + AU);
    fifo |= (unsigned long long)(n_issued & 0xffu) << (8 * (AU - 1));
  };

  GramOp Mcur, Mnext;
  GramOp Nwin[NROW];
  auto convert = [&](GramOp &o, float scale) {
    float v[8];
    MFN_UNROLL
    for (int j = 0; j < 8; ++j) v[j] = raw[j] * scale;
    if (TERMS == 3) mfn_split3x8(v, o.h, o.m, o.l);
    else { mfn_split2x8(v, o.h, o.l); o.m = o.l; }
  };

  wait_unit();
  read_raw(1); convert(Nwin[0], 1.0f);
  read_raw(2); convert(Nwin[1], 1.0f);
  read_raw(0); convert(Mcur, p.inv_c);     // block -MD: zeros (fill iteration), unused
  refill(0);
  cslot = 1 % AU;

  const unsigned dplane4 = (unsigned)(D * plane) * 4u;   // bytes between displacement rows of the output
  for (int qb = 0; qb < Q; qb += NP) {
    MFN_UNROLL
    for (int qq = 0; qq < NP; ++qq) {
      const int q = qb + qq;
      if (q < Q) {
        wait_unit();                       // unit q+1; past the item's end: a unit of zeros
        const int t = q - MD;
        const int s0 = (2 * (qq + 1)) % NROW, s1 = (2 * (qq + 1) + 1) % NROW;   // window slots of rows 2(q+1), +1 = this block's e = 0, 1
        if (t >= 0) {
          const bool odd_end = 2 * t + 1 >= R;                 // last block of an item with an odd row count
          const unsigned vo_mid = odd_end ? voffS_up : voffS;
          const unsigned vo_last = odd_end ? INVALID : voffS_lo;
          // one descriptor per block: base = plane -D (displacement row -1) of output row ys+2t; chain e adds e*D*plane*4 as soffset
          const mfn_rsrc_t rs = mfn_make_rsrc(outn + ((long long)(ys + 2 * t) * W - (long long)D * plane), 0x80000000u);
          auto chain = [&](int e, int ws) {
            const GramOp &No = Nwin[ws];
            f32x4 acc;
            acc[0] = 0.f; acc[1] = 0.f; acc[2] = 0.f; acc[3] = 0.f;
            if (TERMS == 3) {   // smallest terms first
              acc = MFN_MFMA_16x16x32_BF16(Mcur.l, No.h, acc);
              acc = MFN_MFMA_16x16x32_BF16(Mcur.h, No.l, acc);
              acc = MFN_MFMA_16x16x32_BF16(Mcur.m, No.m, acc);
              acc = MFN_MFMA_16x16x32_BF16(Mcur.m, No.h, acc);
              acc = MFN_MFMA_16x16x32_BF16(Mcur.h, No.m, acc);
              acc = MFN_MFMA_16x16x32_BF16(Mcur.h, No.h, acc);
            } else {
              acc = MFN_MFMA_16x16x32_BF16(Mcur.l, No.h, acc);
              acc = MFN_MFMA_16x16x32_BF16(Mcur.h, No.l, acc);
              acc = MFN_MFMA_16x16x32_BF16(Mcur.h, No.h, acc);
            }
            // de-skew: register i of lane n holds (x = 4h+i, dx = n-XOFF-4h-i); lane n0 collects dx0 = n0-XOFF-4h from lanes n0+i
            f32x4 v;
            v[0] = acc[0];
            v[1] = mfn_dpp_row_shl<1>(acc[1], acc[1]);
            v[2] = mfn_dpp_row_shl<2>(acc[2], acc[2]);
            v[3] = mfn_dpp_row_shl<3>(acc[3], acc[3]);
            if (LEAKY) {
              MFN_UNROLL
              for (int i = 0; i < 4; ++i) v[i] = mfn_leaky01(v[i]);
            }
            const unsigned vo = e == 0 ? voffS_up : (e == NROW - 1 ? vo_last : vo_mid);
            mfn_bstore4_so(rs, vo, (unsigned)e * dplane4, v, POL);
          };
          MFN_UNROLL
          for (int e = 0; e < NROW; ++e) {
            const int ws = (((2 * (qq - MD) + e) % NROW) + NROW) % NROW;   // window slot of row 2t+e: compile-time (qb % NP == 0)
            // stations (see above); NROW >= 6
            if (e == 2) convert(Nwin[s0], 1.0f);
            if (e == 4) convert(Nwin[s1], 1.0f);
            if (e == (NROW >= 8 ? 6 : 5)) refill(q + 1);
            if (e == (NROW >= 8 ? 7 : 5)) convert(Mnext, p.inv_c);
            chain(e, ws);
            if (e == 0) read_raw(1);
            if (e == 3) read_raw(2);
            if (e == (NROW >= 8 ? 5 : 4)) read_raw(0);
            if (e == 1 || e == 3 || e == 5) MFN_SCHED_BARRIER();
          }
          n_issued += NROW;
          Mcur = Mnext;
        } else {   // fill iteration: the window is not complete yet
          read_raw(1); convert(Nwin[s0], 1.0f);
          read_raw(2); convert(Nwin[s1], 1.0f);
          read_raw(0); convert(Mcur, p.inv_c);
          refill(q + 1);
        }
        cslot = cslot + 1 == AU ? 0 : cslot + 1;
      }
    }
  }
}

template <int D, int AU, int NWV, int TERMS, int POL, bool LEAKY>
inline int corr_gram_launch(CorrGramParams p, hipStream_t stream, const char *name) {
  p.strips = cdiv(p.W, 8);
  p.segs = cdiv(p.H, p.rows);
  p.bx_per_row = cdiv(p.strips, NWV);
  const long nblk = (long)p.N * p.segs * p.bx_per_row;
  if (nblk <= 0) return 0;
  const size_t lds = (size_t)NWV * AU * 3 * 512 * sizeof(float);
  return launch(name, corr_gram_kernel<D, AU, NWV, TERMS, POL, LEAKY>, dim3((unsigned)nblk), dim3(NWV * 64), lds, stream, p);
}

inline bool corr_variant_gram(int v) { return v == 40 || v == 41; }
// Output rows per work item (even).  One wave per item and eight resident waves per CU (two blocks of four, 72 KB of LDS each):
// the whole launch is one residency round when the items number 2048; fewer rows per item mean more halo (an item converts
// rows + 2*md f2 rows).  corr.rows overrides.
inline int corr_gram_rows(int N, int H, int W, int override_rows) {
  if (override_rows >= 2) return override_rows & ~1;
  const long strip_rows = (long)N * cdiv(W, 8) * H;
  int rows = (int)((strip_rows + 2047) / 2048);
  rows = (rows + 1) & ~1;
  if (rows < 4) rows = 4;
  if (rows > H) rows = (H + 1) & ~1;
  return rows;
}
// corr.variant 40: exact (three terms, six products); 41: two terms, three products (measured variant, ~1e-5 relative; md = 4,
// no fused activation).  The store policy is write-through (sc0 sc1) when the caller's policy has that bit, plain otherwise.
template <int D>
inline int corr_gram_variant(const CorrGramParams &p, int variant, hipStream_t s) {
  const bool wt = (p.store_policy & 2) != 0;
  if (variant == 41 && D == 9 && !p.leaky)
    return wt ? corr_gram_launch<9, 3, 4, 2, 2, false>(p, s, "corr_gram_v41") : corr_gram_launch<9, 3, 4, 2, 0, false>(p, s, "corr_gram_v41");
  if (p.leaky)
    return wt ? corr_gram_launch<D, 3, 4, 3, 2, true>(p, s, "corr_gram_v40") : corr_gram_launch<D, 3, 4, 3, 0, true>(p, s, "corr_gram_v40");
  return wt ? corr_gram_launch<D, 3, 4, 3, 2, false>(p, s, "corr_gram_v40") : corr_gram_launch<D, 3, 4, 3, 0, false>(p, s, "corr_gram_v40");
}

}  // namespace mfn
