// conv_wgrad.h -- weight gradient of the network's 3x3 / stride 1 convolutions (dilation 1, 2 or a multiple of 4; pad = dilation)
// as a GEMM over the PIXELS on the fp32 matrix instruction, for gfx950.
//
// Caller row: the backward of Gluon nn.Conv2D (/root/reference/network/MaskFlownet.py:79-163) inside pipeline.py:112-113;
// semantics: gw[o][c][ky][kx] = sum over (n, y, x) of g[n][o][y][x] * x[n][c][y + (ky-1) d][x + (kx-1) d] (zero outside the image).
// Until round 4 mfn_conv2d_bwd formed it with the DEFORMABLE convolution's weight-gradient kernels and a zero offset tensor
// (bilinear machinery at integer positions, a zero-filled offset buffer per call: 17.3 ms of a 37.5 ms training step).
//
// v_mfma_f32_32x32x2_f32: A = 32 filters x 2 pixels, B = 2 pixels x 32 channels, D[filter][channel] per tap -- the exact fp32
// FMA chain, K = pixels.  A lane (row, half) holds FOUR consecutive pixels of its filter / channel plane (one aligned 16-byte
// load): the two halves take pixels P+0..3 and P+4..7 of an 8-pixel run of one image row, and step e = 0..3 of the run
// contracts pixel e of both halves.  The nine taps read the same three input rows: per row three aligned 16-byte loads (the
// quad left of the lane's, its own, the one right of it) give the quads shifted by -d, 0, +d in registers (d = 1, 2; for d % 4 == 0
// the shifted quads are themselves aligned loads).  Columns outside the image: the borrowed elements are zeroed (d < 4) or the
// whole quad is (d % 4 == 0: W % 4 == 0 keeps a quad on one side of the border); rows outside: the tap's quads are zero.
// Per 32-pixel run and wave: 22 loads of 16 bytes per lane, 144 matrix instructions (9216 ALU cycles).
// One wave = one (32-filter, 32-channel) tile of the gradient x all nine taps (144 accumulator registers) x one slice of the
// runs; the slices' partial tiles go to slabs and conv_wgrad_reduce_kernel adds them in slice order (deterministic, no atomics).
#pragma once
#include "../mfn_rt.h"

namespace mfn {

struct ConvWgradParams {
  const float *g;       // (N, Cout, H, W): gradient w.r.t. the convolution's output (before the fused activation's derivative: after it)
  const float *x;       // (N, Cin, H, W)
  float *slabs;         // [slice][filter tile][channel tile][tap 9][32 filters][32 channels]
  int N, Cin, Cout, H, W, dil;
  int tiles_o, tiles_c, slices;
  int runs;             // N * H * (W / 8)
  int runs_per_slice;
};

// DM: 1, 2 = dilation 1 / 2 (shifted quads from aligned neighbours); 4 = dilation % 4 == 0 (aligned shifted loads).
// QPL = quads per lane and run: a run is 8 QPL pixels of one image row, lane (col, half) owns pixels 4 QPL half .. + 4 QPL - 1 of it.
// QPL = 4 (W % 32 == 0): the two halves of a row of lanes read one whole 128-byte line per (plane, row) -- with QPL = 1 a lane
// uses 16 bytes of the line per load and the line has left the L1 by the time the next runs want it (first form of this kernel:
// 14.8 ms for the training step's weight gradients against 13.8 ms through the deformable kernels).
template <int DM, int QPL>
__global__ __launch_bounds__(64) void conv_wgrad_kernel(ConvWgradParams p) {
  const int lane = threadIdx.x & 63;
  const int col = lane & 31, half = lane >> 5;
  const int slice = blockIdx.x, tc = blockIdx.y, to = blockIdx.z;
  const int H = p.H, W = p.W, d = p.dil;
  const int plane = H * W, wr = W / (8 * QPL);
  const int o = to * 32 + col, c = tc * 32 + col;
  const bool o_ok = o < p.Cout, c_ok = c < p.Cin;
  f32x16 acc[9];
  MFN_UNROLL
  for (int t = 0; t < 9; ++t)
    MFN_UNROLL
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const int r0 = slice * p.runs_per_slice, r1 = min(p.runs, r0 + p.runs_per_slice);
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  auto ld4 = [&](const float *base, int off, bool ok) -> f32x4 {   // aligned quad at element offset `off` (clamped when masked)
    const f32x4 v = *reinterpret_cast<const f32x4 *>(base + (ok ? off : 0));
    return ok ? v : zero4;
  };
  if (DM == 4) {
    // shifted quads are aligned loads (3 per quad and row): plain loop, hipcc's own order (82 TFLOP/s at dilation 4)
    for (int run = r0; run < r1; ++run) {
      const int xr = run % wr, row = run / wr;
      const int y = row % H, n = row / H;
      const int xq = xr * (8 * QPL) + 4 * QPL * half;            // first column of this lane's quads
      const float *gp = p.g + ((size_t)n * p.Cout + (o_ok ? o : 0)) * plane;
      const float *xp = p.x + ((size_t)n * p.Cin + (c_ok ? c : 0)) * plane;
      f32x4 a[QPL];
      MFN_UNROLL
      for (int q = 0; q < QPL; ++q) a[q] = ld4(gp, y * W + xq + 4 * q, o_ok);
      MFN_UNROLL
      for (int ky = 0; ky < 3; ++ky) {
        const int yy = y + (ky - 1) * d;
        const bool rok = c_ok && yy >= 0 && yy < H;
        const int ro = (rok ? yy : 0) * W;
        f32x4 b[QPL][3];
        MFN_UNROLL
        for (int q = 0; q < QPL; ++q) {
          const int xx = xq + 4 * q;
          b[q][0] = ld4(xp, ro + xx - d, rok && xx - d >= 0);
          b[q][1] = ld4(xp, ro + xx, rok);
          b[q][2] = ld4(xp, ro + xx + d, rok && xx + d < W);
        }
        MFN_UNROLL
        for (int q = 0; q < QPL; ++q)
          MFN_UNROLL
          for (int e = 0; e < 4; ++e)
            MFN_UNROLL
            for (int kx = 0; kx < 3; ++kx) acc[ky * 3 + kx] = MFN_MFMA_32x32x2(a[q][e], b[q][kx][e], acc[ky * 3 + kx]);
      }
    }
  } else {
    // Software pipeline over (run, input row): the quads of step s+1 are requested before the 48 QPL / 4 matrix instructions of step
    // s (left alone hipcc requested each quad right before its use -- a dozen exposed round trips per run, 60 TFLOP/s).  Two
    // register sets, the loop unrolled over two runs so that every index is static; a run past the slice loads zeros.
    f32x4 A[2][QPL], C[2][QPL + 2];
    // masks are out-of-range offsets of range-checked buffer loads (zeros from the hardware): a select on the loaded value would
    // need the value -- hipcc then waits for the quads of step s+1 before the matrix instructions of step s
    constexpr unsigned OOR = 0xFFFFFF00u;
    const unsigned gbytes = (unsigned)(p.Cout * plane) * 4u, xbytes = (unsigned)(p.Cin * plane) * 4u;
    auto issue = [&](int run, int ky, f32x4 (&Cb)[QPL + 2], f32x4 (&Ab)[QPL]) {
#ifdef MFN_WGRAD_ABLATE   // measurement builds: operands loaded for the first two runs only
      if (run >= r0 + 2) return;
#endif
      const bool live = run < r1;
      const int rr = live ? run : r0;
      const int xr = rr % wr, row = rr / wr;
      const int y = row % H, n = MFN_UNIFORM(row / H);
      const int xq = xr * (8 * QPL) + 4 * QPL * half;
      if (ky == 0) {
        const float *gimg = p.g + (size_t)n * p.Cout * plane;
        const unsigned go = (unsigned)(o * plane + y * W + xq) * 4u;
        MFN_UNROLL
        for (int q = 0; q < QPL; ++q) Ab[q] = mfn_bload4(gimg, gbytes, (o_ok && live) ? go + 16u * q : OOR);
      }
      const float *ximg = p.x + (size_t)n * p.Cin * plane;
      const int yy = y + (ky - 1) * d;
      const bool rok = live && c_ok && yy >= 0 && yy < H;
      const unsigned xo = (unsigned)(c * plane + yy * W + xq) * 4u;
      Cb[0] = mfn_bload4(ximg, xbytes, (rok && xq >= 4) ? xo - 16u : OOR);     // the quad left of the lane's, its own, the one right of them
      MFN_UNROLL
      for (int q = 0; q < QPL; ++q) Cb[q + 1] = mfn_bload4(ximg, xbytes, rok ? xo + 16u * q : OOR);
      Cb[QPL + 1] = mfn_bload4(ximg, xbytes, (rok && xq + 4 * QPL < W) ? xo + 16u * QPL : OOR);
    };
    auto mma = [&](int ky, const f32x4 (&Cb)[QPL + 2], const f32x4 (&Ab)[QPL]) {
      MFN_UNROLL
      for (int q = 0; q < QPL; ++q) {
        const f32x4 &Lq = Cb[q], &Cq = Cb[q + 1], &Rq = Cb[q + 2];
        const f32x4 b0 = DM == 1 ? f32x4{Lq[3], Cq[0], Cq[1], Cq[2]} : f32x4{Lq[2], Lq[3], Cq[0], Cq[1]};
        const f32x4 b2 = DM == 1 ? f32x4{Cq[1], Cq[2], Cq[3], Rq[0]} : f32x4{Cq[2], Cq[3], Rq[0], Rq[1]};
        MFN_UNROLL
        for (int e = 0; e < 4; ++e) {
          acc[ky * 3 + 0] = MFN_MFMA_32x32x2(Ab[q][e], b0[e], acc[ky * 3 + 0]);
          acc[ky * 3 + 1] = MFN_MFMA_32x32x2(Ab[q][e], Cq[e], acc[ky * 3 + 1]);
          acc[ky * 3 + 2] = MFN_MFMA_32x32x2(Ab[q][e], b2[e], acc[ky * 3 + 2]);
        }
      }
    };
    issue(r0, 0, C[0], A[0]);
    for (int run = r0; run < r1; run += 2) {
      issue(run, 1, C[1], A[0]);     MFN_SCHED_BARRIER(); mma(0, C[0], A[0]); MFN_SCHED_BARRIER();
      issue(run, 2, C[0], A[0]);     MFN_SCHED_BARRIER(); mma(1, C[1], A[0]); MFN_SCHED_BARRIER();
      issue(run + 1, 0, C[1], A[1]); MFN_SCHED_BARRIER(); mma(2, C[0], A[0]); MFN_SCHED_BARRIER();
      issue(run + 1, 1, C[0], A[1]); MFN_SCHED_BARRIER(); mma(0, C[1], A[1]); MFN_SCHED_BARRIER();
      issue(run + 1, 2, C[1], A[1]); MFN_SCHED_BARRIER(); mma(1, C[0], A[1]); MFN_SCHED_BARRIER();
      issue(run + 2, 0, C[0], A[0]); MFN_SCHED_BARRIER(); mma(2, C[1], A[1]); MFN_SCHED_BARRIER();
    }
  }
  // D register r of lane (col, half): filter (r&3) + 8 (r>>2) + 4 half, channel col -- rows of 32 channels: coalesced
  float *sl = p.slabs + (((size_t)slice * p.tiles_o + to) * p.tiles_c + tc) * (9 * 1024);
  MFN_UNROLL
  for (int t = 0; t < 9; ++t)
    MFN_UNROLL
    for (int r = 0; r < 16; ++r) sl[(t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * 32 + col] = acc[t][r];
}

// The same gradient for dilation 1 on v_mfma_f32_32x32x16_bf16, fp32-EQUIVALENT (the library's default arithmetic; conv_wgrad_kernel
// is MFN_ARITH_FP32's): both operands as three bf16 terms, six of the nine partial products, fp32 accumulate -- 6 matrix instructions
// of 32 cycles per 16 pixels and tap against 8 of 64.  K = 16 pixels of one image row: lane (col, half) owns pixels 8 half .. 8 half + 7
// of the run -- two aligned quads of its filter's gradient plane, and per input row the four quads from 4 pixels left of them.  The ten
// values x[-1 .. 8] of a row are split ONCE, as five pairs (55 instructions): the operands of the taps kx = 0 and kx = 2 are four
// consecutive words of the five as they are, the one of kx = 1 starts at an odd element -- four v_alignbit_b32 per term.  Per run and
// wave: 14 loads of 16 bytes, ~250 VALU instructions, 54 matrix instructions (1 728 cycles; the fp32 form: 72 of 64 = 4 608).
// Masks are out-of-range offsets (zeros from the hardware), as in conv_wgrad_kernel.  Tile / slice / slab layout and the reduce are conv_wgrad_kernel's.
__global__ __launch_bounds__(64) void conv_wgrad_mma_kernel(ConvWgradParams p) {
  const int lane = threadIdx.x & 63;
  const int col = lane & 31, half = lane >> 5;
  const int slice = blockIdx.x, tc = blockIdx.y, to = blockIdx.z;
  const int H = p.H, W = p.W;
  const int plane = H * W, wr = W / 16;
  const int o = to * 32 + col, c = tc * 32 + col;
  const bool o_ok = o < p.Cout, c_ok = c < p.Cin;
  f32x16 acc[9];
  MFN_UNROLL
  for (int t = 0; t < 9; ++t)
    MFN_UNROLL
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const int r0 = slice * p.runs_per_slice, r1 = min(p.runs, r0 + p.runs_per_slice);
  constexpr unsigned OOR = 0xFFFFFF00u;
  const unsigned gbytes = (unsigned)(p.Cout * plane) * 4u, xbytes = (unsigned)(p.Cin * plane) * 4u;
  // software pipeline over (run, input row) as in conv_wgrad_kernel: the quads of step s + 1 are requested before the 18 matrix
  // instructions of step s; two register sets (126 -> ~100 registers beside the 144 accumulators: two waves per SIMD)
  f32x4 A[2][2], C[2][4];   // [set][quad] of the gradient; [set][quad L, C0, C1, R] of one input row
  auto coords = [&](int run, bool &live, int &y, int &n, int &x0) {
    live = run < r1;
    const int rr = live ? run : r0;
    const int xr = rr % wr, row = rr / wr;
    y = row % H; n = MFN_UNIFORM(row / H);
    x0 = xr * 16 + 8 * half;
  };
  auto issue = [&](int run, int ky, f32x4 (&Cb)[4], f32x4 (&Ab)[2]) {
#ifdef MFN_WGRAD_ABLATE   // measurement builds: operands loaded for the first two runs only
    if (run >= r0 + 2) return;
#endif
    bool live; int y, n, x0;
    coords(run, live, y, n, x0);
    if (ky == 0) {
      const float *gimg = p.g + (size_t)n * p.Cout * plane;
      const unsigned go = (unsigned)(o * plane + y * W + x0) * 4u;
      Ab[0] = mfn_bload4(gimg, gbytes, (o_ok && live) ? go : OOR);
      Ab[1] = mfn_bload4(gimg, gbytes, (o_ok && live) ? go + 16u : OOR);
    }
    const float *ximg = p.x + (size_t)n * p.Cin * plane;
    const int yy = y + ky - 1;
    const bool rok = live && c_ok && yy >= 0 && yy < H;
    const unsigned xo = (unsigned)(c * plane + yy * W + x0) * 4u;
    Cb[0] = mfn_bload4(ximg, xbytes, (rok && x0 >= 4) ? xo - 16u : OOR);
    Cb[1] = mfn_bload4(ximg, xbytes, rok ? xo : OOR);
    Cb[2] = mfn_bload4(ximg, xbytes, rok ? xo + 16u : OOR);
    Cb[3] = mfn_bload4(ximg, xbytes, (rok && x0 + 8 < W) ? xo + 32u : OOR);
  };
  mfn_bf16x8 ah, am, al;
  auto work = [&](int ky, const f32x4 (&Cb)[4], const f32x4 (&Ab)[2]) {
    if (ky == 0) {
      const float a8[8] = {Ab[0][0], Ab[0][1], Ab[0][2], Ab[0][3], Ab[1][0], Ab[1][1], Ab[1][2], Ab[1][3]};
      mfn_split3x8_scalar(a8, ah, am, al);
    }
    const float v[10] = {Cb[0][3], Cb[1][0], Cb[1][1], Cb[1][2], Cb[1][3], Cb[2][0], Cb[2][1], Cb[2][2], Cb[2][3], Cb[3][0]};   // x[-1 .. 8]
    unsigned wh[5], wm[5], wl[5];
    mfn_split3_pairs<5>(v, wh, wm, wl);
    MFN_UNROLL
    for (int kx = 0; kx < 3; ++kx) {
      mfn_bf16x8 bh, bm, bl;
      if (kx == 1) {   // elements 1 .. 8: every word straddles two pairs
        bh = mfn_words_to_bf16x8(mfn_alignbit16(wh[1], wh[0]), mfn_alignbit16(wh[2], wh[1]), mfn_alignbit16(wh[3], wh[2]), mfn_alignbit16(wh[4], wh[3]));
        bm = mfn_words_to_bf16x8(mfn_alignbit16(wm[1], wm[0]), mfn_alignbit16(wm[2], wm[1]), mfn_alignbit16(wm[3], wm[2]), mfn_alignbit16(wm[4], wm[3]));
        bl = mfn_words_to_bf16x8(mfn_alignbit16(wl[1], wl[0]), mfn_alignbit16(wl[2], wl[1]), mfn_alignbit16(wl[3], wl[2]), mfn_alignbit16(wl[4], wl[3]));
      } else {         // elements 0 .. 7 (kx = 0) / 2 .. 9 (kx = 2): four of the five words
        const int w0 = kx == 0 ? 0 : 1;
        bh = mfn_words_to_bf16x8(wh[w0], wh[w0 + 1], wh[w0 + 2], wh[w0 + 3]);
        bm = mfn_words_to_bf16x8(wm[w0], wm[w0 + 1], wm[w0 + 2], wm[w0 + 3]);
        bl = mfn_words_to_bf16x8(wl[w0], wl[w0 + 1], wl[w0 + 2], wl[w0 + 3]);
      }
      f32x16 &d = acc[ky * 3 + kx];
      d = MFN_MFMA_32x32x16_BF16(al, bh, d);
      d = MFN_MFMA_32x32x16_BF16(ah, bl, d);
      d = MFN_MFMA_32x32x16_BF16(am, bm, d);
      d = MFN_MFMA_32x32x16_BF16(am, bh, d);
      d = MFN_MFMA_32x32x16_BF16(ah, bm, d);
      d = MFN_MFMA_32x32x16_BF16(ah, bh, d);
    }
  };
  issue(r0, 0, C[0], A[0]);
  for (int run = r0; run < r1; run += 2) {
    issue(run, 1, C[1], A[0]);     MFN_SCHED_BARRIER(); work(0, C[0], A[0]); MFN_SCHED_BARRIER();
    issue(run, 2, C[0], A[0]);     MFN_SCHED_BARRIER(); work(1, C[1], A[0]); MFN_SCHED_BARRIER();
    issue(run + 1, 0, C[1], A[1]); MFN_SCHED_BARRIER(); work(2, C[0], A[0]); MFN_SCHED_BARRIER();
    issue(run + 1, 1, C[0], A[1]); MFN_SCHED_BARRIER(); work(0, C[1], A[1]); MFN_SCHED_BARRIER();
    issue(run + 1, 2, C[1], A[1]); MFN_SCHED_BARRIER(); work(1, C[0], A[1]); MFN_SCHED_BARRIER();
    issue(run + 2, 0, C[0], A[0]); MFN_SCHED_BARRIER(); work(2, C[1], A[1]); MFN_SCHED_BARRIER();
  }
  float *sl = p.slabs + (((size_t)slice * p.tiles_o + to) * p.tiles_c + tc) * (9 * 1024);
  MFN_UNROLL
  for (int t = 0; t < 9; ++t)
    MFN_UNROLL
    for (int r = 0; r < 16; ++r) sl[(t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * 32 + col] = acc[t][r];
}

// gw (Cout, Cin, 3, 3) (+)= the slices' partial tiles.  A block = 32 consecutive slab elements x 8 slice groups: thread (element, group
// g) adds slices g, g+8, ... (four requests in flight), the groups meet in LDS and are added in group order -- a fixed order, and
// 1/8 of the serial chain a thread per element had (a one-tile layer such as conv1b has 1300 slices: 300 of its 396 us were this sum).
struct ConvWgradReduceParams { const float *slabs; float *gw; int Cin, Cout, tiles_o, tiles_c, slices, add; };
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(ConvWgradReduceParams p) {
  MFN_DYN_SHARED(float, part);   // [8 groups][32 elements]
  const size_t per_slice = (size_t)p.tiles_o * p.tiles_c * 9 * 1024;
  const int el = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const size_t idx = (size_t)blockIdx.x * 32 + el;      // per_slice is a multiple of 1024
  float s = 0.f;
  for (int k = grp; k < p.slices; k += 32) {
    float v[4];
    MFN_UNROLL
    for (int u = 0; u < 4; ++u) {
      const int kk = k + 8 * u;
      const float x = p.slabs[(size_t)min(kk, p.slices - 1) * per_slice + idx];
      v[u] = kk < p.slices ? x : 0.f;
    }
    s += (v[0] + v[1]) + (v[2] + v[3]);
  }
  part[grp * 32 + el] = s;
  __syncthreads();
  if (grp != 0) return;
  MFN_UNROLL
  for (int g = 1; g < 8; ++g) s += part[g * 32 + el];
  const int ci = (int)(idx & 31), oi = (int)((idx >> 5) & 31), t = (int)((idx >> 10) % 9);
  const size_t tile = idx / (9 * 1024);
  const int tc = (int)(tile % p.tiles_c), to = (int)(tile / p.tiles_c);
  const int o = to * 32 + oi, c = tc * 32 + ci;
  if (o >= p.Cout || c >= p.Cin) return;
  float *dst = p.gw + ((size_t)o * p.Cin + c) * 9 + t;
  *dst = p.add ? *dst + s : s;
}

inline bool conv_wgrad_shape_ok(int Cin, int Cout, int H, int W, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw) {
  return kh == 3 && kw == 3 && sh == 1 && sw == 1 && dh == dw && ph == dh && pw == dw && (dh == 1 || dh == 2 || dh % 4 == 0) &&
         W % 8 == 0 && (size_t)Cin * H * W < ((size_t)1 << 30) && (size_t)Cout * H * W < ((size_t)1 << 30);
}
struct ConvWgradPlan { int qpl, tiles_o, tiles_c, slices, runs, runs_per_slice; size_t slab_bytes; };
// mma: conv_wgrad_mma_kernel's runs of 16 pixels (dilation 1, W % 16 == 0, the default arithmetic)
inline bool conv_wgrad_mma_ok(int W, int dil) { return dil == 1 && W % 16 == 0; }
inline ConvWgradPlan conv_wgrad_plan(int N, int Cin, int Cout, int H, int W, bool mma = false) {
  ConvWgradPlan q;
  q.tiles_o = cdiv(Cout, 32); q.tiles_c = cdiv(Cin, 32);
  q.qpl = mma ? 2 : (W % 32 == 0 ? 4 : 1);
  q.runs = N * H * (W / (8 * q.qpl));
  int s = cdiv(2048, q.tiles_o * q.tiles_c);       // ~2 k waves: two per SIMD
  const size_t tile_bytes = (size_t)q.tiles_o * q.tiles_c * 9 * 1024 * sizeof(float);
  if ((size_t)s * tile_bytes > ((size_t)48 << 20)) s = (int)(((size_t)48 << 20) / tile_bytes);   // slabs of at most 48 MB
  if (s > q.runs / 2) s = q.runs / 2;              // at least two runs per slice (the pipeline's unroll)
  if (s < 1) s = 1;
  q.runs_per_slice = cdiv(q.runs, s);
  q.slices = cdiv(q.runs, q.runs_per_slice);
  q.slab_bytes = (size_t)q.slices * q.tiles_o * q.tiles_c * 9 * 1024 * sizeof(float);
  return q;
}
inline int conv_wgrad_launch(const float *g, const float *x, float *gw, void *slabs, int N, int Cin, int Cout, int H, int W, int dil,
                             int add, hipStream_t s, bool mma = false) {
  mma = mma && conv_wgrad_mma_ok(W, dil);
  const ConvWgradPlan q = conv_wgrad_plan(N, Cin, Cout, H, W, mma);
  ConvWgradParams p{g, x, (float *)slabs, N, Cin, Cout, H, W, dil, q.tiles_o, q.tiles_c, q.slices, q.runs, q.runs_per_slice};
  const dim3 grid((unsigned)q.slices, (unsigned)q.tiles_c, (unsigned)q.tiles_o);
  int rc;
  if (mma) {
    if ((rc = launch("conv_wgrad_bf16x3", conv_wgrad_mma_kernel, grid, dim3(64), 0, s, p))) return rc;
    ConvWgradReduceParams rp{(const float *)slabs, gw, Cin, Cout, q.tiles_o, q.tiles_c, q.slices, add};
    const size_t per_slice = (size_t)q.tiles_o * q.tiles_c * 9 * 1024;
    return launch("conv_wgrad_reduce", conv_wgrad_reduce_kernel, dim3((unsigned)(per_slice / 32)), dim3(256), 256 * sizeof(float), s, rp);
  }
#define MFN_WG_(DM_) (q.qpl == 4 ? launch("conv_wgrad", conv_wgrad_kernel<DM_, 4>, grid, dim3(64), 0, s, p) \
                                 : launch("conv_wgrad", conv_wgrad_kernel<DM_, 1>, grid, dim3(64), 0, s, p))
  rc = dil == 1 ? MFN_WG_(1) : (dil == 2 ? MFN_WG_(2) : MFN_WG_(4));
#undef MFN_WG_
  if (rc) return rc;
  ConvWgradReduceParams rp{(const float *)slabs, gw, Cin, Cout, q.tiles_o, q.tiles_c, q.slices, add};
  const size_t per_slice = (size_t)q.tiles_o * q.tiles_c * 9 * 1024;
  return launch("conv_wgrad_reduce", conv_wgrad_reduce_kernel, dim3((unsigned)(per_slice / 32)), dim3(256), 256 * sizeof(float), s, rp);
}

}  // namespace mfn
