// correlation_band.h -- cost volume of the coarse pyramid levels (few pixels, many channels) in ONE launch.
//
// Same operator as correlation.h (MXNet Correlation, kernel_size=1, stride1=stride2=1, pad=md, multiply;
// /root/reference/network/MaskFlownet.py:193-195); semantics as oracle/mfn_ref_body.inc correlation_fwd.
//
// At levels 6..4 of the 384x512 pyramid an image has 48..768 pixels and 196..96 channels: a pixel tiling alone
// leaves most CUs idle and makes every workgroup walk a long channel loop, and splitting the channels over
// workgroups costs a second launch plus a round trip of partial sums through HBM (corr_reduce_kernel).  Here a
// workgroup owns a band of R output rows of one image and ALL channels: its (up to 16) waves form G channel
// groups, every group stages CK-channel pieces of the band (f1 rows, f2 rows +-md with a 4-column zero halo)
// in its own LDS region and accumulates them, and the groups' accumulators are added through LDS in a fixed
// tree order (deterministic) before group 0 normalises and stores.  No workspace, no second kernel.
//
// Lane task = (displacement row dy, band row r, 4-pixel quad qx): 36 accumulators out(dy, dx=0..D-1, 4 px), fed
// per channel by one ds_read_b128 of f1 and three of f2 -- the packed anti-diagonal FMA scheme of correlation.h.
#pragma once
#include "../mfn_rt.h"
#include "correlation.h"

namespace mfn {

struct CorrBandParams {
  const float *f1;
  const float *f2;
  float *out;
  size_t out_nstride;  // elements between consecutive images of `out`
  int st_policy;       // mfn_store4_stream
  int N, C, H, W;
  int R, bands;      // output rows per workgroup, ceil(H / R)
  int G, WG;         // channel groups per workgroup, waves per group (blockDim = G * WG * 64)
  int cpg;           // channels per group (a multiple of CK; the last group may run short)
  int nchunks;       // cpg / CK: LDS stages every group walks (barriers are block-wide)
  int S;             // stages in flight per group (ring depth)
  int NI;            // LDS-DMA instructions per wave and stage
  int red_groups;    // largest power of two < G (0 when G == 1): first round of the reduction tree
  float inv_sumelems, sumelems;
  int exact_div, leaky;
  unsigned long long *timeline;  // measurement only (mfn_debug_set_timeline)
  float inv_bands, inv_rq, inv_q, inv_r2q2, inv_q2;  // 1/(R*QW), 1/QW, 1/((R+2md)*(QW+2)), 1/(QW+2): division-free index decode
};

// a / b for 0 <= a < 2^22 without the ~40-instruction integer division sequence: the float quotient is off by at
// most one, one correction step makes it exact
__device__ __forceinline__ void corr_band_divmod(int a, int b, float inv_b, int &q, int &r) {
  q = (int)((float)a * inv_b);
  r = a - q * b;
  if (r < 0) { --q; r += b; }
  if (r >= b) { ++q; r -= b; }
}

// s_waitcnt takes an immediate: a wave-uniform switch over the few counts the ring can ask for
__device__ __forceinline__ void corr_band_wait_vm(int n) {
  switch (n) {
    case 0: MFN_WAIT_VM(0); break;
    case 1: MFN_WAIT_VM(1); break;
    case 2: MFN_WAIT_VM(2); break;
    case 3: MFN_WAIT_VM(3); break;
    case 4: MFN_WAIT_VM(4); break;
    case 5: MFN_WAIT_VM(5); break;
    case 6: MFN_WAIT_VM(6); break;
    case 7: MFN_WAIT_VM(7); break;
    case 8: MFN_WAIT_VM(8); break;
    case 9: MFN_WAIT_VM(9); break;
    case 10: MFN_WAIT_VM(10); break;
    case 11: MFN_WAIT_VM(11); break;
    case 12: MFN_WAIT_VM(12); break;
    case 13: MFN_WAIT_VM(13); break;
    case 14: MFN_WAIT_VM(14); break;
    case 15: MFN_WAIT_VM(15); break;
    default: MFN_WAIT_VM(16); break;
  }
}

template <int D, int CK>
__global__ __launch_bounds__(1024) void corr_band_kernel(CorrBandParams p) {
  constexpr int MD = (D - 1) / 2;
  constexpr int MAXI = 8;  // LDS-DMA instructions per wave and stage (the host plan keeps to it)
  MFN_DYN_SHARED(float, lds);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  MFN_STAMP(p.timeline, 0);
  const int NTG = p.WG * 64;                       // threads per group
  const int wave = MFN_UNIFORM(tid >> 6);
  int g = 0, wg = wave;                            // group = WG whole waves (at most 16 waves: a short loop)
  while (wg >= p.WG) { wg -= p.WG; ++g; }
  const int t = tid - g * NTG;                     // thread / task index inside the group
  int n = 0, band = 0;
  corr_band_divmod((int)mfn_xcd_remap(blockIdx.x, gridDim.x), p.bands, p.inv_bands, n, band);
  const int y0 = band * p.R;
  const int H = p.H, W = p.W, C = p.C, R = p.R;
  const int QW = W >> 2;
  const int RS2 = W + 8;                           // staged f2 row: columns -4 .. W+3
  const int ROWS2 = R + 2 * MD;
  const int F1_PER_C = R * W, F2_PER_C = ROWS2 * RS2;
  // a stage = CK channels as two linear images, [CK][R][W] then [CK][ROWS2][RS2], each padded to whole wave
  // transfers (64 items of 16 bytes): LDS-DMA writes lane-linear, so the LDS layout IS the item order
  const int I1 = CK * R * QW, I2 = CK * ROWS2 * (QW + 2);
  const int I1P = (I1 + 63) & ~63, I2P = (I2 + 63) & ~63;
  const int STAGE_F = p.NI * NTG * 4;  // every wave issues exactly NI transfers per stage (uniform vmcnt): >= (I1P+I2P)*4
  float *gbase = lds + (size_t)g * p.S * STAGE_F;  // this group's ring
  const size_t plane = (size_t)H * W;

  // ---- lane task ------------------------------------------------------------------------------------------
  const int T = R * QW * D;
  const bool active = t < T;
  const int tt = active ? t : 0;
  int dyi, rq, r, qx;
  corr_band_divmod(tt, R * QW, p.inv_rq, dyi, rq);
  corr_band_divmod(rq, QW, p.inv_q, r, qx);
  const int f1_off = r * W + 4 * qx;
  const int f2_off = I1P * 4 + (r + dyi) * RS2 + 4 * qx;  // staged column 4*qx = image column 4*qx - 4

  // ---- staging plan: wave transfer i covers items ((i*WG + wg)*64 + lane); byte offset inside the stage's
  // CK-channel slab of image n, or a poisoned offset (outside the image / padding item): the buffer reads zero.
  // Channels past the group's range are cut off by the descriptor's size, rebuilt for every stage.
  unsigned goff[MAXI];
  MFN_UNROLL
  for (int i = 0; i < MAXI; ++i) {
    const int it = (i * p.WG + wg) * 64 + lane;
    goff[i] = 0xFFFFFF00u;
    if (it < I1) {
      int c, rem, rr, q;
      corr_band_divmod(it, R * QW, p.inv_rq, c, rem);
      corr_band_divmod(rem, QW, p.inv_q, rr, q);
      const int y = y0 + rr;
      if (y < H) goff[i] = (unsigned)((c * (int)plane + y * W + 4 * q) * 4);
    } else if (it >= I1P && it - I1P < I2) {
      const int j = it - I1P;
      int c, rem, rr, q;
      corr_band_divmod(j, ROWS2 * (QW + 2), p.inv_r2q2, c, rem);
      corr_band_divmod(rem, QW + 2, p.inv_q2, rr, q);
      const int y = y0 - MD + rr, x = 4 * q - 4;
      if (y >= 0 && y < H && x >= 0 && x < W) goff[i] = (unsigned)((c * (int)plane + y * W + x) * 4);
    }
  }
  const float *f1n = p.f1 + (size_t)n * C * plane;
  const float *f2n = p.f2 + (size_t)n * C * plane;
  const int c_begin = g * p.cpg;
  const int c_end = min(C, c_begin + p.cpg);
  auto issue = [&](int ch) {
    const int c0 = c_begin + ch * CK;
    const int cleft = max(0, min(CK, c_end - c0));
    const unsigned nbytes = (unsigned)((size_t)cleft * plane * 4);
    const mfn_rsrc_t r1 = mfn_make_rsrc(f1n + (size_t)(cleft ? c0 : 0) * plane, nbytes);
    const mfn_rsrc_t r2 = mfn_make_rsrc(f2n + (size_t)(cleft ? c0 : 0) * plane, nbytes);
    float *stage = gbase + (size_t)(ch % p.S) * STAGE_F;
    MFN_UNROLL
    for (int i = 0; i < MAXI; ++i) {
      if (i < p.NI) {
        const int it0 = (i * p.WG + wg) * 64;  // uniform
        if (it0 < I1P) mfn_dma16(r1, stage + (size_t)it0 * 4, goff[i]);
        else mfn_dma16(r2, stage + (size_t)it0 * 4, goff[i]);
      }
    }
  };

  f32x2 accp[D - 1][2];
  float accs[4];
  MFN_UNROLL
  for (int d = 0; d < D - 1; ++d) { accp[d][0] = mfn_f2(0.f, 0.f); accp[d][1] = mfn_f2(0.f, 0.f); }
  MFN_UNROLL
  for (int q = 0; q < 4; ++q) accs[q] = 0.f;

  auto consume = [&](const float *stage) {
    MFN_UNROLL
    for (int c = 0; c < CK; ++c) {
      const float4 a = *reinterpret_cast<const float4 *>(stage + c * F1_PER_C + f1_off);
      const float *b = stage + c * F2_PER_C + f2_off;
      const float4 b0 = *reinterpret_cast<const float4 *>(b);
      const float4 b1 = *reinterpret_cast<const float4 *>(b + 4);
      const float4 b2 = *reinterpret_cast<const float4 *>(b + 8);
      const float bv[12] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w};
      const f32x2 asw[2] = {mfn_f2(a.y, a.x), mfn_f2(a.w, a.z)};
      constexpr int OFF = 4 - MD;  // staged column of displacement 0 for pixel 0
      MFN_UNROLL
      for (int d = 0; d < D - 1; ++d)
        MFN_UNROLL
        for (int h = 0; h < 2; ++h) {
          const float bb = bv[2 * h + 1 + d + OFF];
          accp[d][h] = mfn_fma2(asw[h], mfn_f2(bb, bb), accp[d][h]);
        }
      accs[0] = fmaf(a.x, bv[0 + OFF], accs[0]);
      accs[1] = fmaf(a.z, bv[2 + OFF], accs[1]);
      accs[2] = fmaf(a.y, bv[1 + (D - 1) + OFF], accs[2]);
      accs[3] = fmaf(a.w, bv[3 + (D - 1) + OFF], accs[3]);
    }
  };

  // ---- channel loop: S stages of LDS-DMA in flight per group ---------------------------------------------
  MFN_STAMP(p.timeline, 1);
  for (int ch = 0; ch < p.S - 1 && ch < p.nchunks; ++ch) issue(ch);
  for (int ch = 0; ch < p.nchunks; ++ch) {
    const int ahead = ch + p.S - 1;
    if (ahead < p.nchunks) issue(ahead);  // into the buffer whose readers all passed the barrier below
    // stage ch landed for this wave: only the newer stages' transfers may still be outstanding
    corr_band_wait_vm(min(p.nchunks - 1 - ch, p.S - 1) * p.NI);
    __syncthreads();                      // ... and for every wave of the group
    consume(gbase + (size_t)(ch % p.S) * STAGE_F);
    MFN_WAIT_LGKM0();
    __syncthreads();                      // everyone is done reading before a later issue() overwrites it
  }

  MFN_STAMP(p.timeline, 2);
  // ---- reduction tree over the channel groups (fixed order: deterministic) ----------------------------------
  constexpr int NACC = 4 * (D - 1) + 4;
  float *red = lds;  // [round's upper groups][NACC][NTG]
  int cur = p.G;
  for (int half = p.red_groups; half >= 1; half >>= 1) {
    if (g >= half && g < cur) {
      float *dst = red + ((size_t)(g - half) * NACC) * NTG + t;
      MFN_UNROLL
      for (int d = 0; d < D - 1; ++d)
        MFN_UNROLL
        for (int h = 0; h < 2; ++h) {
          dst[(size_t)(d * 4 + h * 2) * NTG] = accp[d][h].x;
          dst[(size_t)(d * 4 + h * 2 + 1) * NTG] = accp[d][h].y;
        }
      MFN_UNROLL
      for (int q = 0; q < 4; ++q) dst[(size_t)(4 * (D - 1) + q) * NTG] = accs[q];
    }
    __syncthreads();
    if (g < half && g + half < cur) {
      const float *src = red + ((size_t)g * NACC) * NTG + t;
      MFN_UNROLL
      for (int d = 0; d < D - 1; ++d)
        MFN_UNROLL
        for (int h = 0; h < 2; ++h) {
          accp[d][h].x += src[(size_t)(d * 4 + h * 2) * NTG];
          accp[d][h].y += src[(size_t)(d * 4 + h * 2 + 1) * NTG];
        }
      MFN_UNROLL
      for (int q = 0; q < 4; ++q) accs[q] += src[(size_t)(4 * (D - 1) + q) * NTG];
    }
    __syncthreads();
    cur = half;
  }

  // ---- epilogue (group 0): normalise, optional LeakyReLU, 16-byte stores --------------------------------------
  MFN_STAMP(p.timeline, 3);
  const int y = y0 + r;
  if (g != 0 || !active || y >= H) return;
#define ACCB(d, q)                                                                                  \
  (((q) & 1) ? ((d) < D - 1 ? accp[(d) < D - 1 ? (d) : 0][((q) - 1) / 2].x : accs[2 + ((q) - 1) / 2]) \
             : ((d) > 0 ? accp[(d) > 0 ? (d) - 1 : 0][(q) / 2].y : accs[(q) / 2]))
  float *dst = p.out + (size_t)n * p.out_nstride + (size_t)(dyi * D) * plane + (size_t)y * W + 4 * qx;
  const float slope = p.leaky ? 0.1f : 1.f;  // LeakyReLU(0.1)(v) == max(v, 0.1 v)
  MFN_UNROLL
  for (int d = 0; d < D; ++d) {
    float v[4];
    MFN_UNROLL
    for (int q = 0; q < 4; ++q) {
      const float s = ACCB(d, q);
      const float rr = p.exact_div ? s / p.sumelems : s * p.inv_sumelems;
      v[q] = fmaxf(rr, slope * rr);
    }
    mfn_store4_stream(dst + (size_t)d * plane, v[0], v[1], v[2], v[3], p.st_policy);
  }
#undef ACCB
}

// ---- host plan: band height, waves per group and channel groups ----------------------------------------------
struct CorrBandPlan {
  int ok, R, bands, G, WG, cpg, nchunks, S, NI, red_groups;
  size_t lds_bytes;
};
inline CorrBandPlan corr_band_plan(int N, int C, int H, int W, int D, int CK) {
  CorrBandPlan best;
  best.ok = 0;
  if (W % 4 || W < 4) return best;
  const int MD = (D - 1) / 2, QW = W / 4, NACC = 4 * (D - 1) + 4;
  const size_t kLds = 160 * 1024;
  double best_score = 1e30;
  for (int R = 1; R <= H; ++R) {
    const int T = R * QW * D;
    const int WG = (T + 63) / 64;
    if (WG > 8) break;
    const int i1p = (CK * R * QW + 63) / 64 * 64, i2p = (CK * (R + 2 * MD) * (QW + 2) + 63) / 64 * 64;
    const int NI = ((i1p + i2p) / 64 + WG - 1) / WG;
    if (NI > 8) continue;
    const size_t stage = (size_t)NI * WG * 64 * 16;  // whole transfers of every wave
    for (int G = 16 / WG; G >= 1; --G) {
      if (G > (C + CK - 1) / CK) continue;
      int cpg = (C + G - 1) / G;
      cpg = (cpg + CK - 1) / CK * CK;      // whole stages per group ...
      if ((C + cpg - 1) / cpg != G) continue;  // ... and no empty group
      const int nchunks = cpg / CK;
      int S = nchunks < 3 ? nchunks : 3;
      while (S > 1 && (size_t)G * S * stage > kLds) --S;
      if ((size_t)G * S * stage > kLds) continue;
      if (S * NI > 16) continue;           // the vmcnt switch covers 16 outstanding transfers
      int red_groups = 0;
      for (int h = 1; h < G; h <<= 1) red_groups = h;
      const size_t red = (size_t)red_groups * NACC * WG * 64 * sizeof(float);
      if (red > kLds) continue;
      const size_t lds = (size_t)G * S * stage > red ? (size_t)G * S * stage : red;
      const int bands = (H + R - 1) / R;
      const double eff = (double)T / (WG * 64) * (double)H / (bands * R);
      const double waves = (double)N * bands * WG * G;
      const double over = waves > 4096.0 ? waves / 4096.0 : 1.0;  // more than 16 waves per CU: extra rounds
      // A CU pulls only ~12 B/cycle through LDS-DMA (MI355X_MICROARCH.md, "prologue HBM burst"), so what a
      // workgroup must load (f1 band + f2 band with its +-md rows, all channels) sets its duration: small bands
      // on many CUs win even though each re-reads the 2*md halo rows.  Cost in cycles, plus the per-stage
      // hand-off and the lane efficiency of the FMA part.
      const int rows2 = R + 2 * MD < H ? R + 2 * MD : H;
      const double bytes = (double)C * W * 4.0 * (R + rows2);
      const double rounds = (double)((long)N * bands + 255) / 256;
      const double score = (bytes / 12.0 + nchunks * (S > 1 ? 400.0 : 1500.0) + nchunks * CK * 150.0 / eff) *
                           (rounds < 1.0 ? 1.0 : rounds) * over;
      if (score < best_score) {
        best_score = score;
        best.ok = 1; best.R = R; best.bands = bands; best.G = G; best.WG = WG; best.cpg = cpg; best.nchunks = nchunks;
        best.S = S; best.NI = NI; best.red_groups = red_groups; best.lds_bytes = lds;
      }
    }
  }
  return best;
}

template <int D>
inline int corr_band_launch(CorrBandParams p, const CorrBandPlan &pl, hipStream_t stream) {
  p.R = pl.R; p.bands = pl.bands; p.G = pl.G; p.WG = pl.WG; p.cpg = pl.cpg; p.nchunks = pl.nchunks;
  p.red_groups = pl.red_groups; p.S = pl.S; p.NI = pl.NI;
  {
    const int MDh = (D - 1) / 2, QW = p.W / 4;
    p.inv_bands = 1.0f / (float)pl.bands;
    p.inv_rq = 1.0f / (float)(pl.R * QW);
    p.inv_q = 1.0f / (float)QW;
    p.inv_r2q2 = 1.0f / (float)((pl.R + 2 * MDh) * (QW + 2));
    p.inv_q2 = 1.0f / (float)(QW + 2);
  }
  if (p.N * pl.bands <= 0) return 0;
  return launch("corr_band", corr_band_kernel<D, 8>, dim3(p.N * pl.bands), dim3(pl.G * pl.WG * 64), pl.lds_bytes, stream, p);
}

}  // namespace mfn

// ---- corr_direct_kernel: the tiniest levels (level 6: 6x8 pixels, 196 channels) ---------------------------------
// Everything fits in L1/L2 and the job is launch- and latency-bound, so no LDS staging at all: a workgroup owns
// (image, displacement row, band of output rows) and up to 1024 threads = (4-pixel quad, dx) x S channel slices; a
// thread walks its channels with one 16-byte load of f1 and four (clamped, masked) loads of f2, the slices meet in
// LDS in index order (deterministic), slice 0 normalises and stores.  One launch, no workspace.
namespace mfn {
struct CorrDirectParams {
  const float *f1, *f2;
  float *out;
  size_t out_nstride;
  int st_policy;
  int N, C, H, W;
  int R, bands, S, Q;   // rows per band, bands per image, channel slices, outputs (quad, dx, row) per slice
  int cps;              // channels per slice
  int max_slices;       // cap on S: the serial slice sum in the epilogue grows with it
  float inv_sumelems, sumelems;
  int exact_div, leaky;
  float inv_bands, inv_rq, inv_q;  // 1/bands, 1/(R*QW), 1/QW for the division-free index decode
};

template <int D>
__global__ __launch_bounds__(1024) void corr_direct_kernel(CorrDirectParams p) {
  constexpr int MD = (D - 1) / 2;
  MFN_DYN_SHARED(float, red);  // [S-1][Q][4]
  const int tid = threadIdx.x;
  const int H = p.H, W = p.W, C = p.C, QW = W >> 2;
  const size_t plane = (size_t)H * W;
  // block -> (image, displacement row, band)
  int b = (int)mfn_xcd_remap(blockIdx.x, gridDim.x), n, rem, dyi, band;  // the 9 rows of an image on one XCD
  corr_band_divmod(b, D * p.bands, 1.0f / (float)(D * p.bands), n, rem);
  corr_band_divmod(rem, p.bands, p.inv_bands, dyi, band);
  // thread -> (slice, dx, row, quad)
  const int s = tid / p.Q, q = tid - s * p.Q;
  const bool active = s < p.S;
  int dxi, rq, r, qx;
  corr_band_divmod(active ? q : 0, p.R * QW, p.inv_rq, dxi, rq);
  corr_band_divmod(rq, QW, p.inv_q, r, qx);
  const int y = band * p.R + r, x = 4 * qx;
  const int y2 = y + dyi - MD;
  const bool row_ok = active && y < H && y2 >= 0 && y2 < H;
  const int yc = min(y, H - 1), y2c = min(max(y2, 0), H - 1);
  const int xs = x + dxi - MD;  // first f2 column of the quad
  int xi[4];
  bool xo[4];
  MFN_UNROLL
  for (int e = 0; e < 4; ++e) { xo[e] = xs + e >= 0 && xs + e < W; xi[e] = min(max(xs + e, 0), W - 1); }
  const int c0 = s * p.cps, c1 = min(C, c0 + p.cps);
  const float *a = p.f1 + ((size_t)n * C + c0) * plane + (size_t)yc * W + x;
  const float *bp = p.f2 + ((size_t)n * C + c0) * plane + (size_t)y2c * W;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (row_ok) {
    MFN_NOUNROLL
    for (int c = c0; c < c1; c += 4) {  // four channels per trip: 20 unconditional loads in flight (clamped addresses)
      float4 av[4];
      float bv[4][4];
      MFN_UNROLL
      for (int u = 0; u < 4; ++u) {
        const size_t o = (size_t)min(u, c1 - 1 - c) * plane;
        av[u] = *reinterpret_cast<const float4 *>(a + o);
        MFN_UNROLL
        for (int e = 0; e < 4; ++e) bv[u][e] = bp[o + xi[e]];
      }
      MFN_UNROLL
      for (int u = 0; u < 4; ++u) {
        const bool cok = c + u < c1;
        acc[0] = fmaf(cok ? av[u].x : 0.f, xo[0] ? bv[u][0] : 0.f, acc[0]);
        acc[1] = fmaf(cok ? av[u].y : 0.f, xo[1] ? bv[u][1] : 0.f, acc[1]);
        acc[2] = fmaf(cok ? av[u].z : 0.f, xo[2] ? bv[u][2] : 0.f, acc[2]);
        acc[3] = fmaf(cok ? av[u].w : 0.f, xo[3] ? bv[u][3] : 0.f, acc[3]);
      }
      a += 4 * plane;
      bp += 4 * plane;
    }
  }
  if (active && s > 0) {
    float *dst = red + ((size_t)(s - 1) * p.Q + q) * 4;
    dst[0] = acc[0]; dst[1] = acc[1]; dst[2] = acc[2]; dst[3] = acc[3];
  }
  __syncthreads();
  if (!active || s != 0 || y >= H) return;
  for (int s2 = 1; s2 < p.S; ++s2) {
    const float *src = red + ((size_t)(s2 - 1) * p.Q + q) * 4;
    acc[0] += src[0]; acc[1] += src[1]; acc[2] += src[2]; acc[3] += src[3];
  }
  const float slope = p.leaky ? 0.1f : 1.f;
  float v[4];
  MFN_UNROLL
  for (int e = 0; e < 4; ++e) {
    const float rr = p.exact_div ? acc[e] / p.sumelems : acc[e] * p.inv_sumelems;
    v[e] = fmaxf(rr, slope * rr);
  }
  mfn_store4_stream(p.out + (size_t)n * p.out_nstride + (size_t)(dyi * D + dxi) * plane + (size_t)y * W + x, v[0], v[1], v[2],
                    v[3], p.st_policy);
}

// plan: the band height R that gives the most channel slices within 1024 threads (latency is what counts here)
template <int D>
inline int corr_direct_launch(CorrDirectParams p, hipStream_t stream) {
  const int QW = p.W / 4;
  int bestR = 0, bestS = 0;
  for (int R = 1; R <= p.H; ++R) {
    const int Q = R * QW * D;
    if (Q > 1024) break;
    int S = 1024 / Q;
    if (S > (p.C + 3) / 4) S = (p.C + 3) / 4;
    if (S > p.max_slices) S = p.max_slices;
    // prefer many slices (short channel loops), then tall bands (fewer workgroups re-reading rows)
    if (S > bestS || (S == bestS && R > bestR && (long)p.N * D * ((p.H + R - 1) / R) >= 256)) { bestS = S; bestR = R; }
  }
  if (!bestR) return -1;
  p.R = bestR; p.S = bestS; p.Q = bestR * QW * D;
  p.bands = (p.H + p.R - 1) / p.R;
  p.cps = (((p.C + p.S - 1) / p.S) + 3) / 4 * 4;
  p.S = (p.C + p.cps - 1) / p.cps;
  p.inv_bands = 1.0f / (float)p.bands;
  p.inv_rq = 1.0f / (float)(p.R * QW);
  p.inv_q = 1.0f / (float)QW;
  const int threads = ((p.S * p.Q + 63) / 64) * 64;
  const size_t lds = (size_t)(p.S > 1 ? p.S - 1 : 1) * p.Q * 4 * sizeof(float);
  return launch("corr_direct", corr_direct_kernel<D>, dim3(p.N * D * p.bands), dim3(threads), lds, stream, p);
}
}  // namespace mfn
