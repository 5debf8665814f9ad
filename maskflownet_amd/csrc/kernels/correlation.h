// correlation.h -- cost-volume kernels for gfx950.
//
// Replaces MXNet Correlation at /root/reference/network/MaskFlownet.py:193-195 (md=4, 81 ch)
// and :440-441 (md=2, 25 ch); semantics as oracle/mfn_ref_body.inc correlation_fwd.
//
// corr_tiled_kernel<D, TW, NCH, CK>  (kernel_size=1, stride1=stride2=1, pad=md, W%4==0)
//   HBM-bound op (4*N*h*w*(2C + D*D) bytes), close to the fp32 ridge, so the
//   design goal is to touch HBM once and keep VALU + LDS pressure low:
//   * one workgroup = D waves; wave `wv` owns displacement row dy = wv - md, so all D*D outputs
//     of a pixel live in registers of D different waves and never move;
//   * a lane owns NCH chunks of 4 adjacent pixels and the D displacements dx of its dy:
//     NCH*4*D fp32 accumulators, fed per channel by one ds_read_b128 of f1 and three of f2
//     (12 consecutive f2 values serve 4 pixels x up to 9 dx) -> 9 FMA per 16-byte LDS read;
//   * a tile is TW x (256*NCH/TW) pixels; channels stream through LDS in chunks of CK with the
//     next chunk's global loads in flight while the current chunk is consumed (register
//     prefetch); the f2 window carries a +-md row halo and a +-4 column halo so every global
//     and LDS access is an aligned 16-byte vector;
//   * lane -> (row, 4-px group) uses the ds_read_b128 service-group order of gfx950
//     (MI355X_MICROARCH.md LDS table) so the four 16-lane groups of one read hit 16 distinct
//     16-byte bank slots: conflict-free for every TW;
//   * stores are 16 B per lane, 256 B contiguous per tile row.
// corr_generic_kernel: any valid MXNet parameter set, one thread per output element.
#pragma once
#include "../mfn_rt.h"

// measurement builds only (tools/corr_ablate_build.py): bit mask -- 1 no output stores, 2 no global loads, 4 no LDS reads / FMAs
#ifndef MFN_CORR_ABLATE_MASK
#define MFN_CORR_ABLATE_MASK 0
#endif

namespace mfn {

// ds_read_b128 is serviced in four 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, and the
// same +32.  Returns the lane's position in service-group-major order (group g -> 16g..16g+15).
__device__ __forceinline__ int b128_service_pos(int lane) {
  const unsigned long long tbl = 0x1C0C081804141000ull;  // per 4-lane quad: base position
  const int q = (lane >> 2) & 7;
  const int base = (int)((tbl >> (8 * q)) & 0xFFull);
  return base + (lane & 3) + (lane & 32);
}

// component-wise select (a struct-level `ok ? v : zero` makes hipcc build the pair in scratch)
__device__ __forceinline__ float4 zero_unless(bool ok, float4 v) {
  return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
}

template <int TW>
struct CorrGeom {
  static constexpr int GX = TW / 4;     // 4-px groups per tile row
  static constexpr int RPC = 256 / TW;  // tile rows covered by one chunk plane (64 lanes x 4 px)
  // LDS row stride in floats: window is TW+8 columns; padded where the lane map needs it
  static constexpr int RS = (TW == 64) ? 72 : (TW == 32 ? 40 : 24);
  static constexpr int CW4 = (TW + 8) / 4;  // float4 per f2 window row
  // position (service-group-major) -> (row within chunk plane, group within row), chosen so that
  // the 16 lanes of one service group touch 16 distinct 16-byte slots mod 16 with stride RS/4
  __device__ static __forceinline__ void map(int pos, int &row, int &gx) {
    const int sg = pos >> 4, wi = pos & 15;
    if (TW == 64) { row = sg; gx = wi; }
    else if (TW == 32) { row = sg + 4 * (wi >> 3); gx = wi & 7; }
    else if (TW == 16) { row = 2 * (wi >> 2) + (sg & 1) + 8 * (sg >> 1); gx = wi & 3; }
    else { row = 8 * sg + (wi >> 1); gx = wi & 1; }
  }
};

struct CorrParams {
  const float *f1;
  const float *f2;
  float *out;
  int N, C, H, W;
  int tiles_x, tiles_y;
  float inv_sumelems;  // 1/C
  float sumelems;      // C
  int exact_div;       // 1: divide (C not a power of two), 0: multiply by the exact reciprocal
  size_t out_nstride;  // elements between consecutive images of `out` (D*D*H*W when dense; larger = a channel slice
                       // of the decoder's concat buffer, MaskFlownet.py:235)
  int nt_store;        // cache policy of the output stores (mfn_store4_stream; LDS-DMA kernel)
  int xcd_swizzle;     // 1: remap blockIdx so that neighbouring tiles share an XCD's L2
  int leaky;           // fused epilogue (f-1): LeakyReLU(0.1) on the output (MaskFlownet.py:217)
  // channel slicing for levels with few pixels and many channels: blockIdx.y = slice, each slice
  // reduces `slice_channels` channels and writes RAW partial sums to partial + slice*N*D*D*H*W;
  // corr_reduce_kernel then sums the slices in a fixed order and normalises (deterministic).
  int nslices, slice_channels;
  unsigned long long *timeline;  // measurement only: 4 wall-clock stamps (100 MHz) per block, or NULL
  float *partial;
};

// Variant knobs (swept on the GPU, see tools/sweep.py):
//   NCH  4-px chunks per lane            DYW  displacement rows per wave (block = ceil(D/DYW) waves)
//   CK   channels per LDS stage          PF   register prefetch of the next stage's global loads
//   WPE  minimum waves per SIMD the register allocator must leave room for
template <int D, int TW, int NCH, int CK, int DYW, bool PF, int WPE>
__global__ __launch_bounds__(((D + DYW - 1) / DYW) * 64, WPE) void corr_tiled_kernel(CorrParams p) {
  constexpr int MD = (D - 1) / 2;
  constexpr int NW = (D + DYW - 1) / DYW;
  constexpr int NT = NW * 64;
  using G = CorrGeom<TW>;
  constexpr int RS = G::RS;
  constexpr int TH = G::RPC * NCH;
  constexpr int ROWS2 = TH + 2 * MD;
  constexpr int F1_PER_C = TH * RS;
  constexpr int F2_PER_C = ROWS2 * RS;
  constexpr int ITEMS1 = CK * TH * (TW / 4);
  constexpr int ITEMS2 = CK * ROWS2 * G::CW4;
  constexpr int NI1 = (ITEMS1 + NT - 1) / NT;
  constexpr int NI2 = (ITEMS2 + NT - 1) / NT;

  MFN_DYN_SHARED(float, lds);
  float *f1s = lds;                  // [CK][TH][RS]
  float *f2s = lds + CK * F1_PER_C;  // [CK][ROWS2][RS]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  // wave index through readfirstlane: keeps dy0 (and every test on it) in SGPRs / wave-uniform
  const int dy0 = MFN_UNIFORM(tid >> 6) * DYW;  // first displacement row of this wave: dy = dy0 + e - MD
  constexpr bool ALL_ROWS = (D % DYW) == 0;       // no wave has a displacement row past D-1

  // ---- which tile ---------------------------------------------------------------------------
  int bid = blockIdx.x;
  const int nblk = gridDim.x;
  if (p.xcd_swizzle) bid = (int)mfn_xcd_remap((unsigned)bid, (unsigned)nblk);
  const int tiles_per_img = p.tiles_x * p.tiles_y;
  const int n = bid / tiles_per_img;
  const int t = bid - n * tiles_per_img;
  const int ty = t / p.tiles_x;
  const int tx = t - ty * p.tiles_x;
  const int y0 = ty * TH, x0 = tx * TW;
  const int H = p.H, W = p.W, C = p.C;
  const size_t plane = (size_t)H * W;
  const int c_begin = blockIdx.y * p.slice_channels;              // this block's channel slice
  const int c_end = min(C, c_begin + p.slice_channels);
  const float *f1n = p.f1 + (size_t)n * C * plane;
  const float *f2n = p.f2 + (size_t)n * C * plane;

  // ---- lane geometry ------------------------------------------------------------------------
  int row, gx;
  G::map(b128_service_pos(lane), row, gx);
  const int f1_off = row * RS + 4 * gx;          // chunk plane 0; plane k adds k*RPC*RS
  const int f2_off = (row + dy0) * RS + 4 * gx;  // window row = row + dy + MD; row e adds e*RS

  // Accumulators.  VALU issue is what bounds this kernel (a wave64 v_fma_f32 occupies its SIMD for 4
  // cycles, measured), so the 4 px x D dx products of a unit are issued as v_pk_fma_f32: out(d,q) for
  // pixel q and displacement d needs a[q]*b[q+d]; the anti-diagonal pair {(d,q), (d+1,q-1)} shares
  // ONE b value and takes its two a values from one aligned register pair, so every packed FMA is
  // fed by op_sel swizzles alone.  accp[d][h] = {out(d, 2h+1), out(d+1, 2h)}, d < D-1; the four
  // corner products out(0,0), out(0,2), out(D-1,1), out(D-1,3) stay scalar in accs.
  f32x2 accp[DYW][NCH][D - 1][2];
  float accs[DYW][NCH][4];
  MFN_UNROLL
  for (int e = 0; e < DYW; ++e)
    MFN_UNROLL
    for (int k = 0; k < NCH; ++k) {
      MFN_UNROLL
      for (int d = 0; d < D - 1; ++d) { accp[e][k][d][0] = mfn_f2(0.f, 0.f); accp[e][k][d][1] = mfn_f2(0.f, 0.f); }
      MFN_UNROLL
      for (int q = 0; q < 4; ++q) accs[e][k][q] = 0.f;
    }

  // ---- staging plan, computed once: item -> (global element offset inside a CK-channel stage,
  // LDS float offset, channel within the stage).  Spatially invalid items (MXNet's pad_size
  // border, ragged tiles, padding slots of the item grid) get goff = -1 and are zero-filled.
  // Per stage the source is then one uniform base pointer + a 32-bit per-lane offset, so the loads
  // need no 64-bit VALU address math and no branches (a branch per load makes hipcc drain vmcnt
  // between the loads -- cdna_hip_programming.md section 5, trap (c)).
  float4 pre1[NI1], pre2[NI2];
  int goff1[NI1], goff2[NI2], lofc1[NI1], lofc2[NI2];  // lofc = LDS float offset | (channel << 20)
  MFN_UNROLL
  for (int i = 0; i < NI1; ++i) {
    const int it = tid + i * NT;
    const int c = it / (TH * (TW / 4));
    const int rem = it - c * (TH * (TW / 4));
    const int r = rem / (TW / 4);
    const int q = rem - r * (TW / 4);
    const int y = y0 + r, x = x0 + 4 * q;
    const bool ok = (i < NI1 - 1 || ITEMS1 % NT == 0 || it < ITEMS1) && y < H && x < W;
    goff1[i] = ok ? c * (int)plane + y * W + x : -1;
    lofc1[i] = (c * F1_PER_C + r * RS + 4 * q) | (c << 20);
  }
  MFN_UNROLL
  for (int i = 0; i < NI2; ++i) {
    const int it = tid + i * NT;
    const int c = it / (ROWS2 * G::CW4);
    const int rem = it - c * (ROWS2 * G::CW4);
    const int r = rem / G::CW4;
    const int q = rem - r * G::CW4;
    const int y = y0 - MD + r, x = x0 - 4 + 4 * q;
    const bool ok = (i < NI2 - 1 || ITEMS2 % NT == 0 || it < ITEMS2) && y >= 0 && y < H && x >= 0 && x < W;
    goff2[i] = ok ? c * (int)plane + y * W + x : -1;
    lofc2[i] = (c * F2_PER_C + r * RS + 4 * q) | (c << 20);
  }
  const bool loads_on = MFN_CORR_ABLATE_MASK != 2;

  auto fetch = [&](int c0) {
    const float *b1 = f1n + (size_t)c0 * plane;  // wave-uniform
    const float *b2 = f2n + (size_t)c0 * plane;
    const int cleft = c_end - c0;                 // channels of this stage that exist
    MFN_UNROLL
    for (int i = 0; i < NI1; ++i) {
      const bool ok = goff1[i] >= 0 && (lofc1[i] >> 20) < cleft && loads_on;
      pre1[i] = zero_unless(ok, *reinterpret_cast<const float4 *>(b1 + (ok ? goff1[i] : 0)));
    }
    MFN_UNROLL
    for (int i = 0; i < NI2; ++i) {
      const bool ok = goff2[i] >= 0 && (lofc2[i] >> 20) < cleft && loads_on;
      pre2[i] = zero_unless(ok, *reinterpret_cast<const float4 *>(b2 + (ok ? goff2[i] : 0)));
    }
  };
  // registers -> LDS
  auto stash = [&]() {
    MFN_UNROLL
    for (int i = 0; i < NI1; ++i)
      if (i < NI1 - 1 || ITEMS1 % NT == 0 || tid + i * NT < ITEMS1)
        *reinterpret_cast<float4 *>(f1s + (lofc1[i] & 0xFFFFF)) = pre1[i];
    MFN_UNROLL
    for (int i = 0; i < NI2; ++i)
      if (i < NI2 - 1 || ITEMS2 % NT == 0 || tid + i * NT < ITEMS2)
        *reinterpret_cast<float4 *>(f2s + (lofc2[i] & 0xFFFFF)) = pre2[i];
  };
  // One "unit" = (channel c, chunk plane k, displacement row e): 3 ds_read_b128 of f2 (+1 of f1 when
  // e == 0) feeding 4*D FMAs.  Operands are double-buffered in registers by hand and fenced with
  // sched_barrier so the compiler overlaps exactly one unit of LDS latency with one unit of FMAs
  // instead of hoisting a whole stage of reads (which costs 40-60 VGPRs and a wave of occupancy).
  auto consume = [&]() {
    constexpr int NU = CK * NCH * DYW;
    float4 A[2];
    float4 B[2][3];
    auto load_unit = [&](int u) {
      const int e = u % DYW, k = (u / DYW) % NCH, c = u / (DYW * NCH);
      if (e == 0)
        A[(u / DYW) & 1] = *reinterpret_cast<const float4 *>(f1s + c * F1_PER_C + k * G::RPC * RS + f1_off);
      if (ALL_ROWS || dy0 + e < D) {  // wave-uniform
        const float *b = f2s + c * F2_PER_C + (k * G::RPC + e) * RS + f2_off;
        B[u & 1][0] = *reinterpret_cast<const float4 *>(b);
        B[u & 1][1] = *reinterpret_cast<const float4 *>(b + 4);
        B[u & 1][2] = *reinterpret_cast<const float4 *>(b + 8);
      }
    };
    load_unit(0);
    MFN_UNROLL
    for (int u = 0; u < NU; ++u) {
      if (u + 1 < NU) load_unit(u + 1);
      const int e = u % DYW, k = (u / DYW) % NCH;
      if (ALL_ROWS || dy0 + e < D) {
        const float4 a = A[(u / DYW) & 1];
        const float4 b0 = B[u & 1][0], b1 = B[u & 1][1], b2 = B[u & 1][2];
        const float bv[12] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w};
        const f32x2 asw[2] = {mfn_f2(a.y, a.x), mfn_f2(a.w, a.z)};
        constexpr int OFF = 4 - MD;  // window column of displacement 0 for pixel 0
        MFN_UNROLL
        for (int d = 0; d < D - 1; ++d)
          MFN_UNROLL
          for (int h = 0; h < 2; ++h) {
            const float bb = bv[2 * h + 1 + d + OFF];
            accp[e][k][d][h] = mfn_fma2(asw[h], mfn_f2(bb, bb), accp[e][k][d][h]);
          }
        accs[e][k][0] = fmaf(a.x, bv[0 + OFF], accs[e][k][0]);
        accs[e][k][1] = fmaf(a.z, bv[2 + OFF], accs[e][k][1]);
        accs[e][k][2] = fmaf(a.y, bv[1 + (D - 1) + OFF], accs[e][k][2]);
        accs[e][k][3] = fmaf(a.w, bv[3 + (D - 1) + OFF], accs[e][k][3]);
      }
      MFN_SCHED_BARRIER();
    }
  };

  const int nchunks = (c_end - c_begin + CK - 1) / CK;
  if (PF) {
    fetch(c_begin);
    for (int ch = 0; ch < nchunks; ++ch) {
      stash();
      __syncthreads();
      if (ch + 1 < nchunks) fetch(c_begin + (ch + 1) * CK);  // in flight while this stage is consumed
      consume();
      __syncthreads();  // everyone is done reading before the next stash overwrites
    }
  } else {
    for (int ch = 0; ch < nchunks; ++ch) {
      fetch(c_begin + ch * CK);
      if (ch) __syncthreads();  // previous stage fully consumed
      stash();
      __syncthreads();
      consume();
    }
  }

  // ---- epilogue: normalise, optional LeakyReLU, 16-byte stores -----------------------------------
  // The mode is decided once (uniform), the per-element work is branch-free:
  //   raw partial sums (channel slices)  -> scale 1, no activation
  //   C a power of two                   -> multiply by the exact reciprocal
  //   otherwise                          -> true division, as CorrelationForward does
  // out(d,q) from the pair layout (all indices are compile-time after unrolling)
#define ACC(e, k, d, q)                                                                                   \
  (((q) & 1) ? ((d) < D - 1 ? accp[e][k][(d) < D - 1 ? (d) : 0][((q) - 1) / 2].x : accs[e][k][2 + ((q) - 1) / 2]) \
             : ((d) > 0 ? accp[e][k][(d) > 0 ? (d) - 1 : 0][(q) / 2].y : accs[e][k][(q) / 2]))
  const bool raw = p.nslices > 1;
  float *outn = raw ? p.partial + ((size_t)blockIdx.y * p.N + n) * (D * D) * plane : p.out + (size_t)n * p.out_nstride;
  const bool use_div = p.exact_div && !raw;
  const float scale = raw ? 1.f : p.inv_sumelems;
  const float slope = (p.leaky && !raw) ? 0.1f : 1.f;  // LeakyReLU(0.1)(r) == max(r, 0.1 r)
  MFN_UNROLL
  for (int e = 0; e < DYW; ++e) {
    if (ALL_ROWS || dy0 + e < D) {
      MFN_UNROLL
      for (int k = 0; k < NCH; ++k) {
        const int y = y0 + k * G::RPC + row;
        const int x = x0 + 4 * gx;
        if (y < H && x < W) {
          float *dst = outn + (size_t)((dy0 + e) * D) * plane + (size_t)y * W + x;
          if (use_div) {
            MFN_UNROLL
            for (int d = 0; d < D; ++d) {
              float v[4];
              MFN_UNROLL
              for (int q = 0; q < 4; ++q) { const float r = ACC(e, k, d, q) / p.sumelems; v[q] = fmaxf(r, slope * r); }
              *reinterpret_cast<float4 *>(dst + (size_t)d * plane) = make_float4(v[0], v[1], v[2], v[3]);
            }
          } else {
            MFN_UNROLL
            for (int d = 0; d < D; ++d) {
              float v[4];
              MFN_UNROLL
              for (int q = 0; q < 4; ++q) { const float r = ACC(e, k, d, q) * scale; v[q] = fmaxf(r, slope * r); }
              if (MFN_CORR_ABLATE_MASK != 1 || v[0] != v[0])  // ablation keeps the value live but never stores
                *reinterpret_cast<float4 *>(dst + (size_t)d * plane) = make_float4(v[0], v[1], v[2], v[3]);
            }
          }
        }
      }
    }
  }
#undef ACC
}

template <int D, int TW, int NCH, int CK>
inline size_t corr_tiled_lds_bytes() {
  constexpr int MD = (D - 1) / 2;
  using G = CorrGeom<TW>;
  constexpr int TH = G::RPC * NCH;
  return (size_t)CK * (TH * G::RS + (TH + 2 * MD) * G::RS) * sizeof(float);
}

template <int D, int TW, int NCH, int CK, int DYW, bool PF, int WPE>
inline int corr_tiled_launch(CorrParams p, hipStream_t stream, const char *name) {
  using G = CorrGeom<TW>;
  constexpr int TH = G::RPC * NCH;
  constexpr int NW = (D + DYW - 1) / DYW;
  p.tiles_x = cdiv(p.W, TW);
  p.tiles_y = cdiv(p.H, TH);
  const int nblk = p.N * p.tiles_x * p.tiles_y;
  if (nblk <= 0) return 0;
  return launch(name, corr_tiled_kernel<D, TW, NCH, CK, DYW, PF, WPE>, dim3(nblk, p.nslices), dim3(NW * 64),
                corr_tiled_lds_bytes<D, TW, NCH, CK>(), stream, p);
}

// The one point of (NCH, CK, DYW, PF, WPE) that is still built: one displacement row per wave, one 4-px chunk per lane,
// 4-channel stages, no register prefetch, >= 4 waves per SIMD -- the kernel of images narrower than 32 columns
// (corr.variant 6).  Rounds 1 / 2 swept eight points and a half-wave form (corr_hw_kernel); none of them is selected by a plan.
constexpr int kCorrVariants = 48;  // valid values of corr.variant: 6 (corr_tiled_kernel), 16 / 20 / 22 / 26 / 31 (corr_dma_kernel), 40 / 41 / 42 / 43 (corr_gram_kernel), 44 / 45 (corr_gramk_kernel)
inline bool corr_variant_known(int v) { return v == 6 || v == 16 || v == 20 || v == 22 || v == 26 || v == 31 || (v >= 40 && v <= 48); }
template <int D, int TW>
inline int corr_tiled_variant(const CorrParams &p, int /*variant*/, hipStream_t s) {
  return corr_tiled_launch<D, TW, 1, 4, 1, false, 4>(p, s, "corr_tiled_v6");
}
inline int corr_variant_tile_h(int tw, int variant) { return variant >= 8 ? 4 : (256 / tw); }

#ifndef MFN_CORR_ABLATE
#define MFN_CORR_ABLATE 0  // tools/corr_ablate_build.py: 1 no LDS operand reads, 2 no FMAs (single-buffered consume only)
#endif
// One stage of the half-wave kernels: CK channels, operands double-buffered in registers so the
// ds_read_b128 of channel c+1 are in flight while channel c's 16 v_pk_fma_f32 + 4 v_fma_f32 issue
// (measured: without this a lone wave spends ~300 cycles per channel, 80 of them issuing VALU).
template <int D, int CK, int F1_PER_C, int F2_PER_C, bool DBUF = true>
__device__ __forceinline__ void corr_hw_consume(const float *f1p, const float *f2p, f32x2 (&accp)[D - 1][2],
                                                float (&accs)[4]) {
  constexpr int MD = (D - 1) / 2, OFF = 4 - MD;
  // one channel: pairs {out(d,2h+1), out(d+1,2h)} += {a[2h+1], a[2h]} * b[2h+1+d+OFF]; the four corner products stay scalar
  auto channel = [&](f32x4v a, f32x4v b0, f32x4v b1, f32x4v b2) {
    const f32x2 ap[2] = {mfn_lo2(a), mfn_hi2(a)};
    const f32x2 bp[6] = {mfn_lo2(b0), mfn_hi2(b0), mfn_lo2(b1), mfn_hi2(b1), mfn_lo2(b2), mfn_hi2(b2)};
    const float av[4] = {a.x, a.y, a.z, a.w};
    const float bv[12] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w};
    mfn_fmac_inorder(accs[0], av[0], bv[0 + OFF]);   // operands in read order: b0 first, b2 last
    mfn_fmac_inorder(accs[1], av[2], bv[2 + OFF]);
    MFN_UNROLL
    for (int d = 0; d < D - 1; ++d)
      MFN_UNROLL
      for (int h = 0; h < 2; ++h) {
        const int i = 2 * h + 1 + d + OFF;
        if (i & 1) mfn_pk_fma_swbc<true>(accp[d][h], ap[h], bp[i >> 1]);
        else mfn_pk_fma_swbc<false>(accp[d][h], ap[h], bp[i >> 1]);
      }
    mfn_fmac_inorder(accs[2], av[1], bv[1 + (D - 1) + OFF]);
    mfn_fmac_inorder(accs[3], av[3], bv[3 + (D - 1) + OFF]);
  };
  if (!DBUF) {  // single operand set: ~16 fewer VGPRs (one more resident block per CU); LDS latency is hidden by TLP
    MFN_UNROLL
    for (int c = 0; c < CK; ++c) {
#if MFN_CORR_ABLATE & 1  // measurement build: operands without LDS traffic (loop-carried, so nothing folds)
      f32x4v a; a.x = accs[0]; a.y = accs[1]; a.z = accs[2]; a.w = accs[3];
      const f32x4v b0 = a, b1 = a, b2 = a;
#else
      const f32x4v a = mfn_lds_read4(f1p + c * F1_PER_C);
      const f32x4v b0 = mfn_lds_read4(f2p + c * F2_PER_C);
      const f32x4v b1 = mfn_lds_read4(f2p + c * F2_PER_C + 4);
      const f32x4v b2 = mfn_lds_read4(f2p + c * F2_PER_C + 8);
#endif
#if MFN_CORR_ABLATE & 2  // measurement build: the four reads stay live, the 36 FMAs are gone
      accs[0] += a.x + b0.x;
      accs[1] += b1.y + b2.z;
      MFN_SCHED_BARRIER();
      continue;
#endif
      channel(a, b0, b1, b2);
      MFN_SCHED_BARRIER();
    }
    return;
  }
  f32x4v A[2], B[2][3];
  A[0] = mfn_lds_read4(f1p);
  B[0][0] = mfn_lds_read4(f2p);
  B[0][1] = mfn_lds_read4(f2p + 4);
  B[0][2] = mfn_lds_read4(f2p + 8);
  MFN_UNROLL
  for (int c = 0; c < CK; ++c) {
    if (c + 1 < CK) {
      A[(c + 1) & 1] = mfn_lds_read4(f1p + (c + 1) * F1_PER_C);
      B[(c + 1) & 1][0] = mfn_lds_read4(f2p + (c + 1) * F2_PER_C);
      B[(c + 1) & 1][1] = mfn_lds_read4(f2p + (c + 1) * F2_PER_C + 4);
      B[(c + 1) & 1][2] = mfn_lds_read4(f2p + (c + 1) * F2_PER_C + 8);
    }
    channel(A[c & 1], B[c & 1][0], B[c & 1][1], B[c & 1][2]);
    MFN_SCHED_BARRIER();
  }
}

// ---- corr_dma_kernel: 32x4-pixel tiles, two displacement rows per wave (one per half-wave), LDS-DMA staging ring ----
//   * tile = 32 x 4 px (128 px): lanes 0-31 own the 32 four-pixel groups for displacement row 2*wave, lanes 32-63 the same
//     groups for row 2*wave+1 -> ceil(D/2) waves per block (5 for D=9); 768 blocks for the level-2 shape = 3 per CU, all resident;
//   * LDS row stride 48 floats: the two rows {r, r+2} of one ds_read_b128 service group land on 16 distinct 16-byte slots
//     (2*12 = 24 = 8 mod 16) -> conflict-free.
// The register-staged kernels pay one full global-load latency per channel stage (measured: ~1.2-1.5
// us per stage, 8-16 stages per block).  Here the loads of NS-1 stages are always in flight:
//   * a stage = CK channels of the f1 tile (4 x 48-float rows) and the f2 window (12 rows), laid
//     out exactly in item order, so buffer_load_dwordx4 ... lds (16 B per lane, 1 KB per wave
//     instruction, no VGPRs) fills it; padding columns / the pad_size border / ragged tiles /
//     missing tail channels are items whose byte offset is out of the descriptor's range -> the
//     hardware writes zeros;
//   * per stage: counted s_waitcnt vmcnt, ONE raw s_barrier, issue stage ch+NS-1, consume stage ch.
//   * G > 1: G channel groups inside the block (each its own ring over 1/G of the channels), accumulators added
//     through LDS at the end -- for levels whose tile count leaves CUs idle (level 3: 192 tiles), no workspace and
//     no second launch.
//   * NWB < ceil(D/2): the displacement rows of a tile are split over blockIdx.z -- a block runs NWB of the ceil(D/2)
//     row-pair waves (per channel group) and stages only the TH + 2*NWB - 1 window rows they read.  For the coarse levels
//     (48 / 192 tiles on 256 CUs) the work of a tile is what ONE CU can pull through its LDS and VALU; split five ways it
//     runs on five CUs, at the price of staging the f1 tile five times.
template <int D, int CK, int NS, int WPE, bool DBUF, int G = 1, int NWB = (D + 1) / 2>
__global__ __launch_bounds__(NWB * 64 * G, WPE) void corr_dma_kernel(CorrParams p) {
  constexpr int MD = (D - 1) / 2;
  constexpr int NW = NWB;                   // row-pair waves of this block (per channel group)
  constexpr int NT = NW * 64;
  constexpr int TH = 4, RS = 48, R4 = 12;  // 12 float4 per LDS row (f1 uses 8, f2 uses 10)
  constexpr int ROWS2 = (TH + 2 * NWB - 1) < (TH + 2 * MD) ? (TH + 2 * NWB - 1) : (TH + 2 * MD);
  constexpr int F1_PER_C = TH * RS, F2_PER_C = ROWS2 * RS;
  constexpr int ITEMS1 = CK * TH * R4;     // multiple of 64 for CK % 4 == 0 (CK*48)
  constexpr int ITEMS2 = CK * ROWS2 * R4;
  constexpr int ITEMS = ITEMS1 + ITEMS2;
  constexpr int NI = (ITEMS + NT - 1) / NT;  // DMA instructions per thread per stage
  constexpr int STAGE_F = NI * NT * 4;       // floats per stage slot (tail = write-only padding)
  constexpr int OFF = 4 - MD;
  static_assert(CK % 4 == 0 && (ITEMS1 % 64) == 0, "f1/f2 split must fall on a wave boundary");

  MFN_DYN_SHARED(float, lds_all);
  const int lane = threadIdx.x & 63;
  const int wave_all = MFN_UNIFORM(threadIdx.x >> 6);
  const int grp = G > 1 ? wave_all / NW : 0;       // channel group of this wave
  const int wave = wave_all - grp * NW;            // wave inside the group
  const int tid = wave * 64 + lane;                // thread inside the group
  float *lds = lds_all + (size_t)grp * NS * STAGE_F;
  MFN_STAMP(p.timeline, 0);
  const int dy_first = (int)blockIdx.z * (2 * NWB);   // first displacement row of this block
  const int dyl = wave * 2 + (lane >> 5);             // displacement row inside the block's window
  const int dyi = dy_first + dyl;
  const bool live = dyi < D;

  int bid = blockIdx.x;
  const int nblk = gridDim.x;
  if (p.xcd_swizzle) bid = (int)mfn_xcd_remap((unsigned)bid, (unsigned)nblk);
  const int tiles_per_img = p.tiles_x * p.tiles_y;
  const int n = bid / tiles_per_img;
  const int t = bid - n * tiles_per_img;
  const int ty = t / p.tiles_x;
  const int tx = t - ty * p.tiles_x;
  const int y0 = ty * TH, x0 = tx * 32;
  const int H = p.H, W = p.W, C = p.C;
  const size_t plane = (size_t)H * W;
  const int s_begin = blockIdx.y * p.slice_channels;
  const int s_end = min(C, s_begin + p.slice_channels);
  // every group walks the same number of CK-channel stages (block-wide barriers); a short last group reads zeros
  const int gch = G > 1 ? ((s_end - s_begin + G * CK - 1) / (G * CK)) * CK : s_end - s_begin;
  const int c_begin = s_begin + grp * gch;
  const int c_end = min(s_end, c_begin + gch);
  const float *f1n = p.f1 + (size_t)n * C * plane;
  const float *f2n = p.f2 + (size_t)n * C * plane;

  const int pos = b128_service_pos(lane) & 31;
  const int sg = pos >> 4, wi = pos & 15;
  const int row = sg + 2 * (wi >> 3), gx = wi & 7;
  const int f1_off = row * RS + 4 * gx;
  const int f2_off = ITEMS1 * 4 + (row + (live ? dyl : 0)) * RS + 4 * gx;

  f32x2 accp[D - 1][2];
  float accs[4];
  MFN_UNROLL
  for (int d = 0; d < D - 1; ++d) { accp[d][0] = mfn_f2(0.f, 0.f); accp[d][1] = mfn_f2(0.f, 0.f); }
  MFN_UNROLL
  for (int q = 0; q < 4; ++q) accs[q] = 0.f;

  // per-thread byte offsets of its NI items inside a stage's channel group (stage-invariant).  A wave instruction
  // lies entirely in the f1 or the f2 part (ITEMS1 % 64 == 0), so that choice is a scalar branch and the rest is
  // straight-line select code (the prologue runs on all 15 waves of a CU at once: every 100 VALU instructions
  // here cost ~0.6 us of kernel time).
  constexpr unsigned INVALID = 0xFFFFFF00u;
  unsigned voff[NI];
  MFN_UNROLL
  for (int i = 0; i < NI; ++i) {
    const int first = (i * NW + wave) * 64;  // uniform
    const int it = first + lane;
    int c, r, q, ybase, xbase, qmax;
    if (first < ITEMS1) {
      c = it / (TH * R4);
      const int rem = it - c * (TH * R4);
      r = rem / R4; q = rem - r * R4;
      ybase = y0; xbase = x0; qmax = 8;
    } else {
      const int it2 = it - ITEMS1;
      c = it2 / (ROWS2 * R4);
      const int rem = it2 - c * (ROWS2 * R4);
      r = rem / R4; q = rem - r * R4;
      ybase = y0 - MD + dy_first; xbase = x0 - 4; qmax = 10;
    }
    const int y = ybase + r, x = xbase + 4 * q;
    const bool ok = first < ITEMS && q < qmax && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
    const unsigned v = (unsigned)(c * (int)plane + y * W + x) * 4u;
    voff[i] = (ok && !(MFN_CORR_ABLATE_MASK & 2)) ? v : INVALID;
  }

  auto issue = [&](int ch) {  // stage ch -> ring slot ch % NS
    const int c0 = c_begin + ch * CK;
    const int cleft = max(0, min(CK, c_end - c0));
    const unsigned nrec = (unsigned)((size_t)cleft * plane * 4);
    const int cb = cleft ? c0 : 0;  // an empty stage (short last group) reads nothing: every lane out of range
    const mfn_rsrc_t r1 = mfn_make_rsrc(f1n + (size_t)cb * plane, nrec);
    const mfn_rsrc_t r2 = mfn_make_rsrc(f2n + (size_t)cb * plane, nrec);
    float *slot = lds + (ch % NS) * STAGE_F;
    MFN_UNROLL
    for (int i = 0; i < NI; ++i) {
      const int first = (i * NW + wave) * 64;  // first item of this wave instruction (uniform)
      mfn_dma16(first < ITEMS1 ? r1 : r2, slot + first * 4, voff[i]);
    }
  };
  auto consume = [&](int ch) {
    const float *slot = lds + (ch % NS) * STAGE_F;
    corr_hw_consume<D, CK, F1_PER_C, F2_PER_C, DBUF>(slot + f1_off, slot + f2_off, accp, accs);
  };

  const int nchunks = G > 1 ? gch / CK : (c_end - c_begin + CK - 1) / CK;
  MFN_UNROLL
  for (int s0 = 0; s0 < NS - 1; ++s0)
    if (s0 < nchunks) issue(s0);
  for (int ch = 0; ch < nchunks; ++ch) {
    // stages issued so far: 0 .. min(nchunks, ch+NS-1)-1; stage ch must have landed
    const int after = min(nchunks - 1, ch + NS - 2) - ch;  // stages issued after stage ch (uniform)
    if (NS >= 4 && after >= 2) MFN_WAIT_VM(2 * NI);
    else if (after >= 1) MFN_WAIT_VM(NI);
    else MFN_WAIT_VM(0);
    MFN_WAIT_LGKM0();      // this wave's reads of the previous stage are complete
    MFN_RAW_BARRIER();     // stage ch landed for every wave; slot (ch-1)%NS is free
    if (ch == 0) MFN_STAMP(p.timeline, 1);
    if (ch + NS - 1 < nchunks) issue(ch + NS - 1);
    if (!(MFN_CORR_ABLATE_MASK & 4)) consume(ch);
  }
  MFN_STAMP(p.timeline, 2);
  if (G > 1) {  // add the groups' accumulators in index order (deterministic); group 0 owns the epilogue
    constexpr int NACC = 4 * (D - 1) + 4;
    float *red = lds_all;  // [G-1][NACC][NT]
    MFN_WAIT_LGKM0();
    __syncthreads();       // every ring is dead
    if (grp > 0) {
      float *dst = red + (size_t)(grp - 1) * NACC * NT + tid;
      MFN_UNROLL
      for (int d = 0; d < D - 1; ++d)
        MFN_UNROLL
        for (int h = 0; h < 2; ++h) { dst[(d * 4 + h * 2) * NT] = accp[d][h].x; dst[(d * 4 + h * 2 + 1) * NT] = accp[d][h].y; }
      MFN_UNROLL
      for (int q = 0; q < 4; ++q) dst[(4 * (D - 1) + q) * NT] = accs[q];
    }
    __syncthreads();
    if (grp > 0) return;
    for (int g2 = 1; g2 < G; ++g2) {
      const float *src = red + (size_t)(g2 - 1) * NACC * NT + tid;
      MFN_UNROLL
      for (int d = 0; d < D - 1; ++d)
        MFN_UNROLL
        for (int h = 0; h < 2; ++h) { accp[d][h].x += src[(d * 4 + h * 2) * NT]; accp[d][h].y += src[(d * 4 + h * 2 + 1) * NT]; }
      MFN_UNROLL
      for (int q = 0; q < 4; ++q) accs[q] += src[(4 * (D - 1) + q) * NT];
    }
  }

#define ACC1(d, q)                                                                        \
  (((q) & 1) ? ((d) < D - 1 ? accp[(d) < D - 1 ? (d) : 0][((q) - 1) / 2].x : accs[2 + ((q) - 1) / 2]) \
             : ((d) > 0 ? accp[(d) > 0 ? (d) - 1 : 0][(q) / 2].y : accs[(q) / 2]))
  const bool raw = p.nslices > 1;
  float *outn = raw ? p.partial + ((size_t)blockIdx.y * p.N + n) * (D * D) * plane : p.out + (size_t)n * p.out_nstride;
  const bool use_div = p.exact_div && !raw;
  const bool leaky = p.leaky && !raw;
  const float scale = raw ? 1.f : p.inv_sumelems;
  const int y = y0 + row, x = x0 + 4 * gx;
  // the normalisation / activation choice is uniform: one straight-line copy of the 9 stores per case instead of
  // a branch per element (the epilogue, like the prologue, runs on every wave of the CU at the same time)
  if (live && y < H && x < W && !(MFN_CORR_ABLATE_MASK & 1)) {
    float *dst = outn + (size_t)(dyi * D) * plane + (size_t)y * W + x;
    // POL: the store policy as a constant for the two the plans use (2 = written through for >= 4 MB outputs, 0 = plain);
    // -1 = whatever the store.policy key asked for, decided per store
    auto emit = [&](auto div_c, auto leaky_c, auto pol_c) {
      constexpr bool DIV = decltype(div_c)::value, LEAKY = decltype(leaky_c)::value;
      constexpr int POL = decltype(pol_c)::value;
      if (!DIV) {  // packed multiplies on the accumulator pairs
        const f32x2 s2 = mfn_f2(scale, scale);
        MFN_UNROLL
        for (int d = 0; d < D - 1; ++d) { accp[d][0] = mfn_mul2(accp[d][0], s2); accp[d][1] = mfn_mul2(accp[d][1], s2); }
        MFN_UNROLL
        for (int q = 0; q < 4; ++q) accs[q] *= scale;
        if (LEAKY) {  // LeakyReLU(0.1)(r) = max(r, 0.1 r): the products packed as well
          const f32x2 k2 = mfn_f2(0.1f, 0.1f);
          MFN_UNROLL
          for (int d = 0; d < D - 1; ++d)
            MFN_UNROLL
            for (int h = 0; h < 2; ++h) {
              const f32x2 t = mfn_mul2(accp[d][h], k2);
              accp[d][h] = mfn_f2(fmaxf(accp[d][h].x, t.x), fmaxf(accp[d][h].y, t.y));
            }
          MFN_UNROLL
          for (int q = 0; q < 4; ++q) accs[q] = fmaxf(accs[q], 0.1f * accs[q]);
        }
      }
      MFN_UNROLL
      for (int d = 0; d < D; ++d) {
        float v[4];
        MFN_UNROLL
        for (int q = 0; q < 4; ++q) {
          const float r = DIV ? ACC1(d, q) / p.sumelems : ACC1(d, q);
          v[q] = (DIV && LEAKY) ? fmaxf(r, 0.1f * r) : r;
        }
        mfn_store4_stream(dst + (size_t)d * plane, v[0], v[1], v[2], v[3], POL < 0 ? p.nt_store : POL);
      }
    };
    using T_ = std::integral_constant<bool, true>;
    using F_ = std::integral_constant<bool, false>;
    auto emit_pol = [&](auto pol_c) {
      if (use_div) { if (leaky) emit(T_{}, T_{}, pol_c); else emit(T_{}, F_{}, pol_c); }
      else { if (leaky) emit(F_{}, T_{}, pol_c); else emit(F_{}, F_{}, pol_c); }
    };
    if (p.nt_store == 2) emit_pol(std::integral_constant<int, 2>{});
    else if (p.nt_store == 0) emit_pol(std::integral_constant<int, 0>{});
    else emit_pol(std::integral_constant<int, -1>{});
  }
  MFN_STAMP(p.timeline, 3);
#undef ACC1
}

template <int D, int CK, int NS, int WPE, bool DBUF, int G = 1, int NWB = (D + 1) / 2>
inline int corr_dma_launch(CorrParams p, hipStream_t stream, const char *name) {
  constexpr int MD = (D - 1) / 2;
  constexpr int NT = NWB * 64;
  constexpr int ROWS2 = (4 + 2 * NWB - 1) < (4 + 2 * MD) ? (4 + 2 * NWB - 1) : (4 + 2 * MD);
  constexpr int ITEMS = CK * (4 + ROWS2) * 12;
  constexpr int NI = (ITEMS + NT - 1) / NT;
  constexpr int DS = ((D + 1) / 2 + NWB - 1) / NWB;   // blocks per tile along the displacement rows
  p.tiles_x = cdiv(p.W, 32);
  p.tiles_y = cdiv(p.H, 4);
  const int nblk = p.N * p.tiles_x * p.tiles_y;
  if (nblk <= 0) return 0;
  const size_t ring = (size_t)G * NS * NI * NT * 16;
  const size_t red = (size_t)(G - 1) * (4 * (D - 1) + 4) * NT * sizeof(float);
  return launch(name, corr_dma_kernel<D, CK, NS, WPE, DBUF, G, NWB>, dim3(nblk, p.nslices, DS), dim3(NT * G),
                ring > red ? ring : red, stream, p);
}
// corr.variant 16 / 20 / 22: (CK, ring stages, min waves/SIMD, double-buffered operands, channel groups) as the plans pick
// them (api_impl.inc corr_plan: >= 400 tiles / 128-399 / fewer).  Rounds 1 / 2 measured sixteen more points (deeper rings,
// 8- and 16-channel stages, operand double buffering at level 2, staggered block starts): DESIGN.md 4.1.
template <int D>
inline int corr_dma_variant(const CorrParams &p, int variant, hipStream_t s) {
  switch (variant) {
    case 20: return corr_dma_launch<D, 8, 2, 2, true, 2>(p, s, "corr_dma_v20");   // two channel groups
    case 22: return corr_dma_launch<D, 8, 2, 1, true, 3>(p, s, "corr_dma_v22");   // three channel groups (86..127 tiles: one 15-wave block per tile)
    case 26: return corr_dma_launch<D, 4, 2, 2, true, 4, 1>(p, s, "corr_dma_v26");  // one row pair per block (blockIdx.z), 4 channel groups
    case 31: return corr_dma_launch<D, 4, 2, 2, true, 2, 2>(p, s, "corr_dma_v31");  // two row pairs per block, 2 channel groups
    default: return corr_dma_launch<D, 4, 2, 5, false>(p, s, "corr_dma_v16");     // level 2: every block of the launch resident
  }
}

// ---- slice reduction: out = (sum_s partial[s]) / C, slices summed in index order ------------------
struct CorrReduceParams { const float *partial; float *out; size_t n4; int nslices; float inv, sumelems; int exact_div, leaky;
                          size_t img4, out_nstride4; };  // float4 per image in the partials / between images of `out`
__global__ __launch_bounds__(256) void corr_reduce_kernel(CorrReduceParams p) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= p.n4) return;
  const float4 *src = reinterpret_cast<const float4 *>(p.partial);
  float4 s = src[i];
  for (int k = 1; k < p.nslices; ++k) {
    const float4 v = src[(size_t)k * p.n4 + i];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  float r[4] = {s.x, s.y, s.z, s.w};
  MFN_UNROLL
  for (int q = 0; q < 4; ++q) {
    r[q] = p.exact_div ? r[q] / p.sumelems : r[q] * p.inv;
    if (p.leaky) r[q] = r[q] > 0.f ? r[q] : 0.1f * r[q];
  }
  size_t o = i;
  if (p.out_nstride4 != p.img4) { const size_t n = i / p.img4; o = n * p.out_nstride4 + (i - n * p.img4); }
  reinterpret_cast<float4 *>(p.out)[o] = make_float4(r[0], r[1], r[2], r[3]);
}
inline int corr_reduce_launch(CorrReduceParams p, hipStream_t stream) {
  if (!p.n4) return 0;
  return launch("corr_reduce", corr_reduce_kernel, dim3((unsigned)((p.n4 + 255) / 256)), dim3(256), 0, stream, p);
}

// ---- generic fallback: any MXNet-valid parameter set ----------------------------------------------
struct CorrGenericParams {
  const float *f1;
  const float *f2;
  float *out;
  int N, C, H, W;
  int md, kernel, stride1, stride2, pad, is_multiply;
  int top_c, top_h, top_w, radius, gw;
  int leaky;  // fused LeakyReLU(0.1)
  size_t out_nstride;
};

__global__ __launch_bounds__(256) void corr_generic_kernel(CorrGenericParams p) {
  const size_t total = (size_t)p.N * p.top_c * p.top_h * p.top_w;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int j = (int)(idx % p.top_w);
  const int i = (int)((idx / p.top_w) % p.top_h);
  const int tc = (int)((idx / ((size_t)p.top_w * p.top_h)) % p.top_c);
  const int n = (int)(idx / ((size_t)p.top_w * p.top_h * p.top_c));
  // padded coordinates as in CorrelationForward; unpadded = padded - pad, zero outside
  const int x1 = j * p.stride1 + p.md, y1 = i * p.stride1 + p.md;
  const int x2 = x1 + (tc % p.gw - p.radius) * p.stride2;
  const int y2 = y1 + (tc / p.gw - p.radius) * p.stride2;
  const size_t plane = (size_t)p.H * p.W;
  const float *a = p.f1 + (size_t)n * p.C * plane;
  const float *b = p.f2 + (size_t)n * p.C * plane;
  float s = 0.f;
  for (int h = 0; h < p.kernel; ++h)
    for (int w = 0; w < p.kernel; ++w) {
      const int ya = y1 + h - p.pad, xa = x1 + w - p.pad;
      const int yb = y2 + h - p.pad, xb = x2 + w - p.pad;
      const bool ina = ya >= 0 && ya < p.H && xa >= 0 && xa < p.W;
      const bool inb = yb >= 0 && yb < p.H && xb >= 0 && xb < p.W;
      if (p.is_multiply) {
        if (ina && inb)
          for (int c = 0; c < p.C; ++c)
            s = fmaf(a[c * plane + (size_t)ya * p.W + xa], b[c * plane + (size_t)yb * p.W + xb], s);
      } else {
        for (int c = 0; c < p.C; ++c) {
          const float va = ina ? a[c * plane + (size_t)ya * p.W + xa] : 0.f;
          const float vb = inb ? b[c * plane + (size_t)yb * p.W + xb] : 0.f;
          s += fabsf(va - vb);
        }
      }
    }
  const float r = s / (float)(p.kernel * p.kernel * p.C);
  p.out[(size_t)n * p.out_nstride + ((size_t)tc * p.top_h + i) * p.top_w + j] = p.leaky ? fmaxf(r, 0.1f * r) : r;
}

inline int corr_generic_launch(CorrGenericParams p, hipStream_t stream) {
  const size_t total = (size_t)p.N * p.top_c * p.top_h * p.top_w;
  if (total == 0) return 0;
  return launch("corr_generic", corr_generic_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                stream, p);
}

}  // namespace mfn
