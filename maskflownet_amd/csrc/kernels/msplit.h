// msplit.h -- the bf16 x 3 operand split with its residuals formed ON THE MATRIX CORES (round 6), shared by the Gram cost volumes
// (correlation_gram.h) and the matrix-core deformable convolution / convolution (deform_conv_mma.h).
#pragma once
#include "../mfn_rt.h"

namespace mfn {

// The VALU form of the operand split (mfn_split3x8 of mfn_rt.h, correlation_gram.h TERMS == 3) is nine
// VALU instructions per pair of values: three v_cvt_pk_bf16_f32 and, to get each residual x - float(bf16(x)), a shift, a mask and a
// packed subtract -- 646 of a wave's 912 VALU instructions at level 2 (profiles/r04_corr_pmc.md).  But a matrix instruction's
// accumulator registers can BE the values to split: with a lane's eight raw values held as two accumulator tiles C0 = raw[0..3],
// C1 = raw[4..7] and their bf16 roundings as the B operand,
//   C <- Sel * Hb + C,   Sel = -identity
// subtracts from every register exactly the bf16 value of the same slot: the residual, exact in fp32 (products -1 * h and 0 * h are
// exact, the sum has one non-zero term and x - h is representable).  A tile's split is then 3 x 4 v_cvt_pk_bf16_f32 + 2 x 2 matrix
// instructions instead of 36 VALU instructions; the terms are bit-identical to the VALU split's (same roundings).  Non-finite inputs:
// 0 * inf = NaN spreads an inf to the residuals of the three other values of its lane's tile -- the same pixel's, in both users of
// this header -- so every output that pixel takes part in is NaN (VALU split: NaN as well, through inf - inf in the slot itself): the
// documented behaviour (include/mfn_hip.h "Arithmetic") is unchanged.
// The instruction: v_mfma_f32_4x4x4_16B_bf16 -- sixteen independent 4 x 4 x 4 products, block = lane / 4; A: lane (b, i) holds row i's
// four K values, B / C / D: lane (b, j) holds column j (four K values / registers i = 0..3).  With A = -identity in every block (lane l:
// -1.0 in K slot l % 4) and B = the lane's own four bf16 roundings, D[i] = C[i] - B[i]: the subtraction never leaves the lane -- a
// non-finite value poisons the other three values of ITS lane's tile (0 * inf), nobody else's (the first form of this header used
// v_mfma_f32_16x16x32_bf16, whose columns span four lanes: fine for the cost volume, where those are one pixel, wrong for the deformable
// convolution, where they are two) -- and the instruction is 2 passes: 0.65 of a 16 x 16 x 32's issue cost
// (tools/ubench/msplit_4x4.hip: bit-identical to the VALU residual on 256 values over twelve orders of magnitude; 7.2 against
// 10.8 cycles per instruction at four waves per SIMD).
// The WIDE form (the first one of round 6) does the same on v_mfma_f32_16x16x32_bf16, whose accumulator layout (lane = column n + 16 g,
// register i = row 4 g + i) is its operand layout (lane = column + 16 k-block, eight K values) with rows read as K slots:
// Sel0[r][k] = -1 if k == 8 (r / 4) + r % 4 (Sel1: ... + 4).  Its columns span the four lanes n, n + 16, n + 32, n + 48 -- one pixel in
// the cost volume's layout, so it is legal there (and only there) -- and it costs a full 16 x 16 x 32 per tile and level; the level-2 cost
// volume keeps it because INSIDE THE PASS it is the faster one there (rocprofv3 medians, same box: 11.40 us against 11.80 with the
// narrow form, although back to back the narrow form wins 9.41 against 9.6; level 3: narrow 9.44 against 9.64).
struct GramSel { mfn_bf16x4 s; mfn_bf16x8 s0, s1; };
template <bool WIDE = false>
__device__ __forceinline__ GramSel gram_make_sel(int lane) {
  GramSel r;
  if (WIDE) {
    const int m = lane & 15, kb = lane >> 4;
    const bool on = kb == (m >> 2);
    const unsigned one = 0xBF80u << (16 * (m & 1));        // -1.0 as bf16, in K slot m % 4 of the lane's k-block
    const unsigned w0 = (on && (m & 2) == 0) ? one : 0u, w1 = (on && (m & 2) != 0) ? one : 0u;
    r.s0 = mfn_words_to_bf16x8(w0, w1, 0u, 0u);
    r.s1 = mfn_words_to_bf16x8(0u, 0u, w0, w1);
    r.s = mfn_words_to_bf16x4(0u, 0u);
  } else {
    const unsigned one = 0xBF80u << (16 * (lane & 1));     // -1.0 as bf16, in K slot lane % 4
    r.s = mfn_words_to_bf16x4((lane & 2) == 0 ? one : 0u, (lane & 2) != 0 ? one : 0u);
    r.s0 = mfn_words_to_bf16x8(0u, 0u, 0u, 0u);
    r.s1 = r.s0;
  }
  return r;
}
// eight fp32 values (two accumulator tiles) -> their bf16 roundings as one operand: 4 x v_cvt_pk_bf16_f32
__device__ __forceinline__ mfn_bf16x8 gram_cvt8(const f32x4 &a, const f32x4 &b) {
#if defined(MFN_EMU)
  unsigned w[4];
  for (int q = 0; q < 2; ++q) {
    w[q] = (unsigned)hipemu_f32_to_bf16(a[2 * q]) | ((unsigned)hipemu_f32_to_bf16(a[2 * q + 1]) << 16);
    w[2 + q] = (unsigned)hipemu_f32_to_bf16(b[2 * q]) | ((unsigned)hipemu_f32_to_bf16(b[2 * q + 1]) << 16);
  }
  return mfn_words_to_bf16x8(w[0], w[1], w[2], w[3]);
#else
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  const f32x2 p0 = {a[0], a[1]}, p1 = {a[2], a[3]}, p2 = {b[0], b[1]}, p3 = {b[2], b[3]};
  return mfn_words_to_bf16x8(__builtin_bit_cast(unsigned, __builtin_convertvector(p0, bf2)), __builtin_bit_cast(unsigned, __builtin_convertvector(p1, bf2)),
                             __builtin_bit_cast(unsigned, __builtin_convertvector(p2, bf2)), __builtin_bit_cast(unsigned, __builtin_convertvector(p3, bf2)));
#endif
}
// one stage of the matrix-core split of a tile held as (x0, x1): even stages round the present residual into term st / 2,
// odd stages subtract that term
template <bool WIDE = false>
__device__ __forceinline__ void gram_msplit_stage(int st, const GramSel &sel, f32x4 &x0, f32x4 &x1, mfn_bf16x8 (&term)[3]) {
  if ((st & 1) == 0) term[st >> 1] = gram_cvt8(x0, x1);
  else if (WIDE) {
    x0 = MFN_MFMA_16x16x32_BF16(sel.s0, term[st >> 1], x0);
    x1 = MFN_MFMA_16x16x32_BF16(sel.s1, term[st >> 1], x1);
  } else {
    x0 = MFN_MFMA_4x4x4_BF16(sel.s, mfn_bf16x8_half(term[st >> 1], 0), x0);
    x1 = MFN_MFMA_4x4x4_BF16(sel.s, mfn_bf16x8_half(term[st >> 1], 1), x1);
  }
}

// all five stages: eight fp32 values of a lane (any 64-lane register set: the selector subtracts lane-locally) -> the three terms
template <bool WIDE = false>
__device__ __forceinline__ void gram_msplit8(const float (&x)[8], const GramSel &sel, mfn_bf16x8 &h, mfn_bf16x8 &m, mfn_bf16x8 &l) {
  f32x4 x0, x1;
  MFN_UNROLL
  for (int q = 0; q < 4; ++q) { x0[q] = x[q]; x1[q] = x[4 + q]; }
  mfn_bf16x8 term[3];
  MFN_UNROLL
  for (int st = 0; st < 5; ++st) gram_msplit_stage<WIDE>(st, sel, x0, x1, term);
  h = term[0]; m = term[1]; l = term[2];
}

}  // namespace mfn
