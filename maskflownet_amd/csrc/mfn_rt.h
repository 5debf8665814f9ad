// mfn_rt.h -- the one place where the kernels meet the runtime.
//
// Product build (hipcc --offload-arch=gfx950): plain HIP for CDNA4, launches go through
// mfn::launch() which optionally brackets each kernel with HIP events on its own stream
// (hipExtLaunchKernelGGL) for the built-in profiler of include/mfn_hip.h.
// Test build (g++ -DMFN_EMU): the same kernel sources run on tests/emu/hipemu.h so that the
// CPU-only CI can check their logic against the oracle.  There is no other dual path: no CUDA,
// no hipify, no fallback inside the product library.
#pragma once

#if defined(MFN_EMU)
#include "hipemu.h"
typedef f32x16_emu f32x16;
typedef f32x4_emu f32x4;
struct f4u { float x, y, z, w; };  // 16 bytes at 4-byte alignment
static inline f4u mfn_load4u(const float *p) { f4u v; memcpy(&v, p, 16); return v; }
struct f3u { float x, y, z; };      // 12 bytes at 4-byte alignment
static inline f3u mfn_load3u(const float *p) { f3u v; memcpy(&v, p, 12); return v; }
static inline int mfn_f2i(float f) { int i; memcpy(&i, &f, 4); return i; }  // bit pattern
struct f2u { float x, y; };  // 8 bytes at 4-byte alignment
static inline f2u mfn_load2u(const float *p) { f2u v; memcpy(&v, p, 8); return v; }
struct f32x2 { float x, y; };
static inline f32x2 mfn_f2(float x, float y) { return f32x2{x, y}; }
static inline f32x2 mfn_fma2(f32x2 a, f32x2 b, f32x2 c) { return f32x2{fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y)}; }
// acc = {a.y, a.x} * b[HI] + acc (see the device definition)
template <bool HI> static inline void mfn_pk_fma_swbc(f32x2 &acc, f32x2 a, f32x2 b) {
  const float bb = HI ? b.y : b.x;
  acc = f32x2{fmaf(a.y, bb, acc.x), fmaf(a.x, bb, acc.y)};
}
static inline void mfn_fmac_inorder(float &acc, float a, float b) { acc = fmaf(a, b, acc); }
struct f32x4v { float x, y, z, w; };
static inline f32x4v mfn_lds_read4(const float *p) { return f32x4v{p[0], p[1], p[2], p[3]}; }
static inline f32x2 mfn_lo2(f32x4v v) { return f32x2{v.x, v.y}; }
static inline f32x2 mfn_hi2(f32x4v v) { return f32x2{v.z, v.w}; }
static inline f32x2 mfn_mul2(f32x2 a, f32x2 b) { return f32x2{a.x * b.x, a.y * b.y}; }
#define MFN_DYN_SHARED(T, name) T *name = reinterpret_cast<T *>(hipemu::dyn_shared())
#define MFN_MFMA_32x32x2(a, b, c) hipemu_mfma_32x32x2((a), (b), (c))
#define MFN_MFMA_16x16x4(a, b, c) hipemu_mfma_16x16x4((a), (b), (c))
// bf16 x 3 operand split (dc.mma = 1): see the device definitions
typedef bf16x8_emu mfn_bf16x8;
#define MFN_MFMA_32x32x16_BF16(a, b, c) hipemu_mfma_32x32x16_bf16((a), (b), (c))
static inline void mfn_split3(float x, unsigned short &h, unsigned short &m, unsigned short &l) {
  h = hipemu_f32_to_bf16(x);
  const float r1 = x - hipemu_bf16_to_f32(h);
  m = hipemu_f32_to_bf16(r1);
  l = hipemu_f32_to_bf16(r1 - hipemu_bf16_to_f32(m));
}
static inline void mfn_split3x8(const float (&x)[8], mfn_bf16x8 &h, mfn_bf16x8 &m, mfn_bf16x8 &l) {
  for (int e = 0; e < 8; ++e) mfn_split3(x[e], h.v[e], m.v[e], l.v[e]);
}
#define mfn_split3x8_scalar mfn_split3x8   // (device: the residuals by scalar subtractions instead of v_pk_add_f32)
// NP pairs of values -> NP words per term (low half = the pair's first value); words back into a matrix operand; a word from two neighbours
template <int NP> static inline void mfn_split3_pairs(const float (&x)[2 * NP], unsigned (&h)[NP], unsigned (&m)[NP], unsigned (&l)[NP]) {
  for (int q = 0; q < NP; ++q) {
    unsigned short h0, m0, l0, h1, m1, l1;
    mfn_split3(x[2 * q], h0, m0, l0);
    mfn_split3(x[2 * q + 1], h1, m1, l1);
    h[q] = (unsigned)h0 | ((unsigned)h1 << 16); m[q] = (unsigned)m0 | ((unsigned)m1 << 16); l[q] = (unsigned)l0 | ((unsigned)l1 << 16);
  }
}
static inline mfn_bf16x8 mfn_words_to_bf16x8(unsigned w0, unsigned w1, unsigned w2, unsigned w3) {
  const unsigned w[4] = {w0, w1, w2, w3};
  mfn_bf16x8 v; memcpy(&v, w, 16); return v;
}
static inline unsigned mfn_alignbit16(unsigned hi, unsigned lo) { return (lo >> 16) | (hi << 16); }
// the same split one term at a time (the kernel places a matrix instruction between the stages)
struct mfn_split_state { float r[8]; };
static inline void mfn_split_stage_h(const float (&x)[8], mfn_bf16x8 &h, mfn_split_state &st) {
  for (int e = 0; e < 8; ++e) { h.v[e] = hipemu_f32_to_bf16(x[e]); st.r[e] = x[e] - hipemu_bf16_to_f32(h.v[e]); }
}
static inline void mfn_split_stage_m(mfn_split_state &st, mfn_bf16x8 &m) {
  for (int e = 0; e < 8; ++e) { m.v[e] = hipemu_f32_to_bf16(st.r[e]); st.r[e] = st.r[e] - hipemu_bf16_to_f32(m.v[e]); }
}
static inline void mfn_split_stage_l(const mfn_split_state &st, mfn_bf16x8 &l) {
  for (int e = 0; e < 8; ++e) l.v[e] = hipemu_f32_to_bf16(st.r[e]);
}
static inline mfn_bf16x8 mfn_read_bf16x8(const float *p) { mfn_bf16x8 v; memcpy(&v, p, 16); return v; }
static inline void mfn_write_bf16x8(float *p, mfn_bf16x8 v) { memcpy(p, &v, 16); }
static inline float mfn_bf16_at(const float *base, int idx) {   // element idx of a bf16 array, as fp32
  unsigned short h; memcpy(&h, (const char *)base + 2 * (size_t)idx, 2); return hipemu_bf16_to_f32(h);
}
// Gram-band cost volume (correlation_gram.h): bf16 matrix-core tile, DPP row shift, range-checked buffer store, counted wait
#define MFN_MFMA_16x16x32_BF16(a, b, c) hipemu_mfma_16x16x32_bf16((a), (b), (c))
typedef bf16x4_emu mfn_bf16x4;
#define MFN_MFMA_4x4x4_BF16(a, b, c) hipemu_mfma_4x4x4_bf16((a), (b), (c))
static inline mfn_bf16x4 mfn_bf16x8_half(const mfn_bf16x8 &v, int hi) { mfn_bf16x4 r; memcpy(&r, &v.v[4 * hi], 8); return r; }
static inline mfn_bf16x4 mfn_words_to_bf16x4(unsigned w0, unsigned w1) { unsigned w[2] = {w0, w1}; mfn_bf16x4 r; memcpy(&r, w, 8); return r; }
template <int N> static inline float mfn_dpp_row_shl(float old, float src) { return hipemu_dpp_row_shl(old, src, N); }
static inline float mfn_dpp_wave_shr1(float old, float src) { return hipemu_dpp_wave_shift(old, src, -1); }   // lane i <- lane i-1
static inline float mfn_dpp_wave_shl1(float old, float src) { return hipemu_dpp_wave_shift(old, src, +1); }   // lane i <- lane i+1
static inline float mfn_leaky01(float v) { return fmaxf(v, 0.1f * v); }
static inline void mfn_split2x8(const float (&x)[8], mfn_bf16x8 &h, mfn_bf16x8 &l) {
  for (int e = 0; e < 8; ++e) {
    h.v[e] = hipemu_f32_to_bf16(x[e]);
    l.v[e] = hipemu_f32_to_bf16(x[e] - hipemu_bf16_to_f32(h.v[e]));
  }
}
#define MFN_LANE_ID() ((int)hipemu::t_lane)
#define MFN_UNROLL
#define MFN_NOUNROLL
#define MFN_OPAQUE(x) ((void)(x))
#define MFN_SCHED_BARRIER() ((void)0)
#define MFN_UNIFORM(x) (x)
// wave-wide integer min / max, same value in every lane
#define MFN_WAVE_REDUCE_EMU(name, op)                                  \
  static inline int name(int v) {                                     \
    for (int s = 32; s >= 1; s >>= 1) v = op(v, __shfl_xor(v, s));     \
    return v;                                                          \
  }
// sum over the 32 lanes of each half-wave; valid in the top lane (31 / 63) of the half
static inline float mfn_half_sum_top(float v) {
  for (int s = 16; s >= 1; s >>= 1) v += __shfl_xor(v, s, 32);
  return v;
}
// emulated lanes are free-running threads: where the hardware's in-order LDS pipe orders two lanes' accesses, the
// emulation needs a wave barrier
#define MFN_WAVE_SYNC_EMU() (hipemu::wave().bar.arrive_and_wait())
// a lane's sequence of LDS read-add-writes that the hardware interleaves with the other lanes' instruction by instruction
// (in order): the free-running emulated lanes run the whole sequence one lane at a time instead -- same sums
#define MFN_EMU_LOCK() (hipemu::t_block->mu.lock())
#define MFN_EMU_UNLOCK() (hipemu::t_block->mu.unlock())
static inline int mfn_readlane_i32(int v, int lane) { return __shfl(v, lane); }   // the value of one lane, wave-uniform
MFN_WAVE_REDUCE_EMU(mfn_wave_min_i32, std::min)
MFN_WAVE_REDUCE_EMU(mfn_wave_max_i32, std::max)
// LDS-DMA emulation: synchronous copy (ordering of the real asynchronous engine is checked on the GPU)
struct mfn_rsrc_t { const char *base; unsigned nrec; };
static inline mfn_rsrc_t mfn_make_rsrc(const void *p, unsigned nbytes) { return mfn_rsrc_t{(const char *)p, nbytes}; }
static inline void mfn_dma16(mfn_rsrc_t r, float *lds_wave_base, unsigned voff) {
  char *dst = (char *)lds_wave_base + hipemu::t_lane * 16;
  if ((unsigned long long)voff + 16 <= r.nrec) memcpy(dst, r.base + voff, 16);
  else memset(dst, 0, 16);
}
static inline void mfn_dma16_so(mfn_rsrc_t r, float *lds_wave_base, unsigned voff, unsigned soff) {
  mfn_dma16(r, lds_wave_base, voff > 0xFFFFFF00u - soff ? 0xFFFFFF00u : voff + soff);
}
// 16 bytes per lane through a raw buffer descriptor: lanes whose byte offset is out of the descriptor's range store nothing
static inline void mfn_bstore4(mfn_rsrc_t r, unsigned voff, f32x4_emu v, int /*policy*/) {
  if ((unsigned long long)voff + 16 <= r.nrec) memcpy(const_cast<char *>(r.base) + voff, &v, 16);
}
// the same with a wave-uniform byte offset (soffset): added to the address AND to the offset the range check sees
static inline void mfn_bstore4_so(mfn_rsrc_t r, unsigned voff, unsigned soff, f32x4_emu v, int /*policy*/) {
  if ((unsigned long long)voff + soff + 16 <= r.nrec) memcpy(const_cast<char *>(r.base) + voff + soff, &v, 16);
}
// one float per lane through a raw buffer descriptor + wave-uniform soffset: out of range (or !valid) reads 0
static inline float mfn_bload1_row(const void *base, unsigned full_bytes, unsigned soff, bool valid, unsigned voff) {
  float v = 0.f;
  if (valid && (unsigned long long)voff + soff + 4 <= full_bytes) memcpy(&v, (const char *)base + voff + soff, 4);
  return v;
}
// sixteen bytes per lane through a raw buffer descriptor: a lane whose byte offset is out of the range reads zeros
static inline f32x4_emu mfn_bload4(const void *base, unsigned range_bytes, unsigned voff) {
  f32x4_emu v = {{0.f, 0.f, 0.f, 0.f}};
  if ((unsigned long long)voff + 16 <= range_bytes) memcpy(&v, (const char *)base + voff, 16);
  return v;
}
// DMA of one row of a tensor: a fixed base, the row's byte offset as a wave-uniform soffset, and a range check that is exact for
// base .. base+full_bytes (lanes whose voff+soff+16 exceeds it, or every lane when !valid, write zeros)
static inline void mfn_dma16_row(const void *base, unsigned full_bytes, unsigned soff, bool valid, float *lds_wave_base, unsigned voff) {
  char *dst = (char *)lds_wave_base + hipemu::t_lane * 16;
  if (valid && (unsigned long long)voff + soff + 16 <= full_bytes) memcpy(dst, (const char *)base + voff + soff, 16);
  else memset(dst, 0, 16);
}
// four consecutive floats at dword alignment from a wave-uniform base + a per-lane byte offset; synchronous here
static inline void mfn_gload4_async(f32x4_emu &dst, const float *base_uniform, unsigned byteoff) {
  memcpy(&dst, (const char *)base_uniform + byteoff, 16);
}
// four consecutive floats, one request each, through a raw buffer descriptor: out of range reads 0.  Synchronous here.
static inline void mfn_bload1x4_async(float &d0, float &d1, float &d2, float &d3, mfn_rsrc_t r, unsigned voff, unsigned soff,
                                      unsigned long long lanes) {
  if (!((lanes >> hipemu::t_lane) & 1ull)) return;
  float *d[4] = {&d0, &d1, &d2, &d3};
  for (int k = 0; k < 4; ++k) {
    const unsigned off = voff + 4u * k;
    if ((unsigned long long)off + soff + 4 <= r.nrec) memcpy(d[k], r.base + off + soff, 4);
    else *d[k] = 0.f;
  }
}
#define MFN_LANDED4(a, b, c, d, n) ((void)0)
#define MFN_REGFENCE4(a, b, c, d) ((void)0)
#define MFN_REGFENCE9(v) ((void)0)
#define MFN_REGFENCE8(a, b, c, d, e, f, g, h) ((void)0)
#define MFN_SCHED_GROUP(mask, n) ((void)0)
// emulated lanes are independent threads: a wave-private DMA hand-off needs a wave barrier where the
// hardware needs only the issuing wave's vmcnt wait (lock-step lanes)
#define MFN_WAIT_VM(n) (hipemu::wave().bar.arrive_and_wait())
#define MFN_WAIT_LGKM0() (hipemu::wave().bar.arrive_and_wait())
static inline void mfn_wait_vm_dyn(unsigned) { hipemu::wave().bar.arrive_and_wait(); }
#define MFN_RAW_BARRIER() __syncthreads()
#define MFN_LDS_BARRIER() __syncthreads()
#define MFN_COMPILER_FENCE() ((void)0)
#define MFN_STAMP(buf, k) ((void)0)
#define MFN_STAMP2(buf, k) ((void)0)
#define MFN_STAMP_INFO(buf, val) ((void)0)
#define MFN_CYCLES() 0ull
#else
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// packed fp32: v_pk_fma_f32 does two FMAs per lane per issue; hipcc folds the element swizzles of
// its operands into op_sel/op_sel_hi, so building pairs from registers costs no moves
// four consecutive floats at 4-byte alignment: one global_load_dwordx4 (gfx950 allows dword-aligned
// vector loads) instead of four global_load_dword -- a quarter of the L1 (TA/TCP) accesses
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ f4u mfn_load4u(const float *p) { return *reinterpret_cast<const f4u *>(p); }
typedef float f3u __attribute__((ext_vector_type(3), aligned(4)));   // one global_load_dwordx3 at dword alignment
__device__ __forceinline__ f3u mfn_load3u(const float *p) { return *reinterpret_cast<const f3u *>(p); }
__device__ __forceinline__ int mfn_f2i(float f) { return __builtin_bit_cast(int, f); }  // bit pattern
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
__device__ __forceinline__ f2u mfn_load2u(const float *p) { return *reinterpret_cast<const f2u *>(p); }
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 mfn_f2(float x, float y) { return (f32x2){x, y}; }
#define mfn_fma2(a, b, c) __builtin_elementwise_fma((a), (b), (c))
#define mfn_mul2(a, b) ((a) * (b))
// acc = {a.y, a.x} * b[HI] + acc as ONE v_pk_fma_f32 with the swizzles spelled out in op_sel / op_sel_hi.  hipcc folds most
// but not all of them from the generic form (the cost volume's channel loop kept 4 v_mov_b32 per 20 FMA instructions); a and b
// are the aligned halves of a ds_read_b128 result, so nothing has to move.
template <bool HI> __device__ __forceinline__ void mfn_pk_fma_swbc(f32x2 &acc, f32x2 a, f32x2 b) {
  if (HI) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(a), "v"(b));
  else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[0,0,1]" : "+v"(acc) : "v"(a), "v"(b));
}
// a scalar FMA that stays where it is written among the packed ones (the scheduler otherwise hoists it above them and with it
// the wait for the LAST operand read)
__device__ __forceinline__ void mfn_fmac_inorder(float &acc, float a, float b) { asm("v_fmac_f32 %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b)); }
typedef f32x4 f32x4v;
__device__ __forceinline__ f32x4v mfn_lds_read4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
__device__ __forceinline__ f32x2 mfn_lo2(f32x4v v) { return __builtin_shufflevector(v, v, 0, 1); }
__device__ __forceinline__ f32x2 mfn_hi2(f32x4v v) { return __builtin_shufflevector(v, v, 2, 3); }
// all dynamic LDS hangs off ONE 16-byte aligned symbol (cdna_hip_programming.md G17)
extern __shared__ __attribute__((aligned(16))) unsigned char mfn_lds_raw[];
#define MFN_DYN_SHARED(T, name) T *name = reinterpret_cast<T *>(mfn_lds_raw)
#define MFN_MFMA_32x32x2(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define MFN_MFMA_16x16x4(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
// ---- bf16 x 3 operand split (dc.mma = 1, measured non-default variant) -------------------------------------------------
// An fp32 value is written as hi + mid + lo with three bf16 terms (8 significant bits each: 24 together, exact unless the
// low terms underflow); a product a * b is then the sum of nine bf16 x bf16 products, each EXACT in fp32, of which the six
// with weight >= 2^-16 (hh, hm, mh, mm, hl, lh) are formed on the matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulate):
// the omitted terms are below 2^-23 of the product, the size of one fp32 rounding.  Conversions round to nearest even
// (v_cvt_pk_bf16_f32); x - float(hi) is exact in fp32.
typedef __bf16 mfn_bf16x8 __attribute__((ext_vector_type(8)));
#define MFN_MFMA_32x32x16_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
// two values at a time: one v_cvt_pk_bf16_f32 per term, the terms widened back by a shift / a mask, the residuals by one
// v_pk_add_f32 -- 38 VALU instructions for eight values
__device__ __forceinline__ void mfn_split3x8(const float (&x)[8], mfn_bf16x8 &h, mfn_bf16x8 &m, mfn_bf16x8 &l) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 hw, mw, lw;
  _Pragma("unroll")
  for (int q = 0; q < 4; ++q) {
    const f32x2 v = {x[2 * q], x[2 * q + 1]};
    const unsigned hp = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2));
    const f32x2 hf = {__builtin_bit_cast(float, hp << 16), __builtin_bit_cast(float, hp & 0xffff0000u)};
    const f32x2 r1 = v - hf;
    const unsigned mp = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf2));
    const f32x2 mf = {__builtin_bit_cast(float, mp << 16), __builtin_bit_cast(float, mp & 0xffff0000u)};
    const f32x2 r2 = r1 - mf;
    hw[q] = hp; mw[q] = mp; lw[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf2));
  }
  h = __builtin_bit_cast(mfn_bf16x8, hw); m = __builtin_bit_cast(mfn_bf16x8, mw); l = __builtin_bit_cast(mfn_bf16x8, lw);
}
// the same with the residuals formed by scalar v_sub_f32 (46 instructions): packed fp32 arithmetic issued while matrix
// instructions of the wave are in flight stalls (MI355X_MICROARCH.md, "anti-lever beside MFMAs"), so kernels that place the
// split BETWEEN their matrix instructions use this form
__device__ __forceinline__ void mfn_split3x8_scalar(const float (&x)[8], mfn_bf16x8 &h, mfn_bf16x8 &m, mfn_bf16x8 &l) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 hw, mw, lw;
  _Pragma("unroll")
  for (int q = 0; q < 4; ++q) {
    const f32x2 v = {x[2 * q], x[2 * q + 1]};
    const unsigned hp = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2));
    float r0 = v.x - __builtin_bit_cast(float, hp << 16), r1 = v.y - __builtin_bit_cast(float, hp & 0xffff0000u);
    asm volatile("" : "+v"(r0), "+v"(r1));   // keeps the two subtractions scalar (hipcc would pack them again)
    const f32x2 rr = {r0, r1};
    const unsigned mp = __builtin_bit_cast(unsigned, __builtin_convertvector(rr, bf2));
    float s0 = r0 - __builtin_bit_cast(float, mp << 16), s1 = r1 - __builtin_bit_cast(float, mp & 0xffff0000u);
    asm volatile("" : "+v"(s0), "+v"(s1));
    const f32x2 ss = {s0, s1};
    hw[q] = hp; mw[q] = mp; lw[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(ss, bf2));
  }
  h = __builtin_bit_cast(mfn_bf16x8, hw); m = __builtin_bit_cast(mfn_bf16x8, mw); l = __builtin_bit_cast(mfn_bf16x8, lw);
}
// NP pairs of values -> NP words per term (low half = the pair's first value), residuals by scalar subtractions: 11 instructions per pair
template <int NP> __device__ __forceinline__ void mfn_split3_pairs(const float (&x)[2 * NP], unsigned (&h)[NP], unsigned (&m)[NP], unsigned (&l)[NP]) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  _Pragma("unroll")
  for (int q = 0; q < NP; ++q) {
    const f32x2 v = {x[2 * q], x[2 * q + 1]};
    const unsigned hp = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2));
    float r0 = v.x - __builtin_bit_cast(float, hp << 16), r1 = v.y - __builtin_bit_cast(float, hp & 0xffff0000u);
    asm volatile("" : "+v"(r0), "+v"(r1));
    const f32x2 rr = {r0, r1};
    const unsigned mp = __builtin_bit_cast(unsigned, __builtin_convertvector(rr, bf2));
    float s0 = r0 - __builtin_bit_cast(float, mp << 16), s1 = r1 - __builtin_bit_cast(float, mp & 0xffff0000u);
    asm volatile("" : "+v"(s0), "+v"(s1));
    const f32x2 ss = {s0, s1};
    h[q] = hp; m[q] = mp; l[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(ss, bf2));
  }
}
__device__ __forceinline__ mfn_bf16x8 mfn_words_to_bf16x8(unsigned w0, unsigned w1, unsigned w2, unsigned w3) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 w = {w0, w1, w2, w3};
  return __builtin_bit_cast(mfn_bf16x8, w);
}
// the word made of the high half of `lo` and the low half of `hi` (one v_alignbit_b32): a pair that starts at an odd element
__device__ __forceinline__ unsigned mfn_alignbit16(unsigned hi, unsigned lo) { return __builtin_amdgcn_alignbit(hi, lo, 16); }
// the same split one term at a time (4 + 16 + 16 VALU instructions; the kernel places a matrix instruction between the stages):
// h: the hi terms, the state keeps the values; m: widen hi, first residual, mid terms; l: widen mid, second residual, lo terms
struct mfn_split_state { f32x2 v[4]; unsigned hp[4], mp[4]; };
__device__ __forceinline__ void mfn_split_stage_h(const float (&x)[8], mfn_bf16x8 &h, mfn_split_state &st) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 hw;
  _Pragma("unroll")
  for (int q = 0; q < 4; ++q) {
    st.v[q] = f32x2{x[2 * q], x[2 * q + 1]};
    st.hp[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(st.v[q], bf2));
    hw[q] = st.hp[q];
  }
  h = __builtin_bit_cast(mfn_bf16x8, hw);
}
__device__ __forceinline__ void mfn_split_stage_m(mfn_split_state &st, mfn_bf16x8 &m) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 mw;
  _Pragma("unroll")
  for (int q = 0; q < 4; ++q) {
    const f32x2 hf = {__builtin_bit_cast(float, st.hp[q] << 16), __builtin_bit_cast(float, st.hp[q] & 0xffff0000u)};
    st.v[q] = st.v[q] - hf;
    st.mp[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(st.v[q], bf2));
    mw[q] = st.mp[q];
  }
  m = __builtin_bit_cast(mfn_bf16x8, mw);
}
__device__ __forceinline__ void mfn_split_stage_l(const mfn_split_state &st, mfn_bf16x8 &l) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 lw;
  _Pragma("unroll")
  for (int q = 0; q < 4; ++q) {
    const f32x2 mf = {__builtin_bit_cast(float, st.mp[q] << 16), __builtin_bit_cast(float, st.mp[q] & 0xffff0000u)};
    const f32x2 r2 = st.v[q] - mf;
    lw[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf2));
  }
  l = __builtin_bit_cast(mfn_bf16x8, lw);
}
__device__ __forceinline__ mfn_bf16x8 mfn_read_bf16x8(const float *p) { return *reinterpret_cast<const mfn_bf16x8 *>(p); }
__device__ __forceinline__ void mfn_write_bf16x8(float *p, mfn_bf16x8 v) { *reinterpret_cast<mfn_bf16x8 *>(p) = v; }
__device__ __forceinline__ float mfn_bf16_at(const float *base, int idx) {   // element idx of a bf16 array, as fp32
  return __builtin_bit_cast(float, (unsigned)reinterpret_cast<const unsigned short *>(base)[idx] << 16);
}
// ---- Gram-band cost volume on the bf16 matrix cores (correlation_gram.h) ------------------------------------------------
#define MFN_MFMA_16x16x32_BF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
// sixteen independent 4 x 4 x 4 products (block = lane / 4; A: lane (b, i) = row i, B / C / D: lane (b, j) = column j): 2 passes
typedef short mfn_bf16x4 __attribute__((ext_vector_type(4)));
#define MFN_MFMA_4x4x4_BF16(a, b, c) __builtin_amdgcn_mfma_f32_4x4x4bf16_1k((a), (b), (c), 0, 0, 0)
__device__ __forceinline__ mfn_bf16x4 mfn_bf16x8_half(const mfn_bf16x8 &v, int hi) {
  typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
  typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
  const u32x4_ w = __builtin_bit_cast(u32x4_, v);
  const u32x2_ h = {hi ? w[2] : w[0], hi ? w[3] : w[1]};
  return __builtin_bit_cast(mfn_bf16x4, h);
}
__device__ __forceinline__ mfn_bf16x4 mfn_words_to_bf16x4(unsigned w0, unsigned w1) {
  typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
  const u32x2_ h = {w0, w1};
  return __builtin_bit_cast(mfn_bf16x4, h);
}
// v_mov_b32_dpp row_shl:N -- lane i of every 16-lane row receives lane i+N of its row; lanes whose source is outside the row keep `old`
template <int N> __device__ __forceinline__ float mfn_dpp_row_shl(float old, float src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), 0x100 + N, 0xf, 0xf, false));
}
// v_mov_b32_dpp wave_shr:1 / wave_shl:1 (gfx9 family): lane i receives lane i-1 / i+1 of the whole wave; lane 0 / 63 keeps `old`
__device__ __forceinline__ float mfn_dpp_wave_shr1(float old, float src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float mfn_dpp_wave_shl1(float old, float src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), 0x130, 0xf, 0xf, false));
}
// LeakyReLU(0.1)(v) = max(v, 0.1 v) as v_mul + v_max (fmaxf() puts a canonicalising v_max in front of each operand)
__device__ __forceinline__ float mfn_leaky01(float v) {
  const float t = 0.1f * v;
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(t));
  return r;
}
// two-term split (hi + lo, 16 significant bits): 20 VALU instructions for eight values -- the measured, inexact variant
__device__ __forceinline__ void mfn_split2x8(const float (&x)[8], mfn_bf16x8 &h, mfn_bf16x8 &l) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 hw, lw;
  _Pragma("unroll")
  for (int q = 0; q < 4; ++q) {
    const f32x2 v = {x[2 * q], x[2 * q + 1]};
    const unsigned hp = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2));
    const f32x2 hf = {__builtin_bit_cast(float, hp << 16), __builtin_bit_cast(float, hp & 0xffff0000u)};
    hw[q] = hp; lw[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(v - hf, bf2));
  }
  h = __builtin_bit_cast(mfn_bf16x8, hw); l = __builtin_bit_cast(mfn_bf16x8, lw);
}
#define MFN_LANE_ID() ((int)(threadIdx.x & 63))
#define MFN_UNROLL _Pragma("unroll")
#define MFN_NOUNROLL _Pragma("nounroll")
#define MFN_OPAQUE(x) asm volatile("" : "+v"(x))
#define MFN_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
// a scheduling group inside the region between two MFN_SCHED_BARRIERs: `n` instructions of the classes in `mask`
// (0x002 VALU, 0x008 MFMA, 0x100 LDS read) are placed next, in the order the groups are written
#define MFN_SCHED_GROUP(mask, n) __builtin_amdgcn_sched_group_barrier((mask), (n), 0)
#define MFN_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
// wave-wide integer min / max as a DPP scan (row_shr 1,2,4,8, row_bcast 15/31: six v_min/max_i32_dpp, no LDS
// round trips as ds_bpermute shuffles would need); the result comes back wave-uniform from lane 63
__device__ __forceinline__ int mfn_wave_min_i32(int v) {
  const int id = 0x7fffffff;
  v = min(v, __builtin_amdgcn_update_dpp(id, v, 0x111, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(id, v, 0x112, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(id, v, 0x114, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(id, v, 0x118, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(id, v, 0x142, 0xa, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(id, v, 0x143, 0xc, 0xf, false));
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int mfn_readlane_i32(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }   // wave-uniform
#define MFN_WAVE_SYNC_EMU() ((void)0)
#define MFN_EMU_LOCK() ((void)0)
#define MFN_EMU_UNLOCK() ((void)0)
// sum over the 32 lanes of each half-wave as a DPP scan (no LDS); valid in the top lane (31 / 63) of the half
__device__ __forceinline__ float mfn_half_sum_top(float v) {
#define MFN_DPP_ADD_(ctrl, rmask) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, 0xf, false))
  MFN_DPP_ADD_(0x111, 0xf);  // row_shr:1
  MFN_DPP_ADD_(0x112, 0xf);  // row_shr:2
  MFN_DPP_ADD_(0x114, 0xf);  // row_shr:4
  MFN_DPP_ADD_(0x118, 0xf);  // row_shr:8  -> lane 15 of every 16-lane row holds the row's sum
  MFN_DPP_ADD_(0x142, 0xa);  // row_bcast:15 into rows 1 and 3 -> lanes 31 and 63 hold the half-wave sums
#undef MFN_DPP_ADD_
  return v;
}
__device__ __forceinline__ int mfn_wave_max_i32(int v) {
  const int id = (int)0x80000000;
  v = max(v, __builtin_amdgcn_update_dpp(id, v, 0x111, 0xf, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(id, v, 0x112, 0xf, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(id, v, 0x114, 0xf, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(id, v, 0x118, 0xf, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(id, v, 0x142, 0xa, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(id, v, 0x143, 0xc, 0xf, false));
  return __builtin_amdgcn_readlane(v, 63);
}
// ---- LDS-DMA (buffer_load_dwordx4 ... lds): global -> LDS without a VGPR round trip ------------------
// Issued through inline asm on purpose: hipcc waits vmcnt(0) before any ds_read that follows a DMA it
// knows about, which would serialise the staging ring.  Counting is therefore ours: MFN_WAIT_VM(n).
// M0 (the LDS destination) is written and read INSIDE one statement and not named a clobber: hipcc reserves the register (naming
// it only drew 1 690 "clobber list contains reserved registers" warnings per build and changed nothing), and it never keeps a
// value of its own in it across our statements -- tests/test_abi.py disassembles the shipped library and asserts that every
// instruction that touches M0 is one of these `s_mov_b32 m0, ...`, immediately followed by its `buffer_load ... lds`.
// A raw buffer descriptor gives zero fill for free: lanes whose byte offset is >= num_records read 0.
typedef int mfn_rsrc_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ mfn_rsrc_t mfn_make_rsrc(const void *p, unsigned nbytes) {
  const unsigned long long a = (unsigned long long)p;
  mfn_rsrc_t r;
  r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xFFFFu));
  r.z = __builtin_amdgcn_readfirstlane((int)nbytes);
  r.w = 0x00020000;
  return r;
}
// every lane of the wave writes 16 bytes at lds_wave_base + lane*16 (the base must be wave-uniform)
__device__ __forceinline__ void mfn_dma16(mfn_rsrc_t rsrc, float *lds_wave_base, unsigned voff) {
  const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds_wave_base);
  asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc)
               : "memory");
}
// same with a wave-uniform byte offset added by the instruction's soffset operand
__device__ __forceinline__ void mfn_dma16_so(mfn_rsrc_t rsrc, float *lds_wave_base, unsigned voff, unsigned soff) {
  const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds_wave_base);
  const unsigned so = __builtin_amdgcn_readfirstlane(soff);
  asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc), "s"(so)
               : "memory");
}
// DMA of one row of a tensor: ONE descriptor per tensor, the row's byte offset in soffset.  The hardware's range check of a raw
// buffer is offset >= num_records - soffset, i.e. voff + soff against the descriptor's range: exact for the tensor (measured:
// shrinking num_records by soff as well zero-fills the lower half of the image).  !valid: num_records 0, every lane reads zeros.
__device__ __forceinline__ void mfn_dma16_row(const void *base, unsigned full_bytes, unsigned soff, bool valid, float *lds_wave_base,
                                              unsigned voff) {
  const unsigned long long a = (unsigned long long)base;
  mfn_rsrc_t r;
  r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xFFFFu));
  r.z = __builtin_amdgcn_readfirstlane((int)(valid ? full_bytes : 0u));
  r.w = 0x00020000;
  const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds_wave_base);
  const unsigned so = __builtin_amdgcn_readfirstlane(valid ? soff : 0u);
  asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(r), "s"(so)
               : "memory");
}
// one float per lane into a REGISTER through the same kind of descriptor (the compiler sees this load and places its wait):
// lanes whose voff + soff is out of the tensor, or every lane when !valid, read 0
__device__ __forceinline__ float mfn_bload1_row(const void *base, unsigned full_bytes, unsigned soff, bool valid, unsigned voff) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)(valid ? full_bytes : 0u), 0x00020000);
  const unsigned so = __builtin_amdgcn_readfirstlane(valid ? soff : 0u);
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)so, 0));
}
// sixteen bytes per lane through a raw buffer descriptor (wave-uniform base, per-lane byte offset): out-of-range lanes read
// zeros -- a mask that costs no VALU instruction and, unlike a select on the loaded value, does not pull the wait up to the load
__device__ __forceinline__ f32x4 mfn_bload4(const void *base, unsigned range_bytes, unsigned voff) {
  typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)range_bytes, 0x00020000);
  return __builtin_bit_cast(f32x4, (u32x4_)__builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0));
}
#define MFN_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define MFN_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
// 16 bytes per lane through a raw buffer descriptor (wave-uniform base, per-lane byte offset): lanes whose offset is out of
// the descriptor's range store nothing -- the band mask of the Gram kernel costs no exec juggling.  Unknown to hipcc like the
// LDS-DMA loads: it joins the in-order vmcnt queue (gfx9 family: vector memory operations complete in issue order), so the
// counted waits of a pipeline that stores while it loads have to count it.  s_nop 1: as mfn_store4_stream.
__device__ __forceinline__ void mfn_bstore4(mfn_rsrc_t rsrc, unsigned voff, f32x4 v, int policy) {
  if (policy == 2) asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen sc0 sc1\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(rsrc) : "memory");
  else if (policy == 1) asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen nt\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(rsrc) : "memory");
  else if (policy == 3) asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen sc0 sc1 nt\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(rsrc) : "memory");
  else asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(rsrc) : "memory");
}
// the same with a wave-uniform byte offset in the instruction's soffset operand (added to the address and to the range-checked offset)
__device__ __forceinline__ void mfn_bstore4_so(mfn_rsrc_t rsrc, unsigned voff, unsigned soff, f32x4 v, int policy) {
  const unsigned so = __builtin_amdgcn_readfirstlane(soff);
  if (policy == 2) asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen sc0 sc1\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(rsrc), "s"(so) : "memory");
  else if (policy == 1) asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen nt\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(rsrc), "s"(so) : "memory");
  else if (policy == 3) asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen sc0 sc1 nt\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(rsrc), "s"(so) : "memory");
  else asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(rsrc), "s"(so) : "memory");
}
// s_waitcnt vmcnt(k) for a wave-uniform RUN-TIME k (the instruction only takes an immediate): a compare tree over the
// values a pipeline can ask for; larger k wait for 48 (waiting for more than asked is always safe)
__device__ __forceinline__ void mfn_wait_vm_dyn(unsigned k) {
#define MFN_WVD_(n) case n: MFN_WAIT_VM(n); break;
  switch (k) {
    MFN_WVD_(0) MFN_WVD_(1) MFN_WVD_(2) MFN_WVD_(3) MFN_WVD_(4) MFN_WVD_(5) MFN_WVD_(6) MFN_WVD_(7)
    MFN_WVD_(8) MFN_WVD_(9) MFN_WVD_(10) MFN_WVD_(11) MFN_WVD_(12) MFN_WVD_(13) MFN_WVD_(14) MFN_WVD_(15)
    MFN_WVD_(16) MFN_WVD_(17) MFN_WVD_(18) MFN_WVD_(19) MFN_WVD_(20) MFN_WVD_(21) MFN_WVD_(22) MFN_WVD_(23)
    MFN_WVD_(24) MFN_WVD_(25) MFN_WVD_(26) MFN_WVD_(27) MFN_WVD_(28) MFN_WVD_(29) MFN_WVD_(30) MFN_WVD_(31)
    MFN_WVD_(32) MFN_WVD_(33) MFN_WVD_(34) MFN_WVD_(35) MFN_WVD_(36) MFN_WVD_(37) MFN_WVD_(38) MFN_WVD_(39)
    MFN_WVD_(40) MFN_WVD_(41) MFN_WVD_(42) MFN_WVD_(43) MFN_WVD_(44) MFN_WVD_(45) MFN_WVD_(46) MFN_WVD_(47)
    default: MFN_WAIT_VM(48); break;
  }
#undef MFN_WVD_
}
// A register load hipcc neither counts nor waits for: four consecutive floats at dword alignment from a wave-uniform
// base (SGPR pair) + a per-lane 32-bit byte offset.  It joins the same in-order vmcnt queue as the LDS-DMA transfers, so a
// software pipeline can keep loads of later steps in flight across its counted waits (a load hipcc knows about would make
// it wait for every LDS-DMA issued before the load's use as well).  The destination holds data only after MFN_LANDED4 on
// it: that statement carries the wait and names the registers read-write, so no use is scheduled above it
// (cdna_hip_programming.md 5.7, form (ii)).
__device__ __forceinline__ void mfn_gload4_async(f32x4 &dst, const float *base_uniform, unsigned byteoff) {
  const unsigned long long a = (unsigned long long)base_uniform;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  const unsigned long long sb = ((unsigned long long)hi << 32) | lo;
  // keep every request UNCONDITIONAL: a request inside a branch makes the destination a phi value, and the copies hipcc
  // resolves those with can land between the request and its wait
  // s_nop 4: the base may have been written by a v_readfirstlane just before (VALU-written SGPR -> VMEM: 5 wait states,
  // and hipcc pads nothing inside an asm statement)
  asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(byteoff), "s"(sb) : "memory");
}
// Four consecutive floats -- four requests, the instruction's immediate offsets 0, 4, 8, 12 -- through a raw buffer descriptor with a
// wave-uniform soffset, unknown to hipcc like mfn_gload4_async, for the lanes of the (wave-uniform) mask only: the texture
// addresser spends a cycle per ACTIVE lane on a dword request whose lanes do not share cache lines, so sixteen full-wave requests
// are ~1000 cycles of its time per step whoever needed them (measured: level 4, 7.1 -> 10.5 us of loop with 4 of 64 lanes outside
// the window).  The other lanes' registers keep their contents; with an empty mask the requests still issue and still count.
// Lanes whose offset is out of the descriptor's range get 0 and touch no memory.  Early-clobber destinations: the address
// register is read by the later requests.
__device__ __forceinline__ void mfn_bload1x4_async(float &d0, float &d1, float &d2, float &d3, mfn_rsrc_t rsrc, unsigned voff, unsigned soff,
                                                   unsigned long long lanes) {
  const unsigned so = __builtin_amdgcn_readfirstlane(soff);
  unsigned long long saved;
  asm volatile("s_and_saveexec_b64 %4, %8\n\ts_nop 4\n\t"
               "buffer_load_dword %0, %5, %6, %7 offen\n\tbuffer_load_dword %1, %5, %6, %7 offen offset:4\n\t"
               "buffer_load_dword %2, %5, %6, %7 offen offset:8\n\tbuffer_load_dword %3, %5, %6, %7 offen offset:12\n\t"
               "s_mov_b64 exec, %4"
               : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&s"(saved) : "v"(voff), "s"(rsrc), "s"(so), "s"(lanes) : "memory", "scc");
}
#define MFN_LANDED4(a, b, c, d, n) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(n) : "memory")
// the same in two parts, for waits whose count is chosen by a (uniform) branch: MFN_WAIT_VM(n) in the branches, then ONE
// fence on the registers behind the join -- a LANDED4 per branch makes the destinations phi values, and hipcc resolved
// those with register copies placed BEFORE the waits (copies of data that has not landed)
#define MFN_REGFENCE4(a, b, c, d) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "memory")
// nine values are COMPUTED here (hipcc otherwise sinks the arithmetic that forms them next to its use, one pipeline step
// later, and keeps the registers it reads alive across the request that is about to overwrite them)
#define MFN_REGFENCE9(v) asm volatile("" : "+v"((v)[0]), "+v"((v)[1]), "+v"((v)[2]), "+v"((v)[3]), "+v"((v)[4]), "+v"((v)[5]), "+v"((v)[6]), "+v"((v)[7]), "+v"((v)[8]))
// eight values (any register class width) are opaque from here on: what is computed from them cannot move above this point
#define MFN_REGFENCE8(a, b, c, d, e, f, g, h) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h))
#define MFN_RAW_BARRIER() __builtin_amdgcn_s_barrier()
// the compiler keeps memory accesses on their side of this point (no instruction): LDS accesses whose ORDER matters to other
// lanes of the wave -- the in-order LDS pipe does the rest
#define MFN_COMPILER_FENCE() asm volatile("" ::: "memory")
// block barrier for data handed over through LDS: this wave's LDS operations are complete before it, and the compiler keeps
// every memory access on its side of it (the bare s_barrier builtin does not stop hipcc from hoisting later LDS reads above it)
#define MFN_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// measurement only: constant-rate (100 MHz) wall clock stamps, one writer per block
#define MFN_CYCLES() ((unsigned long long)clock64())
// measurement only: bit 0 of the buffer address selects the shader-cycle counter instead of the 100 MHz wall clock.
// Compiled in only with -DMFN_TIMELINE=1 (tools/timeline*.py build their own library): the stamps cost registers and a
// branch inside the kernels' main loops.
#ifndef MFN_TIMELINE
#define MFN_TIMELINE 0
#endif
#if !MFN_TIMELINE
#define MFN_STAMP(buf, k) ((void)0)
#define MFN_STAMP2(buf, k) ((void)0)
#define MFN_STAMP_INFO(buf, val) ((void)0)
#else
// one more word per block behind the stamps of 16384 blocks (tools/timeline_dc_blocks.py): bits 0..15 the caller's value,
// 16..31 HW_ID (wave / SIMD / CU / SH / SE), 32..35 XCC_ID
#define MFN_STAMP_INFO(buf, val)                                                                            \
  do {                                                                                                      \
    if ((buf) && threadIdx.x == 0) {                                                                        \
      unsigned long long *b_ = (unsigned long long *)(((unsigned long long)(buf)) & ~1ull);                 \
      const unsigned long long hw_ = (unsigned)__builtin_amdgcn_s_getreg(63492) & 0xFFFFu;                  \
      const unsigned long long xcc_ = (unsigned)__builtin_amdgcn_s_getreg(63508) & 0xFu;                    \
      b_[65536 + ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] =                 \
          ((unsigned long long)(val) & 0xFFFFull) | (hw_ << 16) | (xcc_ << 32);                             \
    }                                                                                                       \
  } while (0)
// eight more stamps per block behind those (set-up phases): [3 * 65536 + block * 8 + k]
#define MFN_STAMP2(buf, k)                                                                                  \
  do {                                                                                                      \
    if ((buf) && threadIdx.x == 0) {                                                                        \
      const bool cyc_ = ((unsigned long long)(buf)) & 1ull;                                                 \
      unsigned long long *b_ = (unsigned long long *)(((unsigned long long)(buf)) & ~1ull);                 \
      b_[3 * 65536 + (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + (k)] =  \
          cyc_ ? (unsigned long long)clock64() : (unsigned long long)wall_clock64();                        \
    }                                                                                                       \
  } while (0)
#define MFN_STAMP(buf, k)                                                                                   \
  do {                                                                                                      \
    if ((buf) && threadIdx.x == 0) {                                                                        \
      const bool cyc_ = ((unsigned long long)(buf)) & 1ull;                                                 \
      unsigned long long *b_ = (unsigned long long *)(((unsigned long long)(buf)) & ~1ull);                 \
      b_[(((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + (k)] =              \
          cyc_ ? (unsigned long long)clock64() : (unsigned long long)wall_clock64();                        \
    }                                                                                                       \
  } while (0)
#endif
#endif

#include <stddef.h>
#include <stdint.h>
#include <type_traits>
#include <utility>

// f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>) in order: an unrolled loop by construction (a `#pragma unroll`
// loop is a request that hipcc drops above its size threshold -- and a kernel whose counted waits fold only when unrolled
// cannot live with that)
template <int... I, class F>
__host__ __device__ __forceinline__ void mfn_static_for_(std::integer_sequence<int, I...>, F &&f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__host__ __device__ __forceinline__ void mfn_static_for(F &&f) { mfn_static_for_(std::make_integer_sequence<int, N>{}, f); }

namespace mfn {

// Stores of a kernel's OUTPUT (cost volumes, warped features, offsets: nothing in the same kernel reads them again, the
// next kernel does, from whichever XCD its block order puts it on).  A plain store leaves the line dirty in the XCD's
// write-back L2 and the end-of-kernel release flushes all of it at once; `sc0 sc1` writes it through while the other
// blocks still compute (level-2 correlation, 31.8 MB of stores: 18.8 -> 14.8 us; tools/corr_nt_time.py).
// policy: 0 plain, 1 nt, 2 sc0 sc1 (default, tuning key store.policy), 3 sc0 sc1 nt.
__device__ __forceinline__ void mfn_store4_stream(float *dst, float a, float b, float c, float d, int policy) {
#if defined(MFN_EMU)
  (void)policy;
  dst[0] = a; dst[1] = b; dst[2] = c; dst[3] = d;
#else
  typedef float mfn_v4f __attribute__((ext_vector_type(4)));
  const mfn_v4f v = {a, b, c, d};
  // s_nop 1: a store of more than 64 bits reads its data registers after issue, and gfx940+ wants 2 wait states before a VALU
  // instruction overwrites them; the compiler's hazard recognizer does not look inside an asm statement (seen: straight-line
  // epilogues re-filling the same four registers for the next store -> ~1 % wrong elements)
  if (policy == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
  else if (policy == 1) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
  else if (policy == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
  else *reinterpret_cast<mfn_v4f *>(dst) = v;
#endif
}
__device__ __forceinline__ void mfn_store1_stream(float *dst, float v, int policy) {
#if defined(MFN_EMU)
  (void)policy;
  *dst = v;
#else
  if (policy == 2) asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(dst), "v"(v) : "memory");
  else if (policy == 1) asm volatile("global_store_dword %0, %1, off nt" ::"v"(dst), "v"(v) : "memory");
  else if (policy == 3) asm volatile("global_store_dword %0, %1, off sc0 sc1 nt" ::"v"(dst), "v"(v) : "memory");
  else *dst = v;
#endif
}

// Workgroup b of a 1-D grid runs on XCD b % 8 (round-robin dispatch) and every XCD has its own L2.  Kernels whose
// neighbouring tiles read overlapping data remap their block id with this so that XCD k works on one contiguous
// range of tiles, [k*q + min(k, r), ...) with q = nb / 8, r = nb % 8: the overlap is then served by one L2
// instead of being fetched over the fabric by several (full-resolution warp: 15.2 -> 11.6 us).
// x / d for a divisor the host knows: magic = ceil(2^32 / d) (0 for d == 1), exact while x * d < 2^32
__host__ __device__ __forceinline__ unsigned mfn_div_magic(unsigned x, unsigned magic) {
  return magic ? (unsigned)(((unsigned long long)x * magic) >> 32) : x;
}
inline unsigned mfn_make_magic(unsigned d) { return d <= 1 ? 0u : (unsigned)(((1ULL << 32) + d - 1) / d); }
__host__ __device__ __forceinline__ unsigned mfn_xcd_remap(unsigned b, unsigned nb) {
  const unsigned q = nb >> 3, r = nb & 7u, xcd = b & 7u, i = b >> 3;
  return xcd * q + (xcd < r ? xcd : r) + i;
}

// ---- launch plumbing -------------------------------------------------------------------------
#if defined(MFN_EMU)
template <class K, class... Args>
inline int launch(const char *name, K kernel, dim3 grid, dim3 block, size_t shmem,
                  hipStream_t /*stream*/, Args... args) {
  hipemu::note_launch(name);  // tests read the sequence of kernels a call dispatched to
  hipemu::launch(grid, block, shmem, [=]() { kernel(args...); });
  return 0;
}
#else
// profiler hooks (api.hip)
bool profile_enabled();
void profile_record(const char *name, hipEvent_t start, hipEvent_t stop);

template <class K, class... Args>
inline int launch(const char *name, K kernel, dim3 grid, dim3 block, size_t shmem,
                  hipStream_t stream, Args... args) {
  if (shmem > 65536) {  // gfx950 has 160 KiB per CU; above 64 KiB the runtime wants an opt-in
    hipError_t a = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (a != hipSuccess) return (int)a;
  }
  if (profile_enabled()) {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return (int)hipGetLastError();
    hipExtLaunchKernelGGL(kernel, grid, block, (unsigned)shmem, stream, e0, e1, 0, args...);
    profile_record(name, e0, e1);
  } else {
    hipLaunchKernelGGL(kernel, grid, block, (unsigned)shmem, stream, args...);
  }
  return (int)hipGetLastError();
}
#endif

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

}  // namespace mfn
