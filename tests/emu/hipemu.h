// hipemu.h -- a tiny CPU emulation of the HIP execution model (TEST INFRASTRUCTURE ONLY).
//
// The kernels under maskflownet_amd/csrc/kernels/ are written against a small subset of HIP
// (threadIdx/blockIdx, dynamic LDS, __syncthreads, wave shuffles, fp32 MFMA).  This header
// lets g++ compile the SAME kernel sources for the host so that tests/test_emu_*.py can check
// the index arithmetic, tiling, LDS staging and barrier placement of the real kernels
// against the oracle in the CPU-only container, before any GPU minute is spent.  It is not a
// fallback: the product (maskflownet_amd/_lib.py) only ever loads libmfn_hip.so and fails
// loudly without it; libmfn_emu.so exports mfn_emu_* symbols and is loaded by tests only.
//
// Model: one OS thread per HIP thread of a block, blocks run one after another.
// __syncthreads = std::barrier over the block; wave collectives (shuffle, ballot, MFMA)
// exchange through a per-wave buffer guarded by a 64-thread barrier.  Wave = 64 lanes.
// The MFMA emulation follows the operand / accumulator lane maps of
// /opt/skills/guides/cdna_hip_programming.md section 3 (v_mfma_f32_32x32x2_f32, 16x16x4_f32).
#pragma once
#include <algorithm>
#include <atomic>
#include <barrier>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <pthread.h>
#include <sched.h>
#include <unistd.h>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#ifndef __restrict__
#define __restrict__ __restrict
#endif

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }

struct f32x4_emu { float v[4]; float &operator[](int i) { return v[i]; } const float &operator[](int i) const { return v[i]; } };
struct f32x16_emu { float v[16]; float &operator[](int i) { return v[i]; } const float &operator[](int i) const { return v[i]; } };

typedef void *hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return 0; }
static inline const char *hipGetErrorString(hipError_t) { return "emu"; }

namespace hipemu {

struct Wave {
  std::barrier<> bar{64};
  float f[64];
  float g[64];
  unsigned short a8[64][8], b8[64][8];  // operands of the 8-element (bf16) MFMA forms
  unsigned long long mask;
  int votes[64];
};

struct Block {
  unsigned nthreads = 0;
  std::unique_ptr<std::barrier<>> bar;
  std::vector<std::unique_ptr<Wave>> waves;
  std::vector<unsigned char> lds;
  std::mutex mu;  // MFN_EMU_LOCK: sections that the hardware orders through its in-order LDS pipe
};

extern thread_local dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
extern thread_local Block *t_block;
extern thread_local unsigned t_lane, t_wave;

inline void *dyn_shared() { return t_block->lds.data(); }
inline Wave &wave() { return *t_block->waves[t_wave]; }

// names of the kernels launched since the last query (emu_api.cpp: mfn_emu_test_launch_log)
inline std::string &launch_log() { static std::string log; return log; }
inline void note_launch(const char *name) { launch_log() += name; launch_log() += ';'; }

template <class F>
void launch(dim3 grid, dim3 block, size_t shmem, F &&body) {
  const unsigned nthreads = block.x * block.y * block.z;
  if (nthreads % 64 != 0) {
    fprintf(stderr, "hipemu: block size %u is not a multiple of 64\n", nthreads);
    abort();
  }
  Block blk;
  blk.nthreads = nthreads;
  blk.bar.reset(new std::barrier<>(nthreads));
  for (unsigned w = 0; w < nthreads / 64; ++w) blk.waves.emplace_back(new Wave());
  blk.lds.assign(shmem + 64, 0);
  std::vector<std::thread> pool;
  pool.reserve(nthreads);
  // The lanes spend their time handing barriers to each other: on ONE core a hand-over is a context switch, across cores it
  // is a futex wake-up plus a migration (measured on 8 cores: the sequential CPU suite 6 min -> 2 min 11 s, system time
  // 12 min -> 1.5 min).  Every process picks its core from its pid (pytest -n workers spread out); MFN_EMU_PIN=0 leaves the
  // scheduler alone.
  static const int pin_core = []() {
    const char *e = getenv("MFN_EMU_PIN");
    if (e && e[0] == '0') return -1;
    cpu_set_t cur;
    CPU_ZERO(&cur);
    if (sched_getaffinity(0, sizeof(cur), &cur) != 0) return -1;
    const int want = (int)(getpid() % CPU_SETSIZE);
    int k = 0, n = CPU_COUNT(&cur);
    if (n <= 0) return -1;
    for (int c = 0; c < CPU_SETSIZE; ++c)
      if (CPU_ISSET(c, &cur) && k++ == want % n) return c;
    return -1;
  }();
  for (unsigned t = 0; t < nthreads; ++t) {
    pool.emplace_back([&, t]() {
      if (pin_core >= 0) {
        cpu_set_t one;
        CPU_ZERO(&one);
        CPU_SET(pin_core, &one);
        pthread_setaffinity_np(pthread_self(), sizeof(one), &one);
      }
      t_block = &blk;
      t_blockDim = block;
      t_gridDim = grid;
      t_threadIdx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
      t_lane = t % 64;
      t_wave = t / 64;
      for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
          for (unsigned bx = 0; bx < grid.x; ++bx) {
            t_blockIdx = dim3(bx, by, bz);
            body();
            blk.bar->arrive_and_wait();  // block boundary: LDS is reused by the next block
          }
    });
  }
  for (auto &th : pool) th.join();
}

}  // namespace hipemu

#define threadIdx (hipemu::t_threadIdx)
#define blockIdx (hipemu::t_blockIdx)
#define blockDim (hipemu::t_blockDim)
#define gridDim (hipemu::t_gridDim)

static inline void __syncthreads() { hipemu::t_block->bar->arrive_and_wait(); }

static inline float __shfl(float v, int src, int width = 64) {
  hipemu::Wave &w = hipemu::wave();
  const unsigned lane = hipemu::t_lane;
  w.f[lane] = v;
  w.bar.arrive_and_wait();
  const int base = (int)(lane / width) * width;
  const float r = w.f[base + (src % width)];
  w.bar.arrive_and_wait();
  return r;
}
static inline int __shfl(int v, int src, int width = 64) {
  float f;
  memcpy(&f, &v, 4);
  f = __shfl(f, src, width);
  memcpy(&v, &f, 4);
  return v;
}
template <class T> static inline T __shfl_xor(T v, int m, int width = 64) { return __shfl(v, (int)(hipemu::t_lane % width) ^ m, width); }
template <class T> static inline T __shfl_down(T v, int d, int width = 64) {
  const int l = (int)(hipemu::t_lane % width);
  return __shfl(v, (l + d < width) ? l + d : l, width);
}
static inline unsigned long long __ballot(int pred) {
  hipemu::Wave &w = hipemu::wave();
  w.votes[hipemu::t_lane] = pred ? 1 : 0;
  w.bar.arrive_and_wait();
  unsigned long long m = 0;
  for (int i = 0; i < 64; ++i) m |= (unsigned long long)w.votes[i] << i;
  w.bar.arrive_and_wait();
  return m;
}
static inline int __all(int pred) { return __ballot(pred) == ~0ull; }
static inline int __any(int pred) { return __ballot(pred) != 0ull; }
static inline int __syncthreads_and(int pred) {
  // emulated through LDS-free voting: ballot per wave, then a block-wide reduction
  static thread_local int dummy;
  (void)dummy;
  hipemu::Block *b = hipemu::t_block;
  int wave_ok = __all(pred);
  // publish per-wave result in the wave buffer, then read all after a block barrier
  hipemu::wave().votes[0] = wave_ok;
  b->bar->arrive_and_wait();
  int ok = 1;
  for (auto &w : b->waves) ok &= w->votes[0];
  b->bar->arrive_and_wait();
  return ok;
}

static inline float atomicAdd(float *p, float v) { return std::atomic_ref<float>(*p).fetch_add(v, std::memory_order_relaxed); }

// v_mfma_f32_32x32x2_f32: A lane l = A[i=l&31][k=l>>5], B lane l = B[k=l>>5][j=l&31],
// D reg r of lane l = D[row=(r&3)+8*(r>>2)+4*(l>>5)][col=l&31].  k-ordered fmaf chain.
static inline f32x16_emu hipemu_mfma_32x32x2(float a, float b, f32x16_emu c) {
  hipemu::Wave &w = hipemu::wave();
  const unsigned lane = hipemu::t_lane;
  w.f[lane] = a;
  w.g[lane] = b;
  w.bar.arrive_and_wait();
  const int col = lane & 31, hi = lane >> 5;
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
    float acc = c[r];
    for (int k = 0; k < 2; ++k) acc = fmaf(w.f[row + 32 * k], w.g[col + 32 * k], acc);
    c[r] = acc;
  }
  w.bar.arrive_and_wait();
  return c;
}
// v_mfma_f32_32x32x16_bf16: lane l holds eight bf16 of A[i=l&31][.] resp. B[.][j=l&31] for k-block l>>5; element e of a
// k-block of A meets element e of the same k-block of B.  Products of two bf16 are exact in fp32; the hardware's summation
// order inside the instruction is not documented -- here k-ordered fp32 adds (tests compare with a tolerance).
struct bf16x8_emu { unsigned short v[8]; };
static inline float hipemu_bf16_to_f32(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
static inline unsigned short hipemu_f32_to_bf16(float f) {   // round to nearest even, as v_cvt_pk_bf16_f32
  unsigned u; memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
static inline f32x16_emu hipemu_mfma_32x32x16_bf16(bf16x8_emu a, bf16x8_emu b, f32x16_emu c) {
  hipemu::Wave &w = hipemu::wave();
  const unsigned lane = hipemu::t_lane;
  for (int e = 0; e < 8; ++e) { w.a8[lane][e] = a.v[e]; w.b8[lane][e] = b.v[e]; }
  w.bar.arrive_and_wait();
  const int col = lane & 31, hi = lane >> 5;
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
    float acc = c[r];
    for (int kb = 0; kb < 2; ++kb)
      for (int e = 0; e < 8; ++e)
        acc += hipemu_bf16_to_f32(w.a8[row + 32 * kb][e]) * hipemu_bf16_to_f32(w.b8[col + 32 * kb][e]);
    c[r] = acc;
  }
  w.bar.arrive_and_wait();
  return c;
}
// v_mfma_f32_16x16x32_bf16: lane l holds eight bf16 of A[i=l&15][.] resp. B[.][j=l&15] for k-block l>>4 (four k-blocks of
// eight); element e of a k-block of A meets element e of the same k-block of B.  D reg r of lane l = D[row=(l>>4)*4+r][col=l&15].
static inline f32x4_emu hipemu_mfma_16x16x32_bf16(bf16x8_emu a, bf16x8_emu b, f32x4_emu c) {
  hipemu::Wave &w = hipemu::wave();
  const unsigned lane = hipemu::t_lane;
  for (int e = 0; e < 8; ++e) { w.a8[lane][e] = a.v[e]; w.b8[lane][e] = b.v[e]; }
  w.bar.arrive_and_wait();
  const int col = lane & 15, grp = lane >> 4;
  for (int r = 0; r < 4; ++r) {
    const int row = grp * 4 + r;
    float acc = c[r];
    for (int kb = 0; kb < 4; ++kb)
      for (int e = 0; e < 8; ++e)
        acc += hipemu_bf16_to_f32(w.a8[row + 16 * kb][e]) * hipemu_bf16_to_f32(w.b8[col + 16 * kb][e]);
    c[r] = acc;
  }
  w.bar.arrive_and_wait();
  return c;
}
// v_mfma_f32_4x4x4_16B_bf16: sixteen independent 4 x 4 x 4 products, block b = lane / 4.  A: lane (b, i) holds row i's four K values;
// B: lane (b, j) holds column j's four K values; C / D: lane (b, j), register i = element (i, j).
struct bf16x4_emu { unsigned short v[4]; };
static inline f32x4_emu hipemu_mfma_4x4x4_bf16(bf16x4_emu a, bf16x4_emu b, f32x4_emu c) {
  hipemu::Wave &w = hipemu::wave();
  const unsigned lane = hipemu::t_lane;
  for (int e = 0; e < 4; ++e) { w.a8[lane][e] = a.v[e]; w.b8[lane][e] = b.v[e]; }
  w.bar.arrive_and_wait();
  const unsigned blk = lane & ~3u;
  for (int i = 0; i < 4; ++i) {
    float acc = c[i];
    for (int k = 0; k < 4; ++k) acc += hipemu_bf16_to_f32(w.a8[blk + i][k]) * hipemu_bf16_to_f32(w.b8[lane][k]);
    c[i] = acc;
  }
  w.bar.arrive_and_wait();
  return c;
}
// DPP row_shl:n -- lane i of every 16-lane row receives the value of lane i+n of the same row; lanes whose source is
// outside the row keep `old` (bound_ctrl off)
static inline float hipemu_dpp_row_shl(float old, float src, int n) {
  hipemu::Wave &w = hipemu::wave();
  const unsigned lane = hipemu::t_lane;
  w.f[lane] = src;
  w.bar.arrive_and_wait();
  const float r = ((lane & 15) + n <= 15) ? w.f[lane + n] : old;
  w.bar.arrive_and_wait();
  return r;
}
// DPP wave_shr:1 (dir = -1: lane i receives lane i-1) / wave_shl:1 (dir = +1: lane i+1); the lane without a source keeps `old`
static inline float hipemu_dpp_wave_shift(float old, float src, int dir) {
  hipemu::Wave &w = hipemu::wave();
  const int lane = (int)hipemu::t_lane;
  w.f[lane] = src;
  w.bar.arrive_and_wait();
  const int from = lane + dir;
  const float r = (from >= 0 && from < 64) ? w.f[from] : old;
  w.bar.arrive_and_wait();
  return r;
}
// v_mfma_f32_16x16x4_f32: A lane l = A[i=l&15][k=l>>4], B lane l = B[k=l>>4][j=l&15],
// D reg r of lane l = D[row=(l>>4)*4+r][col=l&15].
static inline f32x4_emu hipemu_mfma_16x16x4(float a, float b, f32x4_emu c) {
  hipemu::Wave &w = hipemu::wave();
  const unsigned lane = hipemu::t_lane;
  w.f[lane] = a;
  w.g[lane] = b;
  w.bar.arrive_and_wait();
  const int col = lane & 15, grp = lane >> 4;
  for (int r = 0; r < 4; ++r) {
    const int row = grp * 4 + r;
    float acc = c[r];
    for (int k = 0; k < 4; ++k) acc = fmaf(w.f[row + 16 * k], w.g[col + 16 * k], acc);
    c[r] = acc;
  }
  w.bar.arrive_and_wait();
  return c;
}
