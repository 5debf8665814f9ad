// Microbenchmark: fp32 atomic-add throughput on gfx950 by ACCESS PATTERN of the wave instruction, global and LDS.
//   global: a wave instruction's 64 lanes cover R rows x (64/R) consecutive floats of a 2-D buffer (row pitch W floats) at a
//           pseudo-random origin -- the shape of a gradient-window flush -- R = 1, 2, 4, 8, 64 (64 = fully scattered)
//   lds:    ds_add_f32 (no return) with lanes on distinct banks, with a plane stride of 241 floats (the layout of the
//           privatised scatter), and ds_read + add + ds_write of the same cells for comparison
// Build: hipcc --offload-arch=gfx950 -O3 -o atomic_patterns atomic_patterns.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int R, bool STORE, int KEEP = 1>
__global__ __launch_bounds__(256) void gk(float *buf, int H, int W, int iters) {
  const int lane = threadIdx.x & 63;
  const int wv = (blockIdx.x * 256 + threadIdx.x) >> 6;
  const int row = lane / (64 / R), col = lane % (64 / R);
  unsigned s = wv * 2654435761u + 12345u;
  for (int i = 0; i < iters; ++i) {
    s = s * 1664525u + 1013904223u;
    size_t idx;
    if (R == 64) {
      unsigned s2 = (s + lane * 40503u) * 2246822519u;
      idx = (size_t)((s2 >> 7) % (unsigned)(H * W));
    } else {
      const int y0 = (int)((s >> 8) % (unsigned)(H - R)), x0 = (int)(((s >> 3) * 7u) % (unsigned)(W - 64 / R));
      idx = (size_t)(y0 + row) * W + x0 + col;
    }
    if (KEEP > 1 && (lane % KEEP) != 0) continue;  // only every KEEP-th lane of the pattern is active
    if (STORE) buf[idx] = 1.0f;
    else atomicAdd(buf + idx, 1.0f);
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void lk(float *out, int iters) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float *pl = lds + wave * 32 * 241 + (lane & 31) * 241 + (lane >> 5) * 97;  // lane = channel plane, the halves 97 cells apart
  for (int e = threadIdx.x; e < 4 * 32 * 241; e += 256) lds[e] = 0.f;
  __syncthreads();
  float g = 1.0f;
  for (int i = 0; i < iters; ++i) {
    const int cell = (i * 7) % 120;
    if (MODE == 0) {
#pragma unroll
      for (int u = 0; u < 16; ++u) atomicAdd(pl + cell + (u >> 2) * 24 + (u & 3), g);
    } else if (MODE == 1) {
      float o[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) o[u] = pl[cell + (u >> 2) * 24 + (u & 3)];
#pragma unroll
      for (int u = 0; u < 16; ++u) pl[cell + (u >> 2) * 24 + (u & 3)] = o[u] + g;
    } else {
      // lanes = pixels of one plane: consecutive lanes, consecutive cells (distinct addresses, distinct banks)
      float *pp = lds + wave * 1024 + (lane & 7) + (lane >> 3) * 24;
#pragma unroll
      for (int u = 0; u < 16; ++u) atomicAdd(pp + (cell & 63) + (u >> 2) * 24 + (u & 3), g);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = lds[5];
}

template <int R, bool STORE, int KEEP = 1> void grun(const char *name) {
  const int H = 8 * 32 * 96, W = 128;  // the level-2 input gradient: 12.6 MB
  float *d;
  (void)hipMalloc(&d, (size_t)H * W * 4);
  (void)hipMemset(d, 0, (size_t)H * W * 4);
  const int blocks = 2048, iters = 128;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((gk<R, STORE, KEEP>), dim3(blocks), dim3(256), 0, 0, d, H, W, iters);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL((gk<R, STORE, KEEP>), dim3(blocks), dim3(256), 0, 0, d, H, W, iters);
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  const double instr = (double)blocks * 4 * iters, lanes = instr * 64;
  printf("%-44s %8.3f ms  %7.1f G lanes/s  %6.2f G wave-instr/s  (%d row segment%s of %d B per instruction)\n", name, ms,
         lanes / ms / 1e6, instr / ms / 1e6, R == 64 ? 64 : R, R == 1 ? "" : "s", R == 64 ? 4 : 256 / R);
  (void)hipFree(d);
}
template <int MODE> void lrun(const char *name) {
  float *d; (void)hipMalloc(&d, 4096 * 4);
  const int blocks = 1024, iters = 256;
  const size_t shm = 4 * 32 * 241 * 4;
  (void)hipFuncSetAttribute((const void *)lk<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(lk<MODE>, dim3(blocks), dim3(256), shm, 0, d, iters);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(lk<MODE>, dim3(blocks), dim3(256), shm, 0, d, iters);
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  // one block per CU at a time (123 KB of LDS): 4 waves share the CU's LDS pipe
  const double per_cu_instr = (double)blocks / 256.0 * 4 * iters * 16;
  printf("%-44s %8.3f ms  %6.1f shader cycles (at 2.1 GHz) per 64-lane LDS op of a CU\n", name, ms, ms * 1e-3 * 2.1e9 / per_cu_instr);
  (void)hipFree(d);
}
int main() {
  grun<1, false>("global atomicAdd, 1 row x 64 floats");
  grun<2, false>("global atomicAdd, 2 rows x 32 floats");
  grun<4, false>("global atomicAdd, 4 rows x 16 floats");
  grun<8, false>("global atomicAdd, 8 rows x 8 floats");
  grun<64, false>("global atomicAdd, 64 scattered floats");
  grun<4, false, 2>("global atomicAdd, 4 rows x 16, every 2nd lane");
  grun<4, false, 4>("global atomicAdd, 4 rows x 16, every 4th lane");
  grun<1, false, 4>("global atomicAdd, 1 row x 64, every 4th lane");
  grun<4, true>("global store, 4 rows x 16 floats");
  grun<64, true>("global store, 64 scattered floats");
  lrun<0>("LDS atomicAdd, lane = channel plane (stride 241)");
  lrun<1>("LDS read + add + write, same cells");
  lrun<2>("LDS atomicAdd, lanes = 8x8 adjacent cells of one plane");
  return 0;
}
