// Microbenchmark: cost of ds_read_b128 / ds_write_b128 for different lane->address patterns on gfx950.
// Measures wave-cycles per instruction with 4 or 8 waves per CU issuing back-to-back reads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];

__device__ int service_pos(int lane) {
  const unsigned long long tbl = 0x1C0C081804141000ull;
  return (int)((tbl >> (8 * ((lane >> 2) & 7))) & 0xFF) + (lane & 3) + (lane & 32);
}

// pattern -> byte address of this lane's 16-byte slot
__device__ int addr_of(int pattern, int lane, int wave) {
  const int base = wave * 4096;  // each wave its own 4 KB window (+ strides below stay inside 32 KB)
  switch (pattern) {
    case 0: return base + lane * 16;                                   // contiguous 1 KB
    case 1: return base + (lane >> 4) * 288 + (lane & 15) * 16;         // 4 rows x 256 B, row stride 288 (TW=64 natural)
    case 2: { int p = service_pos(lane); return base + (p >> 4) * 288 + (p & 15) * 16; }  // same, service order
    case 3: return base + lane * 32;                                   // 32 B lane stride
    case 4: return base + (lane >> 3) * 160 + (lane & 7) * 16;          // 8 rows x 128 B, stride 160 (TW=32 natural)
    case 5: { int p = lane; int sg = p >> 4, wi = p & 15; int row = sg + 4 * (wi >> 3), gx = wi & 7; return base + row * 160 + gx * 16; }
    case 6: { int p = service_pos(lane); int sg = p >> 4, wi = p & 15; int row = sg + 4 * (wi >> 3), gx = wi & 7; return base + row * 160 + gx * 16; }
    case 7: return base + (lane & 31) * 16 + (lane >> 5) * 1152;        // two halves in different rows (stride 1152 B)
    case 8: return base + (lane >> 4) * 256 + (lane & 15) * 16;         // == contiguous (sanity)
    case 9: return base + (lane >> 4) * 272 + (lane & 15) * 16;         // row stride 272 B (256+16)
    default: return base + lane * 16;
  }
}

template <bool WRITE>
__global__ void k(int pattern, int iters, unsigned long long* out, float* sink) {
  float* lds = reinterpret_cast<float*>(lds_raw);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = (float)i;
  __syncthreads();
  const int a = addr_of(pattern, lane, wave & 7);
  float4 acc = make_float4(0, 0, 0, 0);
  const float4* p = reinterpret_cast<const float4*>(lds_raw + a);
  float4* pw = reinterpret_cast<float4*>(lds_raw + a);
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (WRITE) {
        pw[u * 0] = acc;  // same address each time
        asm volatile("" ::: "memory");
      } else {
        float4 v;
        asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a) : "memory");
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        acc.x += v.x;
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[blockIdx.x * (blockDim.x / 64) + wave] = t1 - t0;
  if (acc.x == 12345.678f) sink[0] = acc.x + acc.y;
}

int main(int argc, char** argv) {
  unsigned long long* d_out; float* d_sink;
  (void)hipMalloc(&d_out, 4096 * 8); (void)hipMalloc(&d_sink, 64);
  const int iters = 200;
  if (argc == 4) {  // single configuration (for rocprofv3 --pmc): write pattern waves
    const int write = atoi(argv[1]), pattern = atoi(argv[2]), waves = atoi(argv[3]);
    for (int r = 0; r < 3; ++r) {
      if (write) hipLaunchKernelGGL(k<true>, dim3(256), dim3(64 * waves), 65536, 0, pattern, iters, d_out, d_sink);
      else hipLaunchKernelGGL(k<false>, dim3(256), dim3(64 * waves), 65536, 0, pattern, iters, d_out, d_sink);
    }
    (void)hipDeviceSynchronize();
    printf("single config done: 256 blocks x %d waves x %d instr\n", waves, iters * 16);
    return 0;
  }
  for (int write = 0; write < 2; ++write)
    for (int waves = 4; waves <= 16; waves *= 2)
      for (int pattern = 0; pattern < 10; ++pattern) {
        std::vector<unsigned long long> h(256 * waves);
        if (write) hipLaunchKernelGGL(k<true>, dim3(256), dim3(64 * waves), 65536, 0, pattern, iters, d_out, d_sink);
        else hipLaunchKernelGGL(k<false>, dim3(256), dim3(64 * waves), 65536, 0, pattern, iters, d_out, d_sink);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
        double s = 0; for (auto v : h) s += (double)v;
        const double per_wave = s / h.size();
        // readcyclecounter on gfx9 = s_memtime (constant 100 MHz?) or shader clock -- report raw and per-CU-instruction
        printf("%s waves/CU=%2d pattern=%d  ticks/wave=%.0f  ticks per wave-instr=%.2f  ticks per CU-instr=%.3f\n",
               write ? "WRITE" : "READ ", waves, pattern, per_wave, per_wave / (iters * 16.0), per_wave / (iters * 16.0 * waves));
      }
  return 0;
}
