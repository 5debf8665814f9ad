// Microbenchmark: on which SIMD of its CU does each wave of a 5-wave (320-thread) workgroup land when 768 such
// groups (3 per CU, the level-2 correlation launch) are resident?  Reads HW_REG_HW_ID / XCC_ID per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
extern __shared__ float lds[];
template <int NW>
__global__ __launch_bounds__(NW * 64) void k(unsigned *out, int spin) {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  lds[threadIdx.x] = (float)hw;
  __syncthreads();
  float acc = lds[(threadIdx.x * 7) % (NW * 64)];
  for (int i = 0; i < spin; ++i) acc = acc * 1.0001f + 0.5f;   // keep the blocks resident together
  if ((threadIdx.x & 63) == 0) {
    out[(blockIdx.x * NW + (threadIdx.x >> 6)) * 2] = hw;
    out[(blockIdx.x * NW + (threadIdx.x >> 6)) * 2 + 1] = xcc;
  }
  if (acc == 1.2345f) out[0] = 0;
}
template <int NW> void run(int blocks, size_t ldsb) {
  unsigned *d; (void)hipMalloc(&d, blocks * NW * 8);
  (void)hipFuncSetAttribute((const void *)k<NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
  hipLaunchKernelGGL(k<NW>, dim3(blocks), dim3(NW * 64), ldsb, 0, d, 20000);
  (void)hipDeviceSynchronize();
  std::vector<unsigned> h(blocks * NW * 2);
  (void)hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  std::map<unsigned, std::vector<int>> per_cu;   // cu key -> waves per SIMD
  std::map<unsigned, int> blocks_per_cu;
  for (int b = 0; b < blocks; ++b)
    for (int w = 0; w < NW; ++w) {
      const unsigned hw = h[(b * NW + w) * 2], xcc = h[(b * NW + w) * 2 + 1] & 0xF;
      const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
      const unsigned key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
      auto &v = per_cu[key];
      if (v.empty()) v.assign(4, 0);
      v[simd]++;
      if (w == 0) blocks_per_cu[key]++;
    }
  std::map<std::vector<int>, int> hist;
  for (auto &kv : per_cu) { auto v = kv.second; hist[v]++; }
  printf("%d-wave groups, %d blocks, LDS %zu B: %zu CUs used\n", NW, blocks, ldsb, per_cu.size());
  for (auto &kv : hist) printf("  waves on SIMD0..3 = %d %d %d %d : %d CUs\n", kv.first[0], kv.first[1], kv.first[2], kv.first[3], kv.second);
  std::map<int, int> bh;
  for (auto &kv : blocks_per_cu) bh[kv.second]++;
  for (auto &kv : bh) printf("  %d blocks on a CU: %d CUs\n", kv.first, kv.second);
  (void)hipFree(d);
}
int main() {
  run<5>(768, 30720);
  run<5>(768, 46080);
  run<4>(768, 30720);
  run<4>(1024, 30720);
  run<8>(512, 30720);
  return 0;
}
