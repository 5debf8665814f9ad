// emu_api.cpp -- libmfn_emu.so: the kernels of maskflownet_amd/csrc/kernels/ compiled by g++ on
// top of hipemu.h, behind the same argument checking / dispatch code as the product library
// (api_impl.inc), exported as mfn_emu_*.  TEST INFRASTRUCTURE ONLY: lets the CPU-only CI check
// kernel logic against the oracle.  Pointers are HOST pointers; `stream` is ignored.
#include "../../include/mfn_hip.h"
#define MFN_API(name) mfn_emu_##name
#include "hipemu.h"

namespace hipemu {
thread_local dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
thread_local Block *t_block = nullptr;
thread_local unsigned t_lane = 0, t_wave = 0;
}  // namespace hipemu
using std::max;
using std::min;

#include "../../maskflownet_amd/csrc/api_impl.inc"

// the kernels the calls since the last query dispatched to, "name;name;..." (cleared by the query): which path a call took
extern "C" int mfn_emu_test_launch_log(char *buf, int cap) {
  std::string &log = hipemu::launch_log();
  const int n = (int)log.size();
  if (buf && cap > 0) {
    const int m = n < cap - 1 ? n : cap - 1;
    memcpy(buf, log.data(), (size_t)m);
    buf[m] = 0;
  }
  log.clear();
  return n;
}
