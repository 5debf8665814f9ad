// api.hip -- libmfn_hip.so: the C ABI of include/mfn_hip.h for gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC (see build.py).
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mfn_hip.h"
#define MFN_API(name) mfn_##name
#include "mfn_rt.h"

namespace mfn {
// ---- built-in kernel timer ------------------------------------------------------------------------
struct ProfRec { std::string name; hipEvent_t e0, e1; };
static std::mutex g_prof_mu;
static std::vector<ProfRec> g_prof;
static bool g_prof_on = false;
static std::string g_prof_tag;  // appended to the names of the records taken while it is set (mfn_profile_tag)
bool profile_enabled() { return g_prof_on; }
void profile_record(const char *name, hipEvent_t e0, hipEvent_t e1) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof.push_back(ProfRec{g_prof_tag.empty() ? std::string(name) : std::string(name) + "@" + g_prof_tag, e0, e1});
}
}  // namespace mfn

#include "api_impl.inc"

extern "C" {

int mfn_abi_version(void) { return MFN_ABI_VERSION; }
#ifndef MFN_SOURCE_HASH
#define MFN_SOURCE_HASH "unhashed"
#endif
// MFN_SOURCE_HASH: sha256 (first 16 hex digits) over csrc/** and include/mfn_hip.h, put on the compiler command line by
// maskflownet_amd/_lib.py:build(); _lib.lib() refuses a library whose hash differs from the sources next to it.
const char *mfn_version_string(void) { return "mfn_hip 0.2 (gfx950) src=" MFN_SOURCE_HASH; }

// ---- hipGraph plumbing -------------------------------------------------------------------------------
int mfn_graph_begin_capture(void *stream) {
  return hipfail((int)hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal), "graph_begin_capture");
}
int mfn_graph_end_capture(void *stream, void **exec_out) {
  if (!exec_out) return fail(MFN_E_NULL, "graph_end_capture: NULL output");
  hipGraph_t graph = nullptr;
  int rc = (int)hipStreamEndCapture((hipStream_t)stream, &graph);
  if (rc) return hipfail(rc, "graph_end_capture");
  hipGraphExec_t exec = nullptr;
  rc = (int)hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (rc) return hipfail(rc, "graph_instantiate");
  *exec_out = (void *)exec;
  return 0;
}
int mfn_graph_launch(void *exec, void *stream) {
  if (!exec) return fail(MFN_E_NULL, "graph_launch: NULL graph");
  return hipfail((int)hipGraphLaunch((hipGraphExec_t)exec, (hipStream_t)stream), "graph_launch");
}
int mfn_graph_destroy(void *exec) {
  if (!exec) return 0;
  return hipfail((int)hipGraphExecDestroy((hipGraphExec_t)exec), "graph_destroy");
}

// ---- profiler ------------------------------------------------------------------------------------------
int mfn_profile_enable(int on) { g_prof_on = on != 0; return 0; }
int mfn_profile_tag(const char *tag) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_tag = tag ? tag : "";
  return 0;
}
int mfn_profile_reset(void) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto &r : g_prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  g_prof.clear();
  return 0;
}
int mfn_profile_query(const char *substr, int *launches, double *total_ms) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  int n = 0;
  double ms = 0.0;
  for (auto &r : g_prof) {
    if (substr && *substr && r.name.find(substr) == std::string::npos) continue;
    if (hipEventSynchronize(r.e1) != hipSuccess) continue;
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.e0, r.e1) != hipSuccess) continue;
    ms += t;
    ++n;
  }
  if (launches) *launches = n;
  if (total_ms) *total_ms = ms;
  return n;
}
int mfn_profile_dump(char *buf, int cap) {
  std::vector<std::string> names;
  {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto &r : g_prof) {
      bool seen = false;
      for (auto &n : names) seen = seen || n == r.name;
      if (!seen) names.push_back(r.name);
    }
  }
  std::string s;
  for (auto &n : names) {
    int c = 0;
    double ms = 0;
    mfn_profile_query(n.c_str(), &c, &ms);
    char line[256];
    snprintf(line, sizeof(line), "%s %d %.6f\n", n.c_str(), c, ms);
    s += line;
  }
  if (buf && cap > 0) {
    const int k = (int)s.size() < cap - 1 ? (int)s.size() : cap - 1;
    memcpy(buf, s.data(), k);
    buf[k] = 0;
  }
  return (int)s.size() + 1;
}

}  // extern "C"
