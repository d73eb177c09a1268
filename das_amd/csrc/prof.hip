// das_prof_*: HIP event pairs recorded inside the library around every entry point's launches (prof.h).
#include <hip/hip_runtime.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "das_hip.h"
#include "prof.h"
#include "tuning.h"

namespace dasprof {
bool g_on = false;

namespace {
struct Rec {
  hipEvent_t e0, e1;
  const char* entry;   // __func__ of the entry point (static storage)
  char kernel[64];     // das_last_kernel() when the scope closed (conv / weight-gradient launchers), else ""
  bool closed;
};
std::mutex g_mu;
std::vector<Rec> g_rec;   // event pairs are created on first use and reused by later passes
long long g_n = 0;        // records of the current pass
thread_local int t_depth = 0;  // an entry point that calls another one is ONE record
}  // namespace

void Scope::open(const char* entry) {
  counted = true;
  if (t_depth++ > 0) return;
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_on) return;
  if ((size_t)g_n == g_rec.size()) {
    Rec r;
    memset(&r, 0, sizeof(r));
    // (no system-scope fence when an event completes: nothing on the host reads device memory behind these events)
    if (hipEventCreateWithFlags(&r.e0, hipEventDisableSystemFence) != hipSuccess ||
        hipEventCreateWithFlags(&r.e1, hipEventDisableSystemFence) != hipSuccess)
      return;
    g_rec.push_back(r);
  }
  Rec& r = g_rec[g_n];
  r.entry = entry;
  r.kernel[0] = 0;
  r.closed = false;
  if (hipEventRecord(r.e0, s) != hipSuccess) return;
  notes = dastune::note_count();
  idx = g_n++;
}

void Scope::close() {
  --t_depth;
  if (idx < 0) return;   // nested scope, or the pass ended / an event failed while opening
  std::lock_guard<std::mutex> lk(g_mu);
  if (idx < g_n) {
    Rec& r = g_rec[idx];
    (void)hipEventRecord(r.e1, s);
    if (dastune::note_count() != notes) {   // a launcher inside this scope picked a kernel: the record carries its name
      const char* k = das_last_kernel();
      if (k) strncpy(r.kernel, k, sizeof(r.kernel) - 1);
    }
    r.closed = true;
  }
}
}  // namespace dasprof

using namespace dasprof;

extern "C" int das_prof_begin(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_n = 0;
  g_on = true;
  return DAS_OK;
}

extern "C" int das_prof_end(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_on = false;
  return DAS_OK;
}

extern "C" long long das_prof_count(void) { return g_n; }

extern "C" int das_prof_read(float* ms, char* names, int name_stride, long long n) {
  if (!ms || n < 0 || (names && name_stride < 16)) return DAS_ERR_ARG;
  std::lock_guard<std::mutex> lk(g_mu);
  if (n > g_n) return DAS_ERR_ARG;
  for (long long i = 0; i < n; ++i) {
    Rec& r = g_rec[i];
    ms[i] = -1.f;
    if (r.closed && hipEventSynchronize(r.e1) == hipSuccess) {
      float t = 0.f;
      if (hipEventElapsedTime(&t, r.e0, r.e1) == hipSuccess) ms[i] = t;
    }
    if (names) {
      char* dst = names + i * (long long)name_stride;
      const char* src = r.kernel[0] ? r.kernel : (r.entry ? r.entry : "");
      strncpy(dst, src, name_stride - 1);
      dst[name_stride - 1] = 0;
    }
  }
  return DAS_OK;
}
