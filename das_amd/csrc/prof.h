// Launch timing inside the library (das_prof_*, include/das_hip.h): when switched on, every C entry point that
// launches kernels records one HIP event right before its first launch and one right after its last, on the stream the
// launches go to — no interpreter or ctypes time between an event and the launch it brackets (bench.py's per-family
// pricing used to record its events from Python around the ctypes call; on a slow host that priced host time as kernel
// time). Off (the default) a scope costs one relaxed load.
#pragma once
#include <hip/hip_runtime.h>

namespace dasprof {
extern bool g_on;
struct Scope {
  long long idx;
  unsigned long long notes;   // dastune::note_count() when the scope opened
  bool counted;   // this scope raised the thread's nesting depth (an entry point that calls another one is ONE record)
  hipStream_t s;
  Scope(void* stream, const char* entry) : idx(-1), notes(0), counted(false), s((hipStream_t)stream) {
    if (g_on) open(entry);
  }
  ~Scope() {
    if (counted) close();
  }
  void open(const char* entry);
  void close();
};
}  // namespace dasprof

#define DAS_PROF(stream) dasprof::Scope das_prof_scope__((stream), __func__)
