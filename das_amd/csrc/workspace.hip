// See workspace.h.
#include "workspace.h"

#include <algorithm>
#include <mutex>

namespace dasws {
float* get(Kind kind, hipStream_t s, size_t bytes, size_t min_bytes) {
  struct Entry { int dev; hipStream_t s; int kind; float* buf; size_t bytes; bool used; };
  static Entry table[32];
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  Entry* e = nullptr;
  for (Entry& t : table)
    if (t.used && t.dev == dev && t.s == s && t.kind == kind) { e = &t; break; }
  if (!e) {
    for (Entry& t : table)
      if (!t.used) { e = &t; break; }
    if (!e) {   // table full (streams that came and went): take over the first entry
      e = &table[0];
      if (hipDeviceSynchronize() != hipSuccess) return nullptr;
      if (e->buf) { (void)hipFree(e->buf); e->buf = nullptr; e->bytes = 0; }
      e->dev = dev; e->s = s; e->kind = kind;
    } else {
      *e = Entry{dev, s, kind, nullptr, 0, true};
    }
  }
  if (e->bytes < bytes) {
    if (e->buf) {
      if (hipStreamSynchronize(s) != hipSuccess || hipFree(e->buf) != hipSuccess) return nullptr;
      e->buf = nullptr; e->bytes = 0;
    }
    const size_t want = std::max(bytes, min_bytes);
    if (hipMalloc(reinterpret_cast<void**>(&e->buf), want) != hipSuccess) { e->buf = nullptr; return nullptr; }
    e->bytes = want;
    if (kind == SPLITK_CNT && hipMemsetAsync(e->buf, 0, want, s) != hipSuccess) return nullptr;   // (ordered before the first user on s)
  }
  return e->buf;
}
}  // namespace dasws
