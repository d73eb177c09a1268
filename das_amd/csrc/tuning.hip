// Host-side dispatch tuning table and last-kernel record (see tuning.h; C ABI in include/das_hip.h).
#include "tuning.h"

#include <atomic>
#include <cstring>

#include "das_hip.h"

#include <hip/hip_runtime.h>

namespace dastune {
namespace {
struct Entry {
  const char* name;
  long long def;
};
// (defaults: break-even points measured with cold operands, tools/dev/conv_cold_bench.py / wgrad_cold_bench.py)
constexpr Entry kTable[N_KEYS] = {
    {"conv.big_minblocks", 100},  {"conv.big_mink", 0},       {"conv.glds3_pp_mink", 1024},
    {"conv.glds4_minblocks", 128}, {"conv.glds4_pp", -1},      {"conv.glds4_mf", 0},
    {"conv.stream_minrows", 16384}, {"conv.stream_percu", 1}, {"conv.tail_split", 1},        {"conv.splitk_target", 256},
    {"conv.splitk_minsteps", 12}, {"conv.splitk_kernels", 3},  {"wgrad.pp_mink", 256},  {"wgrad.shapes", 1},
    {"wgrad.bkm", 32},            {"wgrad.blocks", 0},        {"wgrad.pp_blocks", 0},      {"bn.reduce_blocks", 256},     {"bn.reduce_threads", 256},
    {"bn.vpt", 8},                {"gn.ppb", 0},             {"conv.c64_mintiles", 64},    {"bn.stream_minbytes", 96 << 20},
    {"comm.reserved_cus", 0},  {"elem.upstats_ppb", 0},   {"bn.upmerge_blocks", 512}, {"dcn.fused_minrows", 100000},
    {"conv.balance_rows", 1},     {"conv.glds4_mfma32", 0},  {"conv.kstream", 35},       {"conv.splitk_inkernel", 0}, {"conv.stream_nt", 0},     {"bn.nt_fwd", 0},          {"bn.nt_bwd", 0},
    {"conv.stem7x7", 1},
};
std::atomic<long long> g_val[N_KEYS];
std::atomic<bool> g_init{false};
void init_once() {
  if (g_init.load(std::memory_order_acquire)) return;
  for (int i = 0; i < N_KEYS; ++i) g_val[i].store(kTable[i].def, std::memory_order_relaxed);
  g_init.store(true, std::memory_order_release);
}
thread_local const char* t_last = "";
thread_local unsigned long long t_notes = 0;   // note_kernel calls of this thread
}  // namespace

long long get(Key k) {
  init_once();
  return g_val[k].load(std::memory_order_relaxed);
}
void note_kernel(const char* name) {
  t_last = name;
  ++t_notes;
}
unsigned long long note_count() { return t_notes; }
int device_cus() {
  static std::atomic<int> cus{0};
  int c = cus.load(std::memory_order_relaxed);
  if (c == 0) {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    c = n > 0 ? n : 256;
    cus.store(c, std::memory_order_relaxed);
  }
  return c;
}
int usable_cus() {
  const long long r = get(COMM_RESERVED_CUS);
  const int c = device_cus();
  const long long u = c - (r > 0 ? r : 0);
  return (int)(u < 8 ? 8 : u);
}
static thread_local int t_pp_share_den = 1;
int pp_share_den() { return t_pp_share_den; }
}  // namespace dastune

using namespace dastune;

extern "C" int das_wgrad_pp_share(int den) {
  if (den < 1 || den > 16) return DAS_ERR_ARG;
  dastune::t_pp_share_den = den;
  return DAS_OK;
}

extern "C" int das_tuning_set(const char* key, long long value) {
  if (!key) return DAS_ERR_ARG;
  init_once();
  for (int i = 0; i < N_KEYS; ++i)
    if (!strcmp(key, kTable[i].name)) {
      g_val[i].store(value, std::memory_order_relaxed);
      return DAS_OK;
    }
  return DAS_ERR_ARG;
}

extern "C" int das_tuning_get(const char* key, long long* value) {
  if (!key || !value) return DAS_ERR_ARG;
  init_once();
  for (int i = 0; i < N_KEYS; ++i)
    if (!strcmp(key, kTable[i].name)) {
      *value = g_val[i].load(std::memory_order_relaxed);
      return DAS_OK;
    }
  return DAS_ERR_ARG;
}

extern "C" int das_tuning_reset(void) {
  init_once();
  for (int i = 0; i < N_KEYS; ++i) g_val[i].store(kTable[i].def, std::memory_order_relaxed);
  return DAS_OK;
}

extern "C" const char* das_last_kernel(void) { return t_last; }
