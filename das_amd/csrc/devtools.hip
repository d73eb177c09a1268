// Measurement aids exported through the C ABI (no reference counterpart; nothing on the product path calls them).
#include "common.h"

namespace {
// `blocks` workgroups that hold their CU slot (threads, LDS) for `usec` microseconds of the 100 MHz wall clock.
__global__ void spin_kernel(long long ticks, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned long long t0 = wall_clock64();
  unsigned v = 0;
  while ((long long)(wall_clock64() - t0) < ticks) {
    v += (unsigned)smem[(threadIdx.x * 4) & 1023];    // (keeps the LDS allocation alive)
    __builtin_amdgcn_s_sleep(32);
  }
  if (sink && v == 0xFFFFFFFFu) *sink = v;
}
}  // namespace

extern "C" int das_dev_occupy_cus(int blocks, int threads, int lds_bytes, int usec, void* stream) {
  if (blocks < 1 || threads < 64 || threads > 1024 || threads % 64 || lds_bytes < 1024 || lds_bytes > 160 * 1024 || usec < 1)
    return DAS_ERR_ARG;
  static int attr_bytes = 0;
  if (lds_bytes > attr_bytes) {
    if (hipFuncSetAttribute((const void*)spin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess)
      return DAS_ERR_LAUNCH;
    attr_bytes = lds_bytes;
  }
  hipLaunchKernelGGL(spin_kernel, dim3((unsigned)blocks), dim3((unsigned)threads), (size_t)lds_bytes, (hipStream_t)stream,
                     (long long)usec * 100, (unsigned*)nullptr);
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}
