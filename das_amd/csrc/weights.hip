// Once-per-optimizer-step weight packing for the whole network in ONE launch.
//
// The optimizer keeps every conv weight in the flat f32 master buffer in (Cout, KH, KW, Cin) order
// (channels-last storage of the OIHW parameter), which already is the forward kernels' operand layout.
// This kernel walks a table of such tensors and writes
//   * fwd  : the same layout cast to the activation dtype (bf16 path; the f32 path reads the master itself)
//   * dgrad: (Cin, KH, KW, Cout) with both tap axes flipped — the operand of the data-gradient conv
// both at the tensor's own element offset inside flat destination buffers, so a layer's packed weights are
// plain views. One workgroup transposes a 32x32 (Cout x Cin) tile of one tap through LDS.
#include "common.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void pack_conv_weights_kernel(const float* __restrict__ src, T* __restrict__ fwd,
                                                                T* __restrict__ dgrad,
                                                                const DasPackEntry* __restrict__ tab, int n) {
  __shared__ float tile[32][33];
  const int bid = blockIdx.x;
  int lo = 0, hi = n - 1;  // last entry with tile_start <= bid
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].tile_start <= bid) lo = mid; else hi = mid - 1;
  }
  const DasPackEntry e = tab[lo];
  const int taps = e.KH * e.KW;
  const int tiles_i = (e.I + 31) >> 5, tiles_o = (e.O + 31) >> 5;
  int t = bid - e.tile_start;
  const int ti = t % tiles_i; t /= tiles_i;
  const int to = t % tiles_o;
  const int tap = t / tiles_o;
  const int o0 = to * 32, i0 = ti * 32;
  const int c = threadIdx.x & 31, r = threadIdx.x >> 5;
  const float* s = src + e.off;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int o = o0 + r + k * 8, i = i0 + c;
    float v = 0.f;
    if (o < e.O && i < e.I) {
      const long long idx = ((long long)o * taps + tap) * e.I + i;
      v = s[idx];
      if (fwd) Elem<T>::store(fwd + e.off + idx, v);
    }
    tile[r + k * 8][c] = v;
  }
  __syncthreads();
  const int ftap = taps - 1 - tap;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = i0 + r + k * 8, o = o0 + c;
    if (o < e.O && i < e.I) Elem<T>::store(dgrad + e.off + ((long long)i * taps + ftap) * e.O + o, tile[c][r + k * 8]);
  }
}

}  // namespace

extern "C" int das_pack_conv_weights(const float* flat_src, void* fwd_dst, void* dgrad_dst, int dtype,
                                     const DasPackEntry* entries_dev, int n_entries, int total_tiles, void* stream) {
  if (!flat_src || !dgrad_dst || !entries_dev || n_entries < 1 || total_tiles < 1) return DAS_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == DAS_BF16) {
    hipLaunchKernelGGL(pack_conv_weights_kernel<bf16_t>, dim3(total_tiles), dim3(256), 0, s, flat_src,
                       (bf16_t*)fwd_dst, (bf16_t*)dgrad_dst, entries_dev, n_entries);
  } else if (dtype == DAS_F32) {
    hipLaunchKernelGGL(pack_conv_weights_kernel<float>, dim3(total_tiles), dim3(256), 0, s, flat_src,
                       (float*)fwd_dst, (float*)dgrad_dst, entries_dev, n_entries);
  } else {
    return DAS_ERR_ARG;
  }
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}
