// Once-per-optimizer-step weight packing for the whole network in ONE launch.
//
// The optimizer keeps every conv weight in the flat f32 master buffer in (Cout, KH, KW, Cin) order
// (channels-last storage of the OIHW parameter), which already is the forward kernels' operand layout.
// This kernel walks a table of such tensors and writes
//   * fwd  : the same layout cast to the activation dtype (bf16 path; the f32 path reads the master itself)
//   * dgrad: (Cin, KH, KW, Cout) with both tap axes flipped — the operand of the data-gradient conv
// both at the tensor's own element offset inside flat destination buffers, so a layer's packed weights are
// plain views. One workgroup transposes a 64x64 (Cout x Cin) tile of one tap through LDS.
#include "common.h"
#include "prof.h"

namespace {

// 16-byte global accesses on both sides (round 4: the 32 x 32 version moved 4-byte loads and 2-byte stores and ran at
// 2.5 TB/s of its 1.5 GB per step; a workgroup's binary search over the table — eight dependent loads — also weighed on
// 126 k four-KiB workgroups): one workgroup = a 64 x 64 (Cout x Cin) tile of one tap, float4 loads, packed 4-element
// stores, transposed through LDS for the data-gradient copy. Cout % 4 == 0 and Cin % 4 == 0 (the callers pack tensors
// whose channel counts are multiples of 8 only).
template <typename T>
struct Pack4;
template <>
struct Pack4<bf16_t> {
  static __device__ __forceinline__ void store(bf16_t* p, float a, float b, float c, float d) {
    *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16x2(a, b), pack_bf16x2(c, d));
  }
};
template <>
struct Pack4<float> {
  static __device__ __forceinline__ void store(float* p, float a, float b, float c, float d) {
    *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
  }
};

template <typename T>
__global__ __launch_bounds__(256) void pack_conv_weights_kernel(const float* __restrict__ src, T* __restrict__ fwd,
                                                                T* __restrict__ dgrad,
                                                                const DasPackEntry* __restrict__ tab, int n) {
  __shared__ float tile[64][65];
  const int bid = blockIdx.x;
  int lo = 0, hi = n - 1;  // last entry with tile_start <= bid
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].tile_start <= bid) lo = mid; else hi = mid - 1;
  }
  const DasPackEntry e = tab[lo];
  const int taps = e.KH * e.KW;
  const int tiles_i = (e.I + 63) >> 6, tiles_o = (e.O + 63) >> 6;
  int t = bid - e.tile_start;
  const int ti = t % tiles_i; t /= tiles_i;
  const int to = t % tiles_o;
  const int tap = t / tiles_o;
  const int o0 = to * 64, i0 = ti * 64;
  const int c4 = threadIdx.x & 15, r = threadIdx.x >> 4;
  const float* s = src + e.off;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int o = o0 + r + k * 16, i = i0 + c4 * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (o < e.O && i < e.I) {
      const long long idx = ((long long)o * taps + tap) * e.I + i;
      v = *reinterpret_cast<const float4*>(s + idx);
      if (fwd) Pack4<T>::store(fwd + e.off + idx, v.x, v.y, v.z, v.w);
    }
    tile[r + k * 16][c4 * 4 + 0] = v.x; tile[r + k * 16][c4 * 4 + 1] = v.y;
    tile[r + k * 16][c4 * 4 + 2] = v.z; tile[r + k * 16][c4 * 4 + 3] = v.w;
  }
  __syncthreads();
  const int ftap = taps - 1 - tap;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = i0 + r + k * 16, o = o0 + c4 * 4;
    if (o < e.O && i < e.I)
      Pack4<T>::store(dgrad + e.off + ((long long)i * taps + ftap) * e.O + o, tile[c4 * 4 + 0][r + k * 16], tile[c4 * 4 + 1][r + k * 16],
                      tile[c4 * 4 + 2][r + k * 16], tile[c4 * 4 + 3][r + k * 16]);
  }
}

}  // namespace

extern "C" int das_pack_conv_weights(const float* flat_src, void* fwd_dst, void* dgrad_dst, int dtype,
                                     const DasPackEntry* entries_dev, int n_entries, int total_tiles, void* stream) {
  DAS_PROF(stream);
  if (!flat_src || !dgrad_dst || !entries_dev || n_entries < 1 || total_tiles < 1) return DAS_ERR_ARG;
  if (((uintptr_t)flat_src | (uintptr_t)fwd_dst | (uintptr_t)dgrad_dst) & 15) return DAS_ERR_ARG;   // (16-byte accesses)
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
