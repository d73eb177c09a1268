// Once-per-optimizer-step weight packing for the whole network in ONE launch.
//
// The optimizer keeps every conv weight in the flat f32 master buffer in (Cout, KH, KW, Cin) order
// (channels-last storage of the OIHW parameter), which already is the forward kernels' operand layout.
// This kernel walks a table of such tensors and writes
//   * fwd  : the same layout cast to the activation dtype (bf16 path; the f32 path reads the master itself)
//   * dgrad: (Cin, KH, KW, Cout) with both tap axes flipped — the operand of the data-gradient conv
//   * dgrad_s2 (stride-2 3x3 layers only, entry.s2_pad >= 0): the flipped taps split by output parity — the four
//     (Cin, nth, ntw, Cout) operands of the stride-2 data gradient's sub-grid launches, back to back (their tap counts
//     4 + 2 + 2 + 1 fill the layer's nine-tap region exactly); round 5: 48 strided copies per step before
// all at the tensor's own element offset inside flat destination buffers, so a layer's packed weights are
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
                                                                T* __restrict__ dgrad, T* __restrict__ dgrad_s2,
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
  // parity class of this flipped tap (ops._s2_classes): with pad' = k - 1 - pad, output parity ph takes the taps
  // a = a0 + 2 a', a0 = (ph + pad') % 2 — so tap fa belongs to ph = (fa + pad') % 2 at position a' = fa / 2
  long long s2_base = -1;
  int s2_nt_w = 0, s2_pos = 0;   // taps per input channel of the class, position of this tap among them
  if (dgrad_s2 && e.s2_pad >= 0) {
    const int k = e.KH, padp = k - 1 - e.s2_pad;
    const int fa = ftap / e.KW, fb = ftap % e.KW;
    const int ph = (fa + padp) & 1, pw = (fb + padp) & 1;
    const int a0h = (ph + padp) & 1, a0w = (pw + padp) & 1;
    const int n0 = (k - (padp & 1) + 1) / 2, n1 = (k - ((1 + padp) & 1) + 1) / 2;   // taps of parity 0 / 1
    const int nth = (k - a0h + 1) / 2, ntw = (k - a0w + 1) / 2;
    const int before = ph == 0 ? (pw == 0 ? 0 : n0 * n0) : (pw == 0 ? n0 * n0 + n0 * n1 : n0 * n0 + 2 * n0 * n1);
    s2_base = (long long)before * e.I * e.O;
    s2_pos = (fa / 2) * ntw + fb / 2;
    // element (i, a', b', o) of the class tensor (I, nth, ntw, O): base + ((i * nth + a') * ntw + b') * O + o
    s2_nt_w = nth * ntw;   // taps per input channel in this class
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = i0 + r + k * 16, o = o0 + c4 * 4;
    if (o < e.O && i < e.I) {
      const float a = tile[c4 * 4 + 0][r + k * 16], b = tile[c4 * 4 + 1][r + k * 16], c = tile[c4 * 4 + 2][r + k * 16],
                  d = tile[c4 * 4 + 3][r + k * 16];
      Pack4<T>::store(dgrad + e.off + ((long long)i * taps + ftap) * e.O + o, a, b, c, d);
      if (s2_base >= 0) Pack4<T>::store(dgrad_s2 + e.off + s2_base + ((long long)i * s2_nt_w + s2_pos) * e.O + o, a, b, c, d);
    }
  }
}

}  // namespace

extern "C" int das_pack_conv_weights(const float* flat_src, void* fwd_dst, void* dgrad_dst, void* dgrad_s2_dst, int dtype,
                                     const DasPackEntry* entries_dev, int n_entries, int total_tiles, void* stream) {
  DAS_PROF(stream);
  if (!flat_src || !dgrad_dst || !entries_dev || n_entries < 1 || total_tiles < 1) return DAS_ERR_ARG;
  if (((uintptr_t)flat_src | (uintptr_t)fwd_dst | (uintptr_t)dgrad_dst | (uintptr_t)dgrad_s2_dst) & 15) return DAS_ERR_ARG;   // (16-byte accesses)
  hipStream_t s = (hipStream_t)stream;
  if (dtype == DAS_BF16) {
    hipLaunchKernelGGL(pack_conv_weights_kernel<bf16_t>, dim3(total_tiles), dim3(256), 0, s, flat_src,
                       (bf16_t*)fwd_dst, (bf16_t*)dgrad_dst, (bf16_t*)dgrad_s2_dst, entries_dev, n_entries);
  } else if (dtype == DAS_F32) {
    hipLaunchKernelGGL(pack_conv_weights_kernel<float>, dim3(total_tiles), dim3(256), 0, s, flat_src,
                       (float*)fwd_dst, (float*)dgrad_dst, (float*)dgrad_s2_dst, entries_dev, n_entries);
  } else {
    return DAS_ERR_ARG;
  }
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}
