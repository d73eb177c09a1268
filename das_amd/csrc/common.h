// Shared device helpers for the DAS HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "das_hip.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef uint16_t bf16_t;  // raw bfloat16 bits

#define DAS_CHECK_LAUNCH()                                    \
  do {                                                        \
    hipError_t e__ = hipGetLastError();                       \
    if (e__ != hipSuccess) return DAS_ERR_LAUNCH;             \
  } while (0)

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// f32 -> bf16, round-to-nearest-even, NaN kept quiet: gfx950's v_cvt_pk_bf16_f32 (one instruction per PAIR; the
// integer sequence it replaces — NaN test, rounding add, shift, merge — was ~8 VALU instructions per element and made
// the C-tile staging of the conv epilogues VALU-bound: 5.4 us of a 256 x 256 tile's 11.5 us epilogue, in-kernel stamps).
typedef __bf16 das_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float das_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const das_f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, das_bf16x2_t));
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return (bf16_t)(pack_bf16x2(f, 0.f) & 0xffffu); }

// Element traits: T is the storage type of activations / weights.
template <typename T>
struct Elem;
template <>
struct Elem<float> {
  static constexpr int EPV = 4;  // elements per 16-byte vector
  static __device__ __forceinline__ void unpack(const uint4& v, float* f) {
    f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y);
    f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
  }
  static __device__ __forceinline__ uint4 pack(const float* f) {
    return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
  }
  static __device__ __forceinline__ float load(const float* p) { return *p; }
  static __device__ __forceinline__ void store(float* p, float v) { *p = v; }
};
template <>
struct Elem<bf16_t> {
  static constexpr int EPV = 8;
  static __device__ __forceinline__ void unpack(const uint4& v, float* f) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
  }
  static __device__ __forceinline__ uint4 pack(const float* f) {
    return make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]),
                      pack_bf16x2(f[6], f[7]));
  }
  static __device__ __forceinline__ float load(const bf16_t* p) { return bf16_to_f32(*p); }
  static __device__ __forceinline__ void store(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

// Train-mode BatchNorm affine, written once so that forward and the backward's recomputed ReLU mask
// round identically.
__device__ __forceinline__ float bn_affine(float x, float mean, float invstd, float gamma, float beta) {
  return __fmaf_rn((x - mean) * invstd, gamma, beta);
}

// ---- ReLU masks as bits. A BatchNorm layer whose ReLU follows a residual add needs (y > 0) in its backward; reading y for
// that costs a full tensor pass per consumer. The forward apply pass can record the mask instead: ONE BYTE PER 16-BYTE
// VECTOR of y (bit j = element j of the vector is > 0; 8 bits for bf16, the low 4 for f32), i.e. 1/16 of y's bytes,
// indexed by the vector's index (rows * C / EPV + vector in row). relu_bits: the byte of a packed (stored) vector;
// mask_vec: a vector whose elements are 1.0 / 0.0 by the byte — what the consumers' `y > 0` tests then see.
template <typename T>
__device__ __forceinline__ unsigned relu_bits(const uint4& packed) {
  constexpr int EPV = Elem<T>::EPV;
  float o[EPV];
  Elem<T>::unpack(packed, o);
  unsigned b = 0;
#pragma unroll
  for (int j = 0; j < EPV; ++j) b |= (o[j] > 0.f ? 1u : 0u) << j;
  return b;
}
template <typename T>
__device__ __forceinline__ uint4 mask_vec(unsigned b);
template <>
__device__ __forceinline__ uint4 mask_vec<float>(unsigned b) {
  return make_uint4((b & 1) ? 0x3F800000u : 0u, (b & 2) ? 0x3F800000u : 0u, (b & 4) ? 0x3F800000u : 0u, (b & 8) ? 0x3F800000u : 0u);
}
template <>
__device__ __forceinline__ uint4 mask_vec<bf16_t>(unsigned b) {
  return make_uint4(((b & 1) ? 0x3F80u : 0u) | ((b & 2) ? 0x3F800000u : 0u), ((b & 4) ? 0x3F80u : 0u) | ((b & 8) ? 0x3F800000u : 0u),
                    ((b & 16) ? 0x3F80u : 0u) | ((b & 32) ? 0x3F800000u : 0u),
                    ((b & 64) ? 0x3F80u : 0u) | ((b & 128) ? 0x3F800000u : 0u));
}

// GroupNorm's affine, written once so that the forward and the backward's recomputed ReLU mask round identically.
__device__ __forceinline__ float gn_affine(float x, float mean, float rstd, float gamma, float beta) {
  return __fmaf_rn((x - mean) * rstd, gamma, beta);
}

// Sum the [slots][n] per-workgroup slot partials into dst[n] (LDS), fixed order. All 16 loads of a value are in flight
// together: a serial `a += src[k * n + i]` chain costs one L2 round trip per slot at the head of every workgroup
// (16 us per launch measured on the BatchNorm apply passes). slots <= 64; ends with a workgroup barrier.
__device__ __forceinline__ void fold_slots_to_lds(const float* __restrict__ src, int slots, int n, float* dst) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    float a = 0.f;
    for (int k0 = 0; k0 < slots; k0 += 16) {
      float v[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) v[k] = (k0 + k < slots) ? src[(size_t)(k0 + k) * n + i] : 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) a += v[k];
    }
    dst[i] = a;
  }
  __syncthreads();
}

// The same fold as a launch of its own (one thread per value, 256 per workgroup), for the passes over large tensors
// that run as many small workgroups (bn_apply_stream_kernel, bn_bwd_apply_dz_stream_kernel). Same summation order
// as fold_slots_to_lds: both give the same bits.
static __global__ __launch_bounds__(256) void fold_slots_kernel(const float* __restrict__ src, int slots, int n,
                                                                 float* __restrict__ dst) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float a = 0.f;
  for (int k0 = 0; k0 < slots; k0 += 16) {
    float v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = (k0 + k < slots) ? src[(size_t)(k0 + k) * n + i] : 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) a += v[k];
  }
  dst[i] = a;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// XCD-aware remap of a linear block id: consecutive logical ids share an XCD (block b runs
// on XCD b % 8), so neighbouring tiles hit the same per-XCD L2. Bijective for any n.
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
  const int q = nblocks >> 3, r = nblocks & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

static inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---- per-channel partial sums of a workgroup, met in LDS WITHOUT atomics.
// The reduction passes keep a channel vector per thread and NT / VC "pixel lanes" side by side; every thread ends with EPV
// (or 2 EPV) partial sums. ds_add_f32 from every lane runs at about half a lane per clock per CU (measured round 4,
// tools/dev/upstats_bench.py: a kernel of many small workgroups spent more time there than on its traffic; a
// 256-thread workgroup pays ~4 us, a 1024-thread one ~17 us). Instead: plain 16-byte stores into part[lane][width] (one
// writer per word), a barrier, and each of the `width` outputs folded by one thread over the lanes.
template <int EPV>
__device__ __forceinline__ void lds_put(float* part, int width, int lane, int col, const float (&s)[EPV]) {
  float* row = part + (size_t)lane * width + col;
#pragma unroll
  for (int j = 0; j < EPV; j += 4) *reinterpret_cast<float4*>(row + j) = make_float4(s[j], s[j + 1], s[j + 2], s[j + 3]);
}
__device__ __forceinline__ float lds_fold(const float* part, int width, int lanes, int i) {
  float a = 0.f;
  for (int r = 0; r < lanes; ++r) a += part[(size_t)r * width + i];
  return a;
}

// ---- ragged multi-level rows (DasLevels): row m -> level, image, (h, w), plane origin
struct LvGeom {
  int l, b, h, w, H, W;
  long long plane0;  // first row of this (level, image) plane
};
__host__ __device__ inline long long lv_total_rows(const DasLevels& lv) {
  long long n = 0;
  for (int l = 0; l < lv.num_levels; ++l) n += (long long)lv.B * lv.H[l] * lv.W[l];
  return n;
}
__device__ __forceinline__ LvGeom lv_geom(const DasLevels& lv, long long m) {
  LvGeom g;
  long long start = 0;
  int l = 0;
  for (; l + 1 < lv.num_levels; ++l) {
    const long long n = (long long)lv.B * lv.H[l] * lv.W[l];
    if (m < start + n) break;
    start += n;
  }
  g.l = l; g.H = lv.H[l]; g.W = lv.W[l];
  const long long local = m - start;
  const int hw = g.H * g.W;
  g.b = (int)(local / hw);
  const int rem = (int)(local - (long long)g.b * hw);
  g.h = rem / g.W;
  g.w = rem - g.h * g.W;
  g.plane0 = start + (long long)g.b * hw;
  return g;
}
// 16-byte vectors per thread of the one-shot BatchNorm passes over large tensors (bn_apply_stream_kernel, bn_bwd_apply_dz_stream_kernel)
#ifndef DAS_BN_STREAM_VPT
#define DAS_BN_STREAM_VPT 4
#endif
// Minimum pixels per workgroup of the GroupNorm passes. `key` = das_tuning gn.ppb; 0 = automatic: as many as keep at least 448
// workgroups (1.75 per CU) in the launch, within 256 ... 1024 — 1024 at the training batch (64 level x image segments: -0.5 ms per
// step against 256, +0.7 with 2048), 256 at the inference batch (32 segments: 1024 there leaves CUs idle, -5 % img/s).
static inline int gn_ppb_min(long long key, int nseg, int maxhw) {
  if (key > 0) return (int)key;
  int ppb = 1024;
  while (ppb > 256 && (long long)nseg * ((maxhw + ppb - 1) / ppb) < 448) ppb /= 2;
  return ppb;
}
static inline bool lv_valid(const DasLevels* lv) {
  if (!lv || lv->num_levels < 1 || lv->num_levels > 5 || lv->B < 1) return false;
  for (int l = 0; l < lv->num_levels; ++l)
    if (lv->H[l] < 1 || lv->W[l] < 1) return false;
  return true;
}

// 16-byte global accesses with a cache-policy switch: nt = non-temporal (`global_load / store ... nt`: streamed data that nobody
// re-reads soon should not displace what the caches hold for the next kernel)
typedef unsigned das_v4u_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld16(const void* p, bool nt) {
  const das_v4u_t v = nt ? __builtin_nontemporal_load(reinterpret_cast<const das_v4u_t*>(p)) : *reinterpret_cast<const das_v4u_t*>(p);
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st16(void* p, const uint4& u, bool nt) {
  das_v4u_t v;
  v.x = u.x; v.y = u.y; v.z = u.z; v.w = u.w;
  if (nt) __builtin_nontemporal_store(v, reinterpret_cast<das_v4u_t*>(p));
  else *reinterpret_cast<das_v4u_t*>(p) = v;
}
