// Implicit-GEMM convolution for NHWC tensors on gfx950 MFMA, with a fused epilogue.
//
// GEMM view: M = B*Ho*Wo output pixels, N = Cout, K = KH*KW*Cin (tap-major, channel-minor,
// which is exactly the memory order of both the NHWC input row and the packed weights).
// Tile 128(M) x BN(N) x BK(K) per 256-thread workgroup (4 waves); BK = 64 bytes of K per row
// (32 bf16 / 16 f32). The weight tile feeds the MFMA "A" operand and the pixel tile the "B"
// operand, so every lane ends up with 4 consecutive output channels of one pixel.
//   bf16: v_mfma_f32_16x16x32_bf16        f32: 4 x v_mfma_f32_16x16x4_f32 (exact f32)
// LDS rows are 64 B, 16-B slots XOR-swizzled so ds_read_b128 fragment reads are conflict
// free under gfx950's non-contiguous b128 lane groups.
// Global->register prefetch of tile k+1 overlaps the MFMAs of tile k; one barrier per K step.
// Epilogue: per-channel scale/shift in registers -> stage the C tile in LDS -> 16-byte
// coalesced NHWC stores with optional residual add, ReLU and per-channel sum / sum-of-squares
// (train-mode BatchNorm statistics) reduced per block and accumulated with f32 atomics.
#include "common.h"

namespace {

struct ConvP {
  const char* x;
  const char* w;
  char* y;
  const float* scale;
  const float* shift;
  const char* res;
  float* stats;
  int H, W, Cin, xps;
  int Ho, Wo, Cout, yps;
  int KH, KW, stride, pad;
  int relu_in, relu, rps;
  int M, K, HoWo, ntiles, nblocks;
};

constexpr int BM = 128;

__device__ __forceinline__ int lds_slot(int row, int kg) { return row * 64 + ((kg ^ ((-(row >> 2)) & 3)) << 4); }

template <typename T>
__device__ __forceinline__ uint4 relu_vec(uint4 v);
template <>
__device__ __forceinline__ uint4 relu_vec<float>(uint4 v) {
  v.x = (v.x >> 31) ? 0u : v.x; v.y = (v.y >> 31) ? 0u : v.y;
  v.z = (v.z >> 31) ? 0u : v.z; v.w = (v.w >> 31) ? 0u : v.w;
  return v;
}
__device__ __forceinline__ uint32_t relu_bf16x2(uint32_t u) {
  uint32_t lo = (u & 0x8000u) ? 0u : (u & 0xffffu);
  uint32_t hi = (u & 0x80000000u) ? 0u : (u & 0xffff0000u);
  return lo | hi;
}
template <>
__device__ __forceinline__ uint4 relu_vec<bf16_t>(uint4 v) {
  return make_uint4(relu_bf16x2(v.x), relu_bf16x2(v.y), relu_bf16x2(v.z), relu_bf16x2(v.w));
}

template <typename T>
__device__ __forceinline__ void mma(const uint4& a, const uint4& b, f32x4_t& c);
template <>
__device__ __forceinline__ void mma<bf16_t>(const uint4& a, const uint4& b, f32x4_t& c) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0,
                                              0, 0);
}
template <>
__device__ __forceinline__ void mma<float>(const uint4& a, const uint4& b, f32x4_t& c) {
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
}

template <typename T, typename OT, int BN, bool ALIGNED>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvP p) {
  constexpr int EPV = Elem<T>::EPV;
  constexpr int BK = 4 * EPV;
  constexpr int TM = (BN == 128) ? 4 : 2;
  constexpr int TN = (BN >= 64) ? 4 : 2;
  constexpr int WROWS = (BN >= 64) ? BN / 64 : 1;  // weight rows staged per thread
  constexpr int A_BYTES = BM * 64, W_BYTES = BN * 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int logical = xcd_remap(blockIdx.x, p.nblocks);
  const int n0 = (logical % p.ntiles) * BN;
  const int m0 = (logical / p.ntiles) * BM;
  const int wave_m0 = (BN == 128) ? (wave & 1) * 64 : wave * 32;
  const int wave_n0 = (BN == 128) ? (wave >> 1) * 64 : 0;

  // ---- per-thread staging coordinates: 2 pixel rows + WROWS weight rows, one 16-B k-group
  const int srow = tid >> 2, kg = tid & 3;
  int hi0[2], wi0[2];
  long long xoff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + srow + 64 * i;
    if (m < p.M) {
      const int b = m / p.HoWo, rem = m - b * p.HoWo;
      const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
      hi0[i] = ho * p.stride - p.pad;
      wi0[i] = wo * p.stride - p.pad;
      xoff[i] = (long long)b * p.H * p.W * p.xps;
    } else {
      hi0[i] = -(1 << 28);
      wi0[i] = 0;
      xoff[i] = 0;
    }
  }
  const bool w_active = (BN >= 64) || (tid < 128);
  const T* xg = reinterpret_cast<const T*>(p.x);
  const T* wg = reinterpret_cast<const T*>(p.w);

  uint4 ra[2], rw[WROWS];
  // incremental (kh, kw, ci) of the tile being fetched, used when Cin % BK == 0
  int f_kh = 0, f_kw = 0, f_ci = 0;

  auto fetch = [&](int kt) {
    const int k0 = kt * BK + kg * EPV;
    int kh, kw, ci;
    bool kok = true;
    if (ALIGNED) {
      kh = f_kh; kw = f_kw; ci = f_ci + kg * EPV;
    } else {
      kok = k0 < p.K;
      const int tap = k0 / p.Cin;
      ci = k0 - tap * p.Cin;
      kh = tap / p.KW;
      kw = tap - kh * p.KW;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int hi = hi0[i] + kh, wi = wi0[i] + kw;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (kok && hi >= 0 && hi < p.H && wi >= 0 && wi < p.W) {
        v = *reinterpret_cast<const uint4*>(xg + xoff[i] + ((long long)hi * p.W + wi) * p.xps + ci);
        if (p.relu_in) v = relu_vec<T>(v);
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < WROWS; ++i) {
      const int n = n0 + srow + 64 * i;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (w_active && kok && n < p.Cout) v = *reinterpret_cast<const uint4*>(wg + (long long)n * p.K + k0);
      rw[i] = v;
    }
    if (ALIGNED) {
      f_ci += BK;
      if (f_ci >= p.Cin) {
        f_ci = 0;
        if (++f_kw == p.KW) { f_kw = 0; ++f_kh; }
      }
    }
  };
  auto stash = [&](int buf) {
    char* sA = smem + buf * A_BYTES;
    char* sW = smem + 2 * A_BYTES + buf * W_BYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<uint4*>(sA + lds_slot(srow + 64 * i, kg)) = ra[i];
    if (w_active) {
#pragma unroll
      for (int i = 0; i < WROWS; ++i) *reinterpret_cast<uint4*>(sW + lds_slot(srow + 64 * i, kg)) = rw[i];
    }
  };

  f32x4_t acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nk = (p.K + BK - 1) / BK;
  fetch(0);
  stash(0);
  __syncthreads();
  const int frow = lane & 15, fkg = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) fetch(kt + 1);
    const char* sA = smem + buf * A_BYTES;
    const char* sW = smem + 2 * A_BYTES + buf * W_BYTES;
    uint4 fb[TM], fa[TN];
#pragma unroll
    for (int t = 0; t < TM; ++t) fb[t] = *reinterpret_cast<const uint4*>(sA + lds_slot(wave_m0 + t * 16 + frow, fkg));
#pragma unroll
    for (int t = 0; t < TN; ++t) fa[t] = *reinterpret_cast<const uint4*>(sW + lds_slot(wave_n0 + t * 16 + frow, fkg));
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int b = 0; b < TM; ++b) mma<T>(fa[a], fb[b], acc[a][b]);
    if (kt + 1 < nk) stash(buf ^ 1);
    __syncthreads();
  }

  // ---------------------------------------------------------------- epilogue
  constexpr int EPVO = 16 / (int)sizeof(OT);
  constexpr int CS = BN * (int)sizeof(OT) + 16;  // padded C-tile row stride in bytes
  const int ch4 = (lane >> 4) * 4;
#pragma unroll
  for (int a = 0; a < TN; ++a) {
    const int nl = wave_n0 + a * 16 + ch4;
    const int n = n0 + nl;
    float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
    if (n < p.Cout) {
      if (p.scale) { const float4 t = *reinterpret_cast<const float4*>(p.scale + n); sc[0] = t.x; sc[1] = t.y; sc[2] = t.z; sc[3] = t.w; }
      if (p.shift) { const float4 t = *reinterpret_cast<const float4*>(p.shift + n); sh[0] = t.x; sh[1] = t.y; sh[2] = t.z; sh[3] = t.w; }
    }
#pragma unroll
    for (int b = 0; b < TM; ++b) {
      const int ml = wave_m0 + b * 16 + (lane & 15);
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = acc[a][b][j] * sc[j] + sh[j];
      char* dst = smem + ml * CS + nl * (int)sizeof(OT);
      if (sizeof(OT) == 2) {
        *reinterpret_cast<uint2*>(dst) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
      } else {
        *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
  }
  __syncthreads();

  constexpr int VR = BN * (int)sizeof(OT) / 16;  // 16-B vectors per C row
  constexpr int RP = 256 / VR;                   // rows per pass
  const int vec = tid % VR, r0 = tid / VR;
  const int n = n0 + vec * EPVO;
  float ssum[EPVO], ssq[EPVO];
#pragma unroll
  for (int j = 0; j < EPVO; ++j) { ssum[j] = 0.f; ssq[j] = 0.f; }
  OT* yg = reinterpret_cast<OT*>(p.y);
  const OT* rg = reinterpret_cast<const OT*>(p.res);
  if (n < p.Cout) {
#pragma unroll 2
    for (int ml = r0; ml < BM; ml += RP) {
      const int m = m0 + ml;
      if (m >= p.M) break;
      const uint4 raw = *reinterpret_cast<const uint4*>(smem + ml * CS + vec * 16);
      float f[EPVO];
      Elem<OT>::unpack(raw, f);
      if (p.stats) {
#pragma unroll
        for (int j = 0; j < EPVO; ++j) { ssum[j] += f[j]; ssq[j] += f[j] * f[j]; }
      }
      if (rg) {
        float r[EPVO];
        Elem<OT>::unpack(*reinterpret_cast<const uint4*>(rg + (long long)m * p.rps + n), r);
#pragma unroll
        for (int j = 0; j < EPVO; ++j) f[j] += r[j];
      }
      if (p.relu) {
#pragma unroll
        for (int j = 0; j < EPVO; ++j) f[j] = fmaxf(f[j], 0.f);
      }
      *reinterpret_cast<uint4*>(yg + (long long)m * p.yps + n) = (rg || p.relu) ? Elem<OT>::pack(f) : raw;
    }
  }
  if (p.stats) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);  // [2][RP][BN]
#pragma unroll
    for (int j = 0; j < EPVO; ++j) {
      red[r0 * BN + vec * EPVO + j] = ssum[j];
      red[(RP + r0) * BN + vec * EPVO + j] = ssq[j];
    }
    __syncthreads();
    if (tid < 2 * BN) {
      const int which = tid / BN, c = tid % BN;
      if (n0 + c < p.Cout) {
        float s = 0.f;
        for (int r = 0; r < RP; ++r) s += red[(which * RP + r) * BN + c];
        atomicAdd(p.stats + which * p.Cout + n0 + c, s);
      }
    }
  }
}

template <typename T, typename OT, int BN>
size_t smem_bytes() {
  size_t stage = 2 * (size_t)(BM + BN) * 64;
  size_t ctile = (size_t)BM * (BN * sizeof(OT) + 16);
  size_t red = 2 * (size_t)(256 / (BN * sizeof(OT) / 16)) * BN * 4;
  size_t m = stage > ctile ? stage : ctile;
  return m > red ? m : red;
}

template <typename T, typename OT, int BN>
int launch(const ConvP& p0, bool aligned, hipStream_t s) {
  ConvP p = p0;
  p.ntiles = (p.Cout + BN - 1) / BN;
  const int mtiles = (p.M + BM - 1) / BM;
  p.nblocks = p.ntiles * mtiles;
  const size_t sm = smem_bytes<T, OT, BN>();
  static bool attr_set = false;  // one flag per instantiation; > 64 KiB dynamic LDS needs the opt-in
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv_igemm_kernel<T, OT, BN, true>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
    (void)hipFuncSetAttribute((const void*)conv_igemm_kernel<T, OT, BN, false>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
    attr_set = true;
  }
  if (aligned) {
    hipLaunchKernelGGL((conv_igemm_kernel<T, OT, BN, true>), dim3(p.nblocks), dim3(256), sm, s, p);
  } else {
    hipLaunchKernelGGL((conv_igemm_kernel<T, OT, BN, false>), dim3(p.nblocks), dim3(256), sm, s, p);
  }
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}

template <typename T, typename OT>
int launch_bn(const ConvP& p, bool aligned, hipStream_t s) {
  if (p.Cout > 64) return launch<T, OT, 128>(p, aligned, s);
  if (p.Cout > 32) return launch<T, OT, 64>(p, aligned, s);
  return launch<T, OT, 32>(p, aligned, s);
}

}  // namespace

extern "C" int das_conv2d_nhwc(const void* x, const void* w, void* y, const DasConvDesc* d, void* stream) {
  if (!x || !w || !y || !d) return DAS_ERR_ARG;
  if (d->Cin % 8 || d->Cout % 8 || d->x_pix_stride % 8 || d->y_pix_stride % 8) return DAS_ERR_ARG;
  if (d->x_pix_stride < d->Cin || d->y_pix_stride < d->Cout) return DAS_ERR_ARG;
  if (d->residual && (d->out_dtype != d->dtype || d->res_pix_stride % 8)) return DAS_ERR_ARG;
  if (d->dtype == DAS_F32 && d->out_dtype != DAS_F32) return DAS_ERR_ARG;
  if (d->KH < 1 || d->KW < 1 || d->stride < 1 || d->B < 1) return DAS_ERR_ARG;
  const long long M = (long long)d->B * d->Ho * d->Wo;
  if (M <= 0 || M > 0x7fffffffLL) return DAS_ERR_ARG;
  ConvP p;
  p.x = (const char*)x; p.w = (const char*)w; p.y = (char*)y;
  p.scale = d->scale; p.shift = d->shift; p.res = (const char*)d->residual; p.stats = d->stats;
  p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.xps = d->x_pix_stride;
  p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout; p.yps = d->y_pix_stride;
  p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad;
  p.relu_in = d->relu_in; p.relu = d->relu; p.rps = d->res_pix_stride;
  p.M = (int)M; p.K = d->KH * d->KW * d->Cin; p.HoWo = d->Ho * d->Wo;
  p.ntiles = p.nblocks = 0;
  hipStream_t s = (hipStream_t)stream;
  if (d->dtype == DAS_BF16) {
    const bool aligned = (d->Cin % 32) == 0;
    if (d->out_dtype == DAS_BF16) return launch_bn<bf16_t, bf16_t>(p, aligned, s);
    return launch_bn<bf16_t, float>(p, aligned, s);
  }
  if (d->dtype == DAS_F32) return launch_bn<float, float>(p, (d->Cin % 16) == 0, s);
  return DAS_ERR_ARG;
}
