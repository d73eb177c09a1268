// HBM-bound NHWC helper kernels: layout pack/unpack, max-pool, bilinear / nearest upsampling,
// adds. One 16-byte channel vector per thread (8 bf16 / 4 f32), grid-stride, fully coalesced.
#include "common.h"
#include "prof.h"
#include "tuning.h"

namespace {

constexpr int TPB = 256;
inline int grid_for(long long n) {
  long long b = (n + TPB - 1) / TPB;
  return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

template <typename T>
__global__ void pack_kernel(const float* __restrict__ x, T* __restrict__ y, int B, int C, int H, int W, int Cpad) {
  const long long npix = (long long)B * H * W;
  const long long total = npix * Cpad;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
    const int c = (int)(i % Cpad);
    const long long pix = i / Cpad;
    float v = 0.f;
    if (c < C) {
      const long long b = pix / ((long long)H * W), hw = pix - b * H * W;
      v = x[(b * C + c) * H * W + hw];
    }
    Elem<T>::store(y + i, v);
  }
}

template <typename T>
__global__ void unpack_kernel(const T* __restrict__ x, float* __restrict__ y, int B, int C, int H, int W, int ps,
                              int c0) {
  const long long HW = (long long)H * W, total = (long long)B * C * HW;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
    const long long hw = i % HW, bc = i / HW;
    const int c = (int)(bc % C);
    const long long b = bc / C;
    y[i] = Elem<T>::load(x + (b * HW + hw) * ps + c0 + c);
  }
}

template <typename T>
__global__ void maxpool_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C, int Ho, int Wo) {
  constexpr int EPV = Elem<T>::EPV;
  const int VC = C / EPV;
  const long long total = (long long)B * Ho * Wo * VC;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
    const int v = (int)(i % VC);
    long long pix = i / VC;
    const int wo = (int)(pix % Wo);
    pix /= Wo;
    const int ho = (int)(pix % Ho);
    const long long b = pix / Ho;
    float m[EPV];
#pragma unroll
    for (int j = 0; j < EPV; ++j) m[j] = -INFINITY;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int hi = ho * 2 - 1 + dy;
      if (hi < 0 || hi >= H) continue;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int wi = wo * 2 - 1 + dx;
        if (wi < 0 || wi >= W) continue;
        float f[EPV];
        Elem<T>::unpack(*reinterpret_cast<const uint4*>(x + ((b * H + hi) * W + wi) * C + v * EPV), f);
#pragma unroll
        for (int j = 0; j < EPV; ++j) m[j] = fmaxf(m[j], f[j]);
      }
    }
    *reinterpret_cast<uint4*>(y + i * EPV) = Elem<T>::pack(m);
  }
}

// The same pooling, also recording WHICH tap won (0..8 in scan order; the first maximum, as torch routes the gradient):
// one byte per output element. The backward pass then gathers from (dy, idx) — 80 MB — instead of re-deriving the maxima of
// every window from x (25 vector loads and 72 compares per four outputs: 338 us per step at B = 16, 1.45 TB/s).
template <typename T>
__global__ void maxpool_argmax_kernel(const T* __restrict__ x, T* __restrict__ y, unsigned char* __restrict__ idx, int B, int H, int W,
                                      int C, int Ho, int Wo) {
  constexpr int EPV = Elem<T>::EPV;
  const int VC = C / EPV;
  const long long total = (long long)B * Ho * Wo * VC;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
    const int v = (int)(i % VC);
    long long pix = i / VC;
    const int wo = (int)(pix % Wo);
    pix /= Wo;
    const int ho = (int)(pix % Ho);
    const long long b = pix / Ho;
    float m[EPV];
    int am[EPV];
#pragma unroll
    for (int j = 0; j < EPV; ++j) { m[j] = 0.f; am[j] = -1; }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int hi = ho * 2 - 1 + k / 3, wi = wo * 2 - 1 + k % 3;
      if (hi < 0 || hi >= H || wi < 0 || wi >= W) continue;
      float f[EPV];
      Elem<T>::unpack(*reinterpret_cast<const uint4*>(x + ((b * H + hi) * W + wi) * C + v * EPV), f);
#pragma unroll
      for (int j = 0; j < EPV; ++j)
        if (am[j] < 0 || f[j] > m[j]) { am[j] = k; m[j] = f[j]; }   // strict >: the first maximum wins (as maxpool_bwd_kernel)
    }
    *reinterpret_cast<uint4*>(y + i * EPV) = Elem<T>::pack(m);
    unsigned char* d = idx + i * EPV;
    if (EPV == 8) {
      *reinterpret_cast<uint2*>(d) = make_uint2((unsigned)am[0] | ((unsigned)am[1] << 8) | ((unsigned)am[2] << 16) | ((unsigned)am[3] << 24),
                                               (unsigned)am[4] | ((unsigned)am[5] << 8) | ((unsigned)am[6] << 16) | ((unsigned)am[7] << 24));
    } else {
      *reinterpret_cast<unsigned*>(d) = (unsigned)am[0] | ((unsigned)am[1] << 8) | ((unsigned)am[2] << 16) | ((unsigned)am[3] << 24);
    }
  }
}

// dx[hi, wi] = sum over the (1, 2 or 4) windows that cover the pixel of dy[window] where that window's winning tap is this
// pixel. Even rows / columns lie in one window (its centre tap), odd ones in two.
template <typename T>
__global__ void maxpool_bwd_argmax_kernel(const T* __restrict__ dy, const unsigned char* __restrict__ idx, T* __restrict__ dx, int B,
                                          int H, int W, int C, int Ho, int Wo) {
  constexpr int EPV = Elem<T>::EPV;
  const int VC = C / EPV;
  const long long total = (long long)B * H * W * VC;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
    const int v = (int)(i % VC);
    long long pix = i / VC;
    const int wi = (int)(pix % W);
    pix /= W;
    const int hi = (int)(pix % H);
    const long long b = pix / H;
    float acc[EPV];
#pragma unroll
    for (int j = 0; j < EPV; ++j) acc[j] = 0.f;
    const int nh = (hi & 1) ? 2 : 1, nw = (wi & 1) ? 2 : 1;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      if (a >= nh) continue;
      const int ho = (hi >> 1) + a, kh = hi - (ho * 2 - 1);
      if (ho >= Ho) continue;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        if (c >= nw) continue;
        const int wo = (wi >> 1) + c, kw = wi - (wo * 2 - 1);
        if (wo >= Wo) continue;
        const int k = kh * 3 + kw;
        const long long o = (((b * Ho + ho) * Wo + wo) * VC + v) * EPV;
        float g[EPV];
        Elem<T>::unpack(*reinterpret_cast<const uint4*>(dy + o), g);
        unsigned w0, w1 = 0;
        if (EPV == 8) { const uint2 t = *reinterpret_cast<const uint2*>(idx + o); w0 = t.x; w1 = t.y; }
        else w0 = *reinterpret_cast<const unsigned*>(idx + o);
#pragma unroll
        for (int j = 0; j < EPV; ++j) {
          const unsigned tap = ((j < 4 ? w0 : w1) >> ((j & 3) * 8)) & 0xffu;
          acc[j] += tap == (unsigned)k ? g[j] : 0.f;
        }
      }
    }
    *reinterpret_cast<uint4*>(dx + i * EPV) = Elem<T>::pack(acc);
  }
}

// torch upsample_bilinear2d, align_corners=True (area_pixel_compute_scale: (in-1)/(out-1) in f32).
// Optionally with the conv epilogue's BatchNorm statistics attached: per-channel sum / sum of squares of the STORED
// (rounded) outputs into stats[slots][2][C] (workgroup b adds into slot b % slots). Used where a bias-free 1x1 conv follows
// the upsampling (MSPN's up_conv, mspn_mmpose.py:385-389): the two are linear and commute, so the conv runs on the
// quarter-size tensor and THIS kernel produces the pre-norm tensor the BatchNorm layer sees.
// One workgroup = a run of PIX output pixels x all channels; a thread keeps its channel vector.
// The per-thread partial sums meet in LDS as plain stores ([pixel lane][2C], one writer per word) and are folded by a second
// sweep (common.h: lds_put / lds_fold). STATS = false is the plain resampling (stats / slots unused, no LDS).
template <typename T, bool STATS>
__global__ __launch_bounds__(TPB) void bilinear_ac_stats_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C,
                                                               int Ho, int Wo, float sh, float sw, int pix_per_block,
                                                               float* __restrict__ stats, int slots) {
#pragma clang fp contract(off)
  constexpr int EPV = Elem<T>::EPV;
  constexpr int NT = TPB;
  extern __shared__ float part[];   // [PL][2C]
  const int VC = C / EPV;
  const int VCB = min(VC, NT), PL = NT / VCB, pl = threadIdx.x / VCB;
  // (pixel indices in 32 bits — the launcher checks — so the row / column split is two 32-bit divisions, not 64-bit ones)
  const unsigned npix = (unsigned)B * Ho * Wo;
  const unsigned p0 = (unsigned)xcd_remap(blockIdx.x, gridDim.x) * (unsigned)pix_per_block, p1 = min(npix, p0 + pix_per_block);   // (neighbouring runs share source rows: same XCD, same L2)
  for (int v = threadIdx.x % VCB; v < VC && pl < PL; v += VCB) {
    float s[EPV], q[EPV];
#pragma unroll
    for (int j = 0; j < EPV; ++j) { s[j] = 0.f; q[j] = 0.f; }
    constexpr int U = 4;   // pixels in flight per thread (the loop is bound by load latency, not by arithmetic)
    for (unsigned pix0 = p0 + pl; pix0 < p1; pix0 += U * PL) {
      uint4 ra[U], rb[U], rc[U], rd[U];
      float h1l[U], w1l[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const unsigned pix = min(pix0 + (unsigned)u * PL, p1 - 1);   // (a clamped duplicate is loaded and dropped)
        const unsigned t = pix / (unsigned)Wo, wo = pix - t * Wo;
        const unsigned b = t / (unsigned)Ho, ho = t - b * Ho;
        const float h1r = sh * (int)ho, w1r = sw * (int)wo;
        const int h1 = (int)h1r, w1 = (int)w1r;
        const int h1p = (h1 < H - 1) ? 1 : 0, w1p = (w1 < W - 1) ? 1 : 0;
        h1l[u] = h1r - h1;
        w1l[u] = w1r - w1;
        const T* base = x + (((long long)b * H + h1) * W + w1) * C + v * EPV;
        ra[u] = *reinterpret_cast<const uint4*>(base);
        rb[u] = *reinterpret_cast<const uint4*>(base + (long long)w1p * C);
        rc[u] = *reinterpret_cast<const uint4*>(base + (long long)h1p * W * C);
        rd[u] = *reinterpret_cast<const uint4*>(base + ((long long)h1p * W + w1p) * C);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const unsigned pix = pix0 + (unsigned)u * PL;
        if (pix >= p1) break;
        const float h0l = 1.f - h1l[u], w0l = 1.f - w1l[u];
        float a[EPV], bb[EPV], c[EPV], d[EPV], o[EPV];
        Elem<T>::unpack(ra[u], a);
        Elem<T>::unpack(rb[u], bb);
        Elem<T>::unpack(rc[u], c);
        Elem<T>::unpack(rd[u], d);
#pragma unroll
        for (int j = 0; j < EPV; ++j) o[j] = h0l * (w0l * a[j] + w1l[u] * bb[j]) + h1l[u] * (w0l * c[j] + w1l[u] * d[j]);
        const uint4 packed = Elem<T>::pack(o);
        if (y) *reinterpret_cast<uint4*>(y + ((long long)pix * VC + v) * EPV) = packed;
        if (STATS) {
          Elem<T>::unpack(packed, o);     // the statistics see the values as stored
#pragma unroll
          for (int j = 0; j < EPV; ++j) { s[j] += o[j]; q[j] += o[j] * o[j]; }
        }
      }
    }
    if (STATS) {
      lds_put<EPV>(part, 2 * C, pl, v * EPV, s);
      lds_put<EPV>(part, 2 * C, pl, C + v * EPV, q);
    }
  }
  if (!STATS) return;
  __syncthreads();
  float* dst = stats + (size_t)(slots > 1 ? blockIdx.x % (unsigned)slots : 0) * 2 * C;
  for (int i = threadIdx.x; i < 2 * C; i += NT) atomicAdd(dst + i, lds_fold(part, 2 * C, PL, i));
}

// y = a + nearest(b): torch nearest index = min(floor(dst * (in/out)), in-1) with f32 scale
template <typename T>
__global__ void add_nearest_kernel(const T* __restrict__ a, const T* __restrict__ bsrc, T* __restrict__ y, int B,
                                   int H, int W, int C, int Hb, int Wb, float sh, float sw) {
  constexpr int EPV = Elem<T>::EPV;
  const int VC = C / EPV;
  const long long total = (long long)B * H * W * VC;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
    const int v = (int)(i % VC);
    long long pix = i / VC;
    const int w = (int)(pix % W);
    pix /= W;
    const int h = (int)(pix % H);
    const long long b = pix / H;
    const int hs = min((int)floorf(h * sh), Hb - 1), ws = min((int)floorf(w * sw), Wb - 1);
    float fa[EPV], fb[EPV];
    Elem<T>::unpack(*reinterpret_cast<const uint4*>(a + i * EPV), fa);
    Elem<T>::unpack(*reinterpret_cast<const uint4*>(bsrc + ((b * Hb + hs) * Wb + ws) * C + v * EPV), fb);
#pragma unroll
    for (int j = 0; j < EPV; ++j) fa[j] += fb[j];
    *reinterpret_cast<uint4*>(y + i * EPV) = Elem<T>::pack(fa);
  }
}

template <typename T>
__global__ void add3_kernel(const T* __restrict__ a, const T* __restrict__ b, const T* __restrict__ c,
                            T* __restrict__ y, long long nvec, int relu) {
  constexpr int EPV = Elem<T>::EPV;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < nvec; i += (long long)gridDim.x * TPB) {
    float fa[EPV], fb[EPV];
    Elem<T>::unpack(*reinterpret_cast<const uint4*>(a + i * EPV), fa);
    Elem<T>::unpack(*reinterpret_cast<const uint4*>(b + i * EPV), fb);
#pragma unroll
    for (int j = 0; j < EPV; ++j) fa[j] += fb[j];
    if (c) {
      // bf16 storage: the reference rounds (a+b) before adding c; keep that order
      uint4 t = Elem<T>::pack(fa);
      Elem<T>::unpack(t, fa);
      Elem<T>::unpack(*reinterpret_cast<const uint4*>(c + i * EPV), fb);
#pragma unroll
      for (int j = 0; j < EPV; ++j) fa[j] += fb[j];
    }
    if (relu) {
#pragma unroll
      for (int j = 0; j < EPV; ++j) fa[j] = fmaxf(fa[j], 0.f);
    }
    *reinterpret_cast<uint4*>(y + i * EPV) = Elem<T>::pack(fa);
  }
}

}  // namespace

#define DISPATCH_T(dtype, CALL)                 \
  if ((dtype) == DAS_BF16) { using T = bf16_t; CALL; } \
  else if ((dtype) == DAS_F32) { using T = float; CALL; } \
  else return DAS_ERR_ARG;

extern "C" int das_pack_nchw_to_nhwc(const float* x, void* y, int dtype, int B, int C, int H, int W, int Cpad,
                                     void* stream) {
  DAS_PROF(stream);
  if (!x || !y || Cpad < C) return DAS_ERR_ARG;
  const long long total = (long long)B * H * W * Cpad;
  DISPATCH_T(dtype, hipLaunchKernelGGL(pack_kernel<T>, dim3(grid_for(total)), dim3(TPB), 0, (hipStream_t)stream, x,
                                       (T*)y, B, C, H, W, Cpad));
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}

extern "C" int das_unpack_nhwc_to_nchw(const void* x, float* y, int dtype, int B, int C, int H, int W, int pix_stride,
                                       int c0, void* stream) {
  DAS_PROF(stream);
  if (!x || !y || c0 + C > pix_stride) return DAS_ERR_ARG;
  const long long total = (long long)B * H * W * C;
  DISPATCH_T(dtype, hipLaunchKernelGGL(unpack_kernel<T>, dim3(grid_for(total)), dim3(TPB), 0, (hipStream_t)stream,
                                       (const T*)x, y, B, C, H, W, pix_stride, c0));
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}

extern "C" int das_maxpool3x3s2(const void* x, void* y, int dtype, int B, int H, int W, int C, void* stream) {
  DAS_PROF(stream);
  if (!x || !y || C % 8) return DAS_ERR_ARG;
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  DISPATCH_T(dtype, {
    const long long total = (long long)B * Ho * Wo * (C / Elem<T>::EPV);
    hipLaunchKernelGGL(maxpool_kernel<T>, dim3(grid_for(total)), dim3(TPB), 0, (hipStream_t)stream, (const T*)x,
                       (T*)y, B, H, W, C, Ho, Wo);
  });
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}

extern "C" int das_maxpool3x3s2_argmax(const void* x, void* y, void* idx, int dtype, int B, int H, int W, int C, void* stream) {
  DAS_PROF(stream);
  if (!x || !y || !idx || C % 8) return DAS_ERR_ARG;
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  DISPATCH_T(dtype, {
    const long long total = (long long)B * Ho * Wo * (C / Elem<T>::EPV);
    hipLaunchKernelGGL(maxpool_argmax_kernel<T>, dim3(grid_for(total)), dim3(TPB), 0, (hipStream_t)stream, (const T*)x, (T*)y,
                       (unsigned char*)idx, B, H, W, C, Ho, Wo);
  });
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}

extern "C" int das_maxpool3x3s2_backward_argmax(const void* dy, const void* idx, void* dx, int dtype, int B, int H, int W, int C,
                                                void* stream) {
  DAS_PROF(stream);
  if (!dy || !idx || !dx || C % 8) return DAS_ERR_ARG;
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  DISPATCH_T(dtype, {
    const long long total = (long long)B * H * W * (C / Elem<T>::EPV);
    hipLaunchKernelGGL(maxpool_bwd_argmax_kernel<T>, dim3(grid_for(total)), dim3(TPB), 0, (hipStream_t)stream, (const T*)dy,
                       (const unsigned char*)idx, (T*)dx, B, H, W, C, Ho, Wo);
  });
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}

// stats == nullptr: the plain resampling
static int launch_bilinear_ac(const void* x, void* y, int dtype, int B, int H, int W, int C, int Ho, int Wo, float* stats,
                              int stats_slots, void* stream) {
  if (!x || (!y && !stats) || C % 8 || B < 1 || H < 1 || W < 1 || Ho < 1 || Wo < 1) return DAS_ERR_ARG;
  if (stats && (C > 4096 || stats_slots < 1 || stats_slots > 64)) return DAS_ERR_ARG;
  const float sh = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f;
  const float sw = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
  const long long npix = (long long)B * Ho * Wo;
  if (npix >= (1ll << 31)) return DAS_ERR_ARG;
  const int vc = C / (dtype == DAS_F32 ? 4 : 8);
  const int pl = TPB / (vc < TPB ? vc : TPB);
  // whole rounds of (pixel lanes x 4 pixels in flight) per workgroup: 128 output pixels (256 channels) for the large maps,
  // 64 below 256 K pixels (measured: tools/dev/upstats_bench.py; the 2C global atomics per workgroup are not what limits)
  const int round_px = pl * 4;
  long long ppb = npix >= (1 << 18) ? 4 * round_px : 2 * round_px;
  if (dastune::get(dastune::ELEM_UPSTATS_PPB) > 0) ppb = dastune::get(dastune::ELEM_UPSTATS_PPB);
  const long long grid = (npix + ppb - 1) / ppb;
  const size_t lds = stats ? (size_t)pl * 2 * C * sizeof(float) : 0;
  DISPATCH_T(dtype, {
    if (stats)
      hipLaunchKernelGGL((bilinear_ac_stats_kernel<T, true>), dim3((unsigned)grid), dim3(TPB), lds, (hipStream_t)stream,
                         (const T*)x, (T*)y, B, H, W, C, Ho, Wo, sh, sw, (int)ppb, stats, stats_slots);
    else
      hipLaunchKernelGGL((bilinear_ac_stats_kernel<T, false>), dim3((unsigned)grid), dim3(TPB), 0, (hipStream_t)stream,
                         (const T*)x, (T*)y, B, H, W, C, Ho, Wo, sh, sw, (int)ppb, nullptr, 1);
  });
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}

extern "C" int das_upsample_bilinear_ac(const void* x, void* y, int dtype, int B, int H, int W, int C, int Ho,
                                        int Wo, void* stream) {
  DAS_PROF(stream);
  if (!y) return DAS_ERR_ARG;
  return launch_bilinear_ac(x, y, dtype, B, H, W, C, Ho, Wo, nullptr, 1, stream);
}

extern "C" int das_upsample_bilinear_ac_stats(const void* x, void* y, int dtype, int B, int H, int W, int C, int Ho, int Wo,
                                              float* stats, int stats_slots, void* stream) {
  DAS_PROF(stream);
  if (!stats) return DAS_ERR_ARG;
  return launch_bilinear_ac(x, y, dtype, B, H, W, C, Ho, Wo, stats, stats_slots, stream);
}

extern "C" int das_add_upsample_nearest(const void* a, const void* b, void* y, int dtype, int B, int H, int W, int C,
                                        int Hb, int Wb, void* stream) {
  DAS_PROF(stream);
  if (!a || !b || !y || C % 8) return DAS_ERR_ARG;
  const float sh = (float)Hb / (float)H, sw = (float)Wb / (float)W;
  DISPATCH_T(dtype, {
    const long long total = (long long)B * H * W * (C / Elem<T>::EPV);
    hipLaunchKernelGGL(add_nearest_kernel<T>, dim3(grid_for(total)), dim3(TPB), 0, (hipStream_t)stream, (const T*)a,
                       (const T*)b, (T*)y, B, H, W, C, Hb, Wb, sh, sw);
  });
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}

extern "C" int das_add3(const void* a, const void* b, const void* c, void* y, int dtype, long long n, int relu,
                        void* stream) {
  DAS_PROF(stream);
  if (!a || !b || !y || n % 8) return DAS_ERR_ARG;
  DISPATCH_T(dtype, {
    const long long nvec = n / Elem<T>::EPV;
    hipLaunchKernelGGL(add3_kernel<T>, dim3(grid_for(nvec)), dim3(TPB), 0, (hipStream_t)stream, (const T*)a,
                       (const T*)b, (const T*)c, (T*)y, nvec, relu);
  });
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}

extern "C" int das_abi_version(void) { return 4; }
extern "C" const char* das_target_arch(void) { return "gfx950"; }
