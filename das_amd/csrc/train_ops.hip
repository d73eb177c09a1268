// Backward kernels of the conv / norm units (training path).
//   conv data-gradient  : das_conv2d_nhwc itself on flipped/transposed weights (+ in_up for strides)
//   conv weight-gradient: conv_wgrad_kernel below — GEMM dW[o][k] = sum_m dY[m][o] * Xcol[m][k] whose
//                         reduction runs over pixel rows, i.e. over the slow axis of both NHWC operands.
//                         bf16: tiles are DMA'd row-major into LDS and transposed for free by
//                         ds_read_b64_tr_b16 while building the MFMA fragments; f32: 16x16x4 MFMA whose
//                         fragments are single dwords, so no transpose is needed.
//   BatchNorm (train) backward, GroupNorm backward, column sums (bias gradients).
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "conv_common.h"

using namespace dasconv;

__device__ uint4 g_das_zero_page_train[8];  // this translation unit's zero page (no -fgpu-rdc)

namespace {
constexpr int TPB = 256;
inline int grid_for(long long n, int cap = 8192) {
  long long b = (n + TPB - 1) / TPB;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

__device__ __forceinline__ void dma16(const void* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

typedef short v4i16_t __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------ weight gradient
template <typename T>
struct WG;
template <>
struct WG<bf16_t> {
  static constexpr int ROWB = 256;  // 128 channels x 2 B
  static __device__ __forceinline__ int swz(int slot, int row) { return slot ^ ((row & 7) << 1); }
};
template <>
struct WG<float> {
  static constexpr int ROWB = 512;
  static __device__ __forceinline__ int swz(int slot, int row) { return slot; }
};

// Walks an output-pixel row index forward without divisions (the pixel-row reduction advances every lane's
// row by a constant each step); falls back to a full decode only when a ragged level boundary is crossed.
struct RowWalk {
  int b, h, w;   // image, output row, output column inside the current level
  RowGeom g;     // input-space geometry of the current row
  int Ho, Wo;    // output plane size of the current level
  long long m;
};
__device__ __forceinline__ void walk_refresh(const ConvP& p, RowWalk& r) {
  r.g = row_geom(p, (int)min(r.m, (long long)p.M));
  if (r.m >= p.M) return;
  if (p.nlev <= 1) {
    const int mm = (int)r.m;
    r.b = mm / p.HoWo;
    const int rem = mm - r.b * p.HoWo;
    r.h = rem / p.Wo; r.w = rem - r.h * p.Wo; r.Ho = p.Ho; r.Wo = p.Wo;
  } else {
    r.Ho = r.g.H; r.Wo = r.g.W;
    r.h = r.g.hi0 + p.pad; r.w = r.g.wi0 + p.pad;
    int l = 0;
    for (int i = 1; i < MAXLV; ++i)
      if (i < p.nlev && r.m >= p.lvStart[i]) l = i;
    r.b = (int)((r.m - p.lvStart[l]) / (r.Ho * r.Wo));
  }
}
__device__ __forceinline__ void walk_advance(const ConvP& p, RowWalk& r, int n) {
  r.m += n;
  if (r.m >= p.M) { r.g.hi0 = -(1 << 28); return; }
  r.w += n;
  while (r.w >= r.Wo) { r.w -= r.Wo; ++r.h; }
  bool crossed = false;
  while (r.h >= r.Ho) { r.h -= r.Ho; ++r.b; crossed = true; }
  if (crossed && (p.nlev > 1 && r.b >= p.B)) { walk_refresh(p, r); return; }   // next ragged level
  if (crossed) r.g.pix0 = (p.nlev <= 1) ? (long long)r.b * p.H * p.W : r.g.pix0;  // (ragged: fixed below)
  if (p.nlev <= 1) {
    r.g.hi0 = r.h * p.stride - p.pad;
    r.g.wi0 = r.w * p.stride - p.pad;
  } else {
    if (crossed) { walk_refresh(p, r); return; }  // new image inside a level: recompute the plane origin
    r.g.hi0 = r.h - p.pad;
    r.g.wi0 = r.w - p.pad;
  }
}

// p.x = forward input X, p.res = dY (pixel stride p.rps), p.y = dW f32 [Cout][K] (atomically added to).
template <typename T, int BKM>   // BKM = pixel rows per step
__global__ __launch_bounds__(256) void conv_wgrad_kernel(ConvP p, int steps_per_block) {
  constexpr int EPV = Elem<T>::EPV;
  constexpr int ROWB = WG<T>::ROWB, SLOTS = ROWB / 16;
  constexpr int TILE = BKM * ROWB;             // bytes per operand tile
  constexpr int RPI = 1024 / ROWB;             // rows per wave-instruction
  constexpr int IPW = BKM / RPI / 4;           // DMA instructions per wave per tile
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntiles = (p.K + 127) / 128;
  // XCD-aware order: all (Cout, K) tiles of one pixel chunk get consecutive logical ids, i.e. run on the same
  // XCD at the same time, so the chunk's dY / X rows are fetched into ONE L2 instead of eight
  const int tiles = gridDim.x;
  const int logical = xcd_remap(blockIdx.x + blockIdx.y * tiles, tiles * gridDim.y);
  const int tile = logical % tiles, chunk = logical / tiles;
  const int o0 = (tile / ntiles) * 128, n0 = (tile % ntiles) * 128;
  const int wave_o0 = (wave >> 1) * 64, wave_n0 = (wave & 1) * 64;
  const long long m_begin = (long long)chunk * steps_per_block * BKM;

  const T* xg = reinterpret_cast<const T*>(p.x);
  const T* dyg = reinterpret_cast<const T*>(p.res);
  const T* zero = reinterpret_cast<const T*>(g_das_zero_page_train);

  // per (lane, instruction) constants: dY channel, X column -> (tap, ci); plus the row walker
  int och[IPW], xkh[IPW], xkw[IPW], xci[IPW];
  bool ook[IPW], nok[IPW];
  RowWalk rw[IPW];
#pragma unroll
  for (int j = 0; j < IPW; ++j) {
    const int row = (wave * IPW + j) * RPI + lane / SLOTS;
    const int logical = WG<T>::swz(lane % SLOTS, row);
    och[j] = o0 + logical * EPV;
    ook[j] = och[j] < p.Cout;
    const int n = n0 + logical * EPV;
    nok[j] = n < p.K;
    const int tap = n / p.Cin;
    xci[j] = n - tap * p.Cin;
    xkh[j] = tap / p.KW;
    xkw[j] = tap - xkh[j] * p.KW;
    rw[j].m = m_begin + row;
    walk_refresh(p, rw[j]);
  }

  auto issue = [&](int buf) {  // DMA the tiles of the walkers' current rows, then advance them one step
    char* sD = smem + buf * 2 * TILE;
    char* sX = sD + TILE;
#pragma unroll
    for (int j = 0; j < IPW; ++j) {
      const bool mok = rw[j].m < p.M;
      const T* sd = (mok && ook[j]) ? dyg + rw[j].m * p.rps + och[j] : zero;
      dma16(sd, sD + (wave * IPW + j) * 1024);
      const RowGeom& g = rw[j].g;
      const int hi = g.hi0 + xkh[j], wi = g.wi0 + xkw[j];
      const bool ok = mok && nok[j] && (unsigned)hi < (unsigned)g.H && (unsigned)wi < (unsigned)g.W;
      const T* sx = ok ? xg + (g.pix0 + (long long)hi * g.W + wi) * p.xps + xci[j] : zero;
      dma16(sx, sX + (wave * IPW + j) * 1024);
      walk_advance(p, rw[j], BKM);
    }
  };

  f32x4_t acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const long long m_left = (long long)p.M - m_begin;
  const int nsteps = (int)std::min<long long>(steps_per_block, (m_left + BKM - 1) / BKM);
  if (nsteps <= 0) return;
  issue(0);
  __syncthreads();
  const int g4 = lane >> 4, q = lane & 15;
  for (int s = 0; s < nsteps; ++s) {
    const int buf = s & 1;
    if (s + 1 < nsteps) issue(buf ^ 1);
    const char* sD = smem + buf * 2 * TILE;
    const char* sX = sD + TILE;
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int half = 0; half < BKM / 32; ++half) {
        uint4 fa[4], fb[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          v4i16_t lo[2], hi[2];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int row = half * 32 + h * 16 + g4 * 4 + (q >> 2);
            const int sub = q & 3;
            const int ca = wave_o0 + t * 16, cb = wave_n0 + t * 16;
            const int sa = WG<T>::swz((ca >> 3) + (sub >> 1), row), sb = WG<T>::swz((cb >> 3) + (sub >> 1), row);
            lo[h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) v4i16_t*)(sD + row * ROWB + sa * 16 + (sub & 1) * 8));
            hi[h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) v4i16_t*)(sX + row * ROWB + sb * 16 + (sub & 1) * 8));
          }
          fa[t] = make_uint4(__builtin_bit_cast(uint2, lo[0]).x, __builtin_bit_cast(uint2, lo[0]).y,
                             __builtin_bit_cast(uint2, lo[1]).x, __builtin_bit_cast(uint2, lo[1]).y);
          fb[t] = make_uint4(__builtin_bit_cast(uint2, hi[0]).x, __builtin_bit_cast(uint2, hi[0]).y,
                             __builtin_bit_cast(uint2, hi[1]).x, __builtin_bit_cast(uint2, hi[1]).y);
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa[a]),
                                                                __builtin_bit_cast(bf16x8_t, fb[b]), acc[a][b], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < BKM / 4; ++kk) {
        float fa[4], fb[4];
        const int row = kk * 4 + g4;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          fa[t] = *reinterpret_cast<const float*>(sD + row * ROWB + (wave_o0 + t * 16 + q) * 4);
          fb[t] = *reinterpret_cast<const float*>(sX + row * ROWB + (wave_n0 + t * 16 + q) * 4);
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[a], fb[b], acc[a][b], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  float* dw = reinterpret_cast<float*>(p.y);
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int n = n0 + wave_n0 + b * 16 + q;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int o = o0 + wave_o0 + a * 16 + g4 * 4 + j;
        if (o < p.Cout && n < p.K) atomicAdd(dw + (long long)o * p.K + n, acc[a][b][j]);
      }
    }
}

// ------------------------------------------------------------------ column sums (bias gradient)
template <typename T>
__global__ void colsum_kernel(const T* __restrict__ x, long long rows, int C, int ps, float* __restrict__ out) {
  constexpr int EPV = Elem<T>::EPV;
  extern __shared__ float sred[];
  const int VC = C / EPV;
  for (int i = threadIdx.x; i < C; i += TPB) sred[i] = 0.f;
  __syncthreads();
  const int VCB = min(VC, TPB);  // channel vectors handled side by side (wider rows loop over v)
  const int pl = threadIdx.x / VCB, PL = TPB / VCB;
  for (int v = threadIdx.x % VCB; v < VC && pl < PL; v += VCB) {
    float s[EPV];
#pragma unroll
    for (int j = 0; j < EPV; ++j) s[j] = 0.f;
    for (long long r = (long long)blockIdx.x * PL + pl; r < rows; r += (long long)gridDim.x * PL) {
      float f[EPV];
      Elem<T>::unpack(*reinterpret_cast<const uint4*>(x + r * ps + v * EPV), f);
#pragma unroll
      for (int j = 0; j < EPV; ++j) s[j] += f[j];
    }
#pragma unroll
    for (int j = 0; j < EPV; ++j) atomicAdd(&sred[v * EPV + j], s[j]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C; i += TPB) atomicAdd(out + i, sred[i]);
}

// ------------------------------------------------------------------ BatchNorm (train) backward
// MASK: 0 = no ReLU, 1 = ReLU mask from the saved output y (needed when a residual was added before the
// ReLU), 2 = ReLU mask recomputed from raw (y > 0 <=> bn_affine(raw) > 0): one tensor read less per pass.
template <typename T, int MASK>
__device__ __forceinline__ void bn_masked_grad(float* g, const float* x, const uint4& yv, const float* mu,
                                               const float* is, const float* ga, const float* be) {
  constexpr int EPV = Elem<T>::EPV;
  if (MASK == 1) {
    float o[EPV];
    Elem<T>::unpack(yv, o);
#pragma unroll
    for (int j = 0; j < EPV; ++j) g[j] = o[j] > 0.f ? g[j] : 0.f;
  } else if (MASK == 2) {
#pragma unroll
    for (int j = 0; j < EPV; ++j) g[j] = bn_affine(x[j], mu[j], is[j], ga[j], be[j]) > 0.f ? g[j] : 0.f;
  }
}

// pass 1: s1[c] = sum dZ, s2[c] = sum dZ * xhat with dZ = dY * mask. Four rows in flight per thread.
template <typename T, int MASK, int NT>
__global__ __launch_bounds__(NT) void bn_bwd_reduce_kernel(const T* __restrict__ dy, const T* __restrict__ y,
                                                           const T* __restrict__ raw, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, long long rows, int C,
                                                           float* __restrict__ sums) {
  constexpr int EPV = Elem<T>::EPV;
  constexpr int U = NT > 256 ? 2 : 4;
  extern __shared__ float sred[];  // [2C]
  const int VC = C / EPV;
  for (int i = threadIdx.x; i < 2 * C; i += NT) sred[i] = 0.f;
  __syncthreads();
  const int VCB = min(VC, NT);
  const int pl = threadIdx.x / VCB, PL = NT / VCB;
  const long long rstride = (long long)gridDim.x * PL;
  for (int v = threadIdx.x % VCB; v < VC && pl < PL; v += VCB) {
    float s1[EPV], s2[EPV], mu[EPV], is[EPV], ga[EPV], be[EPV];
#pragma unroll
    for (int j = 0; j < EPV; ++j) {
      s1[j] = 0.f; s2[j] = 0.f; mu[j] = mean[v * EPV + j]; is[j] = invstd[v * EPV + j];
      ga[j] = MASK == 2 ? gamma[v * EPV + j] : 0.f; be[j] = MASK == 2 ? beta[v * EPV + j] : 0.f;
    }
    for (long long r = (long long)blockIdx.x * PL + pl; r < rows; r += rstride * U) {
      uint4 gv[U], xv[U], yv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long ru = r + u * rstride;
        gv[u] = xv[u] = yv[u] = make_uint4(0, 0, 0, 0);
        if (ru < rows) {
          gv[u] = *reinterpret_cast<const uint4*>(dy + ru * C + v * EPV);
          xv[u] = *reinterpret_cast<const uint4*>(raw + ru * C + v * EPV);
          if (MASK == 1) yv[u] = *reinterpret_cast<const uint4*>(y + ru * C + v * EPV);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {  // rows past the end carry dY = 0 and add nothing
        float g[EPV], x[EPV];
        Elem<T>::unpack(gv[u], g);
        Elem<T>::unpack(xv[u], x);
        bn_masked_grad<T, MASK>(g, x, yv[u], mu, is, ga, be);
#pragma unroll
        for (int j = 0; j < EPV; ++j) { s1[j] += g[j]; s2[j] += g[j] * (x[j] - mu[j]) * is[j]; }
      }
    }
#pragma unroll
    for (int j = 0; j < EPV; ++j) {
      atomicAdd(&sred[v * EPV + j], s1[j]);
      atomicAdd(&sred[C + v * EPV + j], s2[j]);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += NT) atomicAdd(sums + i, sred[i]);
}

// pass 2: dRaw = gamma*invstd*(dZ - s1/N - xhat*s2/N); optionally dRes = dZ. Block 0 also adds the two
// sums into the parameter-gradient accumulators (dbeta += s1, dgamma += s2) when they are given.
template <typename T, int MASK, bool FIXED>
__global__ void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ y, const T* __restrict__ raw,
                                    const float* __restrict__ mean, const float* __restrict__ invstd,
                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                    const float* __restrict__ sums, long long rows, int C, T* __restrict__ draw,
                                    T* __restrict__ dres, float* __restrict__ dgamma_acc,
                                    float* __restrict__ dbeta_acc, float inv_n) {
  constexpr int EPV = Elem<T>::EPV;
  const int VC = C / EPV;
  const long long total = rows * VC;
  const long long stride = (long long)gridDim.x * TPB;
  if (blockIdx.x == 0 && dgamma_acc) {
    for (int c = threadIdx.x; c < C; c += TPB) { dbeta_acc[c] += sums[c]; dgamma_acc[c] += sums[C + c]; }
  }
  float mu[EPV], is[EPV], ga[EPV], be[EPV], k1[EPV], k2[EPV], k3[EPV];
  auto load_consts = [&](int c0) {
#pragma unroll
    for (int j = 0; j < EPV; ++j) {
      const int c = c0 + j;
      mu[j] = mean[c]; is[j] = invstd[c]; ga[j] = gamma[c]; be[j] = MASK == 2 ? beta[c] : 0.f;
      k1[j] = ga[j] * is[j]; k2[j] = sums[c] * inv_n; k3[j] = sums[C + c] * inv_n * is[j];
    }
  };
  auto finish = [&](long long i, const uint4& gv, const uint4& xv, const uint4& yv) {
    float g[EPV], x[EPV], o[EPV];
    Elem<T>::unpack(gv, g);
    Elem<T>::unpack(xv, x);
    bn_masked_grad<T, MASK>(g, x, yv, mu, is, ga, be);
    if (dres) *reinterpret_cast<uint4*>(dres + i * EPV) = Elem<T>::pack(g);
#pragma unroll
    for (int j = 0; j < EPV; ++j) o[j] = k1[j] * (g[j] - k2[j] - (x[j] - mu[j]) * k3[j]);
    *reinterpret_cast<uint4*>(draw + i * EPV) = Elem<T>::pack(o);
  };
  long long i = (long long)blockIdx.x * TPB + threadIdx.x;
  const uint4 z4 = make_uint4(0, 0, 0, 0);
  if (FIXED) {
    if (i < total) load_consts((int)(i % VC) * EPV);
    for (; i + stride < total; i += 2 * stride) {
      const long long i1 = i + stride;
      const uint4 g0 = *reinterpret_cast<const uint4*>(dy + i * EPV), g1 = *reinterpret_cast<const uint4*>(dy + i1 * EPV);
      const uint4 x0 = *reinterpret_cast<const uint4*>(raw + i * EPV), x1 = *reinterpret_cast<const uint4*>(raw + i1 * EPV);
      uint4 y0 = z4, y1 = z4;
      if (MASK == 1) { y0 = *reinterpret_cast<const uint4*>(y + i * EPV); y1 = *reinterpret_cast<const uint4*>(y + i1 * EPV); }
      finish(i, g0, x0, y0);
      finish(i1, g1, x1, y1);
    }
  }
  for (; i < total; i += stride) {
    if (!FIXED) load_consts((int)(i % VC) * EPV);
    const uint4 g0 = *reinterpret_cast<const uint4*>(dy + i * EPV);
    const uint4 x0 = *reinterpret_cast<const uint4*>(raw + i * EPV);
    uint4 y0 = z4;
    if (MASK == 1) y0 = *reinterpret_cast<const uint4*>(y + i * EPV);
    finish(i, g0, x0, y0);
  }
}

template <typename T, int MASK>
void launch_bn_backward(const void* dy, const void* y, const void* raw, long long rows, int C, const float* mean,
                        const float* invstd, const float* gamma, const float* beta, void* draw, void* dres,
                        float* sums, float* dgamma_acc, float* dbeta_acc, int phase, long long stat_rows,
                        hipStream_t s) {
  const int vc = C / Elem<T>::EPV;
  if (phase != 2) {   // pass 1: per-channel sums
    static const char* dev_cfg = getenv("DAS_DEV_BN_REDUCE");  // "<blocks>,<threads>" (tuning only)
    int cap = 256, nt = 256;  // every block ends in 2C global atomics on the same words: ~20 ns per block of tail
    if (dev_cfg) sscanf(dev_cfg, "%d,%d", &cap, &nt);
    const int blocks = (int)std::min<long long>(cap, std::max<long long>(1, rows / 64));
    if (nt == 1024) {
      hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, MASK, 1024>), dim3(blocks), dim3(1024), 2 * C * sizeof(float), s,
                         (const T*)dy, (const T*)y, (const T*)raw, mean, invstd, gamma, beta, rows, C, sums);
    } else {
      hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, MASK, 256>), dim3(blocks), dim3(256), 2 * C * sizeof(float), s,
                         (const T*)dy, (const T*)y, (const T*)raw, mean, invstd, gamma, beta, rows, C, sums);
    }
  }
  if (phase == 1) return;
  const float inv_n = 1.f / (float)stat_rows;   // statistics population (all ranks' rows for SyncBN)
  const int grid = grid_for(rows * vc);
  if (((long long)grid * TPB) % vc == 0) {
    hipLaunchKernelGGL((bn_bwd_apply_kernel<T, MASK, true>), dim3(grid), dim3(TPB), 0, s, (const T*)dy, (const T*)y,
                       (const T*)raw, mean, invstd, gamma, beta, sums, rows, C, (T*)draw, (T*)dres, dgamma_acc,
                       dbeta_acc, inv_n);
  } else {
    hipLaunchKernelGGL((bn_bwd_apply_kernel<T, MASK, false>), dim3(grid), dim3(TPB), 0, s, (const T*)dy, (const T*)y,
                       (const T*)raw, mean, invstd, gamma, beta, sums, rows, C, (T*)draw, (T*)dres, dgamma_acc,
                       dbeta_acc, inv_n);
  }
}
}  // namespace

extern "C" int das_conv2d_wgrad_nhwc(const void* x, const void* dy, float* dw, const DasConvDesc* d, int accumulate,
                                     void* stream) {
  if (!x || !dy || !dw || !d) return DAS_ERR_ARG;
  if (d->Cin % 8 || d->Cout % 8 || d->x_pix_stride % 8 || d->y_pix_stride % 8) return DAS_ERR_ARG;
  if (d->KH < 1 || d->KW < 1 || d->stride < 1 || d->B < 1 || d->in_up > 1) return DAS_ERR_ARG;
  ConvP p;
  long long M = (long long)d->B * d->Ho * d->Wo;
  p.nlev = d->num_levels;
  p.B = d->B;
  if (p.nlev > 1) {
    if (p.nlev > MAXLV || d->stride != 1 || d->KH != d->KW || d->pad != d->KH / 2) return DAS_ERR_ARG;
    M = 0;
    for (int l = 0; l < p.nlev; ++l) {
      p.lvH[l] = d->lvl_H[l]; p.lvW[l] = d->lvl_W[l]; p.lvStart[l] = (int)M;
      M += (long long)d->B * d->lvl_H[l] * d->lvl_W[l];
    }
  }
  for (int l = (p.nlev > 1 ? p.nlev : 0); l < MAXLV; ++l) { p.lvH[l] = 0; p.lvW[l] = 0; p.lvStart[l] = 0x7fffffff; }
  if (M <= 0 || M > 0x7fffffffLL) return DAS_ERR_ARG;
  p.x = (const char*)x; p.w = nullptr; p.y = (char*)dw; p.res = (const char*)dy;
  p.scale = p.shift = nullptr; p.stats = nullptr;
  p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.xps = d->x_pix_stride;
  p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout; p.yps = 0; p.rps = d->y_pix_stride;
  p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad;
  p.relu_in = p.relu = 0; p.up_sh = 0;
  p.M = (int)M; p.K = d->KH * d->KW * d->Cin; p.HoWo = d->Ho * d->Wo;
  p.ntiles = p.nblocks = 0; p.xbytes = 0;
  hipStream_t s = (hipStream_t)stream;
  if (!accumulate && hipMemsetAsync(dw, 0, sizeof(float) * (size_t)d->Cout * p.K, s) != hipSuccess) return DAS_ERR_LAUNCH;
  const int tiles = ((d->Cout + 127) / 128) * ((p.K + 127) / 128);
  // bf16: 32 pixel rows per step = 32 KiB of LDS per workgroup, three workgroups resident per CU (register bound):
  // more independent DMA -> MFMA chains in flight than two workgroups of 64-row steps (+10...14 % measured)
  static const char* dev_bkm = getenv("DAS_DEV_WGRAD_BKM");  // tuning only
  const int bkm = d->dtype == DAS_BF16 ? (dev_bkm ? atoi(dev_bkm) : 32) : 32;
  const long long total_steps = (M + bkm - 1) / bkm;
  // Split the pixel reduction so that the whole grid is ONE resident wave of workgroups (3 per CU for bf16 at
  // 32-row steps, 2 per CU for f32: 64 KiB of LDS each) — a second, partial wave costs a full pass, and every
  // extra split is one more round of atomics on the same dW words. At least 8 steps per workgroup.
  static const char* dev_blocks = getenv("DAS_DEV_WGRAD_BLOCKS");  // tuning only
  const int target = dev_blocks ? atoi(dev_blocks) : (d->dtype == DAS_BF16 && bkm == 32 ? 768 : 512);
  long long splits = std::max<long long>(1, target / tiles);
  long long spb = std::max<long long>(8, (total_steps + splits - 1) / splits);
  splits = (total_steps + spb - 1) / spb;
  if (d->dtype == DAS_BF16) {
    const size_t sm = 2 * 2 * (size_t)bkm * 256;
    if (bkm == 32) {
      hipLaunchKernelGGL((conv_wgrad_kernel<bf16_t, 32>), dim3(tiles, (unsigned)splits), dim3(256), sm, s, p, (int)spb);
    } else {
      hipLaunchKernelGGL((conv_wgrad_kernel<bf16_t, 64>), dim3(tiles, (unsigned)splits), dim3(256), sm, s, p, (int)spb);
    }
  } else if (d->dtype == DAS_F32) {
    const size_t sm = 2 * 2 * 32 * 512;
    hipLaunchKernelGGL((conv_wgrad_kernel<float, 32>), dim3(tiles, (unsigned)splits), dim3(256), sm, s, p, (int)spb);
  } else {
    return DAS_ERR_ARG;
  }
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}

extern "C" int das_colsum(const void* x, int dtype, long long rows, int C, int pix_stride, float* out, void* stream) {
  if (!x || !out || rows <= 0 || C % 8 || pix_stride % 8 || C > 2048) return DAS_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(out, 0, sizeof(float) * C, s) != hipSuccess) return DAS_ERR_LAUNCH;
  const int blocks = (int)std::min<long long>(256, std::max<long long>(1, rows / 64));
  if (dtype == DAS_BF16) {
    hipLaunchKernelGGL(colsum_kernel<bf16_t>, dim3(blocks), dim3(TPB), C * sizeof(float), s, (const bf16_t*)x, rows, C,
                       pix_stride, out);
  } else if (dtype == DAS_F32) {
    hipLaunchKernelGGL(colsum_kernel<float>, dim3(blocks), dim3(TPB), C * sizeof(float), s, (const float*)x, rows, C,
                       pix_stride, out);
  } else {
    return DAS_ERR_ARG;
  }
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}

extern "C" int das_bn_train_backward_phase(const void* dy, const void* y, const void* raw, int dtype, long long rows,
                                           int C, const float* mean, const float* invstd, const float* gamma,
                                           const float* beta, int relu, void* draw, void* dres, float* sums,
                                           int sums_prezeroed, float* dgamma_acc, float* dbeta_acc, int phase,
                                           long long stat_rows, void* stream) {
  if (!dy || !raw || !mean || !invstd || !gamma || !sums || rows <= 0 || C % 8 || C > 2048) return DAS_ERR_ARG;
  if (phase < 0 || phase > 2 || (phase != 1 && !draw) || stat_rows < rows) return DAS_ERR_ARG;
  if (relu && !y && !beta) return DAS_ERR_ARG;
  if ((dgamma_acc == nullptr) != (dbeta_acc == nullptr)) return DAS_ERR_ARG;
  if (dtype != DAS_BF16 && dtype != DAS_F32) return DAS_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (phase != 2 && !sums_prezeroed && hipMemsetAsync(sums, 0, sizeof(float) * 2 * C, s) != hipSuccess)
    return DAS_ERR_LAUNCH;
  const int mask = !relu ? 0 : (y ? 1 : 2);
#define DAS_BN_BWD(T, MASK)                                                                                         \
  launch_bn_backward<T, MASK>(dy, y, raw, rows, C, mean, invstd, gamma, beta, draw, dres, sums, dgamma_acc, dbeta_acc, \
                              phase, stat_rows, s)
  if (dtype == DAS_BF16) {
    if (mask == 0) DAS_BN_BWD(bf16_t, 0); else if (mask == 1) DAS_BN_BWD(bf16_t, 1); else DAS_BN_BWD(bf16_t, 2);
  } else {
    if (mask == 0) DAS_BN_BWD(float, 0); else if (mask == 1) DAS_BN_BWD(float, 1); else DAS_BN_BWD(float, 2);
  }
#undef DAS_BN_BWD
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}

extern "C" int das_bn_train_backward(const void* dy, const void* y, const void* raw, int dtype, long long rows, int C,
                                     const float* mean, const float* invstd, const float* gamma, const float* beta,
                                     int relu, void* draw, void* dres, float* sums, int sums_prezeroed,
                                     float* dgamma_acc, float* dbeta_acc, void* stream) {
  return das_bn_train_backward_phase(dy, y, raw, dtype, rows, C, mean, invstd, gamma, beta, relu, draw, dres, sums,
                                     sums_prezeroed, dgamma_acc, dbeta_acc, 0, rows, stream);
}
