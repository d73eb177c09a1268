// Backward of the HBM-bound NHWC helpers: GroupNorm(+ReLU), 3x3/s2 max-pool, bilinear
// (align_corners=True) upsampling, nearest upsample-add. All in "gather" form (each thread owns one
// 16-byte channel vector of the gradient it produces), so results are deterministic and need no atomics
// except the per-channel / per-group statistic reductions.
#include <algorithm>
#include "prof.h"

#include "common.h"
#include "tuning.h"

namespace {
constexpr int TPB = 256;
inline int grid_for(long long n, int cap = 8192) {
  long long b = (n + TPB - 1) / TPB;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

// ------------------------------------------------------------------ GroupNorm backward
// pass 1: per channel a[c] = sum dZ, b[c] = sum dZ*xhat over the block's pixels of one (level,image)
// segment; dgamma/dbeta get them directly, the group sums s1 = sum gamma*a, s2 = sum gamma*b per segment.
// Both passes: blockIdx.y = segment (level, image), blockIdx.x = a run of pix_per_block pixels; a thread keeps one 16-byte
// channel vector (its groups' constants in registers) and walks the run TPB / VC pixels apart, two pixels' three
// operands in flight (as the forward passes, norm.hip: no per-element geometry, levels picked with static indices).
constexpr int GN_NT = 1024;
struct GnSeg {
  int HW;
  long long row0;
};
__device__ __forceinline__ GnSeg gn_segment(const DasLevels& lv, int seg) {
  const int l = seg / lv.B, b = seg - l * lv.B;
  GnSeg g{0, 0};
  long long start = 0;
#pragma unroll
  for (int i = 0; i < DAS_MAX_LEVELS; ++i) {
    const int hw = i < lv.num_levels ? lv.H[i] * lv.W[i] : 0;
    if (i == l) { g.HW = hw; g.row0 = start + (long long)b * hw; }
    start += (long long)lv.B * hw;
  }
  return g;
}

template <typename T>
__global__ __launch_bounds__(GN_NT) void gn_bwd_reduce_kernel(const T* __restrict__ dy, const T* __restrict__ y,
                                                              const T* __restrict__ x, DasLevels lv, int C, int ps, int G,
                                                              int pix_per_block, const float* __restrict__ fstats,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float eps, int relu,
                                                              float* __restrict__ gsums, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta) {
  constexpr int EPV = Elem<T>::EPV, NT = GN_NT;
  extern __shared__ float part[];  // [rows][2C] (common.h: lds_put / lds_fold), folded in place into row 0
  const int seg = blockIdx.y;
  const GnSeg sg = gn_segment(lv, seg);
  const int HW = sg.HW;
  const int p0 = blockIdx.x * pix_per_block;
  if (p0 >= HW) return;
  const int VC = C / EPV, cpg = C / G;
  const float inv_n = 1.f / ((float)HW * (float)cpg);
  const int v = threadIdx.x % VC, pl = threadIdx.x / VC, PL = NT / VC;
  if (pl < PL) {
    // relu without y: the mask is recomputed from x (y > 0 <=> gn_affine(x) > 0, the forward's own arithmetic) — y is not read
    const bool remask = relu && !y;
    float a[EPV], bb[EPV], mu[EPV], rs[EPV], ga[EPV], be[EPV];
#pragma unroll
    for (int j = 0; j < EPV; ++j) {
      const int g = (v * EPV + j) / cpg;
      const float m = fstats[((long long)seg * G + g) * 2] * inv_n;
      const float var = fmaxf(fstats[((long long)seg * G + g) * 2 + 1] * inv_n - m * m, 0.f);
      mu[j] = m; rs[j] = rsqrtf(var + eps); a[j] = 0.f; bb[j] = 0.f;
      ga[j] = remask ? gamma[v * EPV + j] : 0.f; be[j] = remask ? beta[v * EPV + j] : 0.f;
    }
    const int p1 = min(p0 + pix_per_block, HW);
    const long long off = sg.row0 * ps + v * EPV;
    auto acc = [&](const uint4& rg, const uint4& rx, const uint4& ry) {
      float g[EPV], xx[EPV];
      Elem<T>::unpack(rg, g);
      Elem<T>::unpack(rx, xx);
      if (remask) {
#pragma unroll
        for (int j = 0; j < EPV; ++j) g[j] = gn_affine(xx[j], mu[j], rs[j], ga[j], be[j]) > 0.f ? g[j] : 0.f;
      } else if (relu) {
        float o[EPV];
        Elem<T>::unpack(ry, o);
#pragma unroll
        for (int j = 0; j < EPV; ++j) g[j] = o[j] > 0.f ? g[j] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < EPV; ++j) { a[j] += g[j]; bb[j] += g[j] * (xx[j] - mu[j]) * rs[j]; }
    };
    int p = p0 + pl;
    for (; p + 3 * PL < p1; p += 4 * PL) {   // four pixels' operands in flight (two left this pass at 3.2 TB/s)
      uint4 rg[4], rx[4], ry[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long long o = off + (long long)(p + u * PL) * ps;
        rg[u] = *reinterpret_cast<const uint4*>(dy + o);
        rx[u] = *reinterpret_cast<const uint4*>(x + o);
        ry[u] = (relu && y) ? *reinterpret_cast<const uint4*>(y + o) : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) acc(rg[u], rx[u], ry[u]);
    }
    for (; p < p1; p += PL) {
      const long long o = off + (long long)p * ps;
      acc(*reinterpret_cast<const uint4*>(dy + o), *reinterpret_cast<const uint4*>(x + o),
          (relu && y) ? *reinterpret_cast<const uint4*>(y + o) : make_uint4(0, 0, 0, 0));
    }
    if (VC < 64 && 64 % VC == 0) {
      // the 64 / VC pixel lanes of a wave that share this channel vector: xor-shuffles, then one lane per vector stores the
      // wave's sums (rows = waves)
      for (int m = VC; m < 64; m <<= 1) {
#pragma unroll
        for (int j = 0; j < EPV; ++j) {
          a[j] += __shfl_xor(a[j], m);
          bb[j] += __shfl_xor(bb[j], m);
        }
      }
      if ((threadIdx.x & 63) < VC) {
        lds_put<EPV>(part, 2 * C, threadIdx.x >> 6, v * EPV, a);
        lds_put<EPV>(part, 2 * C, threadIdx.x >> 6, C + v * EPV, bb);
      }
    } else {
      lds_put<EPV>(part, 2 * C, pl, v * EPV, a);
      lds_put<EPV>(part, 2 * C, pl, C + v * EPV, bb);
    }
  }
  __syncthreads();
  const int nrows = (VC < 64 && 64 % VC == 0) ? NT / 64 : PL;
  float* sred = part;   // folded in place: column c is read and written by one thread only
  for (int c = threadIdx.x; c < 2 * C; c += NT) {
    const float t = lds_fold(part, 2 * C, nrows, c);
    sred[c] = t;
    atomicAdd((c < C ? dbeta : dgamma - C) + c, t);
  }
  __syncthreads();
  for (int g = threadIdx.x; g < G; g += NT) {
    float s1 = 0.f, s2 = 0.f;
    for (int j = 0; j < cpg; ++j) { s1 += gamma[g * cpg + j] * sred[g * cpg + j]; s2 += gamma[g * cpg + j] * sred[C + g * cpg + j]; }
    atomicAdd(&gsums[((long long)seg * G + g) * 2], s1);
    atomicAdd(&gsums[((long long)seg * G + g) * 2 + 1], s2);
  }
}

template <typename T>
__global__ __launch_bounds__(GN_NT) void gn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ y,
                                                             const T* __restrict__ x, T* __restrict__ dx, DasLevels lv,
                                                             int C, int ps, int G, int pix_per_block,
                                                             const float* __restrict__ fstats,
                                                             const float* __restrict__ gsums,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float eps, int relu) {
  constexpr int EPV = Elem<T>::EPV, NT = GN_NT;
  const int seg = blockIdx.y;
  const GnSeg sg = gn_segment(lv, seg);
  const int p0 = blockIdx.x * pix_per_block;
  if (p0 >= sg.HW) return;
  const int VC = C / EPV, cpg = C / G;
  const int v = threadIdx.x % VC, pl = threadIdx.x / VC, PL = NT / VC;
  if (pl >= PL) return;
  const float inv_n = 1.f / ((float)sg.HW * (float)cpg);
  const bool remask = relu && !y;
  float mu[EPV], rs[EPV], ga[EPV], be[EPV], k1[EPV], k2[EPV];
#pragma unroll
  for (int j = 0; j < EPV; ++j) {
    const int c = v * EPV + j, gi = c / cpg;
    const float m = fstats[((long long)seg * G + gi) * 2] * inv_n;
    const float var = fmaxf(fstats[((long long)seg * G + gi) * 2 + 1] * inv_n - m * m, 0.f);
    mu[j] = m; rs[j] = rsqrtf(var + eps); ga[j] = gamma[c]; be[j] = remask ? beta[c] : 0.f;
    k1[j] = gsums[((long long)seg * G + gi) * 2] * inv_n;
    k2[j] = gsums[((long long)seg * G + gi) * 2 + 1] * inv_n;
  }
  const int p1 = min(p0 + pix_per_block, sg.HW);
  const long long off = sg.row0 * ps + v * EPV;
  auto one = [&](const uint4& rg, const uint4& rx, const uint4& ry) {
    float g[EPV], xx[EPV], o[EPV];
    Elem<T>::unpack(rg, g);
    Elem<T>::unpack(rx, xx);
    if (remask) {
#pragma unroll
      for (int j = 0; j < EPV; ++j) g[j] = gn_affine(xx[j], mu[j], rs[j], ga[j], be[j]) > 0.f ? g[j] : 0.f;
    } else if (relu) {
      float yy[EPV];
      Elem<T>::unpack(ry, yy);
#pragma unroll
      for (int j = 0; j < EPV; ++j) g[j] = yy[j] > 0.f ? g[j] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < EPV; ++j) {
      const float xhat = (xx[j] - mu[j]) * rs[j];
      o[j] = rs[j] * (g[j] * ga[j] - k1[j] - xhat * k2[j]);
    }
    return Elem<T>::pack(o);
  };
  int p = p0 + pl;
  for (; p + PL < p1; p += 2 * PL) {
    uint4 rg[2], rx[2], ry[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long long o = off + (long long)(p + u * PL) * ps;
      rg[u] = *reinterpret_cast<const uint4*>(dy + o);
      rx[u] = *reinterpret_cast<const uint4*>(x + o);
      ry[u] = (relu && y) ? *reinterpret_cast<const uint4*>(y + o) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) *reinterpret_cast<uint4*>(dx + off + (long long)(p + u * PL) * ps) = one(rg[u], rx[u], ry[u]);
  }
  for (; p < p1; p += PL) {
    const long long o = off + (long long)p * ps;
    *reinterpret_cast<uint4*>(dx + o) = one(*reinterpret_cast<const uint4*>(dy + o), *reinterpret_cast<const uint4*>(x + o),
                                            (relu && y) ? *reinterpret_cast<const uint4*>(y + o) : make_uint4(0, 0, 0, 0));
  }
}

// ------------------------------------------------------------------ max-pool 3x3/s2/p1 backward
// torch routes the gradient to the FIRST maximum in (kh, kw) scan order (ties are common after ReLU).
// One thread = one 2x2 block of input pixels x one channel vector. The four pooling windows that touch the
// block, (a..a+1, b..b+1), read a 5x5 patch of inputs — 25 vector loads for 4 outputs instead of 36 per output
// when every input pixel re-derives the maxima of its own windows. No atomics: each input pixel gathers.
template <typename T>
__global__ void maxpool_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, int B, int H,
                                   int W, int C, int Ho, int Wo, long long total) {
  constexpr int EPV = Elem<T>::EPV;
  const int VC = C / EPV;
  const int H2 = (H + 1) / 2, W2 = (W + 1) / 2;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
    const int v = (int)(i % VC);
    long long blk = i / VC;
    const int b2 = (int)(blk % W2);
    blk /= W2;
    const int a2 = (int)(blk % H2);
    const long long b = blk / H2;
    // first-maximum tap of each of the four windows, per channel; -1 = window does not exist
    int amax[4][EPV];
    float g[4][EPV];
#pragma unroll
    for (int wdx = 0; wdx < 4; ++wdx) {
      const int ho = a2 + (wdx >> 1), wo = b2 + (wdx & 1);
      const bool exists = ho < Ho && wo < Wo;
      float best[EPV];
#pragma unroll
      for (int j = 0; j < EPV; ++j) { amax[wdx][j] = -1; best[j] = 0.f; g[wdx][j] = 0.f; }
      if (!exists) continue;
      Elem<T>::unpack(*reinterpret_cast<const uint4*>(dy + ((b * Ho + ho) * Wo + wo) * C + v * EPV), g[wdx]);
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const int yy = ho * 2 - 1 + k / 3, xx = wo * 2 - 1 + k % 3;
        if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
        float f[EPV];
        Elem<T>::unpack(*reinterpret_cast<const uint4*>(x + ((b * H + yy) * W + xx) * C + v * EPV), f);
#pragma unroll
        for (int j = 0; j < EPV; ++j) {
          if (amax[wdx][j] < 0 || f[j] > best[j]) { amax[wdx][j] = k; best[j] = f[j]; }   // strict >: first maximum wins
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int w2 = 0; w2 < 2; ++w2) {
        const int hi = a2 * 2 + u, wi = b2 * 2 + w2;
        if (hi >= H || wi >= W) continue;
        float acc[EPV];
#pragma unroll
        for (int j = 0; j < EPV; ++j) acc[j] = 0.f;
#pragma unroll
        for (int wdx = 0; wdx < 4; ++wdx) {
          const int da = wdx >> 1, db = wdx & 1;
          if ((da == 1 && u == 0) || (db == 1 && w2 == 0)) continue;   // window (a+da, b+db) does not cover this pixel
          const int myk = (u + 1 - 2 * da) * 3 + (w2 + 1 - 2 * db);
#pragma unroll
          for (int j = 0; j < EPV; ++j) acc[j] += amax[wdx][j] == myk ? g[wdx][j] : 0.f;
        }
        *reinterpret_cast<uint4*>(dx + (((b * H + hi) * W + wi) * (long long)C + v * EPV)) = Elem<T>::pack(acc);
      }
  }
}

// ------------------------------------------------------------------ bilinear (align_corners) backward
__device__ __forceinline__ float bil_w(int dst, int src, int n_src, float scale) {
#pragma clang fp contract(off)
  const float r = scale * dst;
  const int i0 = (int)r;
  const int ip = (i0 < n_src - 1) ? 1 : 0;
  const float l1 = r - i0;
  float w = 0.f;
  if (i0 == src) w += 1.f - l1;
  if (i0 + ip == src) w += ip ? l1 : l1;  // when ip == 0 both taps hit i0: total weight (1-l1)+l1
  return w;
}

template <typename T>
__global__ void bilinear_ac_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int B, int H, int W, int C,
                                       int Ho, int Wo, float sh, float sw, long long total) {
  constexpr int EPV = Elem<T>::EPV;
  const int VC = C / EPV;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
    const int v = (int)(i % VC);
    long long pix = i / VC;
    const int w = (int)(pix % W);
    pix /= W;
    const int h = (int)(pix % H);
    const long long b = pix / H;
    float acc[EPV];
#pragma unroll
    for (int j = 0; j < EPV; ++j) acc[j] = 0.f;
    const int ho_lo = sh > 0.f ? max(0, (int)floorf((h - 1) / sh) - 1) : 0;
    const int ho_hi = sh > 0.f ? min(Ho - 1, (int)ceilf((h + 1) / sh) + 1) : Ho - 1;
    const int wo_lo = sw > 0.f ? max(0, (int)floorf((w - 1) / sw) - 1) : 0;
    const int wo_hi = sw > 0.f ? min(Wo - 1, (int)ceilf((w + 1) / sw) + 1) : Wo - 1;
    for (int ho = ho_lo; ho <= ho_hi; ++ho) {
      const float wh = bil_w(ho, h, H, sh);
      if (wh == 0.f) continue;
      for (int wo = wo_lo; wo <= wo_hi; ++wo) {
        const float ww = bil_w(wo, w, W, sw);
        if (ww == 0.f) continue;
        float g[EPV];
        Elem<T>::unpack(*reinterpret_cast<const uint4*>(dy + ((b * Ho + ho) * Wo + wo) * C + v * EPV), g);
#pragma unroll
        for (int j = 0; j < EPV; ++j) acc[j] += wh * ww * g[j];
      }
    }
    *reinterpret_cast<uint4*>(dx + i * EPV) = Elem<T>::pack(acc);
  }
}

// The same sum for magnifications up to 2.5 x (scale > 0.4: at most five source rows / columns carry weight), a channel
// vector per thread: the weights of the 7 candidate rows and 7 candidate columns are evaluated once (14 bil_w instead of
// 49 + 49 inside the loops), the first live candidate found, and the 5 x 5 taps loaded unconditionally from clamped
// addresses — a tap outside the support has weight 0 and adds +-0. Rows ascending, columns ascending, (wh * ww) * g: the
// order and the products of the kernel above.
struct BilTaps {
  int start;
  float w[5];
};
__device__ __forceinline__ BilTaps bil_taps(int src, int n_src, int n_dst, float scale) {
  const int lo = scale > 0.f ? max(0, (int)floorf((src - 1) / scale) - 1) : 0;
  float w7[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) w7[i] = (lo + i < n_dst) ? bil_w(lo + i, src, n_src, scale) : 0.f;
  const int f = w7[0] != 0.f ? 0 : (w7[1] != 0.f ? 1 : 2);
  BilTaps t;
  t.start = lo + f;
#pragma unroll
  for (int k = 0; k < 5; ++k) t.w[k] = f == 0 ? w7[k] : (f == 1 ? w7[k + 1] : w7[k + 2]);
  return t;
}
constexpr int BWD5_MAXP = 512;   // pixels per workgroup
template <typename T>
__global__ __launch_bounds__(TPB) void bilinear_ac_bwd5_kernel(const T* __restrict__ dy, T* __restrict__ dx, int B, int H, int W,
                                                               int C, int Ho, int Wo, float sh, float sw, int pix_per_block) {
  constexpr int EPV = Elem<T>::EPV;
  const int VC = C / EPV;
  const int VCB = min(VC, TPB), PL = TPB / VCB, pl = threadIdx.x / VCB;
  const unsigned npix = (unsigned)B * H * W;
  const unsigned q0 = (unsigned)xcd_remap(blockIdx.x, gridDim.x) * (unsigned)pix_per_block, q1 = min(npix, q0 + pix_per_block);
  // the taps of this run's pixels, once per workgroup (a third of the kernel's instructions when every thread derived them)
  __shared__ BilTaps s_th[BWD5_MAXP], s_tw[BWD5_MAXP];   // (static: the dynamic-LDS form of this kernel ran 25 % slower)
  for (unsigned i = threadIdx.x; i < 2 * (q1 - q0); i += TPB) {
    const unsigned k = i >> 1, pix = q0 + k;
    const unsigned t = pix / (unsigned)W, w = pix - t * W;
    const unsigned h = t % (unsigned)H;
    if (i & 1) s_tw[k] = bil_taps((int)w, W, Wo, sw);
    else s_th[k] = bil_taps((int)h, H, Ho, sh);
  }
  __syncthreads();
  for (int v = threadIdx.x % VCB; v < VC && pl < PL; v += VCB) {
    for (unsigned pix = q0 + pl; pix < q1; pix += PL) {
      const unsigned b = pix / ((unsigned)H * W);
      const BilTaps th = s_th[pix - q0], tw = s_tw[pix - q0];
      float acc[EPV];
#pragma unroll
      for (int j = 0; j < EPV; ++j) acc[j] = 0.f;
      const T* plane = dy + (long long)b * Ho * Wo * C + v * EPV;
      // all 25 vectors requested before the first is used: five dependent round trips per pixel left the kernel latency bound
      uint4 q[25];
#pragma unroll
      for (int r = 0; r < 5; ++r) {
        const T* row = plane + (long long)min(th.start + r, Ho - 1) * Wo * C;
#pragma unroll
        for (int c = 0; c < 5; ++c) q[r * 5 + c] = *reinterpret_cast<const uint4*>(row + (long long)min(tw.start + c, Wo - 1) * C);
      }
#pragma unroll
      for (int r = 0; r < 5; ++r) {
#pragma unroll
        for (int c = 0; c < 5; ++c) {
          float g[EPV];
          Elem<T>::unpack(q[r * 5 + c], g);
          const float ww = th.w[r] * tw.w[c];
#pragma unroll
          for (int j = 0; j < EPV; ++j) acc[j] += ww * g[j];
        }
      }
      *reinterpret_cast<uint4*>(dx + ((long long)pix * VC + v) * EPV) = Elem<T>::pack(acc);
    }
  }
}

// ------------------------------------------------------------------ nearest upsample backward
// db[hs,ws] = sum of dy[h,w] over the fine pixels whose nearest source is (hs,ws)
template <typename T>
__global__ void nearest_bwd_kernel(const T* __restrict__ dy, T* __restrict__ db, int B, int H, int W, int C, int Hb,
                                   int Wb, float sh, float sw, long long total) {
  constexpr int EPV = Elem<T>::EPV;
  const int VC = C / EPV;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
    const int v = (int)(i % VC);
    long long pix = i / VC;
    const int ws = (int)(pix % Wb);
    pix /= Wb;
    const int hs = (int)(pix % Hb);
    const long long b = pix / Hb;
    float acc[EPV];
#pragma unroll
    for (int j = 0; j < EPV; ++j) acc[j] = 0.f;
    const int h_lo = max(0, (int)floorf(hs / sh) - 1), h_hi = min(H - 1, (int)ceilf((hs + 1) / sh) + 1);
    const int w_lo = max(0, (int)floorf(ws / sw) - 1), w_hi = min(W - 1, (int)ceilf((ws + 1) / sw) + 1);
    for (int h = h_lo; h <= h_hi; ++h) {
      if (min((int)floorf(h * sh), Hb - 1) != hs) continue;
      for (int w = w_lo; w <= w_hi; ++w) {
        if (min((int)floorf(w * sw), Wb - 1) != ws) continue;
        float g[EPV];
        Elem<T>::unpack(*reinterpret_cast<const uint4*>(dy + ((b * H + h) * W + w) * C + v * EPV), g);
#pragma unroll
        for (int j = 0; j < EPV; ++j) acc[j] += g[j];
      }
    }
    *reinterpret_cast<uint4*>(db + i * EPV) = Elem<T>::pack(acc);
  }
}
}  // namespace

#define DISPATCH_T(dtype, CALL)                 \
  if ((dtype) == DAS_BF16) { using T = bf16_t; CALL; } \
  else if ((dtype) == DAS_F32) { using T = float; CALL; } \
  else return DAS_ERR_ARG;

static int groupnorm_backward_impl(const void* dy, const void* y, const void* x, void* dx, int dtype,
                                   const DasLevels* lv, int C, int pix_stride, int G, const float* fwd_stats,
                                   const float* gamma, const float* beta, float eps, int relu, float* gsums_ws,
                                   float* dgamma, float* dbeta, bool accumulate, bool ws_zeroed, void* stream) {
  if (!dy || !x || !dx || !fwd_stats || !gamma || !gsums_ws || !dgamma || !dbeta || !lv_valid(lv)) return DAS_ERR_ARG;
  if (C % 8 || C % G || pix_stride % 8 || C > 2048 || (relu && !y && !beta)) return DAS_ERR_ARG;
  const int epv = dtype == DAS_BF16 ? 8 : 4;
  if ((C / epv) > GN_NT) return DAS_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int nseg = lv->num_levels * lv->B;
  // (one fill when the caller laid the three accumulators out back to back: a fill is a ~4 us launch of its own)
  if (accumulate) {   // dgamma / dbeta already hold gradients (the optimizer's flat buffer): only the group sums start at zero
    if (!ws_zeroed && hipMemsetAsync(gsums_ws, 0, sizeof(float) * 2 * nseg * G, s) != hipSuccess) return DAS_ERR_LAUNCH;
  } else if (dgamma == gsums_ws + 2 * nseg * G && dbeta == dgamma + C) {
    if (hipMemsetAsync(gsums_ws, 0, sizeof(float) * (2 * (size_t)nseg * G + 2 * (size_t)C), s) != hipSuccess) return DAS_ERR_LAUNCH;
  } else {
    if (hipMemsetAsync(gsums_ws, 0, sizeof(float) * 2 * nseg * G, s) != hipSuccess) return DAS_ERR_LAUNCH;
    if (hipMemsetAsync(dgamma, 0, sizeof(float) * C, s) != hipSuccess) return DAS_ERR_LAUNCH;
    if (hipMemsetAsync(dbeta, 0, sizeof(float) * C, s) != hipSuccess) return DAS_ERR_LAUNCH;
  }
  int maxhw = 0;
  for (int l = 0; l < lv->num_levels; ++l) maxhw = std::max(maxhw, lv->H[l] * lv->W[l]);
  int chunks = (256 * 4 + lv->B - 1) / lv->B;
  int ppb = std::max(gn_ppb_min(dastune::get(dastune::GN_PPB), nseg, maxhw), (maxhw + chunks - 1) / chunks);   // (see norm.hip)
  chunks = (maxhw + ppb - 1) / ppb;
  DISPATCH_T(dtype, {
    const int vc = C / Elem<T>::EPV;
    const size_t lds = (size_t)((vc < 64 && 64 % vc == 0) ? GN_NT / 64 : GN_NT / vc) * 2 * C * sizeof(float);
    if (lds > 48 * 1024 && hipFuncSetAttribute((const void*)gn_bwd_reduce_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)lds) != hipSuccess)
      return DAS_ERR_LAUNCH;
    hipLaunchKernelGGL(gn_bwd_reduce_kernel<T>, dim3(chunks, nseg), dim3(GN_NT), lds, s, (const T*)dy,
                       (const T*)y, (const T*)x, *lv, C, pix_stride, G, ppb, fwd_stats, gamma, beta, eps, relu, gsums_ws,
                       dgamma, dbeta);
    hipLaunchKernelGGL(gn_bwd_apply_kernel<T>, dim3(chunks, nseg), dim3(GN_NT), 0, s, (const T*)dy, (const T*)y,
                       (const T*)x, (T*)dx, *lv, C, pix_stride, G, ppb, fwd_stats, gsums_ws, gamma, beta, eps, relu);
  });
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}

extern "C" int das_groupnorm_backward(const void* dy, const void* y, const void* x, void* dx, int dtype,
                                      const DasLevels* lv, int C, int pix_stride, int G, const float* fwd_stats,
                                      const float* gamma, const float* beta, float eps, int relu, float* gsums_ws,
                                      float* dgamma, float* dbeta, void* stream) {
  DAS_PROF(stream);
  return groupnorm_backward_impl(dy, y, x, dx, dtype, lv, C, pix_stride, G, fwd_stats, gamma, beta, eps, relu, gsums_ws,
                                 dgamma, dbeta, false, false, stream);
}
extern "C" int das_groupnorm_backward_acc(const void* dy, const void* y, const void* x, void* dx, int dtype,
                                          const DasLevels* lv, int C, int pix_stride, int G, const float* fwd_stats,
                                          const float* gamma, const float* beta, float eps, int relu, float* gsums_ws,
                                          float* dgamma, float* dbeta, int ws_zeroed, void* stream) {
  DAS_PROF(stream);
  return groupnorm_backward_impl(dy, y, x, dx, dtype, lv, C, pix_stride, G, fwd_stats, gamma, beta, eps, relu, gsums_ws,
                                 dgamma, dbeta, true, ws_zeroed != 0, stream);
}

extern "C" int das_maxpool3x3s2_backward(const void* x, const void* dy, void* dx, int dtype, int B, int H, int W,
                                         int C, void* stream) {
  DAS_PROF(stream);
  if (!x || !dy || !dx || C % 8) return DAS_ERR_ARG;
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  DISPATCH_T(dtype, {
    const long long total = (long long)B * ((H + 1) / 2) * ((W + 1) / 2) * (C / Elem<T>::EPV);   // 2x2 input blocks
    hipLaunchKernelGGL(maxpool_bwd_kernel<T>, dim3(grid_for(total, 65536)), dim3(TPB), 0, (hipStream_t)stream,
                       (const T*)x, (const T*)dy, (T*)dx, B, H, W, C, Ho, Wo, total);
  });
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}

extern "C" int das_upsample_bilinear_ac_backward(const void* dy, void* dx, int dtype, int B, int H, int W, int C,
                                                 int Ho, int Wo, void* stream) {
  DAS_PROF(stream);
  if (!dy || !dx || C % 8) return DAS_ERR_ARG;
  const float sh = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f;
  const float sw = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
  const long long npix = (long long)B * H * W;
  if (sh > 0.4f && sw > 0.4f && npix < (1ll << 31)) {   // at most five live taps per axis
    const int vc = C / (dtype == DAS_F32 ? 4 : 8);
    const int pl = TPB / (vc < TPB ? vc : TPB);
    long long ppb = std::min<long long>(BWD5_MAXP, (long long)pl * (npix >= (1 << 16) ? 4 : 2));   // (measured: tools/dev/upT_bench.py)
    if (dastune::get(dastune::ELEM_UPSTATS_PPB) > 0) ppb = std::min<long long>(BWD5_MAXP, dastune::get(dastune::ELEM_UPSTATS_PPB));
    DISPATCH_T(dtype, {
      hipLaunchKernelGGL(bilinear_ac_bwd5_kernel<T>, dim3((unsigned)((npix + ppb - 1) / ppb)), dim3(TPB), 0, (hipStream_t)stream,
                         (const T*)dy, (T*)dx, B, H, W, C, Ho, Wo, sh, sw, (int)ppb);
    });
    DAS_CHECK_LAUNCH();
    return DAS_OK;
  }
  DISPATCH_T(dtype, {
    const long long total = (long long)B * H * W * (C / Elem<T>::EPV);
    hipLaunchKernelGGL(bilinear_ac_bwd_kernel<T>, dim3(grid_for(total, 65536)), dim3(TPB), 0, (hipStream_t)stream,
                       (const T*)dy, (T*)dx, B, H, W, C, Ho, Wo, sh, sw, total);
  });
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}

extern "C" int das_upsample_nearest_backward(const void* dy, void* db, int dtype, int B, int H, int W, int C, int Hb,
                                             int Wb, void* stream) {
  DAS_PROF(stream);
  if (!dy || !db || C % 8) return DAS_ERR_ARG;
  const float sh = (float)Hb / (float)H, sw = (float)Wb / (float)W;
  DISPATCH_T(dtype, {
    const long long total = (long long)B * Hb * Wb * (C / Elem<T>::EPV);
    hipLaunchKernelGGL(nearest_bwd_kernel<T>, dim3(grid_for(total, 65536)), dim3(TPB), 0, (hipStream_t)stream,
                       (const T*)dy, (T*)db, B, H, W, C, Hb, Wb, sh, sw, total);
  });
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}
