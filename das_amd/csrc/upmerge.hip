// The merge step of an MSPN upsample unit in train mode (mspn_mmpose.py:381-404):
//
//     out = relu( BN1(in_skip(x)) + BN2(up_conv(upsample(up_x))) )
//
// with up_conv already exchanged with the upsampling (elementwise.hip: bilinear_ac_stats_kernel), i.e.
// raw2 = upsample(z), z = up_conv(up_x) at LOW resolution. Neither the normalised lateral BN1(raw1) nor raw2 is ever
// written: a BatchNorm apply is a per-channel affine map and bilinear weights sum to one, so
//
//     out = relu( a1 raw1 + b1 + a2 upsample(z) + b2 ),   a = gamma invstd, b = beta - mean a
//
// is one pass over raw1 (+ the quarter-size z) — against BN1 apply (read, write), upsampling (write) and BN2 apply (two
// reads, write) before. The statistics of raw2 are computed at low resolution too (upstats_lowres_kernel): sum upsample(z) =
// sum_lo w z and sum upsample(z)^2 = sum_lo z (upsample^T upsample z).
//
// Backward, with dZ = dOut * (out > 0) the gradient of BOTH pre-activation branches:
//   pass A (here)    one sweep over dOut, out, raw1 (+ z): writes dZ, reduces s1a = sum dZ, s1b = sum dZ xhat1, s2b = sum dZ xhat2
//                    (xhat2 from upsample(z) recomputed on the fly)
//   pass B           BN1's apply pass (das_bn_backward_apply: dZ, raw1 -> d raw1), as everywhere else
//   pass C           P = upsample^T(dZ) (das_upsample_bilinear_ac_backward)
//   pass D (here)    low resolution: dz = a2 (P - s2a/N w - s2b/N invstd2 (G - mean2 w)), G = upsample^T(upsample(z)),
//                    w = upsample^T(1). upsample = Uh (x) Uw is separable, so upsample^T upsample = (Uh^T Uh) (x) (Uw^T Uw) is a
//                    3 x 3 stencil with position-dependent coefficients (two small tables from the host), and w = wh (x) ww.
// d raw2 = a2 (dZ - s2a/N - xhat2 s2b/N) at high resolution is never formed: dz = upsample^T(d raw2) by linearity.
#include "common.h"
#include "prof.h"
#include "tuning.h"

namespace {
constexpr int TPB = 256;

struct PixGeom {
  long long base;   // element offset of the top-left source pixel (without the channel offset)
  int h1p, w1p;
  float h1l, w1l;
};
// torch upsample_bilinear2d, align_corners = True: the same arithmetic as bilinear_ac_stats_kernel
__device__ __forceinline__ PixGeom pix_geom(unsigned pix, int H, int W, int C, int Ho, int Wo, float sh, float sw) {
#pragma clang fp contract(off)
  PixGeom g;
  const unsigned t = pix / (unsigned)Wo, wo = pix - t * Wo;
  const unsigned b = t / (unsigned)Ho, ho = t - b * Ho;
  const float h1r = sh * (int)ho, w1r = sw * (int)wo;
  const int h1 = (int)h1r, w1 = (int)w1r;
  g.h1p = (h1 < H - 1) ? 1 : 0;
  g.w1p = (w1 < W - 1) ? 1 : 0;
  g.h1l = h1r - h1;
  g.w1l = w1r - w1;
  g.base = (((long long)b * H + h1) * W + w1) * C;
  return g;
}
template <typename T>
__device__ __forceinline__ void load_corners(const T* z, const PixGeom& g, int W, int C, int c0, uint4* r) {
  const T* base = z + g.base + c0;
  r[0] = *reinterpret_cast<const uint4*>(base);
  r[1] = *reinterpret_cast<const uint4*>(base + (long long)g.w1p * C);
  r[2] = *reinterpret_cast<const uint4*>(base + (long long)g.h1p * W * C);
  r[3] = *reinterpret_cast<const uint4*>(base + ((long long)g.h1p * W + g.w1p) * C);
}
// upsample(z) at this pixel (f32: this tensor is never stored, and its statistics are those of the unrounded values)
template <typename T>
__device__ __forceinline__ void interp(const uint4* r, const PixGeom& g, float* up) {
#pragma clang fp contract(off)
  constexpr int EPV = Elem<T>::EPV;
  float a[EPV], b[EPV], c[EPV], d[EPV];
  Elem<T>::unpack(r[0], a);
  Elem<T>::unpack(r[1], b);
  Elem<T>::unpack(r[2], c);
  Elem<T>::unpack(r[3], d);
  const float h0l = 1.f - g.h1l, w0l = 1.f - g.w1l;
#pragma unroll
  for (int j = 0; j < EPV; ++j) up[j] = h0l * (w0l * a[j] + g.w1l * b[j]) + g.h1l * (w0l * c[j] + g.w1l * d[j]);
}

struct BnPar {
  const float *mean, *invstd, *gamma, *beta;
};

// ---------------------------------------------------------------------------------------------- forward
template <typename T>
__global__ __launch_bounds__(TPB) void upmerge_fwd_kernel(const T* __restrict__ raw1, const T* __restrict__ z, T* __restrict__ out,
                                                          int B, int H, int W, int C, int Ho, int Wo, float sh, float sw,
                                                          int pix_per_block, BnPar p1, BnPar p2, unsigned char* __restrict__ bits) {
  constexpr int EPV = Elem<T>::EPV;
  const int VC = C / EPV;
  const int VCB = min(VC, TPB), PL = TPB / VCB, pl = threadIdx.x / VCB;
  const unsigned npix = (unsigned)B * Ho * Wo;
  const unsigned q0 = (unsigned)xcd_remap(blockIdx.x, gridDim.x) * (unsigned)pix_per_block, q1 = min(npix, q0 + pix_per_block);
  for (int v = threadIdx.x % VCB; v < VC && pl < PL; v += VCB) {
    const int c0 = v * EPV;
    float m1[EPV], i1[EPV], g1[EPV], b1[EPV], m2[EPV], i2[EPV], g2[EPV], b2[EPV];
#pragma unroll
    for (int j = 0; j < EPV; ++j) {
      m1[j] = p1.mean[c0 + j]; i1[j] = p1.invstd[c0 + j]; g1[j] = p1.gamma[c0 + j]; b1[j] = p1.beta[c0 + j];
      m2[j] = p2.mean[c0 + j]; i2[j] = p2.invstd[c0 + j]; g2[j] = p2.gamma[c0 + j]; b2[j] = p2.beta[c0 + j];
    }
    constexpr int U = 2;
    for (unsigned pix0 = q0 + pl; pix0 < q1; pix0 += U * PL) {
      PixGeom g[U];
      uint4 rz[U][4], rr[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const unsigned pix = min(pix0 + (unsigned)u * PL, q1 - 1);
        g[u] = pix_geom(pix, H, W, C, Ho, Wo, sh, sw);
        load_corners(z, g[u], W, C, c0, rz[u]);
        rr[u] = *reinterpret_cast<const uint4*>(raw1 + (long long)pix * C + c0);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const unsigned pix = pix0 + (unsigned)u * PL;
        if (pix >= q1) break;
        float up[EPV], x1[EPV], o[EPV];
        interp<T>(rz[u], g[u], up);
        Elem<T>::unpack(rr[u], x1);
#pragma unroll
        for (int j = 0; j < EPV; ++j)
          o[j] = fmaxf(bn_affine(up[j], m2[j], i2[j], g2[j], b2[j]) + bn_affine(x1[j], m1[j], i1[j], g1[j], b1[j]), 0.f);
        const uint4 packed = Elem<T>::pack(o);
        *reinterpret_cast<uint4*>(out + (long long)pix * C + c0) = packed;
        if (bits) bits[(long long)pix * VC + v] = (unsigned char)relu_bits<T>(packed);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------- backward, pass A
// A few workgroups per CU, each over its own run of pixels; every workgroup ends with 3C global atomics.
template <typename T>
__global__ __launch_bounds__(TPB) void upmerge_bwd_reduce_kernel(const T* __restrict__ dy, const T* __restrict__ out,
                                                                 const unsigned char* __restrict__ bits,   // (instead of out)
                                                                 const T* __restrict__ raw1, const T* __restrict__ z,
                                                                 T* __restrict__ dzm, int B, int H, int W, int C, int Ho, int Wo,
                                                                 float sh, float sw, const float* __restrict__ mean1,
                                                                 const float* __restrict__ invstd1,
                                                                 const float* __restrict__ mean2,
                                                                 const float* __restrict__ invstd2, float* __restrict__ sums) {
  constexpr int EPV = Elem<T>::EPV;
  extern __shared__ float part[];   // [PL][3C]
  const int VC = C / EPV;
  const int VCB = min(VC, TPB), PL = TPB / VCB, pl = threadIdx.x / VCB;
  const unsigned npix = (unsigned)B * Ho * Wo;
  // a contiguous run of pixels per workgroup, neighbouring runs on one XCD: the rows of z two runs share stay in one L2
  const unsigned run = (npix + gridDim.x - 1) / gridDim.x;
  const unsigned long long r0 = (unsigned long long)xcd_remap(blockIdx.x, gridDim.x) * run;
  const unsigned long long r1 = r0 + run < npix ? r0 + run : npix;
  const unsigned stride = (unsigned)PL;
  for (int v = threadIdx.x % VCB; v < VC && pl < PL; v += VCB) {
    const int c0 = v * EPV;
    float m1[EPV], i1[EPV], m2[EPV], i2[EPV], sa[EPV], sb1[EPV], sb2[EPV];
#pragma unroll
    for (int j = 0; j < EPV; ++j) {
      m1[j] = mean1[c0 + j]; i1[j] = invstd1[c0 + j]; m2[j] = mean2[c0 + j]; i2[j] = invstd2[c0 + j];
      sa[j] = 0.f; sb1[j] = 0.f; sb2[j] = 0.f;
    }
    constexpr int U = 2;
    // (64-bit loop counter: pix + U * stride may pass 2^32 on the last trip)
    for (unsigned long long pix0 = r0 + pl; pix0 < r1; pix0 += (unsigned long long)U * stride) {
      PixGeom g[U];
      uint4 rz[U][4], rg[U], ro[U], rx[U];
      bool live[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const unsigned long long pu = pix0 + (unsigned long long)u * stride;
        live[u] = pu < r1;
        const unsigned pix = live[u] ? (unsigned)pu : (unsigned)pix0;
        g[u] = pix_geom(pix, H, W, C, Ho, Wo, sh, sw);
        load_corners(z, g[u], W, C, c0, rz[u]);
        const long long o = (long long)pix * C + c0;
        rg[u] = *reinterpret_cast<const uint4*>(dy + o);
        if (bits) ro[u].x = bits[(long long)pix * VC + v];   // (expanded at use)
        else ro[u] = *reinterpret_cast<const uint4*>(out + o);
        rx[u] = *reinterpret_cast<const uint4*>(raw1 + o);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (!live[u]) break;
        const unsigned pix = (unsigned)(pix0 + (unsigned long long)u * stride);
        float up[EPV], x1[EPV], gg[EPV], oo[EPV];
        interp<T>(rz[u], g[u], up);
        Elem<T>::unpack(rx[u], x1);
        Elem<T>::unpack(rg[u], gg);
        Elem<T>::unpack(bits ? mask_vec<T>(ro[u].x) : ro[u], oo);
#pragma unroll
        for (int j = 0; j < EPV; ++j) gg[j] = oo[j] > 0.f ? gg[j] : 0.f;
        const uint4 packed = Elem<T>::pack(gg);
        *reinterpret_cast<uint4*>(dzm + (long long)pix * C + c0) = packed;
        Elem<T>::unpack(packed, gg);   // the sums see dZ as stored (what passes B and C read)
#pragma unroll
        for (int j = 0; j < EPV; ++j) {
          sa[j] += gg[j];
          sb1[j] += gg[j] * (x1[j] - m1[j]) * i1[j];
          sb2[j] += gg[j] * (up[j] - m2[j]) * i2[j];
        }
      }
    }
    lds_put<EPV>(part, 3 * C, pl, c0, sa);
    lds_put<EPV>(part, 3 * C, pl, C + c0, sb1);
    lds_put<EPV>(part, 3 * C, pl, 2 * C + c0, sb2);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * C; i += TPB) atomicAdd(sums + i, lds_fold(part, 3 * C, PL, i));
}

// ---------------------------------------------------------------------------------------------- low-resolution passes
// ah [H][3], aw [W][3]: rows of Uh^T Uh / Uw^T Uw (offsets -1, 0, +1; zero outside the plane); wh [H], ww [W]: Uh^T 1, Uw^T 1.
struct UpTables {
  const float *ah, *aw, *wh, *ww;
};
// G = (upsample^T upsample z) at one low-resolution pixel: the nine taps unconditionally, from clamped addresses (the tables
// hold 0 for a tap outside the plane)
template <typename T>
__device__ __forceinline__ void stencil_g(const T* __restrict__ z, const UpTables& tb, unsigned b, unsigned h, unsigned w, int H,
                                          int W, int C, int c0, float* G, float* zc) {
#pragma clang fp contract(off)
  constexpr int EPV = Elem<T>::EPV;
  uint4 r[9];
  float cc[9];
#pragma unroll
  for (int dh = -1; dh <= 1; ++dh) {
    const int hh = min(max((int)h + dh, 0), H - 1);
#pragma unroll
    for (int dw = -1; dw <= 1; ++dw) {
      const int wx = min(max((int)w + dw, 0), W - 1);
      r[(dh + 1) * 3 + dw + 1] = *reinterpret_cast<const uint4*>(z + (((long long)b * H + hh) * W + wx) * C + c0);
      cc[(dh + 1) * 3 + dw + 1] = tb.ah[h * 3 + dh + 1] * tb.aw[w * 3 + dw + 1];
    }
  }
#pragma unroll
  for (int j = 0; j < EPV; ++j) G[j] = 0.f;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    float f[EPV];
    Elem<T>::unpack(r[k], f);
#pragma unroll
    for (int j = 0; j < EPV; ++j) G[j] += cc[k] * f[j];
    if (k == 4) {
#pragma unroll
      for (int j = 0; j < EPV; ++j) zc[j] = f[j];
    }
  }
}

// BatchNorm 2's batch statistics without visiting the high resolution: sum upsample(z) = sum_lo w z and
// sum upsample(z)^2 = z^T (upsample^T upsample) z = sum_lo z G, per channel, into stats[slots][2C] (workgroup b -> slot b % slots).
template <typename T>
__global__ __launch_bounds__(TPB) void upstats_lowres_kernel(const T* __restrict__ z, int B, int H, int W, int C, UpTables tb,
                                                             int pix_per_block, float* __restrict__ stats, int slots) {
#pragma clang fp contract(off)
  constexpr int EPV = Elem<T>::EPV;
  extern __shared__ float part[];   // [PL][2C]
  const int VC = C / EPV;
  const int VCB = min(VC, TPB), PL = TPB / VCB, pl = threadIdx.x / VCB;
  const unsigned npix = (unsigned)B * H * W;
  const unsigned q0 = (unsigned)xcd_remap(blockIdx.x, gridDim.x) * (unsigned)pix_per_block, q1 = min(npix, q0 + pix_per_block);
  for (int v = threadIdx.x % VCB; v < VC && pl < PL; v += VCB) {
    const int c0 = v * EPV;
    float s[EPV], q[EPV];
#pragma unroll
    for (int j = 0; j < EPV; ++j) { s[j] = 0.f; q[j] = 0.f; }
    for (unsigned pix = q0 + pl; pix < q1; pix += PL) {
      const unsigned t = pix / (unsigned)W, w = pix - t * W;
      const unsigned b = t / (unsigned)H, h = t - b * H;
      float G[EPV], zc[EPV];
      stencil_g<T>(z, tb, b, h, w, H, W, C, c0, G, zc);
      const float wgt = tb.wh[h] * tb.ww[w];
#pragma unroll
      for (int j = 0; j < EPV; ++j) { s[j] += wgt * zc[j]; q[j] += zc[j] * G[j]; }
    }
    lds_put<EPV>(part, 2 * C, pl, c0, s);
    lds_put<EPV>(part, 2 * C, pl, C + c0, q);
  }
  __syncthreads();
  float* dst = stats + (size_t)(slots > 1 ? blockIdx.x % (unsigned)slots : 0) * 2 * C;
  for (int i = threadIdx.x; i < 2 * C; i += TPB) atomicAdd(dst + i, lds_fold(part, 2 * C, PL, i));
}

// backward, pass D. Workgroup 0 also adds BatchNorm 2's parameter gradients (dgamma2 += s2b, dbeta2 += s2a) when
// accumulators are given.
template <typename T>
__global__ __launch_bounds__(TPB) void upmerge_bwd_lowres_kernel(const T* __restrict__ P, const T* __restrict__ z, T* __restrict__ dz,
                                                                 int B, int H, int W, int C, UpTables tb, int pix_per_block,
                                                                 const float* __restrict__ sums,
                                                                 const float* __restrict__ gamma2,
                                                                 const float* __restrict__ mean2,
                                                                 const float* __restrict__ invstd2, float inv_n,
                                                                 float* __restrict__ dgamma2_acc, float* __restrict__ dbeta2_acc) {
#pragma clang fp contract(off)
  constexpr int EPV = Elem<T>::EPV;
  const int VC = C / EPV;
  if (blockIdx.x == 0 && dgamma2_acc && dbeta2_acc) {
    for (int c = threadIdx.x; c < C; c += TPB) {
      dgamma2_acc[c] += sums[2 * C + c];
      dbeta2_acc[c] += sums[c];
    }
  }
  const int VCB = min(VC, TPB), PL = TPB / VCB, pl = threadIdx.x / VCB;
  const unsigned npix = (unsigned)B * H * W;
  const unsigned q0 = (unsigned)xcd_remap(blockIdx.x, gridDim.x) * (unsigned)pix_per_block, q1 = min(npix, q0 + pix_per_block);
  for (int v = threadIdx.x % VCB; v < VC && pl < PL; v += VCB) {
    const int c0 = v * EPV;
    float k1[EPV], k2[EPV], k3[EPV], mu[EPV];
#pragma unroll
    for (int j = 0; j < EPV; ++j) {
      const float is = invstd2[c0 + j];
      k1[j] = gamma2[c0 + j] * is; k2[j] = sums[c0 + j] * inv_n; k3[j] = sums[2 * C + c0 + j] * inv_n * is; mu[j] = mean2[c0 + j];
    }
    for (unsigned pix = q0 + pl; pix < q1; pix += PL) {
      const unsigned t = pix / (unsigned)W, w = pix - t * W;
      const unsigned b = t / (unsigned)H, h = t - b * H;
      const uint4 rp = *reinterpret_cast<const uint4*>(P + (long long)pix * C + c0);
      float G[EPV], zc[EPV], p[EPV], o[EPV];
      stencil_g<T>(z, tb, b, h, w, H, W, C, c0, G, zc);
      const float wgt = tb.wh[h] * tb.ww[w];
      Elem<T>::unpack(rp, p);
#pragma unroll
      for (int j = 0; j < EPV; ++j) o[j] = k1[j] * (p[j] - k2[j] * wgt - k3[j] * (G[j] - mu[j] * wgt));
      *reinterpret_cast<uint4*>(dz + (long long)pix * C + c0) = Elem<T>::pack(o);
    }
  }
}
}  // namespace

#define DISPATCH_T(dtype, CALL)                 \
  if ((dtype) == DAS_BF16) { using T = bf16_t; CALL; } \
  else if ((dtype) == DAS_F32) { using T = float; CALL; } \
  else return DAS_ERR_ARG;

static bool geom_ok(int B, int H, int W, int C, int Ho, int Wo) {
  return B >= 1 && H >= 1 && W >= 1 && Ho >= 1 && Wo >= 1 && C >= 8 && C % 8 == 0 && C <= 4096 &&
         (long long)B * Ho * Wo < (1ll << 31);
}

extern "C" int das_upmerge_forward(const void* raw1, const void* z, void* out, int dtype, int B, int H, int W, int C, int Ho, int Wo,
                                   const float* mean1, const float* invstd1, const float* gamma1, const float* beta1,
                                   const float* mean2, const float* invstd2, const float* gamma2, const float* beta2,
                                   void* relu_bits_out, void* stream) {
  DAS_PROF(stream);
  if (!raw1 || !z || !out || !mean1 || !invstd1 || !gamma1 || !beta1 || !mean2 || !invstd2 || !gamma2 || !beta2 ||
      !geom_ok(B, H, W, C, Ho, Wo))
    return DAS_ERR_ARG;
  const float sh = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f;
  const float sw = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
  const long long npix = (long long)B * Ho * Wo;
  const int vc = C / (dtype == DAS_F32 ? 4 : 8);
  const int pl = TPB / (vc < TPB ? vc : TPB);
  const long long ppb = (long long)pl * 2 * (npix >= (1 << 18) ? 4 : 2);
  const long long grid = (npix + ppb - 1) / ppb;
  const BnPar p1{mean1, invstd1, gamma1, beta1}, p2{mean2, invstd2, gamma2, beta2};
  DISPATCH_T(dtype, {
    hipLaunchKernelGGL(upmerge_fwd_kernel<T>, dim3((unsigned)grid), dim3(TPB), 0, (hipStream_t)stream, (const T*)raw1,
                       (const T*)z, (T*)out, B, H, W, C, Ho, Wo, sh, sw, (int)ppb, p1, p2, (unsigned char*)relu_bits_out);
  });
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}

extern "C" int das_upmerge_backward_reduce(const void* dy, const void* out, const void* out_relu_bits, const void* raw1,
                                           const void* z, void* dzm, int dtype,
                                           int B, int H, int W, int C, int Ho, int Wo, const float* mean1,
                                           const float* invstd1, const float* mean2, const float* invstd2, float* sums,
                                           int sums_zeroed, void* stream) {
  DAS_PROF(stream);
  if (!dy || (!out && !out_relu_bits) || !raw1 || !z || !dzm || !mean1 || !invstd1 || !mean2 || !invstd2 || !sums || !geom_ok(B, H, W, C, Ho, Wo))
    return DAS_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (!sums_zeroed && hipMemsetAsync(sums, 0, sizeof(float) * 3 * C, s) != hipSuccess) return DAS_ERR_LAUNCH;
  const float sh = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f;
  const float sw = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
  const long long npix = (long long)B * Ho * Wo;
  const int vc = C / (dtype == DAS_F32 ? 4 : 8);
  const int pl = TPB / (vc < TPB ? vc : TPB);
  // (a few workgroups per CU: the pass also writes, and its 3C closing atomics per workgroup are cheap next to that)
  const long long cap = dastune::get(dastune::BN_UPMERGE_BLOCKS);
  const int blocks = (int)std::min<long long>(cap, std::max<long long>(1, npix / (2 * pl)));
  const size_t lds = (size_t)pl * 3 * C * sizeof(float);
  DISPATCH_T(dtype, {
    if (lds > 48 * 1024 && hipFuncSetAttribute((const void*)upmerge_bwd_reduce_kernel<T>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return DAS_ERR_LAUNCH;
    hipLaunchKernelGGL(upmerge_bwd_reduce_kernel<T>, dim3(blocks), dim3(TPB), lds, s, (const T*)dy, (const T*)out,
                       out ? nullptr : (const unsigned char*)out_relu_bits, (const T*)raw1, (const T*)z, (T*)dzm, B, H, W, C, Ho, Wo, sh, sw, mean1, invstd1, mean2, invstd2, sums);
  });
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}

extern "C" int das_upmerge_backward_lowres(const void* P, const void* z, void* dz, int dtype, int B, int H, int W, int C,
                                           const float* ah, const float* aw, const float* wh, const float* ww,
                                           const float* sums, const float* gamma2, const float* mean2, const float* invstd2,
                                           long long stat_rows, float* dgamma2_acc, float* dbeta2_acc, void* stream) {
  DAS_PROF(stream);
  if (!P || !z || !dz || !ah || !aw || !wh || !ww || !sums || !gamma2 || !mean2 || !invstd2 || stat_rows < 1 || B < 1 || H < 1 ||
      W < 1 || C % 8 || C < 8 || (dgamma2_acc == nullptr) != (dbeta2_acc == nullptr))
    return DAS_ERR_ARG;
  const long long npix = (long long)B * H * W;
  if (npix >= (1ll << 31)) return DAS_ERR_ARG;
  const int vc = C / (dtype == DAS_F32 ? 4 : 8);
  const int pl = TPB / (vc < TPB ? vc : TPB);
  const long long ppb = (long long)pl * (npix >= (1 << 17) ? 4 : 2);
  const long long grid = (npix + ppb - 1) / ppb;
  const UpTables tb{ah, aw, wh, ww};
  DISPATCH_T(dtype, {
    hipLaunchKernelGGL(upmerge_bwd_lowres_kernel<T>, dim3((unsigned)grid), dim3(TPB), 0, (hipStream_t)stream, (const T*)P,
                       (const T*)z, (T*)dz, B, H, W, C, tb, (int)ppb, sums, gamma2, mean2, invstd2, 1.f / (float)stat_rows,
                       dgamma2_acc, dbeta2_acc);
  });
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}

extern "C" int das_upsample_stats_lowres(const void* z, int dtype, int B, int H, int W, int C, const float* ah, const float* aw,
                                         const float* wh, const float* ww, float* stats, int stats_slots, void* stream) {
  DAS_PROF(stream);
  if (!z || !ah || !aw || !wh || !ww || !stats || B < 1 || H < 1 || W < 1 || C % 8 || C < 8 || C > 4096 || stats_slots < 1 ||
      stats_slots > 64)
    return DAS_ERR_ARG;
  const long long npix = (long long)B * H * W;
  if (npix >= (1ll << 31)) return DAS_ERR_ARG;
  const int vc = C / (dtype == DAS_F32 ? 4 : 8);
  const int pl = TPB / (vc < TPB ? vc : TPB);
  const long long ppb = (long long)pl * (npix >= (1 << 17) ? 8 : 4);
  const long long grid = (npix + ppb - 1) / ppb;
  const size_t lds = (size_t)pl * 2 * C * sizeof(float);
  const UpTables tb{ah, aw, wh, ww};
  DISPATCH_T(dtype, {
    hipLaunchKernelGGL(upstats_lowres_kernel<T>, dim3((unsigned)grid), dim3(TPB), lds, (hipStream_t)stream, (const T*)z, B, H, W,
                       C, tb, (int)ppb, stats, stats_slots);
  });
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}
