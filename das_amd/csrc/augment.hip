// Image half of the pose data pipeline (mmdet3d/datasets/pipelines/transforms_3d.py and the
// mmdet / mmcv / OpenCV ops behind it) as a GPU-side augmentation stage: f32 HWC images (BGR, as mmcv loads them) stay
// on the device from the decoded frame to the network input.
//   resize       mmcv.imrescale / imresize -> cv2.resize(INTER_LINEAR) on float images (half-pixel centres, border clamp,
//                horizontal then vertical interpolation in f32)
//   flip         mmcv.imflip(horizontal)
//   photometric  mmdet PhotoMetricDistortion: brightness, contrast (before or after), BGR->HSV, saturation, hue,
//                HSV->BGR (OpenCV's float formulas), channel permutation — one pass
//   warp_affine  cv2.warpAffine(INTER_LINEAR, BORDER_CONSTANT) as GlobalRotScaleTransPose calls it
//                (transforms_3d.py:973-986): OpenCV's fixed-point coordinate walk (10 fractional bits, rounded to 1/32
//                pixel) and its float bilinear table
//   normalize_pad  mmcv.imnormalize (BGR->RGB swap, (x - mean) / std) + Pad(size_divisor) + HWC->CHW (formating.py:383-442)
// All HBM-bound elementwise / gather passes: one thread per output pixel (three channels).
#include "common.h"
#include "prof.h"

namespace {
constexpr int TPB = 256;
inline unsigned blocks_for(long long n) { return (unsigned)((n + TPB - 1) / TPB); }

__global__ void img_resize_kernel(const float* __restrict__ src, float* __restrict__ dst, int Hs, int Ws, int Hd, int Wd,
                                  int C, double scale_x, double scale_y) {
#pragma clang fp contract(off)
  const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
  if (i >= (long long)Hd * Wd) return;
  const int dy = (int)(i / Wd), dx = (int)(i - (long long)dy * Wd);
  float fx = (float)((dx + 0.5) * scale_x - 0.5);
  int sx = (int)floorf(fx);
  fx -= (float)sx;
  if (sx < 0) { fx = 0.f; sx = 0; }
  if (sx >= Ws - 1) { fx = 0.f; sx = Ws - 1; }
  float fy = (float)((dy + 0.5) * scale_y - 0.5);
  int sy = (int)floorf(fy);
  fy -= (float)sy;
  if (sy < 0) { fy = 0.f; sy = 0; }
  if (sy >= Hs - 1) { fy = 0.f; sy = Hs - 1; }
  const int sx1 = sx + 1 < Ws ? sx + 1 : sx, sy1 = sy + 1 < Hs ? sy + 1 : sy;
  const float a0 = 1.f - fx, a1 = fx, b0 = 1.f - fy, b1 = fy;
  for (int c = 0; c < C; ++c) {
    const float r0 = src[((long long)sy * Ws + sx) * C + c] * a0 + src[((long long)sy * Ws + sx1) * C + c] * a1;
    const float r1 = src[((long long)sy1 * Ws + sx) * C + c] * a0 + src[((long long)sy1 * Ws + sx1) * C + c] * a1;
    dst[i * C + c] = r0 * b0 + r1 * b1;
  }
}

// cv2.resize(INTER_LINEAR) on 8-bit images (the test pipeline resizes the decoded uint8 frame): OpenCV's fixed-point path —
// 11-bit coefficients (saturate_cast<short>(f * 2048)), horizontal pass in int32, vertical pass
// ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2.
__global__ void img_resize_u8_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, int Hs, int Ws,
                                     int Hd, int Wd, int C, double scale_x, double scale_y) {
#pragma clang fp contract(off)
  const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
  if (i >= (long long)Hd * Wd) return;
  const int dy = (int)(i / Wd), dx = (int)(i - (long long)dy * Wd);
  float fx = (float)((dx + 0.5) * scale_x - 0.5);
  int sx = (int)floorf(fx);
  fx -= (float)sx;
  if (sx < 0) { fx = 0.f; sx = 0; }
  if (sx >= Ws - 1) { fx = 0.f; sx = Ws - 1; }
  float fy = (float)((dy + 0.5) * scale_y - 0.5);
  int sy = (int)floorf(fy);
  fy -= (float)sy;
  if (sy < 0) { fy = 0.f; sy = 0; }
  if (sy >= Hs - 1) { fy = 0.f; sy = Hs - 1; }
  const int sx1 = sx + 1 < Ws ? sx + 1 : sx, sy1 = sy + 1 < Hs ? sy + 1 : sy;
  auto coef = [](float v) { const int r = (int)rintf(v * 2048.f); return r > 32767 ? 32767 : r < -32768 ? -32768 : r; };
  const int a0 = coef(1.f - fx), a1 = coef(fx), b0 = coef(1.f - fy), b1 = coef(fy);
  for (int c = 0; c < C; ++c) {
    const int r0 = (int)src[((long long)sy * Ws + sx) * C + c] * a0 + (int)src[((long long)sy * Ws + sx1) * C + c] * a1;
    const int r1 = (int)src[((long long)sy1 * Ws + sx) * C + c] * a0 + (int)src[((long long)sy1 * Ws + sx1) * C + c] * a1;
    const int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
    dst[i * C + c] = (unsigned char)(v < 0 ? 0 : v > 255 ? 255 : v);
  }
}

__global__ void img_flip_kernel(const float* __restrict__ src, float* __restrict__ dst, int H, int W, int C) {
  const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
  if (i >= (long long)H * W) return;
  const int y = (int)(i / W), x = (int)(i - (long long)y * W);
  for (int c = 0; c < C; ++c) dst[i * C + c] = src[((long long)y * W + (W - 1 - x)) * C + c];
}

__global__ void img_photometric_kernel(float* __restrict__ img, long long npix, DasPhotometric p) {
#pragma clang fp contract(off)
  const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
  if (i >= npix) return;
  float b = img[i * 3], g = img[i * 3 + 1], r = img[i * 3 + 2];
  if (p.use_brightness) { b += p.brightness; g += p.brightness; r += p.brightness; }
  if (p.contrast_first && p.use_contrast) { b *= p.contrast; g *= p.contrast; r *= p.contrast; }
  // BGR -> HSV (OpenCV float: h in [0, 360), s = diff / (|v| + eps), v = max)
  float v = fmaxf(r, fmaxf(g, b));
  const float vmin = fminf(r, fminf(g, b));
  float diff = v - vmin;
  float s = diff / (fabsf(v) + 1.1920929e-07f);
  diff = 60.f / (diff + 1.1920929e-07f);
  float h;
  if (v == r) h = (g - b) * diff;
  else if (v == g) h = (b - r) * diff + 120.f;
  else h = (r - g) * diff + 240.f;
  if (h < 0.f) h += 360.f;
  if (p.use_saturation) s *= p.saturation;
  if (p.use_hue) {
    h += p.hue;
    if (h > 360.f) h -= 360.f;
    if (h < 0.f) h += 360.f;
  }
  // HSV -> BGR
  if (s == 0.f) {
    b = g = r = v;
  } else {
    float hh = h * (6.f / 360.f);
    if (hh < 0.f) { do hh += 6.f; while (hh < 0.f); }
    else if (hh >= 6.f) { do hh -= 6.f; while (hh >= 6.f); }
    int sector = (int)floorf(hh);
    hh -= (float)sector;
    if ((unsigned)sector >= 6u) { sector = 0; hh = 0.f; }
    const float tab[4] = {v, v * (1.f - s), v * (1.f - s * hh), v * (1.f - s * (1.f - hh))};
    const int sd[6][3] = {{1, 3, 0}, {1, 0, 2}, {3, 0, 1}, {0, 2, 1}, {0, 1, 3}, {2, 1, 0}};
    b = tab[sd[sector][0]]; g = tab[sd[sector][1]]; r = tab[sd[sector][2]];
  }
  if (!p.contrast_first && p.use_contrast) { b *= p.contrast; g *= p.contrast; r *= p.contrast; }
  const float out[3] = {b, g, r};
  img[i * 3] = out[p.perm[0]]; img[i * 3 + 1] = out[p.perm[1]]; img[i * 3 + 2] = out[p.perm[2]];
}

// round half to even, as cvRound / saturate_cast<int>(double) do (lrint in the default rounding mode)
__device__ __forceinline__ int cv_round(double v) { return (int)rint(v); }

__global__ void img_warp_affine_kernel(const float* __restrict__ src, float* __restrict__ dst, int Hs, int Ws, int Hd,
                                       int Wd, DasAffine a) {
#pragma clang fp contract(off)
  constexpr int AB_BITS = 10, AB_SCALE = 1 << AB_BITS, INTER_BITS = 5, INTER_TAB = 1 << INTER_BITS;
  constexpr int ROUND_DELTA = AB_SCALE / INTER_TAB / 2;
  const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
  if (i >= (long long)Hd * Wd) return;
  const int y = (int)(i / Wd), x = (int)(i - (long long)y * Wd);
  const double* M = a.inv;   // inverse map dst -> src (the host inverts as OpenCV does)
  const int adelta = cv_round(M[0] * x * AB_SCALE), bdelta = cv_round(M[3] * x * AB_SCALE);
  const int X0 = cv_round((M[1] * y + M[2]) * AB_SCALE) + ROUND_DELTA;
  const int Y0 = cv_round((M[4] * y + M[5]) * AB_SCALE) + ROUND_DELTA;
  const int X = (X0 + adelta) >> (AB_BITS - INTER_BITS), Y = (Y0 + bdelta) >> (AB_BITS - INTER_BITS);
  int sx = X >> INTER_BITS, sy = Y >> INTER_BITS;
  sx = sx < -32768 ? -32768 : sx > 32767 ? 32767 : sx;   // saturate_cast<short>
  sy = sy < -32768 ? -32768 : sy > 32767 ? 32767 : sy;
  const float fx = (float)(X & (INTER_TAB - 1)) * (1.f / INTER_TAB), fy = (float)(Y & (INTER_TAB - 1)) * (1.f / INTER_TAB);
  const float w[4] = {(1.f - fy) * (1.f - fx), (1.f - fy) * fx, fy * (1.f - fx), fy * fx};
  for (int c = 0; c < 3; ++c) {
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int yy = sy + (k >> 1), xx = sx + (k & 1);
      v[k] = ((unsigned)yy < (unsigned)Hs && (unsigned)xx < (unsigned)Ws) ? src[((long long)yy * Ws + xx) * 3 + c] : a.border[c];
    }
    const bool all_out = sx >= Ws || sx + 1 < 0 || sy >= Hs || sy + 1 < 0;
    dst[i * 3 + c] = all_out ? a.border[c] : v[0] * w[0] + v[1] * w[1] + v[2] * w[2] + v[3] * w[3];
  }
}

__global__ void img_normalize_pad_kernel(const float* __restrict__ src, float* __restrict__ dst, int H, int W, int Hp,
                                         int Wp, DasNormalize n) {
#pragma clang fp contract(off)
  const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
  if (i >= (long long)Hp * Wp) return;
  const int y = (int)(i / Wp), x = (int)(i - (long long)y * Wp);
  const bool in = y < H && x < W;
  for (int c = 0; c < 3; ++c) {
    float o = 0.f;   // Pad(pad_val = 0) after Normalize
    if (in) {
      const float v = src[((long long)y * W + x) * 3 + (n.to_rgb ? 2 - c : c)];
      // cv2.subtract / cv2.multiply with a Scalar work in f64 when the scalar is not integer-valued, in f32 otherwise
      // (cv::arithm_op, actualScalarDepth), and round to f32 after each call
      const float d = n.mean_f64 ? (float)((double)v - n.mean[c]) : v - (float)n.mean[c];
      o = n.std_f64 ? (float)((double)d * n.stdinv[c]) : d * (float)n.stdinv[c];
    }
    dst[(long long)c * Hp * Wp + i] = o;
  }
}
}  // namespace

extern "C" int das_img_resize_bilinear(const float* src, float* dst, int Hs, int Ws, int Hd, int Wd, int C, void* stream) {
  DAS_PROF(stream);
  if (!src || !dst || Hs < 1 || Ws < 1 || Hd < 1 || Wd < 1 || C < 1) return DAS_ERR_ARG;
  const double sx = 1.0 / ((double)Wd / Ws), sy = 1.0 / ((double)Hd / Hs);   // (as cv2.resize forms them)
  hipLaunchKernelGGL(img_resize_kernel, dim3(blocks_for((long long)Hd * Wd)), dim3(TPB), 0, (hipStream_t)stream, src, dst,
                     Hs, Ws, Hd, Wd, C, sx, sy);
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}

extern "C" int das_img_resize_bilinear_u8(const unsigned char* src, unsigned char* dst, int Hs, int Ws, int Hd, int Wd, int C,
                                          void* stream) {
  DAS_PROF(stream);
  if (!src || !dst || Hs < 1 || Ws < 1 || Hd < 1 || Wd < 1 || C < 1) return DAS_ERR_ARG;
  const double sx = 1.0 / ((double)Wd / Ws), sy = 1.0 / ((double)Hd / Hs);
  hipLaunchKernelGGL(img_resize_u8_kernel, dim3(blocks_for((long long)Hd * Wd)), dim3(TPB), 0, (hipStream_t)stream, src,
                     dst, Hs, Ws, Hd, Wd, C, sx, sy);
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}

extern "C" int das_img_flip_horizontal(const float* src, float* dst, int H, int W, int C, void* stream) {
  DAS_PROF(stream);
  if (!src || !dst || src == dst || H < 1 || W < 1 || C < 1) return DAS_ERR_ARG;
  hipLaunchKernelGGL(img_flip_kernel, dim3(blocks_for((long long)H * W)), dim3(TPB), 0, (hipStream_t)stream, src, dst, H, W, C);
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}

extern "C" int das_img_photometric(float* img, int H, int W, const DasPhotometric* p, void* stream) {
  DAS_PROF(stream);
  if (!img || !p || H < 1 || W < 1) return DAS_ERR_ARG;
  int seen = 0;
  for (int c = 0; c < 3; ++c) {
    if (p->perm[c] < 0 || p->perm[c] > 2) return DAS_ERR_ARG;
    seen |= 1 << p->perm[c];
  }
  if (seen != 7) return DAS_ERR_ARG;
  hipLaunchKernelGGL(img_photometric_kernel, dim3(blocks_for((long long)H * W)), dim3(TPB), 0, (hipStream_t)stream, img,
                     (long long)H * W, *p);
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}

extern "C" int das_img_warp_affine(const float* src, float* dst, int Hs, int Ws, int Hd, int Wd, const double* M,
                                   const float* border, void* stream) {
  DAS_PROF(stream);
  if (!src || !dst || src == dst || !M || !border || Hs < 1 || Ws < 1 || Hd < 1 || Wd < 1) return DAS_ERR_ARG;
  DasAffine a;
  // invert the 2x3 map exactly as cv::warpAffine does (double arithmetic, this order of operations)
  double m[6] = {M[0], M[1], M[2], M[3], M[4], M[5]};
  double D = m[0] * m[4] - m[1] * m[3];
  D = D != 0 ? 1. / D : 0;
  const double A11 = m[4] * D, A22 = m[0] * D;
  m[0] = A11; m[1] *= -D;
  m[3] *= -D; m[4] = A22;
  const double b1 = -m[0] * m[2] - m[1] * m[5];
  const double b2 = -m[3] * m[2] - m[4] * m[5];
  m[2] = b1; m[5] = b2;
  for (int k = 0; k < 6; ++k) a.inv[k] = m[k];
  for (int c = 0; c < 3; ++c) a.border[c] = border[c];
  hipLaunchKernelGGL(img_warp_affine_kernel, dim3(blocks_for((long long)Hd * Wd)), dim3(TPB), 0, (hipStream_t)stream, src,
                     dst, Hs, Ws, Hd, Wd, a);
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}

extern "C" int das_img_normalize_pad_chw(const float* src, float* dst, int H, int W, int Hp, int Wp, const double* mean,
                                         const double* std, int to_rgb, void* stream) {
  DAS_PROF(stream);
  if (!src || !dst || !mean || !std || H < 1 || W < 1 || Hp < H || Wp < W) return DAS_ERR_ARG;
  DasNormalize n;
  n.mean_f64 = n.std_f64 = 0;
  for (int c = 0; c < 3; ++c) {
    if (std[c] == 0.) return DAS_ERR_ARG;
    n.mean[c] = mean[c];
    n.stdinv[c] = 1.0 / std[c];   // (mmcv.imnormalize: stdinv = 1 / np.float64(std))
    if (n.mean[c] != rint(n.mean[c])) n.mean_f64 = 1;
    if (n.stdinv[c] != rint(n.stdinv[c])) n.std_f64 = 1;
  }
  n.to_rgb = to_rgb;
  hipLaunchKernelGGL(img_normalize_pad_kernel, dim3(blocks_for((long long)Hp * Wp)), dim3(TPB), 0, (hipStream_t)stream,
                     src, dst, H, W, Hp, Wp, n);
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}
