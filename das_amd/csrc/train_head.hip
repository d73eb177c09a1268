// Backward of the DASHead-specific ops: DCNv2 deformable im2col (col2im + offset/mask gradients),
// the fused recursive-update offset re-sampling, the sigmoid blend and the head assemble.
// Scatter-type gradients accumulate in f32 with atomics (as mmcv's col2im does); the caller
// hands zeroed f32 buffers.
#include "common.h"

namespace {
constexpr int TPB = 256;
inline int grid_for(long long n, int cap = 16384) {
  long long b = (n + TPB - 1) / TPB;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

// ------------------------------------------------------------------ DCNv2 col2im
// One thread = one (pixel, tap, channel vector): dx += mask*w_corner*dcol (f32 atomics),
// d_om[dy,dx,mask logit] += sum over the vector's channels.
// One WAVE = one (pixel, tap); lane = channel (64 at a time). The four corner scatters of a wave are
// 256-byte contiguous f32 atomic bursts (two cache lines each) instead of 32-byte-strided ones, and the
// offset / mask gradients are a wave reduction written once, without atomics.
template <typename T>
__global__ void deform_col2im_kernel(const T* __restrict__ x, const float* __restrict__ om, const T* __restrict__ dcol,
                                     float* __restrict__ dx, float* __restrict__ dom, DasLevels lv, int C, int xps,
                                     int omps, int domps, long long npairs) {
#pragma clang fp contract(off)
  const int lane = threadIdx.x & 63;
  const long long wave0 = ((long long)blockIdx.x * TPB + threadIdx.x) >> 6;
  const long long nwaves = ((long long)gridDim.x * TPB) >> 6;
  for (long long r = wave0; r < npairs; r += nwaves) {
    const int k = (int)(r % 9);
    const long long m = r / 9;
    const LvGeom g = lv_geom(lv, m);
    const int H = g.H, W = g.W;
    const float* o = om + m * omps;
    const float ody = o[2 * k], odx = o[2 * k + 1];
    const float mask = 1.f / (1.f + expf(-o[18 + k]));
    const float py = (float)(g.h - 1 + k / 3) + ody;
    const float px = (float)(g.w - 1 + k % 3) + odx;
    float* d = dom + m * domps;
    if (!(py > -1.f && px > -1.f && py < (float)H && px < (float)W)) continue;  // wave-uniform
    const float fy = floorf(py), fx = floorf(px);
    const int y0 = (int)fy, x0 = (int)fx;
    const float ly = py - fy, lx = px - fx, hy = 1.f - ly, hx = 1.f - lx;
    const float wts[4] = {hy * hx, hy * lx, ly * hx, ly * lx};
    const float wy[4] = {-hx, -lx, hx, lx};   // d(weight)/d(py)
    const float wx[4] = {-hy, hy, -ly, ly};   // d(weight)/d(px)
    float val = 0.f, gpy = 0.f, gpx = 0.f;
    for (int c0 = 0; c0 < C; c0 += 64) {
      const int ch = c0 + lane;
      if (ch >= C) break;
      const float gc = Elem<T>::load(dcol + (m * 9 + k) * C + ch);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int yy = y0 + (c >> 1), xx = x0 + (c & 1);
        if (yy < 0 || yy > H - 1 || xx < 0 || xx > W - 1) continue;  // wave-uniform
        const long long pix = g.plane0 + (long long)yy * W + xx;
        const float f = Elem<T>::load(x + pix * xps + ch);
        const float dot = gc * f;
        // a corner with zero bilinear weight adds exactly 0.0: skip the atomic (wave-uniform). With samples on
        // the integer grid (zero-initialised offset convs) three of the four corners are such no-ops.
        if (wts[c] != 0.f) atomicAdd(dx + pix * C + ch, gc * mask * wts[c]);
        val += wts[c] * dot;
        gpy += wy[c] * dot;
        gpx += wx[c] * dot;
      }
    }
    val = wave_sum(val);
    gpy = wave_sum(gpy);
    gpx = wave_sum(gpx);
    if (lane == 0) {
      d[2 * k] = gpy * mask;
      d[2 * k + 1] = gpx * mask;
      d[18 + k] = val * mask * (1.f - mask);
    }
  }
}

// ------------------------------------------------------------------ grid-sample helpers (f32 NHWC maps)
struct Bil {
  int x0, y0;
  float w[4], gx[4], gy[4];  // corner weights and their derivatives wrt pixel coordinates
  bool ok[4];
};
__device__ __forceinline__ Bil bil_setup(int H, int W, float locx, float locy) {
#pragma clang fp contract(off)
  Bil b;
  const float gx = 2.f * locx - 1.f, gy = 2.f * locy - 1.f;
  const float ix = ((gx + 1.f) * W - 1.f) / 2.f, iy = ((gy + 1.f) * H - 1.f) / 2.f;
  const float fx = floorf(ix), fy = floorf(iy);
  b.x0 = (int)fx; b.y0 = (int)fy;
  const float tx = ix - fx, ty = iy - fy;
  b.w[0] = (1.f - tx) * (1.f - ty); b.w[1] = tx * (1.f - ty); b.w[2] = (1.f - tx) * ty; b.w[3] = tx * ty;
  b.gx[0] = -(1.f - ty); b.gx[1] = (1.f - ty); b.gx[2] = -ty; b.gx[3] = ty;
  b.gy[0] = -(1.f - tx); b.gy[1] = -tx; b.gy[2] = (1.f - tx); b.gy[3] = tx;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int xx = b.x0 + (c & 1), yy = b.y0 + (c >> 1);
    b.ok[c] = xx >= 0 && xx < W && yy >= 0 && yy < H;
  }
  return b;
}
template <int NCH>
__device__ __forceinline__ void bil_sample(const Bil& b, const float* __restrict__ img, int W, int ps, int c0,
                                           float* out) {
#pragma unroll
  for (int j = 0; j < NCH; ++j) out[j] = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    if (!b.ok[c]) continue;
    const float* p = img + ((long long)(b.y0 + (c >> 1)) * W + b.x0 + (c & 1)) * ps + c0;
#pragma unroll
    for (int j = 0; j < NCH; ++j) out[j] += p[j] * b.w[c];
  }
}
// scatter g[NCH] into dimg at the 4 corners; returns d/d(ix), d/d(iy) of sum_j g[j]*sample_j
template <int NCH>
__device__ __forceinline__ void bil_scatter(const Bil& b, const float* __restrict__ img, float* __restrict__ dimg,
                                            int W, int ps, int dps, int c0, const float* g, float& dix, float& diy) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    if (!b.ok[c]) continue;
    const long long pix = (long long)(b.y0 + (c >> 1)) * W + b.x0 + (c & 1);
    const float* p = img + pix * ps + c0;
    float* d = dimg + pix * dps + c0;
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      dot += g[j] * p[j];
      if (g[j] != 0.f) atomicAdd(d + j, g[j] * b.w[c]);
    }
    dix += b.gx[c] * dot;
    diy += b.gy[c] * dot;
  }
}

__global__ void offset_sample_bwd_kernel(const float* __restrict__ uvd, const float* __restrict__ so,
                                         const float* __restrict__ conf, const float* __restrict__ gout,
                                         float* __restrict__ duvd, float* __restrict__ dso, float* __restrict__ dconf,
                                         DasLevels lv, int J, int uvd_ps, int so_ps, int conf_ps, int gout_ps,
                                         long long total) {
#pragma clang fp contract(off)
  constexpr int HEADS = 4, S = 2 * HEADS;
  const int J3 = 3 * J, J8 = 8 * J;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
    const int j = (int)(i % J);
    const long long pix = i / J;
    const LvGeom gm = lv_geom(lv, pix);
    const int H = gm.H, W = gm.W, px = gm.w, py = gm.h;
    const float* uvd_b = uvd + gm.plane0 * (long long)uvd_ps;
    const float* so_b = so + gm.plane0 * (long long)so_ps;
    const float* conf_b = conf + gm.plane0 * (long long)conf_ps;
    float* duvd_b = duvd + gm.plane0 * (long long)J3;
    float* dso_b = dso + gm.plane0 * (long long)J8;
    float* dconf_b = dconf + gm.plane0 * (long long)J3;
    const float fw = (float)W, fh = (float)H;
    const float cx = (float)px + 0.5f, cy = (float)py + 0.5f;
    const long long own = (long long)py * W + px;
    const float* u = uvd_b + own * uvd_ps + j * 3;
    const float offx = u[0], offy = u[1];
    // ---- forward recompute
    const Bil bt = bil_setup(H, W, (cx + offx) / fw, (cy + offy) / fh);
    float tmp[2 * HEADS];
    bil_sample<2 * HEADS>(bt, so_b, W, so_ps, j * 2 * HEADS, tmp);
    const float* ownso = so_b + own * so_ps + j * 2 * HEADS;
    float sx[S], sy[S];
#pragma unroll
    for (int h = 0; h < HEADS; ++h) {
      sx[h] = tmp[2 * h] + offx; sy[h] = tmp[2 * h + 1] + offy;
      sx[HEADS + h] = ownso[2 * h]; sy[HEADS + h] = ownso[2 * h + 1];
    }
    float val[S][3], cf[S][3];
    Bil bs[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
      bs[s] = bil_setup(H, W, (cx + sx[s]) / fw, (cy + sy[s]) / fh);
      bil_sample<3>(bs[s], uvd_b, W, uvd_ps, j * 3, val[s]);
      bil_sample<3>(bs[s], conf_b, W, conf_ps, j * 3, cf[s]);
      val[s][0] += sx[s];
      val[s][1] += sy[s];
    }
    const float* g = gout + pix * gout_ps + j * 3;
    float dval[S][3], dcf[S][3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      float mx = cf[0][d];
#pragma unroll
      for (int s = 1; s < S; ++s) mx = fmaxf(mx, cf[s][d]);
      float e[S], den = 0.f, out = 0.f;
#pragma unroll
      for (int s = 0; s < S; ++s) { e[s] = expf(cf[s][d] - mx); den += e[s]; }
#pragma unroll
      for (int s = 0; s < S; ++s) { e[s] = e[s] / den; out += val[s][d] * e[s]; }
#pragma unroll
      for (int s = 0; s < S; ++s) { dval[s][d] = g[d] * e[s]; dcf[s][d] = e[s] * g[d] * (val[s][d] - out); }
    }
    // ---- backward through the 8 sites
    float du0 = 0.f, du1 = 0.f;
    float dtmp[2 * HEADS];
#pragma unroll
    for (int s = 0; s < S; ++s) {
      float dix = 0.f, diy = 0.f;
      bil_scatter<3>(bs[s], uvd_b, duvd_b, W, uvd_ps, J3, j * 3, dval[s], dix, diy);
      bil_scatter<3>(bs[s], conf_b, dconf_b, W, conf_ps, J3, j * 3, dcf[s], dix, diy);
      const float dsx = dval[s][0] + dix, dsy = dval[s][1] + diy;
      if (s < HEADS) {
        dtmp[2 * s] = dsx; dtmp[2 * s + 1] = dsy;
        du0 += dsx; du1 += dsy;
      } else {
        atomicAdd(dso_b + own * J8 + j * 2 * HEADS + 2 * (s - HEADS), dsx);
        atomicAdd(dso_b + own * J8 + j * 2 * HEADS + 2 * (s - HEADS) + 1, dsy);
      }
    }
    float dix = 0.f, diy = 0.f;
    bil_scatter<2 * HEADS>(bt, so_b, dso_b, W, so_ps, J8, j * 2 * HEADS, dtmp, dix, diy);
    du0 += dix; du1 += diy;
    atomicAdd(duvd_b + own * J3 + j * 3, du0);
    atomicAdd(duvd_b + own * J3 + j * 3 + 1, du1);
  }
}

__global__ void sigmoid_blend_bwd_kernel(const float* __restrict__ off, const float* __restrict__ w,
                                         const float* __restrict__ nxt, const float* __restrict__ gout,
                                         float* __restrict__ doff, float* __restrict__ dw, float* __restrict__ dnxt,
                                         long long npix, int C, int off_ps, int w_ps, int nxt_ps) {
#pragma clang fp contract(off)
  const long long total = npix * C;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
    const int c = (int)(i % C);
    const long long p = i / C;
    const float g = 1.f / (1.f + expf(-w[p * w_ps + c]));
    const float go = gout[i], a = off[p * off_ps + c], b = nxt[p * nxt_ps + c];
    doff[i] = (1.f - g) * go;
    dnxt[i] = g * go;
    dw[i] = go * (b - a) * g * (1.f - g);
  }
}

__device__ __forceinline__ int level_of(const DasLevels& lv, long long m) {
  long long start = 0;
  int l = 0;
  for (; l + 1 < lv.num_levels; ++l) {
    const long long n = (long long)lv.B * lv.H[l] * lv.W[l];
    if (m < start + n) break;
    start += n;
  }
  return l;
}

// d_raw (rows, raw_ps) f32 (only the off/depth/uvd/sigma slices are written; the caller zero-fills),
// dscale f32[5][4] accumulated with atomics after a block reduction.
__global__ void head_assemble_bwd_kernel(const float* __restrict__ raw, const float* __restrict__ dpose,
                                         const float* __restrict__ duvd, float* __restrict__ draw,
                                         float* __restrict__ dscale, long long npix, DasLevels lv, DasHeadDesc d) {
  __shared__ float sred[20];
  if (threadIdx.x < 20) sred[threadIdx.x] = 0.f;
  __syncthreads();
  const int J3 = 3 * d.J, D = 3 + 6 * d.J;
  const long long total = npix * D;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
    const int c = (int)(i % D);
    const long long p = i / D;
    const int l = level_of(lv, p);
    const float* sc = d.scale[l];
    const float* r = raw + p * d.raw_ps;
    float* dr = draw + p * d.raw_ps;
    float gsc = 0.f;
    int which = -1;
    if (c < 2) {
      dr[d.off_c + c] = dpose[i] * sc[0];
      gsc = dpose[i] * r[d.off_c + c]; which = 0;
    } else if (c == 2) {
      dr[d.depth_c] = dpose[i] * sc[1];
      gsc = dpose[i] * r[d.depth_c]; which = 1;
    } else if (c < 3 + J3) {
      const int k = c - 3, comp = k % 3;
      const bool pinned = (k == d.root_idx * 3 + 2);
      const float g = pinned ? 0.f : dpose[i] + duvd[p * J3 + k];
      dr[d.uvd_c + k] = g * (comp == 2 ? sc[3] : sc[2]);
      gsc = g * r[d.uvd_c + k]; which = comp == 2 ? 3 : 2;
    } else {
      const int k = c - 3 - J3;
      dr[d.sigma_c + k] = (k == d.root_idx * 3 + 2) ? 0.f : dpose[i];
    }
    if (which >= 0 && gsc != 0.f) atomicAdd(&sred[l * 4 + which], gsc);
  }
  __syncthreads();
  if (threadIdx.x < 20 && sred[threadIdx.x] != 0.f) atomicAdd(dscale + threadIdx.x, sred[threadIdx.x]);
}
}  // namespace

extern "C" int das_deform_im2col3x3_backward(const void* x, const float* om, const void* dcol, float* dx, float* dom,
                                             int dtype, const DasLevels* lv, int C, int x_pix_stride,
                                             int om_pix_stride, int dom_pix_stride, void* stream) {
  if (!x || !om || !dcol || !dx || !dom || !lv_valid(lv) || C % 8 || x_pix_stride % 8 || om_pix_stride < 27 ||
      dom_pix_stride < 27)
    return DAS_ERR_ARG;
  const long long npairs = lv_total_rows(*lv) * 9;  // one wave per (pixel, tap)
  const int blocks = grid_for(npairs * 64, 32768);
  if (dtype == DAS_BF16) {
    hipLaunchKernelGGL(deform_col2im_kernel<bf16_t>, dim3(blocks), dim3(TPB), 0, (hipStream_t)stream,
                       (const bf16_t*)x, om, (const bf16_t*)dcol, dx, dom, *lv, C, x_pix_stride, om_pix_stride,
                       dom_pix_stride, npairs);
  } else if (dtype == DAS_F32) {
    hipLaunchKernelGGL(deform_col2im_kernel<float>, dim3(blocks), dim3(TPB), 0, (hipStream_t)stream,
                       (const float*)x, om, (const float*)dcol, dx, dom, *lv, C, x_pix_stride, om_pix_stride,
                       dom_pix_stride, npairs);
  } else {
    return DAS_ERR_ARG;
  }
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}

extern "C" int das_offset_sample_backward(const float* uvd, const float* samp_off, const float* conf,
                                          const float* grad_out, float* d_uvd, float* d_samp_off, float* d_conf,
                                          const DasLevels* lv, int J, int heads, int uvd_ps, int so_ps, int conf_ps,
                                          int gout_ps, void* stream) {
  if (!uvd || !samp_off || !conf || !grad_out || !d_uvd || !d_samp_off || !d_conf || !lv_valid(lv) || heads != 4 ||
      J < 1)
    return DAS_ERR_ARG;
  const long long total = lv_total_rows(*lv) * J;
  hipLaunchKernelGGL(offset_sample_bwd_kernel, dim3(grid_for(total)), dim3(TPB), 0, (hipStream_t)stream, uvd, samp_off,
                     conf, grad_out, d_uvd, d_samp_off, d_conf, *lv, J, uvd_ps, so_ps, conf_ps, gout_ps, total);
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}

extern "C" int das_sigmoid_blend_backward(const float* off, const float* w, const float* nxt, const float* grad_out,
                                          float* d_off, float* d_w, float* d_nxt, long long npix, int C, int off_ps,
                                          int w_ps, int nxt_ps, void* stream) {
  if (!off || !w || !nxt || !grad_out || !d_off || !d_w || !d_nxt || npix <= 0 || C < 1) return DAS_ERR_ARG;
  hipLaunchKernelGGL(sigmoid_blend_bwd_kernel, dim3(grid_for(npix * C)), dim3(TPB), 0, (hipStream_t)stream, off, w, nxt,
                     grad_out, d_off, d_w, d_nxt, npix, C, off_ps, w_ps, nxt_ps);
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}

extern "C" int das_head_assemble_backward(const float* raw, const float* d_pose, const float* d_uvd, float* d_raw,
                                          float* d_scale, const DasLevels* lv, const DasHeadDesc* d, void* stream) {
  if (!raw || !d_pose || !d_uvd || !d_raw || !d_scale || !lv_valid(lv) || !d || d->J < 1) return DAS_ERR_ARG;
  const long long npix = lv_total_rows(*lv);
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(d_scale, 0, sizeof(float) * 20, s) != hipSuccess) return DAS_ERR_LAUNCH;
  hipLaunchKernelGGL(head_assemble_bwd_kernel, dim3(grid_for(npix * (3 + 6 * d->J), 2048)), dim3(TPB), 0, s, raw,
                     d_pose, d_uvd, d_raw, d_scale, npix, *lv, *d);
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}
