// DASHead-specific kernels: DCNv2 deformable im2col, the fused recursive-update offset
// re-sampling, the sigmoid-gated blend and the per-level assemble / finalize of pose_pred.
// All HBM / latency bound gather work in f32 arithmetic. Every kernel works on "ragged" rows
// (DasLevels): one launch covers all FPN levels.
#include "common.h"
#include "prof.h"

namespace {
constexpr int TPB = 256;
inline int grid_for(long long n) {
  long long b = (n + TPB - 1) / TPB;
  return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

// ------------------------------------------------------------------ DCNv2 im2col (3x3, s1, p1)
// One thread = one (pixel, tap, 16-B channel vector). col row = [tap][C].
template <typename T>
__global__ void deform_im2col_kernel(const T* __restrict__ x, const float* __restrict__ om, T* __restrict__ col,
                                     DasLevels lv, int C, int xps, int omps, long long total) {
#pragma clang fp contract(off)
  constexpr int EPV = Elem<T>::EPV;
  const int VC = C / EPV;
  // one thread = one (pixel, channel vector), its nine taps in turn: the row decode (64-bit divisions) is paid once
  // per nine outputs (as one thread per (pixel, tap, vector) the pass ran at 3.6x its HBM floor)
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
    const int v = (int)(i % VC);
    const long long m = i / VC;
    const LvGeom g = lv_geom(lv, m);
    const int H = g.H, W = g.W;
    const float* o = om + m * omps;
#pragma unroll 3
   for (int k = 0; k < 9; ++k) {
    const float dy = o[2 * k], dx = o[2 * k + 1];
    const float mask = 1.f / (1.f + expf(-o[18 + k]));
    const float py = (float)(g.h - 1 + k / 3) + dy;
    const float px = (float)(g.w - 1 + k % 3) + dx;
    float out[EPV];
#pragma unroll
    for (int j = 0; j < EPV; ++j) out[j] = 0.f;
    if (py > -1.f && px > -1.f && py < (float)H && px < (float)W) {
      const float fy = floorf(py), fx = floorf(px);
      const int y0 = (int)fy, x0 = (int)fx;
      const float ly = py - fy, lx = px - fx, hy = 1.f - ly, hx = 1.f - lx;
      const T* base = x + g.plane0 * (long long)xps + v * EPV;
      const float wts[4] = {hy * hx, hy * lx, ly * hx, ly * lx};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int yy = y0 + (c >> 1), xx = x0 + (c & 1);
        if (yy >= 0 && yy <= H - 1 && xx >= 0 && xx <= W - 1) {
          float f[EPV];
          Elem<T>::unpack(*reinterpret_cast<const uint4*>(base + ((long long)yy * W + xx) * xps), f);
#pragma unroll
          for (int j = 0; j < EPV; ++j) out[j] += wts[c] * f[j];
        }
      }
#pragma unroll
      for (int j = 0; j < EPV; ++j) out[j] *= mask;
    }
    *reinterpret_cast<uint4*>(col + (m * 9 + k) * C + v * EPV) = Elem<T>::pack(out);
   }
  }
}

// Wave-level variant (C / EPV in {16, 32, 64} channel vectors per pixel): a wave covers 64 / VC pixels. The tap geometry
// (sigmoid, floor, bilinear fractions, corner validity, corner row) is computed ONCE per (pixel, tap) — by lane
// p * 9 + k — and fetched by the lanes that need it with ds_bpermute; computed per channel vector, as above, it made the
// pass VALU bound (2500 vector instructions per wave against a 652 MB store stream, SQ_INSTS_VALU). Corner vectors of
// three taps are in flight together.
template <typename T, int VC>
__global__ __launch_bounds__(256) void deform_im2col_wave_kernel(const T* __restrict__ x, const float* __restrict__ om,
                                                                 T* __restrict__ col, DasLevels lv, int C, int xps,
                                                                 int omps, long long rows) {
#pragma clang fp contract(off)
  constexpr int EPV = Elem<T>::EPV, PPW = 64 / VC, TG = 3;
  const int lane = threadIdx.x & 63;
  const long long m0 = (((long long)blockIdx.x * 256 + threadIdx.x) >> 6) * PPW;
  if (m0 >= rows) return;
  // ---- producer role: lane p * 9 + k
  float vmask, vly, vlx;
  int vok = 0, vrow = 0, vW = 1;
  {
    const int gp = lane < PPW * 9 ? lane / 9 : 0, gk = lane < PPW * 9 ? lane % 9 : 0;
    const long long gm = m0 + gp < rows ? m0 + gp : rows - 1;
    const LvGeom g = lv_geom(lv, gm);
    const float* o = om + gm * omps;
    const float dy = o[2 * gk], dx = o[2 * gk + 1];
    vmask = 1.f / (1.f + expf(-o[18 + gk]));
    const float py = (float)(g.h - 1 + gk / 3) + dy;
    const float px = (float)(g.w - 1 + gk % 3) + dx;
    const bool live = py > -1.f && px > -1.f && py < (float)g.H && px < (float)g.W;
    const float fy = floorf(py), fx = floorf(px);
    const int y0 = live ? (int)fy : 0, x0 = live ? (int)fx : 0;
    vly = py - fy; vlx = px - fx;
    vok = live ? 16 : 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int yy = y0 + (c >> 1), xx = x0 + (c & 1);
      if (live && yy >= 0 && yy <= g.H - 1 && xx >= 0 && xx <= g.W - 1) vok |= 1 << c;
    }
    vrow = (int)(g.plane0 + (long long)y0 * g.W + x0);   // (row counts < 2^31: checked by the host)
    vW = g.W;
  }
  // ---- consumer role: channel vector v of pixel mp
  const int mp = lane / VC, v = lane - mp * VC;
  const long long m = m0 + mp;
  const bool valid = m < rows;
  const long long msafe = valid ? m : rows - 1;
  auto from = [&](int val, int src) { return __builtin_amdgcn_ds_bpermute(src << 2, val); };
  auto fromf = [&](float val, int src) { return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src << 2, __builtin_bit_cast(int, val))); };
  const int W = from(vW, mp * 9);
  const T* xb = x + v * EPV;
#pragma unroll 1
  for (int k0 = 0; k0 < 9; k0 += TG) {
    uint4 fr[TG][4];
    int ok[TG];
    float ly[TG], lx[TG], mk[TG];
#pragma unroll
    for (int t = 0; t < TG; ++t) {
      const int src = mp * 9 + k0 + t;
      ok[t] = from(vok, src);
      const int row = from(vrow, src);
      ly[t] = fromf(vly, src); lx[t] = fromf(vlx, src); mk[t] = fromf(vmask, src);
#pragma unroll
      for (int c = 0; c < 4; ++c) {   // (a corner outside the plane re-reads this pixel's own row: weight 0)
        const long long pix = (ok[t] >> c & 1) ? (long long)row + (c >> 1) * W + (c & 1) : msafe;
        fr[t][c] = *reinterpret_cast<const uint4*>(xb + pix * xps);
      }
    }
#pragma unroll
    for (int t = 0; t < TG; ++t) {
      const float hy = 1.f - ly[t], hx = 1.f - lx[t];
      const float wts[4] = {hy * hx, hy * lx[t], ly[t] * hx, ly[t] * lx[t]};
      float out[EPV];
#pragma unroll
      for (int j = 0; j < EPV; ++j) out[j] = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (!(ok[t] >> c & 1)) continue;
        float f[EPV];
        Elem<T>::unpack(fr[t][c], f);
#pragma unroll
        for (int j = 0; j < EPV; ++j) out[j] += wts[c] * f[j];
      }
#pragma unroll
      for (int j = 0; j < EPV; ++j) out[j] = (ok[t] & 16) ? out[j] * mk[t] : 0.f;
      if (valid) *reinterpret_cast<uint4*>(col + (m * 9 + k0 + t) * C + v * EPV) = Elem<T>::pack(out);
    }
  }
}

// ------------------------------------------------------------------ grid_sample helper
// torch grid_sample(bilinear, zeros, align_corners=False) on an NHWC f32 map, NCH channels
// starting at `c`, evaluated at normalised location loc in [0,1] units (grid = 2*loc-1).
template <int NCH>
__device__ __forceinline__ void sample_nhwc(const float* __restrict__ img, int H, int W, int ps, int c, float locx,
                                            float locy, float* out) {
#pragma clang fp contract(off)
  const float gx = 2.f * locx - 1.f, gy = 2.f * locy - 1.f;
  const float ix = ((gx + 1.f) * W - 1.f) / 2.f, iy = ((gy + 1.f) * H - 1.f) / 2.f;
  const float fx = floorf(ix), fy = floorf(iy);
  const int x0 = (int)fx, y0 = (int)fy;
  const float tx = ix - fx, ty = iy - fy;
  const float w_nw = (1.f - tx) * (1.f - ty), w_ne = tx * (1.f - ty), w_sw = (1.f - tx) * ty, w_se = tx * ty;
#pragma unroll
  for (int j = 0; j < NCH; ++j) out[j] = 0.f;
  const bool xin0 = x0 >= 0 && x0 < W, xin1 = x0 + 1 >= 0 && x0 + 1 < W;
  const bool yin0 = y0 >= 0 && y0 < H, yin1 = y0 + 1 >= 0 && y0 + 1 < H;
  if (yin0 && xin0) {
    const float* p = img + ((long long)y0 * W + x0) * ps + c;
#pragma unroll
    for (int j = 0; j < NCH; ++j) out[j] += p[j] * w_nw;
  }
  if (yin0 && xin1) {
    const float* p = img + ((long long)y0 * W + x0 + 1) * ps + c;
#pragma unroll
    for (int j = 0; j < NCH; ++j) out[j] += p[j] * w_ne;
  }
  if (yin1 && xin0) {
    const float* p = img + ((long long)(y0 + 1) * W + x0) * ps + c;
#pragma unroll
    for (int j = 0; j < NCH; ++j) out[j] += p[j] * w_sw;
  }
  if (yin1 && xin1) {
    const float* p = img + ((long long)(y0 + 1) * W + x0 + 1) * ps + c;
#pragma unroll
    for (int j = 0; j < NCH; ++j) out[j] += p[j] * w_se;
  }
}

// One thread = one (pixel, joint). heads is fixed at 4 (2*heads = 8 sampling sites).
__global__ void offset_sample_kernel(const float* __restrict__ uvd, const float* __restrict__ so,
                                     const float* __restrict__ conf, float* __restrict__ out, DasLevels lv, int J,
                                     int uvd_ps, int so_ps, int conf_ps, int out_ps, long long total) {
#pragma clang fp contract(off)
  constexpr int HEADS = 4, S = 2 * HEADS;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
    const int j = (int)(i % J);
    const long long pix = i / J;
    const LvGeom g = lv_geom(lv, pix);
    const int H = g.H, W = g.W, px = g.w, py = g.h;
    const float* uvd_b = uvd + g.plane0 * (long long)uvd_ps;
    const float* so_b = so + g.plane0 * (long long)so_ps;
    const float* conf_b = conf + g.plane0 * (long long)conf_ps;
    const float fw = (float)W, fh = (float)H;
    const float cx = (float)px + 0.5f, cy = (float)py + 0.5f;
    const float* u = uvd_b + ((long long)py * W + px) * uvd_ps + j * 3;
    const float offx = u[0], offy = u[1];
    // stage 1: the 4 head offsets as seen from the current target location
    float sx[S], sy[S];
    {
      float tmp[2 * HEADS];
      sample_nhwc<2 * HEADS>(so_b, H, W, so_ps, j * 2 * HEADS, (cx + offx) / fw, (cy + offy) / fh, tmp);
      const float* own = so_b + ((long long)py * W + px) * so_ps + j * 2 * HEADS;
#pragma unroll
      for (int h = 0; h < HEADS; ++h) {
        sx[h] = tmp[2 * h] + offx;
        sy[h] = tmp[2 * h + 1] + offy;
        sx[HEADS + h] = own[2 * h];
        sy[HEADS + h] = own[2 * h + 1];
      }
    }
    // stage 2: sample [uvd(3), conf(3)] at each site, softmax over the 8 sites per dim
    float val[S][3], cf[S][3];
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const float lx = (cx + sx[s]) / fw, ly = (cy + sy[s]) / fh;
      sample_nhwc<3>(uvd_b, H, W, uvd_ps, j * 3, lx, ly, val[s]);
      sample_nhwc<3>(conf_b, H, W, conf_ps, j * 3, lx, ly, cf[s]);
      val[s][0] += sx[s];
      val[s][1] += sy[s];
    }
    float* o = out + pix * out_ps + j * 3;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      float mx = cf[0][d];
#pragma unroll
      for (int s = 1; s < S; ++s) mx = fmaxf(mx, cf[s][d]);
      float e[S], den = 0.f;
#pragma unroll
      for (int s = 0; s < S; ++s) { e[s] = expf(cf[s][d] - mx); den += e[s]; }
      float acc = 0.f;
#pragma unroll
      for (int s = 0; s < S; ++s) acc += val[s][d] * (e[s] / den);
      o[d] = acc;
    }
  }
}

__global__ void sigmoid_blend_kernel(const float* __restrict__ off, const float* __restrict__ w,
                                     const float* __restrict__ nxt, float* __restrict__ out, long long npix, int C,
                                     int off_ps, int w_ps, int nxt_ps, int out_ps) {
#pragma clang fp contract(off)
  const long long total = npix * C;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
    const int c = (int)(i % C);
    const long long p = i / C;
    const float g = 1.f / (1.f + expf(-w[p * w_ps + c]));
    out[p * out_ps + c] = (1.f - g) * off[p * off_ps + c] + g * nxt[p * nxt_ps + c];
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

__global__ void head_assemble_kernel(const float* __restrict__ raw, float* __restrict__ pose, float* __restrict__ uvd,
                                     long long npix, DasLevels lv, DasHeadDesc d) {
  const int J3 = 3 * d.J, D = 3 + 6 * d.J;
  const long long total = npix * D;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
    const int c = (int)(i % D);
    const long long p = i / D;
    const int lvl = level_of(lv, p);
    const float* sc = d.scale_dev ? d.scale_dev + 4 * lvl : d.scale[lvl];
    const float* r = raw + p * d.raw_ps;
    float v;
    if (c < 2) {
      v = r[d.off_c + c] * sc[0];
    } else if (c == 2) {
      v = r[d.depth_c] * sc[1];
    } else if (c < 3 + J3) {
      const int k = c - 3, comp = k % 3;
      v = r[d.uvd_c + k] * (comp == 2 ? sc[3] : sc[2]);
      if (k == d.root_idx * 3 + 2) v = 0.f;
      uvd[p * J3 + k] = v;
    } else {
      const int k = c - 3 - J3;
      v = r[d.sigma_c + k];
      if (k == d.root_idx * 3 + 2) v = 1.f;
    }
    pose[i] = v;
  }
}

__global__ void head_finalize_kernel(float* __restrict__ pose, float* __restrict__ ref, long long npix, DasLevels lv,
                                     DasHeadDesc d, int ref_ps, int eval_mode) {
#pragma clang fp contract(off)
  const int J3 = 3 * d.J, D = 3 + 6 * d.J;
  const long long total = npix * (J3 + 1);
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
    const int k = (int)(i % (J3 + 1));
    const long long p = i / (J3 + 1);
    if (k == J3) {
      if (eval_mode) pose[p * D + 2] = pose[p * D + 2] / d.depth_factor;
      continue;
    }
    float v = ref[p * ref_ps + k];
    if (k == d.root_idx * 3 + 2) {
      v = 0.f;
      ref[p * ref_ps + k] = 0.f;
    }
    if (eval_mode) pose[p * D + 3 + k] = v * ((k % 3) == 2 ? d.z_norm : d.level_stride[level_of(lv, p)]);
  }
}
}  // namespace

extern "C" int das_deform_im2col3x3(const void* x, const float* om, void* col, int dtype, const DasLevels* lv, int C,
                                    int x_pix_stride, int om_pix_stride, void* stream) {
  DAS_PROF(stream);
  if (!x || !om || !col || !lv_valid(lv) || C % 8 || x_pix_stride % 8 || om_pix_stride < 27) return DAS_ERR_ARG;
  const long long npix = lv_total_rows(*lv);
  if (dtype != DAS_BF16 && dtype != DAS_F32) return DAS_ERR_ARG;
  const int vc = C / (dtype == DAS_BF16 ? 8 : 4);
  hipStream_t s = (hipStream_t)stream;
  if ((vc == 16 || vc == 32 || vc == 64) && npix < 0x7fffffffLL) {   // wave-level kernel: 64 / vc pixels per wave
    const long long waves = (npix + 64 / vc - 1) / (64 / vc);
    const unsigned blocks = (unsigned)((waves + 3) / 4);
#define DAS_IM2COL_CASE(TT, VCV)                                                                                   \
    hipLaunchKernelGGL((deform_im2col_wave_kernel<TT, VCV>), dim3(blocks), dim3(256), 0, s, (const TT*)x, om, (TT*)col, \
                       *lv, C, x_pix_stride, om_pix_stride, npix)
    if (dtype == DAS_BF16) {
      if (vc == 16) DAS_IM2COL_CASE(bf16_t, 16); else if (vc == 32) DAS_IM2COL_CASE(bf16_t, 32); else DAS_IM2COL_CASE(bf16_t, 64);
    } else {
      if (vc == 16) DAS_IM2COL_CASE(float, 16); else if (vc == 32) DAS_IM2COL_CASE(float, 32); else DAS_IM2COL_CASE(float, 64);
    }
#undef DAS_IM2COL_CASE
  } else if (dtype == DAS_BF16) {
    const long long total = npix * (C / 8);
    hipLaunchKernelGGL(deform_im2col_kernel<bf16_t>, dim3(grid_for(total)), dim3(TPB), 0, s,
                       (const bf16_t*)x, om, (bf16_t*)col, *lv, C, x_pix_stride, om_pix_stride, total);
  } else {
    const long long total = npix * (C / 4);
    hipLaunchKernelGGL(deform_im2col_kernel<float>, dim3(grid_for(total)), dim3(TPB), 0, s,
                       (const float*)x, om, (float*)col, *lv, C, x_pix_stride, om_pix_stride, total);
  }
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}

extern "C" int das_offset_sample(const float* uvd, const float* samp_off, const float* conf, float* out,
                                 const DasLevels* lv, int J, int heads, int uvd_ps, int so_ps, int conf_ps, int out_ps,
                                 void* stream) {
  DAS_PROF(stream);
  if (!uvd || !samp_off || !conf || !out || !lv_valid(lv) || heads != 4 || J < 1) return DAS_ERR_ARG;
  if (uvd_ps < 3 * J || conf_ps < 3 * J || out_ps < 3 * J || so_ps < 8 * J) return DAS_ERR_ARG;
  const long long total = lv_total_rows(*lv) * J;
  hipLaunchKernelGGL(offset_sample_kernel, dim3(grid_for(total)), dim3(TPB), 0, (hipStream_t)stream, uvd, samp_off,
                     conf, out, *lv, J, uvd_ps, so_ps, conf_ps, out_ps, total);
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}

extern "C" int das_sigmoid_blend(const float* off, const float* w, const float* nxt, float* out, long long npix, int C,
                                 int off_ps, int w_ps, int nxt_ps, int out_ps, void* stream) {
  DAS_PROF(stream);
  if (!off || !w || !nxt || !out || npix <= 0 || C < 1) return DAS_ERR_ARG;
  hipLaunchKernelGGL(sigmoid_blend_kernel, dim3(grid_for(npix * C)), dim3(TPB), 0, (hipStream_t)stream, off, w, nxt,
                     out, npix, C, off_ps, w_ps, nxt_ps, out_ps);
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}

static bool head_desc_ok(const DasHeadDesc* d) { return d && d->J >= 1 && d->root_idx >= 0 && d->root_idx < d->J; }

extern "C" int das_head_assemble(const float* raw, float* pose_pred, float* uvd_out, const DasLevels* lv,
                                 const DasHeadDesc* d, void* stream) {
  DAS_PROF(stream);
  if (!raw || !pose_pred || !uvd_out || !lv_valid(lv) || !head_desc_ok(d)) return DAS_ERR_ARG;
  const long long npix = lv_total_rows(*lv);
  hipLaunchKernelGGL(head_assemble_kernel, dim3(grid_for(npix * (3 + 6 * d->J))), dim3(TPB), 0, (hipStream_t)stream,
                     raw, pose_pred, uvd_out, npix, *lv, *d);
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}

extern "C" int das_head_finalize(float* pose_pred, float* ref_uvd, const DasLevels* lv, const DasHeadDesc* d,
                                 int ref_ps, int eval_mode, void* stream) {
  DAS_PROF(stream);
  if (!pose_pred || !ref_uvd || !lv_valid(lv) || !head_desc_ok(d) || ref_ps < 3 * d->J) return DAS_ERR_ARG;
  const long long npix = lv_total_rows(*lv);
  hipLaunchKernelGGL(head_finalize_kernel, dim3(grid_for(npix * (3 * d->J + 1))), dim3(TPB), 0, (hipStream_t)stream,
                     pose_pred, ref_uvd, npix, *lv, *d, ref_ps, eval_mode);
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}
