// Backward of the DASHead-specific ops: DCNv2 deformable im2col (col2im + offset/mask gradients),
// the fused recursive-update offset re-sampling, the sigmoid blend and the head assemble.
// Scatter-type gradients accumulate in f32 with atomics (as mmcv's col2im does); the caller
// hands zeroed f32 buffers.
#include "common.h"
#include "prof.h"

#ifndef DAS_DCN_TG
#define DAS_DCN_TG 3   // taps per load group of deform_col2im_kernel (1, 3 or 9)
#endif

namespace {
constexpr int TPB = 256;
inline int grid_for(long long n, int cap = 16384) {
  long long b = (n + TPB - 1) / TPB;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

// ------------------------------------------------------------------ DCNv2 col2im
// dx[p] = sum over (position q, tap k, corner) whose bilinear corner is pixel p of mask * w_corner * dcol[q][k].
// As a scatter that is 4 * 9 * C f32 atomics per position, and f32 atomics retire at ~1 lane per clock per L2
// channel (0.33 G/us on the whole chip, tools/dev/probe/atomic_probe.hip): 3.9 ms per head layer with ONE live
// corner per tap (zero-initialised offsets), four times that once the offsets have been trained. So dx is
// computed as a GATHER instead: one wave per pixel p tests the (2R+1)^2 * 9 (position, tap) pairs around p, one
// pair per lane, and sums the hits — plain stores, a fixed summation order, no zero fill. A pair is REGULAR when
// both corner rows and columns of its sample lie within R of its own position (|offset| < R - 1): exactly those
// pairs are complete in the gather. The pair-centric kernel below (offset / mask gradients, a wave reduction
// written once) scatters only the irregular pairs with atomics, after the gather has written dx.
constexpr int DCN_R = 3;
__device__ __forceinline__ void load4(const float* p, float* v) {
  const float4 t = *reinterpret_cast<const float4*>(p);
  v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
__device__ __forceinline__ void load4(const bf16_t* p, float* v) {
  const uint2 t = *reinterpret_cast<const uint2*>(p);
  v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
  v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
}
__device__ __forceinline__ bool dcn_regular(int y0, int x0, int qy, int qx) {
  return y0 >= qy - DCN_R && y0 + 1 <= qy + DCN_R && x0 >= qx - DCN_R && x0 + 1 <= qx + DCN_R;
}

template <typename T>
__global__ __launch_bounds__(TPB) void deform_col2im_gather_kernel(const float* __restrict__ om,
                                                                   const T* __restrict__ dcol, float* __restrict__ dx,
                                                                   DasLevels lv, int C, int omps, long long rows) {
#pragma clang fp contract(off)
  constexpr int S = 2 * DCN_R + 1, NP = S * S * 9, ROUNDS = (NP + 63) / 64;
  const int lane = threadIdx.x & 63;
  const long long m = ((long long)blockIdx.x * TPB + threadIdx.x) >> 6;   // one wave per pixel
  if (m >= rows) return;
  const LvGeom g = lv_geom(lv, m);
  // the candidate pairs of this pixel, one per lane and round: hit / coefficient / dcol row
  float coef[ROUNDS];
  int drow[ROUNDS];
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const int e = r * 64 + lane;
    const int nq = e / 9, k = e - nq * 9;
    const int qy = g.h + nq / S - DCN_R, qx = g.w + nq % S - DCN_R;
    coef[r] = 0.f;
    drow[r] = 0;
    if (e < NP && (unsigned)qy < (unsigned)g.H && (unsigned)qx < (unsigned)g.W) {
      const long long mq = m + (long long)(qy - g.h) * g.W + (qx - g.w);
      const float* o = om + mq * omps;
      const float ody = o[2 * k], odx = o[2 * k + 1], logit = o[18 + k];
      const float py = (float)(qy - 1 + k / 3) + ody;
      const float px = (float)(qx - 1 + k % 3) + odx;
      if (py > -1.f && px > -1.f && py < (float)g.H && px < (float)g.W) {
        const float fy = floorf(py), fx = floorf(px);
        const int y0 = (int)fy, x0 = (int)fx;
        const int cy = g.h - y0, cx = g.w - x0;
        if (dcn_regular(y0, x0, qy, qx) && (unsigned)cy < 2u && (unsigned)cx < 2u) {
          const float ly = py - fy, lx = px - fx, hy = 1.f - ly, hx = 1.f - lx;
          const float wgt = (cy ? ly : hy) * (cx ? lx : hx);
          const float mask = 1.f / (1.f + expf(-logit));
          coef[r] = mask * wgt;
          drow[r] = (int)(mq * 9 + k);
        }
      }
    }
  }
  for (int c0 = 0; c0 < C; c0 += 256) {
    const int ch = c0 + lane * 4;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      unsigned long long hits = __ballot(coef[r] != 0.f);
      // four hits per trip: their dcol vectors are requested together (one load in flight per wave left this kernel
      // latency bound). A missing hit repeats the last one with coefficient 0; the summation order is unchanged.
      while (hits) {
        float cf[4];
        int dr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int l = hits ? __builtin_ctzll(hits) : 0;
          cf[u] = hits ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, coef[r]), l)) : 0.f;
          dr[u] = hits ? __builtin_amdgcn_readlane(drow[r], l) : (u ? dr[u - 1] : 0);
          hits &= hits - 1;
        }
        if (ch < C) {
          float v[4][4];
#pragma unroll
          for (int u = 0; u < 4; ++u) load4(dcol + (long long)dr[u] * C + ch, v[u]);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (cf[u] != 0.f) { a0 += cf[u] * v[u][0]; a1 += cf[u] * v[u][1]; a2 += cf[u] * v[u][2]; a3 += cf[u] * v[u][3]; }
          }
        }
      }
    }
    if (ch < C) *reinterpret_cast<float4*>(dx + m * C + ch) = make_float4(a0, a1, a2, a3);
  }
}

// Sum over the wave by DPP row shifts / row broadcasts (no LDS traffic); the total lands in lane 63.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, true));
}
__device__ __forceinline__ float wave_sum63(float v) {
  v = dpp_add<0x111, 0xF>(v);   // row_shr:1
  v = dpp_add<0x112, 0xF>(v);   // row_shr:2
  v = dpp_add<0x114, 0xF>(v);   // row_shr:4
  v = dpp_add<0x118, 0xF>(v);   // row_shr:8  -> lane 15 of every row holds the row's sum
  v = dpp_add<0x142, 0xA>(v);   // row_bcast:15 into rows 1, 3
  v = dpp_add<0x143, 0xC>(v);   // row_bcast:31 into rows 2, 3 -> lane 63 holds the wave's sum
  return v;
}

// One WAVE = one pixel (its nine taps in turn); lane = four channels (256 at a time): the offset / mask
// gradients, one wave reduction per tap written once, and the corner scatters of the irregular pairs as f32
// atomics (lane = one channel there, so that a wave's atomics are contiguous). Measured: fetching the nine taps'
// dcol vectors up front and batching the 27 reductions costs more in registers than it hides in latency.
// (raw 4-channel vectors: kept packed while in flight, 2 registers for bf16)
template <typename T> struct Raw4;
template <> struct Raw4<float> {
  using type = float4;
  static __device__ __forceinline__ void unpack(const float4& t, float* v) { v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
};
template <> struct Raw4<bf16_t> {
  using type = uint2;
  static __device__ __forceinline__ void unpack(const uint2& t, float* v) {
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
    v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
  }
};

// One WAVE = one pixel: the offset / mask gradients (one wave reduction per tap, written once) and the corner scatters
// of the irregular pairs as f32 atomics (lane = one channel there, so that a wave's atomics are contiguous).
// The per-tap geometry (sigmoid, floor, bilinear weights, corner rows) is computed ONCE, by lane k for tap k, and
// handed to the channel loop through readlane (scalar registers): computed per tap in every lane it made the kernel
// VALU bound (2150 vector instructions per pixel, SQ_INSTS_VALU; 0.5 ms of the 0.7).
__device__ __forceinline__ float rl_f(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
template <typename T>
__global__ __launch_bounds__(TPB) void deform_col2im_kernel(const T* __restrict__ x, const float* __restrict__ om,
                                                            const T* __restrict__ dcol, float* __restrict__ dx,
                                                            float* __restrict__ dom, DasLevels lv, int C, int xps,
                                                            int omps, int domps, long long rows) {
#pragma clang fp contract(off)
  using R4 = typename Raw4<T>::type;
  constexpr int TG = DAS_DCN_TG;   // taps whose 5 vectors (dcol + 4 corners, kept packed) are requested together
  static_assert(9 % TG == 0, "tap groups");
  const int lane = threadIdx.x & 63;
  const long long m = ((long long)blockIdx.x * TPB + threadIdx.x) >> 6;
  if (m >= rows) return;
  const LvGeom g = lv_geom(lv, m);
  const int H = g.H, W = g.W;
  // ---- lane k < 9: tap k
  const int kk = lane < 9 ? lane : 0;
  const float* o = om + m * omps;
  const float ody = o[2 * kk], odx = o[2 * kk + 1], logit = o[18 + kk];
  const float vmask = 1.f / (1.f + expf(-logit));
  const float py = (float)(g.h - 1 + kk / 3) + ody;
  const float px = (float)(g.w - 1 + kk % 3) + odx;
  const bool vlive = py > -1.f && px > -1.f && py < (float)H && px < (float)W;
  const float fy = floorf(py), fx = floorf(px);
  const int vy0 = vlive ? (int)fy : 0, vx0 = vlive ? (int)fx : 0;
  const float vly = py - fy, vlx = px - fx;
  int vok = vlive ? 16 : 0;     // bit c = corner c inside the plane, bit 4 = tap live, bit 5 = irregular pair
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int yy = vy0 + (c >> 1), xx = vx0 + (c & 1);
    if (vlive && yy >= 0 && yy <= H - 1 && xx >= 0 && xx <= W - 1) vok |= 1 << c;
  }
  if (vlive && !dcn_regular(vy0, vx0, g.h, g.w)) vok |= 32;
  const int vrow = (int)(g.plane0 + (long long)vy0 * W + vx0);   // pixel row of corner 0 (rows < 2^31: checked by the host)

  const T* dc = dcol + m * 9 * C;
  float* d = dom + m * domps;
#pragma unroll 1
  for (int k0 = 0; k0 < 9; k0 += TG) {
    float val[TG], gpy[TG], gpx[TG];
#pragma unroll
    for (int t = 0; t < TG; ++t) { val[t] = 0.f; gpy[t] = 0.f; gpx[t] = 0.f; }
    for (int c0 = 0; c0 < C; c0 += 256) {
      const int ch = c0 + lane * 4;
      if (ch >= C) break;
      R4 gr[TG], fr[TG][4];
#pragma unroll
      for (int t = 0; t < TG; ++t) {
        const int ok = __builtin_amdgcn_readlane(vok, k0 + t), row = __builtin_amdgcn_readlane(vrow, k0 + t);
        gr[t] = *reinterpret_cast<const R4*>(dc + (k0 + t) * C + ch);
#pragma unroll
        for (int c = 0; c < 4; ++c) {   // (a corner outside the plane re-reads this pixel's own row: never used)
          const long long pix = (ok >> c & 1) ? (long long)row + (c >> 1) * W + (c & 1) : m;
          fr[t][c] = *reinterpret_cast<const R4*>(x + pix * xps + ch);
        }
      }
#pragma unroll
      for (int t = 0; t < TG; ++t) {
        const int ok = __builtin_amdgcn_readlane(vok, k0 + t);
        const float ly = rl_f(vly, k0 + t), lx = rl_f(vlx, k0 + t);
        float gc[4];
        Raw4<T>::unpack(gr[t], gc);
        const float hy = 1.f - ly, hx = 1.f - lx;
        const float wts[4] = {hy * hx, hy * lx, ly * hx, ly * lx};
        const float wy[4] = {-hx, -lx, hx, lx};     // d(weight)/d(py)
        const float wx[4] = {-hy, hy, -ly, ly};     // d(weight)/d(px)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if (!(ok >> c & 1)) continue;   // wave-uniform
          float f[4];
          Raw4<T>::unpack(fr[t][c], f);
          const float dot = (gc[0] * f[0] + gc[1] * f[1]) + (gc[2] * f[2] + gc[3] * f[3]);
          val[t] += wts[c] * dot;
          gpy[t] += wy[c] * dot;
          gpx[t] += wx[c] * dot;
        }
      }
    }
#pragma unroll
    for (int t = 0; t < TG; ++t) {
      const int k = k0 + t;
      const int ok = __builtin_amdgcn_readlane(vok, k);
      if (!(ok & 16)) continue;   // wave-uniform: the outputs of a dead tap stay as the caller zeroed them
      const float mask = rl_f(vmask, k);
      if (ok & 32) {   // irregular pair (regular pairs: dx comes from the gather kernel)
        const float ly = rl_f(vly, k), lx = rl_f(vlx, k), hy = 1.f - ly, hx = 1.f - lx;
        const float wts[4] = {hy * hx, hy * lx, ly * hx, ly * lx};
        const int row = __builtin_amdgcn_readlane(vrow, k);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          // a corner with zero bilinear weight adds exactly 0.0: skip the atomics (wave-uniform)
          if (!(ok >> c & 1) || wts[c] == 0.f) continue;
          float* dst = dx + ((long long)row + (c >> 1) * W + (c & 1)) * C;
          for (int ch = lane; ch < C; ch += 64) atomicAdd(dst + ch, Elem<T>::load(dc + k * C + ch) * mask * wts[c]);
        }
      }
      const float v = wave_sum63(val[t]), py_ = wave_sum63(gpy[t]), px_ = wave_sum63(gpx[t]);
      if (lane == 63) {
        d[2 * k] = py_ * mask;
        d[2 * k + 1] = px_ * mask;
        d[18 + k] = v * mask * (1.f - mask);
      }
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
    {   // The loss reaches this op at the positive locations only (a fraction of a percent of the pixels): where the
        // upstream gradient of a (pixel, joint) is zero, every term below is zero and the outputs keep the caller's zeros
      const float* g0 = gout + pix * gout_ps + j * 3;
      if (g0[0] == 0.f && g0[1] == 0.f && g0[2] == 0.f) continue;
    }
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
    const float* sc = d.scale_dev ? d.scale_dev + 4 * l : d.scale[l];
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
  DAS_PROF(stream);
  if (!x || !om || !dcol || !dx || !dom || !lv_valid(lv) || C % 8 || x_pix_stride % 8 || om_pix_stride < 27 ||
      dom_pix_stride < 27)
    return DAS_ERR_ARG;
  const long long rows = lv_total_rows(*lv);
  const long long npairs = rows * 9;  // one wave per (pixel, tap)
  if (npairs >= 0x7fffffffLL || rows * 64 / TPB >= 0x7fffffffLL) return DAS_ERR_ARG;
  const int gblocks = (int)((rows * 64 + TPB - 1) / TPB);   // gather: one wave per pixel, writes all of dx
  if (dtype == DAS_BF16) {
    hipLaunchKernelGGL(deform_col2im_gather_kernel<bf16_t>, dim3(gblocks), dim3(TPB), 0, (hipStream_t)stream, om,
                       (const bf16_t*)dcol, dx, *lv, C, om_pix_stride, rows);
  } else if (dtype == DAS_F32) {
    hipLaunchKernelGGL(deform_col2im_gather_kernel<float>, dim3(gblocks), dim3(TPB), 0, (hipStream_t)stream, om,
                       (const float*)dcol, dx, *lv, C, om_pix_stride, rows);
  } else {
    return DAS_ERR_ARG;
  }
  DAS_CHECK_LAUNCH();
  if (dtype == DAS_BF16) {   // offset / mask gradients (+ the irregular pairs' scatter): one wave per pixel too
    hipLaunchKernelGGL(deform_col2im_kernel<bf16_t>, dim3(gblocks), dim3(TPB), 0, (hipStream_t)stream,
                       (const bf16_t*)x, om, (const bf16_t*)dcol, dx, dom, *lv, C, x_pix_stride, om_pix_stride,
                       dom_pix_stride, rows);
  } else if (dtype == DAS_F32) {
    hipLaunchKernelGGL(deform_col2im_kernel<float>, dim3(gblocks), dim3(TPB), 0, (hipStream_t)stream,
                       (const float*)x, om, (const float*)dcol, dx, dom, *lv, C, x_pix_stride, om_pix_stride,
                       dom_pix_stride, rows);
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
  DAS_PROF(stream);
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
  DAS_PROF(stream);
  if (!off || !w || !nxt || !grad_out || !d_off || !d_w || !d_nxt || npix <= 0 || C < 1) return DAS_ERR_ARG;
  hipLaunchKernelGGL(sigmoid_blend_bwd_kernel, dim3(grid_for(npix * C)), dim3(TPB), 0, (hipStream_t)stream, off, w, nxt,
                     grad_out, d_off, d_w, d_nxt, npix, C, off_ps, w_ps, nxt_ps);
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}

extern "C" int das_head_assemble_backward(const float* raw, const float* d_pose, const float* d_uvd, float* d_raw,
                                          float* d_scale, const DasLevels* lv, const DasHeadDesc* d, void* stream) {
  DAS_PROF(stream);
  if (!raw || !d_pose || !d_uvd || !d_raw || !d_scale || !lv_valid(lv) || !d || d->J < 1) return DAS_ERR_ARG;
  const long long npix = lv_total_rows(*lv);
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(d_scale, 0, sizeof(float) * 20, s) != hipSuccess) return DAS_ERR_LAUNCH;
  hipLaunchKernelGGL(head_assemble_bwd_kernel, dim3(grid_for(npix * (3 + 6 * d->J), 2048)), dim3(TPB), 0, s, raw,
                     d_pose, d_uvd, d_raw, d_scale, npix, *lv, *d);
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}
