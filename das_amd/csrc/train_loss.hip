// Dense-head training kernels: point-to-person target assignment, sigmoid focal loss, SmoothL1 / BCE on
// the positives, and the optimizer side (global grad norm, fused SGD-momentum with clipping).
#include <algorithm>
#include "prof.h"

#include "common.h"

namespace {
constexpr int TPB = 256;
inline int grid_for(long long n, int cap = 4096) {
  long long b = (n + TPB - 1) / TPB;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) sh[w] = v;
  __syncthreads();
  float t = 0.f;
  if (threadIdx.x == 0)
    for (int i = 0; i < TPB / 64; ++i) t += sh[i];
  __syncthreads();
  return t;  // valid on thread 0
}

// ------------------------------------------------------------------ target assignment
// das_head.py:551-651 `_get_target_single` for every location of every level and image in one launch.
// One thread per row (level-major, image, h, w). gt: rows [cx, cy, depth, J x (u,v,dz), J x vis].
// centers (optional): rows [centers2d.x, centers2d.y, depths] of the same persons — the reference takes the
// root offsets, the centre box and the depth target from `centers2d` / `depths` and only the joint offsets from
// gt_poses_3d[:, :3] (das_head.py:570-589); NULL = they equal gt[:, :3] (the datasets' invariant).
__global__ void assign_targets_kernel(DasLevels lv, DasTargetDesc d, const float* __restrict__ gt,
                                      const float* __restrict__ centers,
                                      const int* __restrict__ gt_start, int* __restrict__ labels,
                                      float* __restrict__ targets, float* __restrict__ ctr_t, long long rows,
                                      float* __restrict__ counts) {
#pragma clang fp contract(off)
  __shared__ float sh[TPB / 64];
  const int J = d.J, D = 3 + 4 * J;
  float n_pos = 0.f, n_3d = 0.f, n_vis = 0.f;   // (counts: positives, positives with a depth annotation, their visible joints)
  for (long long m = (long long)blockIdx.x * TPB + threadIdx.x; m < rows; m += (long long)gridDim.x * TPB) {
    const LvGeom g = lv_geom(lv, m);
    const int s = d.stride[g.l];
    const float px = (float)(g.w * s + s / 2), py = (float)(g.h * s + s / 2);
    const float rad = (float)s * d.radius;
    const float lo = d.range_lo[g.l], hi = d.range_hi[g.l];
    float best = 1e8f;
    int best_i = -1;
    for (int i = gt_start[g.b]; i < gt_start[g.b + 1]; ++i) {
      const float* p = gt + (long long)i * D;
      float reach = 0.f;  // max over joints of |joint - centre| * vis
      for (int j = 0; j < J; ++j) {
        const float du = p[3 + 3 * j] - p[0], dv = p[4 + 3 * j] - p[1];
        reach = fmaxf(reach, sqrtf(du * du + dv * dv) * p[3 + 3 * J + j]);
      }
      const float cx = centers ? centers[3 * i] : p[0], cy = centers ? centers[3 * i + 1] : p[1];
      const float l = px - (cx - rad), r = (cx + rad) - px, t = py - (cy - rad), b = (cy + rad) - py;
      const bool inside = fminf(fminf(l, t), fminf(r, b)) > 0.f;
      const bool in_range = reach >= lo && reach <= hi;
      const float dx = px - cx, dy = py - cy;
      float dist = sqrtf(dx * dx + dy * dy);
      if (!(inside && in_range)) dist = 1e8f;
      if (best_i < 0 || dist < best) { best = dist; best_i = i; }
    }
    float* o = targets + m * D;
    if (best_i < 0) {  // no GT in this image
      labels[m] = d.background;
      for (int k = 0; k < D; ++k) o[k] = 0.f;
      ctr_t[m] = 0.f;
      continue;
    }
    const float* p = gt + (long long)best_i * D;
    labels[m] = best == 1e8f ? d.background : 0;
    const float bcx = centers ? centers[3 * best_i] : p[0], bcy = centers ? centers[3 * best_i + 1] : p[1];
    const float dx = px - bcx, dy = py - bcy;
    o[0] = dx / (float)s;  // root offsets are stored stride-normalised (das_head.py:547)
    o[1] = dy / (float)s;
    o[2] = centers ? centers[3 * best_i + 2] : p[2];
    for (int j = 0; j < J; ++j) {
      o[3 + 3 * j] = p[3 + 3 * j] - p[0];
      o[4 + 3 * j] = p[4 + 3 * j] - p[1];
      o[5 + 3 * j] = p[5 + 3 * j];
      o[3 + 3 * J + j] = p[3 + 3 * J + j];
    }
    ctr_t[m] = expf(-d.alpha * (sqrtf(dx * dx + dy * dy) / (1.414f * rad)));
    if (counts && best != 1e8f) {
      bool any_z = false;
      float v = 0.f;
      for (int j = 0; j < J; ++j) { any_z = any_z || p[5 + 3 * j] != 0.f; v += p[3 + 3 * J + j]; }
      n_pos += 1.f; n_3d += any_z ? 1.f : 0.f; n_vis += v;
    }
  }
  if (counts) {   // (block-uniform)
    const float a = block_sum(n_pos, sh), b = block_sum(n_3d, sh), c = block_sum(n_vis, sh);
    if (threadIdx.x == 0 && a != 0.f) { atomicAdd(counts, a); atomicAdd(counts + 1, b); atomicAdd(counts + 2, c); }
  }
}

// ------------------------------------------------------------------ the positives' rows of the pose losses
// What DASHead.loss derives from the ground truth for its positive locations (das_head.py:385-409), for `npos` rows listed in
// ascending order in `pos`: the pixel-to-joint targets (image offsets in units of the level's stride, depth in units of z_norm),
// the visibilities, the depth target, the centerness target, 2-D / 3-D kind, and the rank of every positive among the positives
// of its kind (its block of J rows in the flows' input). Kind, rank and the visibility sum need the rows in order: ONE
// workgroup walks them in chunks with a running count (a few thousand positives at most).
__global__ __launch_bounds__(1024) void positives_rank_kernel(const int* __restrict__ pos, int npos, const float* __restrict__ targets,
                                                              int J, int* __restrict__ is2d, int* __restrict__ slot,
                                                              float* __restrict__ nvis, float nvis_scale) {
  __shared__ int scan[1024];
  __shared__ float fsum[16];
  __shared__ int base2, base3;
  const int tid = threadIdx.x, D = 3 + 4 * J;
  if (tid == 0) { base2 = 0; base3 = 0; }
  float vis = 0.f;
  __syncthreads();
  for (int i0 = 0; i0 < npos; i0 += 1024) {
    const int i = i0 + tid;
    int k2 = 0;
    if (i < npos) {
      const float* t = targets + (long long)pos[i] * D;
      bool all0 = true;
      for (int j = 0; j < J; ++j) { all0 = all0 && t[5 + 3 * j] == 0.f; vis += t[3 + 3 * J + j]; }
      k2 = all0 ? 1 : 0;
    }
    scan[tid] = k2;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {   // inclusive scan of the 2-D flags of this chunk
      const int v = tid >= off ? scan[tid - off] : 0;
      __syncthreads();
      scan[tid] += v;
      __syncthreads();
    }
    if (i < npos) {
      const int r2 = scan[tid];                     // 2-D positives up to and including this one, within the chunk
      is2d[i] = k2;
      slot[i] = k2 ? base2 + r2 - 1 : base3 + (tid + 1 - r2) - 1;
    }
    __syncthreads();
    if (tid == 0) {
      const int n = min(1024, npos - i0), c2 = scan[n - 1];
      base2 += c2; base3 += n - c2;
    }
    __syncthreads();
  }
  vis = wave_sum(vis);
  if ((tid & 63) == 0) fsum[tid >> 6] = vis;
  __syncthreads();
  if (tid == 0) {
    float t = 0.f;
    for (int w = 0; w < 16; ++w) t += fsum[w];
    *nvis = t * nvis_scale;
  }
}
__global__ void positives_rows_kernel(const int* __restrict__ pos, int npos, const float* __restrict__ targets,
                                      const float* __restrict__ ctr_t, DasLevels lv, DasTargetDesc d, float z_norm,
                                      float depth_factor, float* __restrict__ real, float* __restrict__ vis,
                                      float* __restrict__ depth_t, float* __restrict__ ctr_pos) {
#pragma clang fp contract(off)
  const int J = d.J, D = 3 + 4 * J;
  const float inv_z = 1.f / z_norm;
  const long long n = (long long)npos * J;
  for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < n; e += (long long)gridDim.x * TPB) {
    const int i = (int)(e / J), j = (int)(e - (long long)i * J);
    const long long m = pos[i];
    const float* t = targets + m * D;
    const float ps = (float)d.stride[lv_geom(lv, m).l];
    // real = (gt_uvd - [root * stride, 0]) / [stride, stride, z_norm]: the reference's order of operations
    const float rx = t[0] * ps, ry = t[1] * ps;
    real[e * 3 + 0] = (t[3 + 3 * j] - rx) / ps;
    real[e * 3 + 1] = (t[4 + 3 * j] - ry) / ps;
    real[e * 3 + 2] = (t[5 + 3 * j] - 0.f) * inv_z;   // (torch divides a tensor by a host scalar as a product with its reciprocal)
    vis[e] = t[3 + 3 * J + j];
    if (j == 0) { depth_t[i] = t[2] * depth_factor; ctr_pos[i] = ctr_t[m]; }
  }
}

// ------------------------------------------------------------------ sigmoid focal loss (num_classes = 1)
// loss_i = BCE(x_i, t_i) * (alpha t + (1-alpha)(1-t)) * pt^gamma, t = (label == 0); writes dloss/dx per
// row and accumulates the sum.
__global__ void focal_kernel(const float* __restrict__ logit, int ps, const int* __restrict__ labels,
                             const float* __restrict__ weight, long long rows, float gamma, float alpha,
                             float* __restrict__ grad, float* __restrict__ sum) {
  __shared__ float sh[TPB / 64];
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < rows; i += (long long)gridDim.x * TPB) {
    const float x = logit[i * ps];
    const float t = labels[i] == 0 ? 1.f : 0.f;
    const float p = 1.f / (1.f + expf(-x));
    const float pt = (1.f - p) * t + p * (1.f - t);
    const float aw = alpha * t + (1.f - alpha) * (1.f - t);
    const float fw = aw * powf(pt, gamma);
    const float bce = fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
    const float wi = weight ? weight[i] : 1.f;   // mmdet's per-sample `weight` (weight_reduce_loss)
    acc += bce * fw * wi;
    const float dpt = p * (1.f - p) * (1.f - 2.f * t);
    const float dfw = aw * gamma * powf(pt, gamma - 1.f) * dpt;
    grad[i] = (dfw * bce + fw * (p - t)) * wi;
  }
  const float tot = block_sum(acc, sh);
  if (threadIdx.x == 0) atomicAdd(sum, tot);
}

// SmoothL1 (beta) and BCE-with-logits over n elements; per-element gradient + sum
__global__ void smooth_l1_kernel(const float* __restrict__ pred, const float* __restrict__ tgt,
                                 const float* __restrict__ weight, long long n, float beta, float* __restrict__ grad,
                                 float* __restrict__ sum) {
  __shared__ float sh[TPB / 64];
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long long)gridDim.x * TPB) {
    const float e = pred[i] - tgt[i], a = fabsf(e);
    const float wi = weight ? weight[i] : 1.f;
    acc += (a < beta ? 0.5f * a * a / beta : a - 0.5f * beta) * wi;
    grad[i] = (a < beta ? e / beta : (e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f))) * wi;
  }
  const float tot = block_sum(acc, sh);
  if (threadIdx.x == 0) atomicAdd(sum, tot);
}
__global__ void bce_logits_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                  const float* __restrict__ weight, long long n, float* __restrict__ grad,
                                  float* __restrict__ sum) {
  __shared__ float sh[TPB / 64];
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long long)gridDim.x * TPB) {
    const float v = x[i], tt = t[i];
    const float wi = weight ? weight[i] : 1.f;
    acc += (fmaxf(v, 0.f) - v * tt + log1pf(expf(-fabsf(v)))) * wi;
    grad[i] = (1.f / (1.f + expf(-v)) - tt) * wi;
  }
  const float tot = block_sum(acc, sh);
  if (threadIdx.x == 0) atomicAdd(sum, tot);
}

// ------------------------------------------------------------------ optimizer
// Sum of squares in two deterministic stages (per-workgroup partials, then one workgroup adds them in a fixed
// order): the clip coefficient derived from it must be bit-identical on every data-parallel rank, or the
// replicas' parameters drift apart by an ulp per step — float atomics would make the order launch-dependent.
constexpr int SUMSQ_MAX_BLOCKS = 2048;
__device__ float g_sumsq_partials[SUMSQ_MAX_BLOCKS];

__global__ void sumsq_kernel(const float* __restrict__ g, long long n, float* __restrict__ partials) {
  __shared__ float sh[TPB / 64];
  float acc = 0.f;
  const long long nv = n / 4;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < nv; i += (long long)gridDim.x * TPB) {
    const float4 v = reinterpret_cast<const float4*>(g)[i];
    acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = g[nv * 4 + threadIdx.x]; acc += v * v; }
  const float tot = block_sum(acc, sh);
  if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

__global__ void sumsq_final_kernel(const float* __restrict__ partials, int nblocks, float* __restrict__ out,
                                   int zero_first) {
  __shared__ float sh[TPB / 64];
  float acc = 0.f;
  for (int i = threadIdx.x; i < nblocks; i += TPB) acc += partials[i];
  const float tot = block_sum(acc, sh);
  if (threadIdx.x == 0) *out = (zero_first ? 0.f : *out) + tot;
}

// torch.optim.SGD (momentum, no dampening/nesterov) with the clip coefficient folded in:
//   g' = g * min(1, max_norm / (sqrt(sumsq) + 1e-6)) * grad_scale ; d = g' + wd*p ; buf = mom*buf + d ; p -= lr*buf
__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, long long n,
                           float lr, float momentum, float wd, float grad_scale, float max_norm,
                           const float* __restrict__ sumsq, int first_step) {
  float coef = grad_scale;
  if (max_norm > 0.f && sumsq) {
    const float norm = sqrtf(*sumsq) * grad_scale;
    coef *= fminf(1.f, max_norm / (norm + 1e-6f));
  }
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long long)gridDim.x * TPB) {
    const float d = g[i] * coef + wd * p[i];
    const float b = first_step ? d : momentum * buf[i] + d;
    buf[i] = b;
    p[i] -= lr * b;
  }
}
}  // namespace

extern "C" int das_assign_targets(const DasLevels* lv, const DasTargetDesc* d, const float* gt, const float* centers,
                                  const int* gt_start, int* labels, float* targets, float* centerness, float* counts,
                                  void* stream) {
  DAS_PROF(stream);
  if (!lv_valid(lv) || !d || !gt_start || !labels || !targets || !centerness || d->J < 1) return DAS_ERR_ARG;
  const long long rows = lv_total_rows(*lv);
  hipLaunchKernelGGL(assign_targets_kernel, dim3(grid_for(rows)), dim3(TPB), 0, (hipStream_t)stream, *lv, *d, gt,
                     centers, gt_start, labels, targets, centerness, rows, counts);
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}

extern "C" int das_positive_rows(const int* pos, int npos, const float* targets, const float* centerness, const DasLevels* lv,
                                 const DasTargetDesc* d, float z_norm, float depth_factor, float nvis_scale, float* real,
                                 float* vis, int* is2d, int* slot, float* depth_t, float* ctr_pos, float* nvis, void* stream) {
  DAS_PROF(stream);
  if (!pos || npos < 1 || !targets || !centerness || !lv_valid(lv) || !d || d->J < 1 || !real || !vis || !is2d || !slot ||
      !depth_t || !ctr_pos || !nvis)
    return DAS_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(positives_rank_kernel, dim3(1), dim3(1024), 0, s, pos, npos, targets, d->J, is2d, slot, nvis, nvis_scale);
  DAS_CHECK_LAUNCH();
  hipLaunchKernelGGL(positives_rows_kernel, dim3(grid_for((long long)npos * d->J)), dim3(TPB), 0, s, pos, npos, targets,
                     centerness, *lv, *d, z_norm, depth_factor, real, vis, depth_t, ctr_pos);
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}

extern "C" int das_sigmoid_focal_loss(const float* logits, int pix_stride, const int* labels, const float* weight,
                                      long long rows, float gamma, float alpha, float* grad, float* loss_sum,
                                      void* stream) {
  DAS_PROF(stream);
  if (!logits || !labels || !grad || !loss_sum || rows <= 0) return DAS_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(loss_sum, 0, sizeof(float), s) != hipSuccess) return DAS_ERR_LAUNCH;
  hipLaunchKernelGGL(focal_kernel, dim3(grid_for(rows, 1024)), dim3(TPB), 0, s, logits, pix_stride, labels, weight, rows,
                     gamma, alpha, grad, loss_sum);
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}

extern "C" int das_smooth_l1_loss(const float* pred, const float* target, const float* weight, long long n, float beta,
                                  float* grad, float* loss_sum, void* stream) {
  DAS_PROF(stream);
  if (!pred || !target || !grad || !loss_sum || n <= 0 || beta <= 0.f) return DAS_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(loss_sum, 0, sizeof(float), s) != hipSuccess) return DAS_ERR_LAUNCH;
  hipLaunchKernelGGL(smooth_l1_kernel, dim3(grid_for(n, 256)), dim3(TPB), 0, s, pred, target, weight, n, beta, grad,
                     loss_sum);
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}

extern "C" int das_bce_logits_loss(const float* logits, const float* target, const float* weight, long long n,
                                   float* grad, float* loss_sum, void* stream) {
  DAS_PROF(stream);
  if (!logits || !target || !grad || !loss_sum || n <= 0) return DAS_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(loss_sum, 0, sizeof(float), s) != hipSuccess) return DAS_ERR_LAUNCH;
  hipLaunchKernelGGL(bce_logits_kernel, dim3(grid_for(n, 256)), dim3(TPB), 0, s, logits, target, weight, n, grad,
                     loss_sum);
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}

extern "C" int das_grad_sumsq(const float* g, long long n, float* out, int zero_first, void* stream) {
  DAS_PROF(stream);
  if (!g || !out || n <= 0) return DAS_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  float* partials = nullptr;
  if (hipGetSymbolAddress((void**)&partials, HIP_SYMBOL(g_sumsq_partials)) != hipSuccess) return DAS_ERR_LAUNCH;
  const int blocks = grid_for(n / 4 + 1, SUMSQ_MAX_BLOCKS);
  hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(TPB), 0, s, g, n, partials);
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(TPB), 0, s, partials, blocks, out, zero_first);
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}

extern "C" int das_sgd_momentum_step(float* p, const float* g, float* buf, long long n, float lr, float momentum,
                                     float weight_decay, float grad_scale, float max_norm, const float* grad_sumsq,
                                     int first_step, void* stream) {
  DAS_PROF(stream);
  if (!p || !g || !buf || n <= 0) return DAS_ERR_ARG;
  hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n, 8192)), dim3(TPB), 0, (hipStream_t)stream, p, g, buf, n, lr, momentum,
                     weight_decay, grad_scale, max_norm, grad_sumsq, first_step);
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}
