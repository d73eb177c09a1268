// The arithmetic of the RLE pose loss AROUND the normalising flows, and the depth term, on the positive locations
// (mmdet3d/models/pose_heads/das_head.py:375-466, mmdet3d/models/losses/residual_log_likelihood_loss.py:17-37).
// Three kernels bracket the two RealNVP launches (train_flow.hip):
//   rle_prepare_kernel   gathers the positive rows of the head outputs, applies the 2-D-sample rules (z offset := 0,
//                        sigma logit := 1), forms bar_mu = (pred - gt) / sigma and writes it in the row order the flow
//                        kernels read (2-D samples -> x2, 3-D samples -> x3, one block of rows per prediction set),
//                        together with every row's weight -3 * vis (d loss / d log_phi up to the scalar upstream);
//   rle_loss_kernel      sum over (positive, joint, set, dim) of vis * (log sigma - log_phi) + vis * logQ and the
//                        smooth-L1 sum of the depth term, as per-workgroup partials (summed in a fixed order by the
//                        caller: the loss value does not depend on the launch);
//   rle_backward_kernel  d / d pose_pred, d / d refined offsets from the flows' dx and the closed-form terms; a
//                        positive row is written by exactly one thread per column (no atomics, deterministic).
// One thread per (positive, joint). Everything is f32, as in the reference.
#include "common.h"
#include "prof.h"

namespace {

constexpr int TPB = 256;
constexpr float SQRT2 = 1.41421356237309504880f;

struct Item {
  long long row;   // row of pose / aux
  int p, j, two_d, slot;
};
__device__ __forceinline__ Item item_of(int i, const DasRleDesc& d, const long long* __restrict__ pos,
                                        const int* __restrict__ is2d, const int* __restrict__ slot) {
  Item it;
  it.p = i / d.J;
  it.j = i - it.p * d.J;
  it.row = pos[it.p];
  it.two_d = is2d[it.p];
  it.slot = slot[it.p];
  return it;
}
// sigma (k) of a positive's joint: sigmoid of the logit (1 for the z column of a 2-D sample) + 1e-9
__device__ __forceinline__ float sigma_of(const float* __restrict__ pose_row, const DasRleDesc& d, int j, int k, int two_d,
                                          float* s_out = nullptr) {
  const float logit = (two_d && k == 2) ? 1.f : pose_row[3 + 3 * d.J + 3 * j + k];
  const float s = 1.f / (1.f + expf(-logit));
  if (s_out) *s_out = s;
  return s + 1e-9f;
}
// prediction of set `set`: 0 = refined offsets (aux), 1 = the head's own uvd; z of a 2-D sample is 0
__device__ __forceinline__ float pred_of(const float* __restrict__ pose_row, const float* __restrict__ aux_row,
                                         const DasRleDesc& d, int set, int j, int k, int two_d) {
  if (two_d && k == 2) return 0.f;
  return set == 0 ? aux_row[3 * j + k] : pose_row[3 + 3 * j + k];
}

__global__ void rle_prepare_kernel(const float* __restrict__ pose, const float* __restrict__ aux,
                                   const long long* __restrict__ pos, const float* __restrict__ real,
                                   const float* __restrict__ vis, const int* __restrict__ is2d,
                                   const int* __restrict__ slot, DasRleDesc d, float* __restrict__ x2,
                                   float* __restrict__ w2, float* __restrict__ x3, float* __restrict__ w3) {
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i >= d.npos * d.J) return;
  const Item it = item_of(i, d, pos, is2d, slot);
  const float* pr = pose + it.row * d.pose_ps;
  const float* ar = aux + it.row * d.aux_ps;
  const float* rl = real + (size_t)i * 3;
  const float w = -3.f * vis[i];
  const int D = it.two_d ? 2 : 3;
  float sg[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) sg[k] = sigma_of(pr, d, it.j, k, it.two_d);
  for (int set = 0; set < d.sets; ++set) {
    const size_t r = (size_t)set * (it.two_d ? d.stride2 : d.stride3) + (size_t)it.slot * d.J + it.j;
    float* x = it.two_d ? x2 + r * 2 : x3 + r * 3;
    for (int k = 0; k < D; ++k) x[k] = (pred_of(pr, ar, d, set, it.j, k, it.two_d) - rl[k]) / sg[k];
    (it.two_d ? w2 : w3)[r] = w;
  }
}

__device__ __forceinline__ float block_sum(float v, float* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) sh[wave] = v;
  __syncthreads();
  float t = 0.f;
  if (threadIdx.x == 0)
    for (int w = 0; w < TPB / 64; ++w) t += sh[w];
  __syncthreads();
  return t;
}

__global__ void rle_loss_kernel(const float* __restrict__ pose, const float* __restrict__ aux,
                                const long long* __restrict__ pos, const float* __restrict__ real,
                                const float* __restrict__ vis, const int* __restrict__ is2d, const int* __restrict__ slot,
                                const float* __restrict__ depth_t, const float* __restrict__ logp2,
                                const float* __restrict__ logp3, DasRleDesc d, float* __restrict__ partials) {
  __shared__ float sh[TPB / 64];
  const int i = blockIdx.x * TPB + threadIdx.x;
  float acc = 0.f, dacc = 0.f;
  if (i < d.npos * d.J) {
    const Item it = item_of(i, d, pos, is2d, slot);
    const float* pr = pose + it.row * d.pose_ps;
    const float* ar = aux + it.row * d.aux_ps;
    const float* rl = real + (size_t)i * 3;
    const float v = vis[i];
    for (int set = 0; set < d.sets; ++set) {
      const size_t r = (size_t)set * (it.two_d ? d.stride2 : d.stride3) + (size_t)it.slot * d.J + it.j;
      const float lphi = it.two_d ? logp2[r] : logp3[r];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float sg = sigma_of(pr, d, it.j, k, it.two_d);
        const float pd = pred_of(pr, ar, d, set, it.j, k, it.two_d);
        const float nf = logf(sg) - lphi;
        const float q = logf(sg / d.amp) + fabsf(rl[k] - pd) / (SQRT2 * sg + 1e-9f);
        acc += nf * v + q * v;
      }
    }
    if (it.j == 0 && !it.two_d) {   // depth term (das_head.py:375-379): smooth L1 of the root depth of 3-D samples
      const float e = pr[2] - depth_t[it.p], a = fabsf(e);
      dacc = a < d.beta ? 0.5f * a * a / d.beta : a - 0.5f * d.beta;
    }
  }
  const float t0 = block_sum(acc, sh), t1 = block_sum(dacc, sh);
  if (threadIdx.x == 0) {
    partials[2 * blockIdx.x] = t0;
    partials[2 * blockIdx.x + 1] = t1;
  }
}

__global__ void rle_backward_kernel(const float* __restrict__ pose, const float* __restrict__ aux,
                                    const long long* __restrict__ pos, const float* __restrict__ real,
                                    const float* __restrict__ vis, const int* __restrict__ is2d,
                                    const int* __restrict__ slot, const float* __restrict__ depth_t,
                                    const float* __restrict__ dx2, const float* __restrict__ dx3,
                                    const float* __restrict__ g_sums, DasRleDesc d, float* __restrict__ dpose,
                                    float* __restrict__ daux) {
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i >= d.npos * d.J) return;
  const Item it = item_of(i, d, pos, is2d, slot);
  const float* pr = pose + it.row * d.pose_ps;
  const float* ar = aux + it.row * d.aux_ps;
  const float* rl = real + (size_t)i * 3;
  const float g0 = g_sums[0], gv = g0 * vis[i];
  const int D = it.two_d ? 2 : 3;
  float* dpr = dpose + it.row * d.pose_ps;
  float* dar = daux + it.row * d.aux_ps;
  for (int k = 0; k < D; ++k) {   // (the z column of a 2-D sample is cut off from the predictions: no gradient)
    float s;
    const float sg = sigma_of(pr, d, it.j, k, it.two_d, &s);
    const float den = SQRT2 * sg + 1e-9f;
    float dsig = 0.f;
    for (int set = 0; set < d.sets; ++set) {
      const size_t r = (size_t)set * (it.two_d ? d.stride2 : d.stride3) + (size_t)it.slot * d.J + it.j;
      const float gbar = it.two_d ? dx2[r * 2 + k] : dx3[r * 3 + k];
      const float pd = pred_of(pr, ar, d, set, it.j, k, it.two_d);
      const float e = pd - rl[k];
      const float sgn = e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f);
      const float dpred = gbar / sg + gv * sgn / den;
      // d/d sigma of: bar = e / sigma (through the flow), vis * log sigma (nf), vis * log(sigma / amp) and vis * |e| / den (logQ)
      dsig += -gbar * e / (sg * sg) + gv * (2.f / sg - fabsf(e) * SQRT2 / (den * den));
      if (set == 0) dar[3 * it.j + k] = dpred;
      else dpr[3 + 3 * it.j + k] = dpred;
    }
    dpr[3 + 3 * d.J + 3 * it.j + k] = dsig * s * (1.f - s);
  }
  if (it.j == 0 && !it.two_d) {
    const float e = pr[2] - depth_t[it.p], a = fabsf(e);
    dpr[2] = g_sums[1] * (a < d.beta ? e / d.beta : (e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f)));
  }
}

bool desc_ok(const DasRleDesc* d) {
  return d && d->J >= 1 && (d->sets == 1 || d->sets == 2) && d->npos >= 1 && d->pose_ps >= 3 + 6 * d->J &&
         d->aux_ps >= 3 * d->J && d->stride2 >= 0 && d->stride3 >= 0 && d->amp > 0.f && d->beta > 0.f;
}
int blocks_of(const DasRleDesc* d) { return (d->npos * d->J + TPB - 1) / TPB; }

}  // namespace

extern "C" int das_rle_blocks(const DasRleDesc* d) { return desc_ok(d) ? blocks_of(d) : 0; }

extern "C" int das_rle_prepare(const float* pose, const float* aux, const long long* pos, const float* real,
                               const float* vis, const int* is2d, const int* slot, const DasRleDesc* d, float* x2,
                               float* w2, float* x3, float* w3, void* stream) {
  DAS_PROF(stream);
  if (!pose || !aux || !pos || !real || !vis || !is2d || !slot || !desc_ok(d)) return DAS_ERR_ARG;
  if ((d->stride2 > 0 && (!x2 || !w2)) || (d->stride3 > 0 && (!x3 || !w3))) return DAS_ERR_ARG;
  hipLaunchKernelGGL(rle_prepare_kernel, dim3(blocks_of(d)), dim3(TPB), 0, (hipStream_t)stream, pose, aux, pos, real, vis,
                     is2d, slot, *d, x2, w2, x3, w3);
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}

extern "C" int das_rle_loss(const float* pose, const float* aux, const long long* pos, const float* real,
                            const float* vis, const int* is2d, const int* slot, const float* depth_t,
                            const float* logp2, const float* logp3, const DasRleDesc* d, float* partials, void* stream) {
  DAS_PROF(stream);
  if (!pose || !aux || !pos || !real || !vis || !is2d || !slot || !depth_t || !partials || !desc_ok(d)) return DAS_ERR_ARG;
  if ((d->stride2 > 0 && !logp2) || (d->stride3 > 0 && !logp3)) return DAS_ERR_ARG;
  hipLaunchKernelGGL(rle_loss_kernel, dim3(blocks_of(d)), dim3(TPB), 0, (hipStream_t)stream, pose, aux, pos, real, vis,
                     is2d, slot, depth_t, logp2, logp3, *d, partials);
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}

extern "C" int das_rle_backward(const float* pose, const float* aux, const long long* pos, const float* real,
                                const float* vis, const int* is2d, const int* slot, const float* depth_t,
                                const float* dx2, const float* dx3, const float* g_sums, const DasRleDesc* d,
                                float* dpose, float* daux, void* stream) {
  DAS_PROF(stream);
  if (!pose || !aux || !pos || !real || !vis || !is2d || !slot || !depth_t || !g_sums || !dpose || !daux || !desc_ok(d))
    return DAS_ERR_ARG;
  if ((d->stride2 > 0 && !dx2) || (d->stride3 > 0 && !dx3)) return DAS_ERR_ARG;
  hipLaunchKernelGGL(rle_backward_kernel, dim3(blocks_of(d)), dim3(TPB), 0, (hipStream_t)stream, pose, aux, pos, real,
                     vis, is2d, slot, depth_t, dx2, dx3, g_sums, *d, dpose, daux);
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}
