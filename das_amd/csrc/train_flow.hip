// RealNVP log-density of the RLE pose loss (mmdet3d/models/losses/real_nvp.py:29-88 used by
// residual_log_likelihood_loss.py): 6 affine coupling layers, each with a t-net and an s-net
// (Linear(D,64) LeakyReLU Linear(64,64) LeakyReLU Linear(64,D) [Tanh for s]), D = 3 (or 2).
//   log_prob(x): for i = L-1..0:  z_ = m_i z;  s = snet_i(z_) (1-m_i);  t = tnet_i(z_) (1-m_i);
//                                 z = (1-m_i) (z - t) exp(-s) + z_;  logdet -= sum s
//                log p = -|z|^2/2 - D/2 log 2pi + logdet
// The reference runs this as ~1000 tiny torch kernels per step; here it is two kernels (forward, backward) over
// N = positives x joints rows. N is small (about 10^4): with one thread per row the launch is 176 waves on 1024 SIMDs,
// each walking 80 000 (forward) / 180 000 (backward) dependent instructions alone on its SIMD — 0.65 / 1.45 ms. So a
// row is spread over a GROUP OF 16 LANES (one DPP row): lane q owns hidden units 4q..4q+3 of both hidden layers.
//   layer 1 (D -> 64)   : 4 x D FMAs per lane
//   layer 2 (64 -> 64)  : the 64 inputs are gathered from the group (ds_bpermute), the lane's 4 x 64 weights come from
//                         LDS as one 16-byte read per input (W2 staged transposed: conflict-free)
//   layer 3 (64 -> D)   : 4 partial products per lane and output, summed over the group (4 xor-shuffles)
// A block is 1024 threads = 64 rows; a net's parameters are staged in LDS once per block. 16 x the waves and a
// sixteenth of the instructions per wave: latency is hidden by the other waves of the SIMD.
//   backward: the coupling layers are invertible, so nothing but the final z is saved — each layer's input is
//             recovered from its output while walking back. Per net the row phase above is re-run and back-propagated
//             (delta2, delta1 per lane: its 4 hidden units), alternating with weight phases in which the block's 64
//             rows of (delta, activation) pairs, parked in LDS, are reduced per weight: one atomic per weight per block.
// params per flow (f32): layer i = [t-net | s-net], net = W1[64][D] b1[64] W2[64][64] b2[64] W3[D][64] b3[D].
#include "common.h"
#include "prof.h"

namespace {
constexpr int FH = 64;          // hidden width
constexpr int FR = 256;         // job row ranges are aligned to this (python pads)
constexpr int GL = 16;          // lanes per row
constexpr int NT = 1024;        // threads per block
constexpr int RPB = NT / GL;    // rows per block
constexpr float SLOPE = 0.01f;  // nn.LeakyReLU default

template <int D>
struct Net {
  static constexpr int W1 = 0, B1 = W1 + FH * D, W2 = B1 + FH, B2 = W2 + FH * FH, W3 = B2 + FH, B3 = W3 + D * FH,
                       SIZE = B3 + D;
};
// LDS image of one net (floats): W1 rows padded to 4, W2 both ways (row-major for the backward's W2^T delta)
struct Lds {
  static constexpr int W1 = 0, B1 = W1 + FH * 4, W2T = B1 + FH, W2 = W2T + FH * FH, B2 = W2 + FH * FH, W3 = B2 + FH,
                       B3 = W3 + 4 * FH, SIZE = B3 + 4;
};

__device__ __forceinline__ float lrelu(float v) { return v > 0.f ? v : SLOPE * v; }

// Several flows (e.g. the `flow3d` and `flow3d_update` of the head) run in ONE launch: each job owns a
// block-aligned range of rows and brings its own parameters / gradient destinations.
struct FlowJobs {
  int n;
  DasFlowJob j[DAS_FLOW_MAX_JOBS];
};
__device__ __forceinline__ DasFlowJob job_of_block(const FlowJobs& jobs) {
  const int row0 = blockIdx.x * RPB;
  int ji = 0;
  for (int q = 1; q < DAS_FLOW_MAX_JOBS; ++q)
    if (q < jobs.n && row0 >= jobs.j[q].row_start) ji = q;
  return jobs.j[ji];
}

// value of lane `src` (0..15) of this lane's group
__device__ __forceinline__ float grp(float v, int src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(((threadIdx.x & ~(GL - 1) & 63) + src) << 2,
                                                                __builtin_bit_cast(int, v)));
}
__device__ __forceinline__ float grp_sum(float v) {
#pragma unroll
  for (int m = 8; m >= 1; m >>= 1)
    v += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(((threadIdx.x & 63) ^ m) << 2, __builtin_bit_cast(int, v)));
  return v;
}

// global -> LDS image of one net (all threads of the block; callers put barriers around it)
template <int D>
__device__ __forceinline__ void stage_net(const float* __restrict__ w, float* __restrict__ sw, bool with_rowmajor) {
  const int t = threadIdx.x;
  {   // W2: thread t brings W2[o][i0..i0+3], o = t >> 4, i0 = (t & 15) * 4
    const int o = t >> 4, i0 = (t & 15) * 4;
    const float4 v = *reinterpret_cast<const float4*>(w + Net<D>::W2 + o * FH + i0);
    sw[Lds::W2T + (i0 + 0) * FH + o] = v.x; sw[Lds::W2T + (i0 + 1) * FH + o] = v.y;
    sw[Lds::W2T + (i0 + 2) * FH + o] = v.z; sw[Lds::W2T + (i0 + 3) * FH + o] = v.w;
    if (with_rowmajor) *reinterpret_cast<float4*>(sw + Lds::W2 + o * FH + i0) = v;
  }
  if (t < FH) {
#pragma unroll
    for (int d = 0; d < 4; ++d) sw[Lds::W1 + t * 4 + d] = d < D ? w[Net<D>::W1 + t * D + (d < D ? d : 0)] : 0.f;
    sw[Lds::B1 + t] = w[Net<D>::B1 + t];
    sw[Lds::B2 + t] = w[Net<D>::B2 + t];
#pragma unroll
    for (int c = 0; c < D; ++c) sw[Lds::W3 + c * FH + t] = w[Net<D>::W3 + c * FH + t];
    if (t < D) sw[Lds::B3 + t] = w[Net<D>::B3 + t];
  }
}

// One row per group of 16 lanes; lane q holds hidden units 4q..4q+3. zin: the masked input (the same in all lanes of
// the group). Returns h1[4], h2[4] (this lane's units) and out[D] (replicated).
template <int D>
__device__ __forceinline__ void mlp_forward(const float* __restrict__ sw, const float* zin, float* h1, float* h2,
                                            float* out) {
  const int q = threadIdx.x & (GL - 1);
#pragma unroll
  for (int oo = 0; oo < 4; ++oo) {
    const float4 w1 = *reinterpret_cast<const float4*>(sw + Lds::W1 + (4 * q + oo) * 4);
    float a = sw[Lds::B1 + 4 * q + oo];
    a += w1.x * zin[0];
    a += w1.y * zin[1];
    if (D > 2) a += w1.z * zin[D > 2 ? 2 : 0];
    h1[oo] = lrelu(a);
  }
  const float4 b2 = *reinterpret_cast<const float4*>(sw + Lds::B2 + 4 * q);
  float acc[4] = {b2.x, b2.y, b2.z, b2.w};
#pragma unroll 2
  for (int s = 0; s < GL; ++s) {   // (not fully unrolled: the compiler would hoist all 64 LDS reads and spill)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const float hv = grp(h1[v], s);
      const float4 w = *reinterpret_cast<const float4*>(sw + Lds::W2T + (4 * s + v) * FH + 4 * q);
      acc[0] += w.x * hv; acc[1] += w.y * hv; acc[2] += w.z * hv; acc[3] += w.w * hv;
    }
  }
#pragma unroll
  for (int oo = 0; oo < 4; ++oo) h2[oo] = lrelu(acc[oo]);
#pragma unroll
  for (int c = 0; c < D; ++c) {
    const float4 w3 = *reinterpret_cast<const float4*>(sw + Lds::W3 + c * FH + 4 * q);
    out[c] = grp_sum(w3.x * h2[0] + w3.y * h2[1] + w3.z * h2[2] + w3.w * h2[3]) + sw[Lds::B3 + c];
  }
}

// ------------------------------------------------------------------ forward
template <int D>
__global__ __launch_bounds__(NT) void realnvp_fwd_kernel(const float* __restrict__ x, FlowJobs jobs, int layers,
                                                         unsigned mask_bits, float* __restrict__ logp,
                                                         float* __restrict__ zout) {
  __shared__ __attribute__((aligned(16))) float sw[Lds::SIZE];
  const DasFlowJob job = job_of_block(jobs);
  const float* __restrict__ params = job.params;
  const int r = blockIdx.x * RPB + (threadIdx.x >> 4);
  const bool live = r >= job.row_start && r < job.row_end;
  float z[D], logdet = 0.f;
#pragma unroll
  for (int d = 0; d < D; ++d) z[d] = live ? x[(size_t)r * D + d] : 0.f;
  for (int i = layers - 1; i >= 0; --i) {
    const unsigned m = (mask_bits >> (i * D)) & ((1u << D) - 1u);
    float zin[D], s[D], t[D];
#pragma unroll
    for (int d = 0; d < D; ++d) zin[d] = ((m >> d) & 1u) ? z[d] : 0.f;
    for (int which = 1; which >= 0; --which) {   // s-net, then t-net
      float h1[4], h2[4], out[D];
      __syncthreads();   // the previous net's image is no longer read
      stage_net<D>(params + (size_t)(i * 2 + which) * Net<D>::SIZE, sw, false);
      __syncthreads();
      mlp_forward<D>(sw, zin, h1, h2, out);
#pragma unroll
      for (int d = 0; d < D; ++d) {
        if (which == 1) s[d] = out[d]; else t[d] = out[d];
      }
    }
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (!((m >> d) & 1u)) {
        const float sd = tanhf(s[d]);
        z[d] = (z[d] - t[d]) * expf(-sd);
        logdet -= sd;
      }
    }
  }
  if (live && (threadIdx.x & (GL - 1)) == 0) {
    float q = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) { q += z[d] * z[d]; zout[(size_t)r * D + d] = z[d]; }
    logp[r] = -0.5f * q - 0.5f * D * 1.8378770664093453f + logdet;
  }
}

// ------------------------------------------------------------------ backward
template <int D>
__global__ __launch_bounds__(NT) void realnvp_bwd_kernel(const float* __restrict__ zfin, const float* __restrict__ glogp,
                                                         FlowJobs jobs, int layers, unsigned mask_bits,
                                                         float* __restrict__ dx) {
  const DasFlowJob job = job_of_block(jobs);
  const float* __restrict__ params = job.params;
  float* __restrict__ dparams = job.dparams;
  float* const* __restrict__ dst_table = job.dst_table;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* sw = sm;                         // Lds image of the net
  float* bufA = sw + Lds::SIZE;           // [RPB][FH] activations (h2, then h1, then delta1)
  float* bufD = bufA + RPB * FH;          // [RPB][FH] delta2
  float* sm3 = bufD + RPB * FH;           // [RPB][4] delta3 rows
  float* smz = sm3 + RPB * 4;             // [RPB][4] masked inputs
  const int tid = threadIdx.x, q = tid & (GL - 1), rg = tid >> 4;
  const int r = blockIdx.x * RPB + rg;
  const bool live = r >= job.row_start && r < job.row_end;
  const float g = live ? glogp[r] : 0.f;
  float z[D], dz[D];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    z[d] = live ? zfin[(size_t)r * D + d] : 0.f;
    dz[d] = -g * z[d];                    // prior: d(-|z|^2/2)
  }
  for (int i = 0; i < layers; ++i) {      // forward went L-1..0, so walk 0..L-1
    const unsigned m = (mask_bits >> (i * D)) & ((1u << D) - 1u);
    float zin[D], tval[D];
#pragma unroll
    for (int d = 0; d < D; ++d) zin[d] = ((m >> d) & 1u) ? z[d] : 0.f;
    float dzin_m[D];                      // gradient reaching the masked inputs through the two nets
#pragma unroll
    for (int d = 0; d < D; ++d) dzin_m[d] = 0.f;
    float sd[D], es[D];
    // two passes: s-net first (its output is needed to invert the layer), then t-net
    for (int which = 1; which >= 0; --which) {
      float h1[4], h2[4], out[D], d3[D];
      __syncthreads();   // the previous net's weight phase has finished reading the image / bufA / smz
      stage_net<D>(params + (size_t)(i * 2 + which) * Net<D>::SIZE, sw, true);
      __syncthreads();
      mlp_forward<D>(sw, zin, h1, h2, out);
      if (which == 1) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
          sd[d] = ((m >> d) & 1u) ? 0.f : tanhf(out[d]);
          es[d] = expf(-sd[d]);
          // ds = -dz * z_out - g on the transformed dims; through tanh
          d3[d] = ((m >> d) & 1u) ? 0.f : (-dz[d] * z[d] - g) * (1.f - sd[d] * sd[d]);
        }
      } else {
#pragma unroll
        for (int d = 0; d < D; ++d) {
          tval[d] = out[d];
          d3[d] = ((m >> d) & 1u) ? 0.f : -dz[d] * es[d];
        }
      }
      if (!live) {
#pragma unroll
        for (int d = 0; d < D; ++d) d3[d] = 0.f;
      }
      // gradient destinations of this net's six tensors: slices of dparams, or the caller's table of
      // pointers (the optimizer's flat gradient buffer: accumulate in place)
      float* gp = dparams ? dparams + (size_t)(i * 2 + which) * Net<D>::SIZE : nullptr;
      float* const* tb = dst_table ? dst_table + (i * 2 + which) * 6 : nullptr;
      float* gW1 = tb ? tb[0] : gp + Net<D>::W1;
      float* gB1 = tb ? tb[1] : gp + Net<D>::B1;
      float* gW2 = tb ? tb[2] : gp + Net<D>::W2;
      float* gB2 = tb ? tb[3] : gp + Net<D>::B2;
      float* gW3 = tb ? tb[4] : gp + Net<D>::W3;
      float* gB3 = tb ? tb[5] : gp + Net<D>::B3;
      // delta2 = (W3^T d3) * lrelu'(a2)   (sign(a2) = sign(h2)); delta1 = (W2^T delta2) * lrelu'(a1)
      float d2[4], d1[4];
#pragma unroll
      for (int oo = 0; oo < 4; ++oo) {
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < D; ++c) a += sw[Lds::W3 + c * FH + 4 * q + oo] * d3[c];
        d2[oo] = a * (h2[oo] > 0.f ? 1.f : SLOPE);
        d1[oo] = 0.f;
      }
#pragma unroll 2
      for (int s = 0; s < GL; ++s) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const float dv = grp(d2[v], s);   // delta2 of hidden unit o = 4 s + v
          const float4 w = *reinterpret_cast<const float4*>(sw + Lds::W2 + (4 * s + v) * FH + 4 * q);
          d1[0] += w.x * dv; d1[1] += w.y * dv; d1[2] += w.z * dv; d1[3] += w.w * dv;
        }
      }
#pragma unroll
      for (int oo = 0; oo < 4; ++oo) d1[oo] *= h1[oo] > 0.f ? 1.f : SLOPE;
      // gradient wrt the masked inputs: W1^T delta1, summed over the group
#pragma unroll
      for (int d = 0; d < D; ++d) {
        if ((m >> d) & 1u) {
          float a = 0.f;
#pragma unroll
          for (int oo = 0; oo < 4; ++oo) a += sw[Lds::W1 + (4 * q + oo) * 4 + d] * d1[oo];
          dzin_m[d] += grp_sum(a);
        }
      }
      // ---- weight phases. Layer 3: dW3 = sum_r d3 (x) h2; layer 2: dW2 = sum_r delta2 (x) h1
      *reinterpret_cast<float4*>(bufA + rg * FH + 4 * q) = make_float4(h2[0], h2[1], h2[2], h2[3]);
      *reinterpret_cast<float4*>(bufD + rg * FH + 4 * q) = make_float4(d2[0], d2[1], d2[2], d2[3]);
      if (q < 4) {
        sm3[rg * 4 + q] = q < D ? d3[q < D ? q : 0] : 0.f;
        smz[rg * 4 + q] = q < D ? zin[q < D ? q : 0] : 0.f;
      }
      __syncthreads();
      if (tid < D * FH) {
        const int c = tid / FH, k = tid - c * FH;
        float acc = 0.f;
        for (int rr = 0; rr < RPB; ++rr) acc += sm3[rr * 4 + c] * bufA[rr * FH + k];
        atomicAdd(gW3 + c * FH + k, acc);
      } else if (tid < D * FH + D) {
        const int c = tid - D * FH;
        float acc = 0.f;
        for (int rr = 0; rr < RPB; ++rr) acc += sm3[rr * 4 + c];
        atomicAdd(gB3 + c, acc);
      }
      __syncthreads();   // bufA (h2) no longer read
      *reinterpret_cast<float4*>(bufA + rg * FH + 4 * q) = make_float4(h1[0], h1[1], h1[2], h1[3]);
      __syncthreads();
      {   // thread t: dW2[o][i0..i0+3], o = t >> 4, i0 = (t & 15) * 4
        const int o = tid >> 4, i0 = (tid & 15) * 4;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, bs = 0.f;
#pragma unroll 4
        for (int rr = 0; rr < RPB; ++rr) {
          const float dv = bufD[rr * FH + o];
          const float4 h = *reinterpret_cast<const float4*>(bufA + rr * FH + i0);
          a0 += dv * h.x; a1 += dv * h.y; a2 += dv * h.z; a3 += dv * h.w;
          bs += dv;
        }
        atomicAdd(gW2 + o * FH + i0, a0); atomicAdd(gW2 + o * FH + i0 + 1, a1);
        atomicAdd(gW2 + o * FH + i0 + 2, a2); atomicAdd(gW2 + o * FH + i0 + 3, a3);
        if ((tid & 15) == 0) atomicAdd(gB2 + o, bs);
      }
      __syncthreads();   // bufA (h1) no longer read
      // ---- layer 1: dW1 = sum_r delta1 (x) z_
      *reinterpret_cast<float4*>(bufA + rg * FH + 4 * q) = make_float4(d1[0], d1[1], d1[2], d1[3]);
      __syncthreads();
      if (tid < FH * D) {
        const int k = tid / D, d = tid - k * D;
        float acc = 0.f;
        for (int rr = 0; rr < RPB; ++rr) acc += bufA[rr * FH + k] * smz[rr * 4 + d];
        atomicAdd(gW1 + k * D + d, acc);
      } else if (tid < FH * D + FH) {
        const int k = tid - FH * D;
        float acc = 0.f;
        for (int rr = 0; rr < RPB; ++rr) acc += bufA[rr * FH + k];
        atomicAdd(gB1 + k, acc);
      }
    }
    // invert the layer and pass the gradient to its input
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if ((m >> d) & 1u) {
        dz[d] += dzin_m[d];
      } else {
        z[d] = z[d] / es[d] + tval[d];   // z_in = z_out * exp(s) + t
        dz[d] = dz[d] * es[d];
      }
    }
  }
  if (live && q == 0) {
#pragma unroll
    for (int d = 0; d < D; ++d) dx[(size_t)r * D + d] = dz[d];
  }
}

template <int D>
constexpr size_t bwd_smem() {
  return (size_t)(Lds::SIZE + 2 * RPB * FH + 2 * RPB * 4) * sizeof(float);
}

}  // namespace

static bool jobs_ok(const DasFlowJob* jobs, int njobs, int rows_total, bool backward) {
  if (!jobs || njobs < 1 || njobs > DAS_FLOW_MAX_JOBS) return false;
  for (int q = 0; q < njobs; ++q) {
    const DasFlowJob& j = jobs[q];
    if (!j.params || j.row_start % FR || j.row_start < 0 || j.row_end < j.row_start || j.row_end > rows_total) return false;
    if (q > 0 && j.row_start < jobs[q - 1].row_end) return false;
    if (backward && (!j.dparams == !j.dst_table)) return false;
  }
  return jobs[0].row_start == 0;
}

extern "C" int das_realnvp_log_prob_multi(const float* x, int rows_total, int D, const DasFlowJob* jobs, int njobs,
                                          int layers, unsigned mask_bits, float* logp, float* z_out, void* stream) {
  DAS_PROF(stream);
  if (!x || !logp || !z_out || rows_total < 1 || layers < 1 || layers * D > 32 || (D != 2 && D != 3)) return DAS_ERR_ARG;
  if (!jobs_ok(jobs, njobs, rows_total, false)) return DAS_ERR_ARG;
  FlowJobs fj;
  fj.n = njobs;
  for (int q = 0; q < DAS_FLOW_MAX_JOBS; ++q) fj.j[q] = jobs[q < njobs ? q : 0];
  const int blocks = (rows_total + RPB - 1) / RPB;
  if (D == 3) {
    hipLaunchKernelGGL(realnvp_fwd_kernel<3>, dim3(blocks), dim3(NT), 0, (hipStream_t)stream, x, fj, layers, mask_bits,
                       logp, z_out);
  } else {
    hipLaunchKernelGGL(realnvp_fwd_kernel<2>, dim3(blocks), dim3(NT), 0, (hipStream_t)stream, x, fj, layers, mask_bits,
                       logp, z_out);
  }
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}

extern "C" int das_realnvp_log_prob_multi_backward(const float* z_final, const float* grad_logp, int rows_total, int D,
                                                   const DasFlowJob* jobs, int njobs, int layers, unsigned mask_bits,
                                                   float* dx, void* stream) {
  DAS_PROF(stream);
  if (!z_final || !grad_logp || !dx || rows_total < 1 || layers < 1 || layers * D > 32 || (D != 2 && D != 3))
    return DAS_ERR_ARG;
  if (!jobs_ok(jobs, njobs, rows_total, true)) return DAS_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  FlowJobs fj;
  fj.n = njobs;
  for (int q = 0; q < DAS_FLOW_MAX_JOBS; ++q) fj.j[q] = jobs[q < njobs ? q : 0];
  const int blocks = (rows_total + RPB - 1) / RPB;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)realnvp_bwd_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)bwd_smem<3>());
    (void)hipFuncSetAttribute((const void*)realnvp_bwd_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)bwd_smem<2>());
    attr_set = true;
  }
  const size_t psize = sizeof(float) * 2 * layers * (D == 3 ? Net<3>::SIZE : Net<2>::SIZE);
  for (int q = 0; q < njobs; ++q)
    if (jobs[q].dparams && hipMemsetAsync(jobs[q].dparams, 0, psize, s) != hipSuccess) return DAS_ERR_LAUNCH;
  if (D == 3) {
    hipLaunchKernelGGL(realnvp_bwd_kernel<3>, dim3(blocks), dim3(NT), bwd_smem<3>(), s, z_final, grad_logp, fj, layers,
                       mask_bits, dx);
  } else {
    hipLaunchKernelGGL(realnvp_bwd_kernel<2>, dim3(blocks), dim3(NT), bwd_smem<2>(), s, z_final, grad_logp, fj, layers,
                       mask_bits, dx);
  }
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}

extern "C" int das_realnvp_log_prob(const float* x, int N, int D, const float* params, int layers, unsigned mask_bits,
                                    float* logp, float* z_out, void* stream) {
  DAS_PROF(stream);
  DasFlowJob j;
  j.params = params; j.dparams = nullptr; j.dst_table = nullptr; j.row_start = 0; j.row_end = N;
  return das_realnvp_log_prob_multi(x, N, D, &j, 1, layers, mask_bits, logp, z_out, stream);
}

extern "C" int das_realnvp_log_prob_backward(const float* z_final, const float* grad_logp, int N, int D,
                                             const float* params, int layers, unsigned mask_bits, float* dx,
                                             float* dparams, float* const* dst_table, void* stream) {
  DAS_PROF(stream);
  DasFlowJob j;
  j.params = params; j.dparams = dparams; j.dst_table = dst_table; j.row_start = 0; j.row_end = N;
  return das_realnvp_log_prob_multi_backward(z_final, grad_logp, N, D, &j, 1, layers, mask_bits, dx, stream);
}
