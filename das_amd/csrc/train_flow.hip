// RealNVP log-density of the RLE pose loss (mmdet3d/models/losses/real_nvp.py:29-88 used by
// residual_log_likelihood_loss.py): 6 affine coupling layers, each with a t-net and an s-net
// (Linear(D,64) LeakyReLU Linear(64,64) LeakyReLU Linear(64,D) [Tanh for s]), D = 3 (or 2).
//   log_prob(x): for i = L-1..0:  z_ = m_i z;  s = snet_i(z_) (1-m_i);  t = tnet_i(z_) (1-m_i);
//                                 z = (1-m_i) (z - t) exp(-s) + z_;  logdet -= sum s
//                log p = -|z|^2/2 - D/2 log 2pi + logdet
// The reference runs this as ~1000 tiny torch kernels per step; here it is two kernels (forward, backward)
// over N = positives x joints rows:
//   forward : one thread per row, activations in registers; every lane uses the same weights, so they are
//             fetched with scalar loads (wave-uniform addresses) and the FMAs read them from SGPRs.
//   backward: the coupling layers are invertible, so nothing but the final z is saved — each layer's input is
//             recovered from its output while walking back. Per net: a row phase (thread = row: recompute the
//             MLP, back-propagate) alternates with a weight phase (thread = 16 weights of W2, or one weight of
//             W1 / W3: reduce  sum_r delta[r] (x) h[r]  over the block's 256 rows from LDS) — no per-row
//             atomics, one atomic per weight per block.
// params per flow (f32): layer i = [t-net | s-net], net = W1[64][D] b1[64] W2[64][64] b2[64] W3[D][64] b3[D].
#include "common.h"

namespace {
constexpr int FH = 64;        // hidden width
constexpr int FR = 256;       // rows per block = threads per block
constexpr float SLOPE = 0.01f;  // nn.LeakyReLU default

template <int D>
struct Net {
  static constexpr int W1 = 0, B1 = W1 + FH * D, W2 = B1 + FH, B2 = W2 + FH * FH, W3 = B2 + FH, B3 = W3 + D * FH,
                       SIZE = B3 + D;
};

__device__ __forceinline__ float lrelu(float v) { return v > 0.f ? v : SLOPE * v; }

// Several flows (e.g. the `flow3d` and `flow3d_update` of the head) run in ONE launch: each job owns a
// block-aligned range of rows and brings its own parameters / gradient destinations.
struct FlowJobs {
  int n;
  DasFlowJob j[DAS_FLOW_MAX_JOBS];
};
__device__ __forceinline__ DasFlowJob job_of_block(const FlowJobs& jobs) {
  const int row0 = blockIdx.x * FR;
  int ji = 0;
  for (int q = 1; q < DAS_FLOW_MAX_JOBS; ++q)
    if (q < jobs.n && row0 >= jobs.j[q].row_start) ji = q;
  return jobs.j[ji];
}

// a1 -> h1 -> a2 -> h2 -> out for one row; w = the net's parameters in LDS. zin: masked input (D values).
template <int D>
__device__ __forceinline__ void mlp_forward(const float* __restrict__ w, const float* zin, float* h1, float* h2,
                                            float* out) {
#pragma unroll
  for (int k = 0; k < FH; ++k) {
    float a = w[Net<D>::B1 + k];
#pragma unroll
    for (int d = 0; d < D; ++d) a += w[Net<D>::W1 + k * D + d] * zin[d];
    h1[k] = lrelu(a);
    if ((k & 15) == 15) __builtin_amdgcn_sched_barrier(0);
  }
  // Fully unrolled on purpose: h1/h2 live in registers, which only works with compile-time indices. The
  // weight row of output o+1 is requested from LDS before the 64 FMAs of output o; the scheduling barrier
  // per output keeps the compiler from hoisting hundreds of LDS loads (and spilling).
  float4 wrow[2][FH / 4];
#pragma unroll
  for (int i4 = 0; i4 < FH / 4; ++i4) wrow[0][i4] = reinterpret_cast<const float4*>(w + Net<D>::W2)[i4];
#pragma unroll
  for (int o = 0; o < FH; ++o) {
    if (o + 1 < FH) {
#pragma unroll
      for (int i4 = 0; i4 < FH / 4; ++i4)
        wrow[(o + 1) & 1][i4] = reinterpret_cast<const float4*>(w + Net<D>::W2 + (o + 1) * FH)[i4];
    }
    float a = w[Net<D>::B2 + o];
#pragma unroll
    for (int i4 = 0; i4 < FH / 4; ++i4) {
      const float4 v = wrow[o & 1][i4];
      a += v.x * h1[i4 * 4] + v.y * h1[i4 * 4 + 1] + v.z * h1[i4 * 4 + 2] + v.w * h1[i4 * 4 + 3];
    }
    h2[o] = lrelu(a);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int c = 0; c < D; ++c) {
    float a = w[Net<D>::B3 + c];
    const float4* row = reinterpret_cast<const float4*>(w + Net<D>::W3 + c * FH);
#pragma unroll
    for (int i4 = 0; i4 < FH / 4; ++i4) {
      const float4 v = row[i4];
      a += v.x * h2[i4 * 4] + v.y * h2[i4 * 4 + 1] + v.z * h2[i4 * 4 + 2] + v.w * h2[i4 * 4 + 3];
    }
    out[c] = a;
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ------------------------------------------------------------------ forward
template <int D>
__global__ __launch_bounds__(FR) void realnvp_fwd_kernel(const float* __restrict__ x, FlowJobs jobs, int layers,
                                                         unsigned mask_bits, float* __restrict__ logp,
                                                         float* __restrict__ zout) {
  const DasFlowJob job = job_of_block(jobs);
  const float* __restrict__ params = job.params;
  const int r = blockIdx.x * FR + threadIdx.x;
  const bool live = r >= job.row_start && r < job.row_end;
  float z[D], logdet = 0.f;
#pragma unroll
  for (int d = 0; d < D; ++d) z[d] = live ? x[(size_t)r * D + d] : 0.f;
  for (int i = layers - 1; i >= 0; --i) {
    const unsigned m = (mask_bits >> (i * D)) & ((1u << D) - 1u);
    float zin[D], s[D], t[D];
#pragma unroll
    for (int d = 0; d < D; ++d) zin[d] = ((m >> d) & 1u) ? z[d] : 0.f;
    for (int which = 1; which >= 0; --which) {   // s-net, then t-net (one inlined copy of the MLP code)
      float h1[FH], h2[FH], out[D];
      // every lane uses the same weights: wave-uniform addresses -> scalar loads, the FMAs take them from SGPRs
      const float* w = params + (size_t)(i * 2 + which) * Net<D>::SIZE;
      mlp_forward<D>(w, zin, h1, h2, out);
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
  if (live) {
    float q = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) { q += z[d] * z[d]; zout[(size_t)r * D + d] = z[d]; }
    logp[r] = -0.5f * q - 0.5f * D * 1.8378770664093453f + logdet;
  }
}

// ------------------------------------------------------------------ backward
// Weight phase: dW2[o][i] (+db2), dW1[k][d] (+db1), dW3[c][k] (+db3) of one net from the block's rows in LDS.
//   bufH: [FR][FH] left factor rows, bufD: [FR][FH] right factor rows.
__device__ __forceinline__ void reduce_w2(const float* __restrict__ H1, const float* __restrict__ D2, float* dW2,
                                          float* db2) {
  const int t = threadIdx.x, o = t >> 2, i0 = (t & 3) * 16;
  float acc[16], bsum = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = 0.f;
#pragma unroll 2
  for (int r = 0; r < FR; ++r) {
    const float dv = D2[r * FH + o];
    const float4* h = reinterpret_cast<const float4*>(H1 + r * FH + i0);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = h[q];
      acc[q * 4] += dv * v.x; acc[q * 4 + 1] += dv * v.y; acc[q * 4 + 2] += dv * v.z; acc[q * 4 + 3] += dv * v.w;
    }
    bsum += dv;
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) atomicAdd(dW2 + o * FH + i0 + j, acc[j]);
  if ((t & 3) == 0) atomicAdd(db2 + o, bsum);
}

template <int D>
__global__ __launch_bounds__(FR) void realnvp_bwd_kernel(const float* __restrict__ zfin, const float* __restrict__ glogp,
                                                         FlowJobs jobs, int layers, unsigned mask_bits,
                                                         float* __restrict__ dx) {
  const DasFlowJob job = job_of_block(jobs);
  const float* __restrict__ params = job.params;
  float* __restrict__ dparams = job.dparams;
  float* const* __restrict__ dst_table = job.dst_table;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* buf0 = sm;                       // [FR][FH]
  float* buf1 = buf0 + FR * FH;           // [FR][FH]
  float* sm3 = buf1 + FR * FH;            // [FR][4] delta3 rows
  float* smz = sm3 + FR * 4;              // [FR][4] masked inputs
  const int tid = threadIdx.x;
  const int r = blockIdx.x * FR + tid;
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
    float zin[D], sraw[D], tval[D];
#pragma unroll
    for (int d = 0; d < D; ++d) zin[d] = ((m >> d) & 1u) ? z[d] : 0.f;
    float dzin_m[D];                      // gradient reaching the masked inputs through the two nets
#pragma unroll
    for (int d = 0; d < D; ++d) dzin_m[d] = 0.f;
    float sd[D], es[D];
    // two passes: s-net first (its output is needed to invert the layer), then t-net
    for (int which = 1; which >= 0; --which) {
      float A[FH], Bv[FH], out[D], d3[D];
      // the net's weights: wave-uniform global addresses -> scalar loads (no LDS staging)
      const float* wl = params + (size_t)(i * 2 + which) * Net<D>::SIZE;
      __syncthreads();   // the previous net's weight phase has finished reading buf0 / smz
      mlp_forward<D>(wl, zin, A, Bv, out);       // A = h1, Bv = h2
      if (which == 1) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
          sraw[d] = out[d];
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
      // ---- layer 3: dW3 = sum_r d3 (x) h2
#pragma unroll
      for (int k = 0; k < FH; ++k) buf1[tid * FH + k] = Bv[k];
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        sm3[tid * 4 + d] = d < D ? d3[d < D ? d : 0] : 0.f;
        smz[tid * 4 + d] = d < D ? zin[d < D ? d : 0] : 0.f;
      }
      __syncthreads();
      if (tid < D * FH) {
        const int c = tid / FH, k = tid - c * FH;
        float acc = 0.f;
        for (int rr = 0; rr < FR; ++rr) acc += sm3[rr * 4 + c] * buf1[rr * FH + k];
        atomicAdd(gW3 + c * FH + k, acc);
      } else if (tid < D * FH + D) {
        const int c = tid - D * FH;
        float acc = 0.f;
        for (int rr = 0; rr < FR; ++rr) acc += sm3[rr * 4 + c];
        atomicAdd(gB3 + c, acc);
      }
      // delta2 = (W3^T d3) * lrelu'(a2)   (sign(a2) = sign(h2)), overwriting h2
#pragma unroll
      for (int k = 0; k < FH; ++k) {
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < D; ++c) a += wl[Net<D>::W3 + c * FH + k] * d3[c];
        Bv[k] = a * (Bv[k] > 0.f ? 1.f : SLOPE);
      }
      __syncthreads();   // buf1 (h2) no longer read
      // ---- layer 2: dW2 = sum_r delta2 (x) h1
#pragma unroll
      for (int k = 0; k < FH; ++k) { buf0[tid * FH + k] = A[k]; buf1[tid * FH + k] = Bv[k]; }
      __syncthreads();
      reduce_w2(buf0, buf1, gW2, gB2);
      // delta1 = (W2^T delta2) * lrelu'(a1), overwriting h1
      unsigned long long pos = 0ull;
#pragma unroll
      for (int k = 0; k < FH; ++k) pos |= (A[k] > 0.f ? 1ull : 0ull) << k;
#pragma unroll
      for (int k = 0; k < FH; ++k) A[k] = 0.f;
      {
        float4 wrow[2][FH / 4];
#pragma unroll
        for (int i4 = 0; i4 < FH / 4; ++i4) wrow[0][i4] = reinterpret_cast<const float4*>(wl + Net<D>::W2)[i4];
#pragma unroll
        for (int o = 0; o < FH; ++o) {
          if (o + 1 < FH) {
#pragma unroll
            for (int i4 = 0; i4 < FH / 4; ++i4)
              wrow[(o + 1) & 1][i4] = reinterpret_cast<const float4*>(wl + Net<D>::W2 + (o + 1) * FH)[i4];
          }
          const float dv = Bv[o];
#pragma unroll
          for (int i4 = 0; i4 < FH / 4; ++i4) {
            const float4 v = wrow[o & 1][i4];
            A[i4 * 4] += v.x * dv; A[i4 * 4 + 1] += v.y * dv; A[i4 * 4 + 2] += v.z * dv; A[i4 * 4 + 3] += v.w * dv;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int k = 0; k < FH; ++k) A[k] *= ((pos >> k) & 1ull) ? 1.f : SLOPE;
      __syncthreads();   // buf0 / buf1 no longer read by reduce_w2
      // ---- layer 1: dW1 = sum_r delta1 (x) z_;  dz_ = W1^T delta1
#pragma unroll
      for (int k = 0; k < FH; ++k) buf0[tid * FH + k] = A[k];
      __syncthreads();
      if (tid < FH * D) {
        const int k = tid / D, d = tid - k * D;
        float acc = 0.f;
        for (int rr = 0; rr < FR; ++rr) acc += buf0[rr * FH + k] * smz[rr * 4 + d];
        atomicAdd(gW1 + k * D + d, acc);
      } else if (tid < FH * D + FH) {
        const int k = tid - FH * D;
        float acc = 0.f;
        for (int rr = 0; rr < FR; ++rr) acc += buf0[rr * FH + k];
        atomicAdd(gB1 + k, acc);
      }
#pragma unroll
      for (int d = 0; d < D; ++d) {
        if ((m >> d) & 1u) {
          float a = 0.f;
#pragma unroll
          for (int k = 0; k < FH; ++k) a += wl[Net<D>::W1 + k * D + d] * A[k];
          dzin_m[d] += a;
        }
      }
    }
    (void)sraw;
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
  if (live) {
#pragma unroll
    for (int d = 0; d < D; ++d) dx[(size_t)r * D + d] = dz[d];
  }
}

template <int D>
constexpr size_t bwd_smem() {
  return (size_t)(2 * FR * FH + 2 * FR * 4) * sizeof(float);
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
  if (!x || !logp || !z_out || rows_total < 1 || layers < 1 || layers * D > 32 || (D != 2 && D != 3)) return DAS_ERR_ARG;
  if (!jobs_ok(jobs, njobs, rows_total, false)) return DAS_ERR_ARG;
  FlowJobs fj;
  fj.n = njobs;
  for (int q = 0; q < DAS_FLOW_MAX_JOBS; ++q) fj.j[q] = jobs[q < njobs ? q : 0];
  const int blocks = (rows_total + FR - 1) / FR;
  if (D == 3) {
    hipLaunchKernelGGL(realnvp_fwd_kernel<3>, dim3(blocks), dim3(FR), 0, (hipStream_t)stream, x, fj, layers, mask_bits,
                       logp, z_out);
  } else {
    hipLaunchKernelGGL(realnvp_fwd_kernel<2>, dim3(blocks), dim3(FR), 0, (hipStream_t)stream, x, fj, layers, mask_bits,
                       logp, z_out);
  }
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}

extern "C" int das_realnvp_log_prob_multi_backward(const float* z_final, const float* grad_logp, int rows_total, int D,
                                                   const DasFlowJob* jobs, int njobs, int layers, unsigned mask_bits,
                                                   float* dx, void* stream) {
  if (!z_final || !grad_logp || !dx || rows_total < 1 || layers < 1 || layers * D > 32 || (D != 2 && D != 3))
    return DAS_ERR_ARG;
  if (!jobs_ok(jobs, njobs, rows_total, true)) return DAS_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  FlowJobs fj;
  fj.n = njobs;
  for (int q = 0; q < DAS_FLOW_MAX_JOBS; ++q) fj.j[q] = jobs[q < njobs ? q : 0];
  const int blocks = (rows_total + FR - 1) / FR;
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
    hipLaunchKernelGGL(realnvp_bwd_kernel<3>, dim3(blocks), dim3(FR), bwd_smem<3>(), s, z_final, grad_logp, fj, layers,
                       mask_bits, dx);
  } else {
    hipLaunchKernelGGL(realnvp_bwd_kernel<2>, dim3(blocks), dim3(FR), bwd_smem<2>(), s, z_final, grad_logp, fj, layers,
                       mask_bits, dx);
  }
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}

extern "C" int das_realnvp_log_prob(const float* x, int N, int D, const float* params, int layers, unsigned mask_bits,
                                    float* logp, float* z_out, void* stream) {
  DasFlowJob j;
  j.params = params; j.dparams = nullptr; j.dst_table = nullptr; j.row_start = 0; j.row_end = N;
  return das_realnvp_log_prob_multi(x, N, D, &j, 1, layers, mask_bits, logp, z_out, stream);
}

extern "C" int das_realnvp_log_prob_backward(const float* z_final, const float* grad_logp, int N, int D,
                                             const float* params, int layers, unsigned mask_bits, float* dx,
                                             float* dparams, float* const* dst_table, void* stream) {
  DasFlowJob j;
  j.params = params; j.dparams = dparams; j.dst_table = dst_table; j.row_start = 0; j.row_end = N;
  return das_realnvp_log_prob_multi_backward(z_final, grad_logp, N, D, &j, 1, layers, mask_bits, dx, stream);
}
