// The cross-stage merge of MSPN in train mode (mspn_mmpose.py:254-275, 381-404): the next stage's feature of a level is
//
//     x' = x + relu(BN1(raw1)) + relu(BN2(raw2)),     raw1 = out_skip1(...), raw2 = out_skip2(...) of the previous stage
//
// Kernel by kernel that is two BatchNorm apply passes (read, write) and a three-operand add (three reads, write); here
// the two normalised tensors are never written: one pass reads x, raw1, raw2 and writes x'. Backward, g = dx' is the
// gradient of all three operands; with g_i = g * (BN_i(raw_i) > 0) (mask recomputed from raw_i, as MASK = 2 of
// bn_bwd_reduce_kernel):
//   reduce   one sweep over g, raw1, raw2 -> sums f32[4C] = [sum g1 | sum g1 xhat1 | sum g2 | sum g2 xhat2]
//   apply    one sweep over g, raw1, raw2 -> d raw_i = gamma_i invstd_i (g_i - sum g_i / N - xhat_i sum g_i xhat_i / N)
// against two reduce passes (two reads each) and two apply passes (two reads, one write each).
#include "common.h"
#include "prof.h"
#include "tuning.h"

#include <algorithm>

namespace {
constexpr int TPB = 256;

struct BnPar {
  const float *mean, *invstd, *gamma, *beta;
};

template <typename T>
__global__ __launch_bounds__(TPB) void bn_relu_add3_fwd_kernel(const T* __restrict__ x, const T* __restrict__ raw1,
                                                               const T* __restrict__ raw2, T* __restrict__ out, long long rows,
                                                               int C, int rows_per_block, BnPar p1, BnPar p2) {
  constexpr int EPV = Elem<T>::EPV;
  const int VC = C / EPV;
  const int VCB = min(VC, TPB), PL = TPB / VCB, pl = threadIdx.x / VCB;
  const long long r0 = (long long)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  for (int v = threadIdx.x % VCB; v < VC && pl < PL; v += VCB) {
    const int c0 = v * EPV;
    float m1[EPV], i1[EPV], g1[EPV], b1[EPV], m2[EPV], i2[EPV], g2[EPV], b2[EPV];
#pragma unroll
    for (int j = 0; j < EPV; ++j) {
      m1[j] = p1.mean[c0 + j]; i1[j] = p1.invstd[c0 + j]; g1[j] = p1.gamma[c0 + j]; b1[j] = p1.beta[c0 + j];
      m2[j] = p2.mean[c0 + j]; i2[j] = p2.invstd[c0 + j]; g2[j] = p2.gamma[c0 + j]; b2[j] = p2.beta[c0 + j];
    }
    constexpr int U = 4;
    for (long long r = r0 + pl; r < r1; r += (long long)U * PL) {
      uint4 rx[U], ra[U], rb[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long o = min(r + (long long)u * PL, r1 - 1) * C + c0;
        rx[u] = *reinterpret_cast<const uint4*>(x + o);
        ra[u] = *reinterpret_cast<const uint4*>(raw1 + o);
        rb[u] = *reinterpret_cast<const uint4*>(raw2 + o);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long ru = r + (long long)u * PL;
        if (ru >= r1) break;
        float fx[EPV], fa[EPV], fb[EPV], o[EPV];
        Elem<T>::unpack(rx[u], fx);
        Elem<T>::unpack(ra[u], fa);
        Elem<T>::unpack(rb[u], fb);
#pragma unroll
        for (int j = 0; j < EPV; ++j)
          o[j] = fx[j] + fmaxf(bn_affine(fa[j], m1[j], i1[j], g1[j], b1[j]), 0.f) +
                 fmaxf(bn_affine(fb[j], m2[j], i2[j], g2[j], b2[j]), 0.f);
        *reinterpret_cast<uint4*>(out + ru * C + c0) = Elem<T>::pack(o);
      }
    }
  }
}

// out = relu(BN1(raw1) + BN2(raw2)): a bottleneck's last BatchNorm with the block's projection shortcut (mspn_mmpose.py:126-157:
// `out = bn3(conv3(..)) + downsample(x)`, then ReLU) — the normalised shortcut is never written.
template <typename T>
__global__ __launch_bounds__(TPB) void bn_dual_apply_kernel(const T* __restrict__ raw1, const T* __restrict__ raw2,
                                                            T* __restrict__ out, long long rows, int C, int rows_per_block,
                                                            BnPar p1, BnPar p2, int relu, unsigned char* __restrict__ bits) {
  constexpr int EPV = Elem<T>::EPV;
  const int VC = C / EPV;
  const int VCB = min(VC, TPB), PL = TPB / VCB, pl = threadIdx.x / VCB;
  const long long r0 = (long long)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  for (int v = threadIdx.x % VCB; v < VC && pl < PL; v += VCB) {
    const int c0 = v * EPV;
    float m1[EPV], i1[EPV], g1[EPV], b1[EPV], m2[EPV], i2[EPV], g2[EPV], b2[EPV];
#pragma unroll
    for (int j = 0; j < EPV; ++j) {
      m1[j] = p1.mean[c0 + j]; i1[j] = p1.invstd[c0 + j]; g1[j] = p1.gamma[c0 + j]; b1[j] = p1.beta[c0 + j];
      m2[j] = p2.mean[c0 + j]; i2[j] = p2.invstd[c0 + j]; g2[j] = p2.gamma[c0 + j]; b2[j] = p2.beta[c0 + j];
    }
    constexpr int U = 4;
    for (long long r = r0 + pl; r < r1; r += (long long)U * PL) {
      uint4 ra[U], rb[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long o = min(r + (long long)u * PL, r1 - 1) * C + c0;
        ra[u] = *reinterpret_cast<const uint4*>(raw1 + o);
        rb[u] = *reinterpret_cast<const uint4*>(raw2 + o);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long ru = r + (long long)u * PL;
        if (ru >= r1) break;
        float fa[EPV], fb[EPV], o[EPV];
        Elem<T>::unpack(ra[u], fa);
        Elem<T>::unpack(rb[u], fb);
#pragma unroll
        for (int j = 0; j < EPV; ++j) {
          const float t = bn_affine(fa[j], m1[j], i1[j], g1[j], b1[j]) + bn_affine(fb[j], m2[j], i2[j], g2[j], b2[j]);
          o[j] = relu ? fmaxf(t, 0.f) : t;
        }
        const uint4 packed = Elem<T>::pack(o);
        *reinterpret_cast<uint4*>(out + ru * C + c0) = packed;
        if (bits) bits[ru * VC + v] = (unsigned char)relu_bits<T>(packed);
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(TPB) void bn_relu_add3_bwd_reduce_kernel(const T* __restrict__ g, const T* __restrict__ raw1,
                                                                      const T* __restrict__ raw2, long long rows, int C,
                                                                      BnPar p1, BnPar p2, float* __restrict__ sums) {
  constexpr int EPV = Elem<T>::EPV;
  extern __shared__ float part[];   // [PL][4C]
  const int VC = C / EPV;
  const int VCB = min(VC, TPB), PL = TPB / VCB, pl = threadIdx.x / VCB;
  const long long run = (rows + gridDim.x - 1) / gridDim.x;
  const long long r0 = (long long)blockIdx.x * run, r1 = min(rows, r0 + run);
  for (int v = threadIdx.x % VCB; v < VC && pl < PL; v += VCB) {
    const int c0 = v * EPV;
    float m1[EPV], i1[EPV], g1[EPV], b1[EPV], m2[EPV], i2[EPV], g2[EPV], b2[EPV], s[4][EPV];
#pragma unroll
    for (int j = 0; j < EPV; ++j) {
      m1[j] = p1.mean[c0 + j]; i1[j] = p1.invstd[c0 + j]; g1[j] = p1.gamma[c0 + j]; b1[j] = p1.beta[c0 + j];
      m2[j] = p2.mean[c0 + j]; i2[j] = p2.invstd[c0 + j]; g2[j] = p2.gamma[c0 + j]; b2[j] = p2.beta[c0 + j];
      s[0][j] = s[1][j] = s[2][j] = s[3][j] = 0.f;
    }
    constexpr int U = 4;
    for (long long r = r0 + pl; r < r1; r += (long long)U * PL) {
      uint4 rg[U], ra[U], rb[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long ru = r + (long long)u * PL;
        rg[u] = ra[u] = rb[u] = make_uint4(0, 0, 0, 0);
        if (ru < r1) {
          const long long o = ru * C + c0;
          rg[u] = *reinterpret_cast<const uint4*>(g + o);
          ra[u] = *reinterpret_cast<const uint4*>(raw1 + o);
          rb[u] = *reinterpret_cast<const uint4*>(raw2 + o);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {   // rows past the end carry g = 0 and add nothing
        float fg[EPV], fa[EPV], fb[EPV];
        Elem<T>::unpack(rg[u], fg);
        Elem<T>::unpack(ra[u], fa);
        Elem<T>::unpack(rb[u], fb);
#pragma unroll
        for (int j = 0; j < EPV; ++j) {
          const float ga = bn_affine(fa[j], m1[j], i1[j], g1[j], b1[j]) > 0.f ? fg[j] : 0.f;
          const float gb = bn_affine(fb[j], m2[j], i2[j], g2[j], b2[j]) > 0.f ? fg[j] : 0.f;
          s[0][j] += ga; s[1][j] += ga * (fa[j] - m1[j]) * i1[j];
          s[2][j] += gb; s[3][j] += gb * (fb[j] - m2[j]) * i2[j];
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) lds_put<EPV>(part, 4 * C, pl, k * C + c0, s[k]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 4 * C; i += TPB) atomicAdd(sums + i, lds_fold(part, 4 * C, PL, i));
}

// Workgroup 0 also adds the parameter gradients (dbeta_i += sum g_i, dgamma_i += sum g_i xhat_i) when accumulators are given.
template <typename T>
__global__ __launch_bounds__(TPB) void bn_relu_add3_bwd_apply_kernel(const T* __restrict__ g, const T* __restrict__ raw1,
                                                                     const T* __restrict__ raw2, T* __restrict__ draw1,
                                                                     T* __restrict__ draw2, long long rows, int C,
                                                                     int rows_per_block, BnPar p1, BnPar p2,
                                                                     const float* __restrict__ sums, float inv_n,
                                                                     float* __restrict__ dgamma1, float* __restrict__ dbeta1,
                                                                     float* __restrict__ dgamma2, float* __restrict__ dbeta2) {
  constexpr int EPV = Elem<T>::EPV;
  const int VC = C / EPV;
  if (blockIdx.x == 0 && dgamma1) {
    for (int c = threadIdx.x; c < C; c += TPB) {
      dbeta1[c] += sums[c]; dgamma1[c] += sums[C + c]; dbeta2[c] += sums[2 * C + c]; dgamma2[c] += sums[3 * C + c];
    }
  }
  const int VCB = min(VC, TPB), PL = TPB / VCB, pl = threadIdx.x / VCB;
  const long long r0 = (long long)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  for (int v = threadIdx.x % VCB; v < VC && pl < PL; v += VCB) {
    const int c0 = v * EPV;
    float m1[EPV], i1[EPV], g1[EPV], b1[EPV], m2[EPV], i2[EPV], g2[EPV], b2[EPV], k2a[EPV], k3a[EPV], k2b[EPV], k3b[EPV];
#pragma unroll
    for (int j = 0; j < EPV; ++j) {
      m1[j] = p1.mean[c0 + j]; i1[j] = p1.invstd[c0 + j]; g1[j] = p1.gamma[c0 + j]; b1[j] = p1.beta[c0 + j];
      m2[j] = p2.mean[c0 + j]; i2[j] = p2.invstd[c0 + j]; g2[j] = p2.gamma[c0 + j]; b2[j] = p2.beta[c0 + j];
      k2a[j] = sums[c0 + j] * inv_n; k3a[j] = sums[C + c0 + j] * inv_n;
      k2b[j] = sums[2 * C + c0 + j] * inv_n; k3b[j] = sums[3 * C + c0 + j] * inv_n;
    }
    constexpr int U = 2;
    for (long long r = r0 + pl; r < r1; r += (long long)U * PL) {
      uint4 rg[U], ra[U], rb[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long o = min(r + (long long)u * PL, r1 - 1) * C + c0;
        rg[u] = *reinterpret_cast<const uint4*>(g + o);
        ra[u] = *reinterpret_cast<const uint4*>(raw1 + o);
        rb[u] = *reinterpret_cast<const uint4*>(raw2 + o);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long ru = r + (long long)u * PL;
        if (ru >= r1) break;
        float fg[EPV], fa[EPV], fb[EPV], oa[EPV], ob[EPV];
        Elem<T>::unpack(rg[u], fg);
        Elem<T>::unpack(ra[u], fa);
        Elem<T>::unpack(rb[u], fb);
#pragma unroll
        for (int j = 0; j < EPV; ++j) {
          const float ga = bn_affine(fa[j], m1[j], i1[j], g1[j], b1[j]) > 0.f ? fg[j] : 0.f;
          const float gb = bn_affine(fb[j], m2[j], i2[j], g2[j], b2[j]) > 0.f ? fg[j] : 0.f;
          oa[j] = g1[j] * i1[j] * (ga - k2a[j] - (fa[j] - m1[j]) * i1[j] * k3a[j]);
          ob[j] = g2[j] * i2[j] * (gb - k2b[j] - (fb[j] - m2[j]) * i2[j] * k3b[j]);
        }
        *reinterpret_cast<uint4*>(draw1 + ru * C + c0) = Elem<T>::pack(oa);
        *reinterpret_cast<uint4*>(draw2 + ru * C + c0) = Elem<T>::pack(ob);
      }
    }
  }
}
}  // namespace

#define DISPATCH_T(dtype, CALL)                 \
  if ((dtype) == DAS_BF16) { using T = bf16_t; CALL; } \
  else if ((dtype) == DAS_F32) { using T = float; CALL; } \
  else return DAS_ERR_ARG;

static bool par_ok(const float* const* p) {
  for (int i = 0; i < 8; ++i)
    if (!p[i]) return false;
  return true;
}

extern "C" int das_bn_relu_add3_forward(const void* x, const void* raw1, const void* raw2, void* out, int dtype, long long rows,
                                        int C, const float* const* bn, void* stream) {
  DAS_PROF(stream);
  if (!x || !raw1 || !raw2 || !out || !bn || !par_ok(bn) || rows < 1 || C % 8 || C < 8 || C > 4096) return DAS_ERR_ARG;
  const int vc = C / (dtype == DAS_F32 ? 4 : 8);
  const int pl = TPB / (vc < TPB ? vc : TPB);
  const long long rpb = (long long)pl * 4 * (rows * C >= (1ll << 24) ? 2 : 1);
  const long long grid = (rows + rpb - 1) / rpb;
  if (grid >= (1ll << 31)) return DAS_ERR_ARG;
  const BnPar p1{bn[0], bn[1], bn[2], bn[3]}, p2{bn[4], bn[5], bn[6], bn[7]};
  DISPATCH_T(dtype, {
    hipLaunchKernelGGL(bn_relu_add3_fwd_kernel<T>, dim3((unsigned)grid), dim3(TPB), 0, (hipStream_t)stream, (const T*)x,
                       (const T*)raw1, (const T*)raw2, (T*)out, rows, C, (int)rpb, p1, p2);
  });
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}

extern "C" int das_bn_dual_apply(const void* raw1, const void* raw2, void* out, int dtype, long long rows, int C,
                                 const float* const* bn, int relu, void* relu_bits_out, void* stream) {
  DAS_PROF(stream);
  if (!raw1 || !raw2 || !out || !bn || !par_ok(bn) || rows < 1 || C % 8 || C < 8 || C > 4096) return DAS_ERR_ARG;
  const int vc = C / (dtype == DAS_F32 ? 4 : 8);
  const int pl = TPB / (vc < TPB ? vc : TPB);
  const long long rpb = (long long)pl * 4 * (rows * C >= (1ll << 24) ? 2 : 1);
  const long long grid = (rows + rpb - 1) / rpb;
  if (grid >= (1ll << 31)) return DAS_ERR_ARG;
  const BnPar p1{bn[0], bn[1], bn[2], bn[3]}, p2{bn[4], bn[5], bn[6], bn[7]};
  DISPATCH_T(dtype, {
    hipLaunchKernelGGL(bn_dual_apply_kernel<T>, dim3((unsigned)grid), dim3(TPB), 0, (hipStream_t)stream, (const T*)raw1,
                       (const T*)raw2, (T*)out, rows, C, (int)rpb, p1, p2, relu, (unsigned char*)relu_bits_out);
  });
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}

extern "C" int das_bn_relu_add3_backward(const void* g, const void* raw1, const void* raw2, void* draw1, void* draw2, int dtype,
                                         long long rows, int C, const float* const* bn, float* sums, int sums_zeroed,
                                         long long stat_rows,
                                         float* dgamma1_acc, float* dbeta1_acc, float* dgamma2_acc, float* dbeta2_acc,
                                         int phase, void* stream) {
  DAS_PROF(stream);
  if (phase < 0 || phase > 2) return DAS_ERR_ARG;
  if (!g || !raw1 || !raw2 || ((!draw1 || !draw2) && phase != 1) || !bn || !par_ok(bn) || !sums || rows < 1 || stat_rows < 1 || C % 8 || C < 8 ||
      C > 4096)
    return DAS_ERR_ARG;
  const int nacc = (dgamma1_acc != nullptr) + (dbeta1_acc != nullptr) + (dgamma2_acc != nullptr) + (dbeta2_acc != nullptr);
  if (nacc != 0 && nacc != 4) return DAS_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (phase != 2 && !sums_zeroed && hipMemsetAsync(sums, 0, sizeof(float) * 4 * C, s) != hipSuccess) return DAS_ERR_LAUNCH;
  const int vc = C / (dtype == DAS_F32 ? 4 : 8);
  const int pl = TPB / (vc < TPB ? vc : TPB);
  const long long cap = dastune::get(dastune::BN_UPMERGE_BLOCKS);
  const int blocks = (int)std::min<long long>(cap, std::max<long long>(1, rows / (4 * pl)));
  const size_t lds = (size_t)pl * 4 * C * sizeof(float);
  const long long rpb = (long long)pl * 2 * (rows * C >= (1ll << 24) ? 2 : 1);
  const long long grid = (rows + rpb - 1) / rpb;
  if (grid >= (1ll << 31)) return DAS_ERR_ARG;
  const BnPar p1{bn[0], bn[1], bn[2], bn[3]}, p2{bn[4], bn[5], bn[6], bn[7]};
  DISPATCH_T(dtype, {
    if (lds > 48 * 1024 && hipFuncSetAttribute((const void*)bn_relu_add3_bwd_reduce_kernel<T>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return DAS_ERR_LAUNCH;
    if (phase != 2)
      hipLaunchKernelGGL(bn_relu_add3_bwd_reduce_kernel<T>, dim3(blocks), dim3(TPB), lds, s, (const T*)g, (const T*)raw1,
                         (const T*)raw2, rows, C, p1, p2, sums);
    if (phase != 1)
      hipLaunchKernelGGL(bn_relu_add3_bwd_apply_kernel<T>, dim3((unsigned)grid), dim3(TPB), 0, s, (const T*)g, (const T*)raw1,
                       (const T*)raw2, (T*)draw1, (T*)draw2, rows, C, (int)rpb, p1, p2, sums, 1.f / (float)stat_rows,
                       dgamma1_acc, dbeta1_acc, dgamma2_acc, dbeta2_acc);
  });
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}
