// Normalisation kernels over NHWC: train-mode BatchNorm finalize/apply and GroupNorm(+ReLU).
// All HBM-bound; statistics in f32.
#include <algorithm>
#include "prof.h"

#include "common.h"
#include "tuning.h"
#include "workspace.h"

namespace {
constexpr int TPB = 256;
inline int grid_for(long long n) {
  long long b = (n + TPB - 1) / TPB;
  return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

// Batch statistics of one channel from the conv epilogue's [sum, sum of squares] (biased variance).
struct BnStat {
  float mean, var, invstd;
};
__device__ __forceinline__ BnStat bn_stat(const float* __restrict__ stats, int C, int c, float n, float eps) {
#pragma clang fp contract(off)
  BnStat s;
  s.mean = stats[c] / n;
  s.var = fmaxf(stats[C + c] / n - s.mean * s.mean, 0.f);
  s.invstd = 1.0f / sqrtf(s.var + eps);
  return s;
}

// The "finalize" duties, done by workgroup 0 of the apply kernel (no separate launch): publish mean / invstd for
// the backward pass, update the running statistics (momentum, unbiased variance) and the batch counter.
__device__ __forceinline__ void bn_publish(const float* __restrict__ stats, long long count, int C, float* running_mean,
                                           float* running_var, float momentum, float eps, float* save_mean,
                                           float* save_invstd, long long* num_batches_tracked) {
#pragma clang fp contract(off)
  const float n = (float)count;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const BnStat s = bn_stat(stats, C, c, n, eps);
    save_mean[c] = s.mean;
    save_invstd[c] = s.invstd;
    if (running_mean) {
      const float unbiased = count > 1 ? s.var * (n / (n - 1.f)) : s.var;
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * s.mean;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
  }
  if (threadIdx.x == 0 && num_batches_tracked) *num_batches_tracked += 1;
}

// Finalize only (das_bn_train_apply with y == NULL): one thread per channel over as many 64-thread workgroups as that
// takes — the same fold order and the same arithmetic as the apply kernels. As workgroup 0 of an apply kernel with an
// empty body this took 6-15 us (one workgroup walking 2C sums and C channels); the fused train-mode passes call it
// ~60 times a step.
__device__ __forceinline__ void bn_finalize_channel(int c, const float* __restrict__ gstats, int slots, long long count, int C,
                                                    float* running_mean, float* running_var, float momentum, float eps,
                                                    float* save_mean, float* save_invstd, long long* num_batches_tracked) {
#pragma clang fp contract(off)
  if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;
  if (c >= C) return;
  float sum[2];
#pragma unroll
  for (int w = 0; w < 2; ++w) {
    float a = 0.f;
    if (slots > 1) {
      for (int k0 = 0; k0 < slots; k0 += 16) {
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = (k0 + k < slots) ? gstats[(size_t)(k0 + k) * 2 * C + w * C + c] : 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) a += v[k];
      }
    } else {
      a = gstats[w * C + c];
    }
    sum[w] = a;
  }
  const float n = (float)count;
  BnStat st;
  st.mean = sum[0] / n;
  st.var = fmaxf(sum[1] / n - st.mean * st.mean, 0.f);
  st.invstd = 1.0f / sqrtf(st.var + eps);
  save_mean[c] = st.mean;
  save_invstd[c] = st.invstd;
  if (running_mean) {
    const float unbiased = count > 1 ? st.var * (n / (n - 1.f)) : st.var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * st.mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
  }
}
__global__ __launch_bounds__(64) void bn_finalize_kernel(const float* __restrict__ gstats, int slots, long long count, int C,
                                                         float* running_mean, float* running_var, float momentum, float eps,
                                                         float* save_mean, float* save_invstd, long long* num_batches_tracked) {
  bn_finalize_channel(blockIdx.x * 64 + threadIdx.x, gstats, slots, count, C, running_mean, running_var, momentum, eps,
                      save_mean, save_invstd, num_batches_tracked);
}
// Several layers in one launch (das_bn_finalize_many): blockIdx.y = layer
struct BnFinalizeMany {
  DasBnFinalize l[4];
};
__global__ __launch_bounds__(64) void bn_finalize_many_kernel(BnFinalizeMany a) {
  const DasBnFinalize& f = a.l[blockIdx.y];
  bn_finalize_channel(blockIdx.x * 64 + threadIdx.x, f.stats, f.stats_slots, f.count, f.C, f.running_mean, f.running_var,
                      f.momentum, f.eps, f.save_mean, f.save_invstd, f.num_batches_tracked);
}

// One 16-byte channel vector per thread and iteration. The grid stride is a multiple of the channel-vector
// count whenever that count is a power of two (every BN on the path), so a thread keeps its channels for the
// whole loop and the per-channel constants live in registers (FIXED); two iterations are kept in flight.
template <typename T, bool FIXED>
__global__ void bn_apply_kernel(const T* __restrict__ x, T* __restrict__ y, long long count, int C,
                                const float* __restrict__ gstats, int slots, long long stat_count, float eps,
                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                const T* __restrict__ res, int relu, float* running_mean, float* running_var,
                                float momentum, float* save_mean, float* save_invstd,
                                long long* num_batches_tracked, unsigned char* __restrict__ bits) {
  constexpr int EPV = Elem<T>::EPV;
  const int VC = C / EPV;
  const long long total = count * VC;
  const long long stride = (long long)gridDim.x * TPB;
  // The conv epilogue's workgroup slots [slots][2C] are summed by every workgroup into LDS (fixed order; no separate
  // fold launch — the host caps the grid so that this stays a small fraction of the loads).
  extern __shared__ float fstat[];   // [2C]
  const float* stats = gstats;
  // The first pair of vectors is requested BEFORE the statistics are folded and the constants derived: the two memory
  // round trips (slots -> LDS -> constants, and x) then overlap instead of following each other — the mid-size
  // layers' launches are a few microseconds of exactly that chain.
  long long i = (long long)blockIdx.x * TPB + threadIdx.x;
  const bool first = FIXED && i + stride < total;
  uint4 pa0 = make_uint4(0, 0, 0, 0), pa1 = pa0, pr0 = pa0, pr1 = pa0;
  if (first) {
    pa0 = *reinterpret_cast<const uint4*>(x + i * EPV);
    pa1 = *reinterpret_cast<const uint4*>(x + (i + stride) * EPV);
    if (res) {
      pr0 = *reinterpret_cast<const uint4*>(res + i * EPV);
      pr1 = *reinterpret_cast<const uint4*>(res + (i + stride) * EPV);
    }
  }
  if (slots > 1) {
    fold_slots_to_lds(gstats, slots, 2 * C, fstat);
    stats = fstat;
  }
  if (blockIdx.x == 0)
    bn_publish(stats, stat_count, C, running_mean, running_var, momentum, eps, save_mean, save_invstd,
               num_batches_tracked);
  // every thread derives mean / invstd of its channels from the raw sums itself (same arithmetic as
  // bn_publish), so nothing waits for workgroup 0
  const float nstat = (float)stat_count;
  float mu[EPV], is[EPV], ga[EPV], be[EPV];
  auto load_consts = [&](int c0) {
#pragma unroll
    for (int j = 0; j < EPV; ++j) {
      const BnStat s = bn_stat(stats, C, c0 + j, nstat, eps);
      mu[j] = s.mean; is[j] = s.invstd; ga[j] = gamma[c0 + j]; be[j] = beta[c0 + j];
    }
  };
  auto finish = [&](long long i, const uint4& raw, const uint4& rv) {
    float f[EPV];
    Elem<T>::unpack(raw, f);
#pragma unroll
    for (int j = 0; j < EPV; ++j) f[j] = bn_affine(f[j], mu[j], is[j], ga[j], be[j]);
    if (res) {
      float r[EPV];
      Elem<T>::unpack(rv, r);
#pragma unroll
      for (int j = 0; j < EPV; ++j) f[j] += r[j];
    }
    if (relu) {
#pragma unroll
      for (int j = 0; j < EPV; ++j) f[j] = fmaxf(f[j], 0.f);
    }
    const uint4 packed = Elem<T>::pack(f);
    *reinterpret_cast<uint4*>(y + i * EPV) = packed;
    if (bits) bits[i] = (unsigned char)relu_bits<T>(packed);   // (common.h: the ReLU mask for the backward, 1/16 of y)
  };
  if (FIXED) {
    if (i < total) load_consts((int)(i % VC) * EPV);
    if (first) {
      finish(i, pa0, pr0);
      finish(i + stride, pa1, pr1);
      i += 2 * stride;
    }
    for (; i + stride < total; i += 2 * stride) {
      const uint4 a0 = *reinterpret_cast<const uint4*>(x + i * EPV);
      const uint4 a1 = *reinterpret_cast<const uint4*>(x + (i + stride) * EPV);
      uint4 r0 = make_uint4(0, 0, 0, 0), r1 = r0;
      if (res) {
        r0 = *reinterpret_cast<const uint4*>(res + i * EPV);
        r1 = *reinterpret_cast<const uint4*>(res + (i + stride) * EPV);
      }
      finish(i, a0, r0);
      finish(i + stride, a1, r1);
    }
  }
  for (; i < total; i += stride) {
    if (!FIXED) load_consts((int)(i % VC) * EPV);
    const uint4 a0 = *reinterpret_cast<const uint4*>(x + i * EPV);
    uint4 r0 = make_uint4(0, 0, 0, 0);
    if (res) r0 = *reinterpret_cast<const uint4*>(res + i * EPV);
    finish(i, a0, r0);
  }
}

// ---- GroupNorm: pass 1 = per-(level, image, group) sum / sumsq; pass 2 = normalise (+ReLU)
// Both passes use one decomposition: blockIdx.y = segment (level, image), blockIdx.x = a run of `pix_per_block` pixels of
// it; a thread keeps one 16-byte channel vector (its groups' constants stay in registers) and walks the run's pixels
// TPB / VC apart, four loads in flight. (Levels are picked with static indices: a dynamically indexed by-value
// argument would be copied to scratch.)
constexpr int GN_NT = 1024;   // threads per workgroup of the GroupNorm passes (16 waves: loads in flight per CU)
struct GnSeg {
  int HW;
  long long row0;
};
__device__ __forceinline__ GnSeg gn_segment(const DasLevels& lv, int seg) {
  const int l = seg / lv.B, b = seg - l * lv.B;
  GnSeg g{0, 0};
  long long start = 0;
#pragma unroll
  for (int i = 0; i < DAS_MAX_LEVELS; ++i) {
    const int hw = i < lv.num_levels ? lv.H[i] * lv.W[i] : 0;
    if (i == l) { g.HW = hw; g.row0 = start + (long long)b * hw; }
    start += (long long)lv.B * hw;
  }
  return g;
}

template <typename T, int NT>
__global__ __launch_bounds__(NT) void gn_stats_kernel(const T* __restrict__ x, DasLevels lv, int C, int ps, int G,
                                                      int pix_per_block, float* __restrict__ stats) {
  constexpr int EPV = Elem<T>::EPV;
  extern __shared__ float part[];  // [PL][width] (common.h: lds_put / lds_fold), folded in place into row 0
  const int seg = blockIdx.y;
  const GnSeg sg = gn_segment(lv, seg);
  const int p0 = blockIdx.x * pix_per_block;
  if (p0 >= sg.HW) return;  // block-uniform
  const int VC = C / EPV, cpg = C / G;
  const bool vec_in_group = cpg % EPV == 0;   // a thread's vector lies inside one group: one pair of sums per thread
  const int width = vec_in_group ? 2 * VC : 2 * C;
  const int v = threadIdx.x % VC, pl = threadIdx.x / VC, PL = NT / VC;
  float s[EPV], q[EPV];
#pragma unroll
  for (int j = 0; j < EPV; ++j) { s[j] = 0.f; q[j] = 0.f; }
  if (pl < PL) {
    const int p1 = min(p0 + pix_per_block, sg.HW);
    const T* base = x + sg.row0 * ps + v * EPV;
    int p = p0 + pl;
    for (; p + 3 * PL < p1; p += 4 * PL) {
      uint4 r[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) r[u] = *reinterpret_cast<const uint4*>(base + (long long)(p + u * PL) * ps);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float f[EPV];
        Elem<T>::unpack(r[u], f);
#pragma unroll
        for (int j = 0; j < EPV; ++j) { s[j] += f[j]; q[j] += f[j] * f[j]; }
      }
    }
    for (; p < p1; p += PL) {
      float f[EPV];
      Elem<T>::unpack(*reinterpret_cast<const uint4*>(base + (long long)p * ps), f);
#pragma unroll
      for (int j = 0; j < EPV; ++j) { s[j] += f[j]; q[j] += f[j] * f[j]; }
    }
    if (vec_in_group) {
      float a = 0.f, c = 0.f;
#pragma unroll
      for (int j = 0; j < EPV; ++j) { a += s[j]; c += q[j]; }
      part[(size_t)pl * width + v] = a;
      part[(size_t)pl * width + VC + v] = c;
    } else {
      lds_put<EPV>(part, width, pl, v * EPV, s);
      lds_put<EPV>(part, width, pl, C + v * EPV, q);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < width; i += NT) part[i] = lds_fold(part, width, PL, i);   // (column i: this thread only)
  __syncthreads();
  const int per = vec_in_group ? cpg / EPV : cpg, half = width / 2;
  for (int g = threadIdx.x; g < G; g += NT) {
    float a = 0.f, c = 0.f;
    for (int j = 0; j < per; ++j) { a += part[g * per + j]; c += part[half + g * per + j]; }
    atomicAdd(&stats[((long long)seg * G + g) * 2], a);
    atomicAdd(&stats[((long long)seg * G + g) * 2 + 1], c);
  }
}

template <typename T, int NT>
__global__ __launch_bounds__(NT) void gn_apply_kernel(const T* __restrict__ x, T* __restrict__ y, DasLevels lv, int C, int ps,
                                                      int G, int pix_per_block, const float* __restrict__ stats,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float eps, int relu) {
  constexpr int EPV = Elem<T>::EPV;
  const int seg = blockIdx.y;
  const GnSeg sg = gn_segment(lv, seg);
  const int p0 = blockIdx.x * pix_per_block;
  if (p0 >= sg.HW) return;
  const int VC = C / EPV, cpg = C / G;
  const int v = threadIdx.x % VC, pl = threadIdx.x / VC, PL = NT / VC;
  if (pl >= PL) return;
  const float inv_n = 1.f / ((float)sg.HW * (float)cpg);
  float mean[EPV], rstd[EPV], ga[EPV], be[EPV];
#pragma unroll
  for (int j = 0; j < EPV; ++j) {
    const int c = v * EPV + j, gi = c / cpg;
    mean[j] = stats[((long long)seg * G + gi) * 2] * inv_n;
    const float var = fmaxf(stats[((long long)seg * G + gi) * 2 + 1] * inv_n - mean[j] * mean[j], 0.f);
    rstd[j] = rsqrtf(var + eps);
    ga[j] = gamma[c]; be[j] = beta[c];
  }
  const int p1 = min(p0 + pix_per_block, sg.HW);
  const long long off = sg.row0 * ps + v * EPV;
  auto norm = [&](uint4 r) {
    float f[EPV];
    Elem<T>::unpack(r, f);
#pragma unroll
    for (int j = 0; j < EPV; ++j) {
      const float o = gn_affine(f[j], mean[j], rstd[j], ga[j], be[j]);
      f[j] = relu ? fmaxf(o, 0.f) : o;
    }
    return Elem<T>::pack(f);
  };
  int p = p0 + pl;
  for (; p + 3 * PL < p1; p += 4 * PL) {
    uint4 r[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) r[u] = *reinterpret_cast<const uint4*>(x + off + (long long)(p + u * PL) * ps);
#pragma unroll
    for (int u = 0; u < 4; ++u) *reinterpret_cast<uint4*>(y + off + (long long)(p + u * PL) * ps) = norm(r[u]);
  }
  for (; p < p1; p += PL)
    *reinterpret_cast<uint4*>(y + off + (long long)p * ps) = norm(*reinterpret_cast<const uint4*>(x + off + (long long)p * ps));
}

// ---- the apply pass of the LARGE layers (tensors of >= ~100 MB: beyond the 256 MiB infinity cache with their residual)
// A plain 2-read-1-write pass over 218 MB tensors reaches 5.9 TB/s on this GPU when it is launched as ~13 000 small
// workgroups, four 16-byte vectors per thread, all loads issued before the first use — and 5.1 TB/s as a 2048-workgroup
// grid-stride loop (tools/dev/probe/triad_probe.hip). bn_apply_kernel above cannot take the many-small-workgroups shape:
// every workgroup folds the conv epilogue's 16 statistic slots first (32 KB of L2 reads for 256 channels against 49 KB of
// payload per workgroup at that grid), which is why its grid is capped at 2048. So for these layers the fold is a launch
// of its own (3 us against a pass of 110-150 us), into a per-stream scratch vector, and the pass is this kernel: the
// per-channel constants are computed ONCE per workgroup (one channel per thread, into LDS) instead of by every thread
// for its eight channels, the arithmetic per element is bn_apply_kernel's.
template <typename T, int VPT>
__global__ __launch_bounds__(TPB) void bn_apply_stream_kernel(const T* __restrict__ x, T* __restrict__ y, long long count,
                                                              int C, const float* __restrict__ stats, long long stat_count,
                                                              float eps, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, const T* __restrict__ res,
                                                              int relu, float* running_mean, float* running_var,
                                                              float momentum, float* save_mean, float* save_invstd,
                                                              long long* num_batches_tracked,
                                                              unsigned char* __restrict__ bits, int ntm) {
  constexpr int EPV = Elem<T>::EPV;
  const bool nt_x = ntm & 1, nt_r = ntm & 2, nt_y = ntm & 4;   // (das_tuning key bn.nt_fwd)
  const int VC = C / EPV;                 // (host: TPB % VC == 0 — a thread keeps its channels for every vector it takes)
  extern __shared__ float cst[];          // [4][C]: mean, invstd, gamma, beta
  const float nstat = (float)stat_count;
  for (int c = threadIdx.x; c < C; c += TPB) {
    const BnStat s = bn_stat(stats, C, c, nstat, eps);
    cst[c] = s.mean; cst[C + c] = s.invstd; cst[2 * C + c] = gamma[c]; cst[3 * C + c] = beta[c];
  }
  if (blockIdx.x == 0)
    bn_publish(stats, stat_count, C, running_mean, running_var, momentum, eps, save_mean, save_invstd,
               num_batches_tracked);
  __syncthreads();
  const int c0 = (threadIdx.x % VC) * EPV;
  float mu[EPV], is[EPV], ga[EPV], be[EPV];
#pragma unroll
  for (int j = 0; j < EPV; ++j) {
    mu[j] = cst[c0 + j]; is[j] = cst[C + c0 + j]; ga[j] = cst[2 * C + c0 + j]; be[j] = cst[3 * C + c0 + j];
  }
  const long long total = count * VC;
  const long long base = (long long)blockIdx.x * (TPB * VPT) + threadIdx.x;
  uint4 a[VPT], r[VPT];
#pragma unroll
  for (int u = 0; u < VPT; ++u) {
    const long long i = base + u * TPB;
    a[u] = make_uint4(0, 0, 0, 0);
    r[u] = a[u];
    if (i < total) {
      a[u] = ld16(x + i * EPV, nt_x);
      if (res) r[u] = ld16(res + i * EPV, nt_r);
    }
  }
#pragma unroll
  for (int u = 0; u < VPT; ++u) {
    const long long i = base + u * TPB;
    if (i >= total) break;
    float f[EPV];
    Elem<T>::unpack(a[u], f);
#pragma unroll
    for (int j = 0; j < EPV; ++j) f[j] = bn_affine(f[j], mu[j], is[j], ga[j], be[j]);
    if (res) {
      float rr[EPV];
      Elem<T>::unpack(r[u], rr);
#pragma unroll
      for (int j = 0; j < EPV; ++j) f[j] += rr[j];
    }
    if (relu) {
#pragma unroll
      for (int j = 0; j < EPV; ++j) f[j] = fmaxf(f[j], 0.f);
    }
    const uint4 packed = Elem<T>::pack(f);
    st16(y + i * EPV, packed, nt_y);
    if (bits) bits[i] = (unsigned char)relu_bits<T>(packed);
  }
}
}  // namespace

extern "C" int das_bn_train_apply(const void* x, void* y, int dtype, long long count, int C, const float* stats,
                                  const float* gamma, const float* beta, float* running_mean, float* running_var,
                                  float momentum, float eps, const void* residual, int relu, float* save_mean,
                                  float* save_invstd, long long* num_batches_tracked, long long stat_count,
                                  int stats_slots, void* relu_bits_out, void* stream) {
  DAS_PROF(stream);
  unsigned char* bits = (unsigned char*)relu_bits_out;
  if (!x || !stats || !gamma || !beta || !save_mean || !save_invstd || C % 8 || count <= 0) return DAS_ERR_ARG;
  if (dtype != DAS_BF16 && dtype != DAS_F32) return DAS_ERR_ARG;
  if (stat_count != 0 && stat_count < count) return DAS_ERR_ARG;
  if (stats_slots < 0 || stats_slots > 64) return DAS_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const long long nstat = stat_count ? stat_count : count;
  // y == NULL: finalize only — mean / invstd published, running statistics and the batch counter advanced, nothing
  // normalised (a layer whose output has no consumer: the statistics are still part of the state dict)
  if (!y) {
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 63) / 64), dim3(64), 0, s, stats, stats_slots < 1 ? 1 : stats_slots, nstat, C,
                       running_mean, running_var, momentum, eps, save_mean, save_invstd, num_batches_tracked);
    DAS_CHECK_LAUNCH();
    dastune::note_kernel("bn_finalize_kernel");
    return DAS_OK;
  }
  const int vc = C / (dtype == DAS_BF16 ? 8 : 4);
  // at least eight vectors per thread (the per-thread mean / invstd / gamma / beta set-up is ~60 instructions and 32
  // loads): the mid-size layers ran at 1.7...3 TB/s with one vector per thread (1024 channels at 32x52: 37 -> 22 us)
  const int vpt = std::max(1, (int)dastune::get(dastune::BN_VPT));
  int grid = count == 0 ? 1 : std::max(1, std::min(grid_for(count * vc), (int)((count * vc + (long long)TPB * vpt - 1) / ((long long)TPB * vpt))));
  const int nslots = stats_slots < 1 ? 1 : stats_slots;
  // large layers: fold launch + many-small-workgroups pass (see bn_apply_stream_kernel)
  const long long stream_from = dastune::get(dastune::BN_STREAM_MINBYTES);
  const long long xbytes = count * C * (dtype == DAS_BF16 ? 2 : 4);
  if (stream_from > 0 && xbytes >= stream_from && TPB % vc == 0 && count * vc / (TPB * 4) < 0x7fffffffLL) {
    const float* folded = stats;
    if (nslots > 1) {
      float* ws = dasws::get(dasws::BN_FOLD, s, 2 * (size_t)C * sizeof(float), 64 << 10);
      if (!ws) return DAS_ERR_LAUNCH;
      hipLaunchKernelGGL(fold_slots_kernel, dim3((2 * C + 255) / 256), dim3(256), 0, s, stats, nslots, 2 * C, ws);
      DAS_CHECK_LAUNCH();
      folded = ws;
    }
    constexpr int VPT = DAS_BN_STREAM_VPT;
    const int sgrid = (int)((count * vc + TPB * VPT - 1) / (TPB * VPT));
    const size_t ssm = 4 * (size_t)C * sizeof(float);
#define DAS_BN_STREAM(T)                                                                                              \
  hipLaunchKernelGGL((bn_apply_stream_kernel<T, VPT>), dim3(sgrid), dim3(TPB), ssm, s, (const T*)x, (T*)y, count, C,   \
                     folded, nstat, eps, gamma, beta, (const T*)residual, relu, running_mean, running_var, momentum,  \
                     save_mean, save_invstd, num_batches_tracked, bits, (int)dastune::get(dastune::BN_NT_FWD))
    if (dtype == DAS_BF16) DAS_BN_STREAM(bf16_t); else DAS_BN_STREAM(float);
#undef DAS_BN_STREAM
    DAS_CHECK_LAUNCH();
    dastune::note_kernel("bn_apply_stream_kernel");
    return DAS_OK;
  }
  dastune::note_kernel("bn_apply_kernel");
  if (nslots > 1) grid = std::min(grid, 2048);   // (every workgroup folds the slots first)
  const bool fixed = ((long long)grid * TPB) % vc == 0;
  const size_t sm = nslots > 1 ? 2 * (size_t)C * sizeof(float) : 0;
#define DAS_BN_APPLY(T, F)                                                                                          \
  hipLaunchKernelGGL((bn_apply_kernel<T, F>), dim3(grid), dim3(TPB), sm, s, (const T*)x, (T*)y, count, C, stats, nslots, nstat, \
                     eps, gamma, beta, (const T*)residual, relu, running_mean, running_var, momentum, save_mean,      \
                     save_invstd, num_batches_tracked, bits)
  if (dtype == DAS_BF16) {
    if (fixed) DAS_BN_APPLY(bf16_t, true); else DAS_BN_APPLY(bf16_t, false);
  } else {
    if (fixed) DAS_BN_APPLY(float, true); else DAS_BN_APPLY(float, false);
  }
#undef DAS_BN_APPLY
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}

extern "C" int das_bn_finalize_many(const DasBnFinalize* layers, int n, void* stream) {
  DAS_PROF(stream);
  if (!layers || n < 1 || n > 4) return DAS_ERR_ARG;
  BnFinalizeMany a;
  int maxc = 0;
  for (int i = 0; i < 4; ++i) {
    a.l[i] = layers[i < n ? i : 0];
    if (i >= n) continue;
    const DasBnFinalize& f = layers[i];
    if (!f.stats || !f.save_mean || !f.save_invstd || f.C < 1 || f.count < 1 || f.stats_slots < 0 || f.stats_slots > 64 ||
        ((f.running_mean == nullptr) != (f.running_var == nullptr)))
      return DAS_ERR_ARG;
    if (a.l[i].stats_slots < 1) a.l[i].stats_slots = 1;
    maxc = std::max(maxc, f.C);
  }
  hipLaunchKernelGGL(bn_finalize_many_kernel, dim3((maxc + 63) / 64, n), dim3(64), 0, (hipStream_t)stream, a);
  DAS_CHECK_LAUNCH();
  dastune::note_kernel("bn_finalize_kernel");
  return DAS_OK;
}

extern "C" int das_groupnorm_nhwc(const void* x, void* y, int dtype, const DasLevels* lv, int C, int pix_stride,
                                  int G, const float* gamma, const float* beta, float eps, int relu, float* stats_ws,
                                  int ws_zeroed, void* stream) {
  DAS_PROF(stream);
  if (!x || !y || !gamma || !beta || !stats_ws || !lv_valid(lv) || C % 8 || C % G || pix_stride % 8 || C > 2048)
    return DAS_ERR_ARG;
  const int epv = dtype == DAS_BF16 ? 8 : 4;
  if ((C / epv) > GN_NT) return DAS_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int nseg = lv->num_levels * lv->B;
  if (!ws_zeroed && hipMemsetAsync(stats_ws, 0, sizeof(float) * 2 * nseg * G, s) != hipSuccess) return DAS_ERR_LAUNCH;
  int maxhw = 0;
  for (int l = 0; l < lv->num_levels; ++l) maxhw = std::max(maxhw, lv->H[l] * lv->W[l]);
  // enough blocks to fill the chip, at least 64 pixels per block
  int chunks = (256 * 4 + lv->B - 1) / lv->B;
  int ppb = (maxhw + chunks - 1) / chunks;
  // (64 pixels per workgroup left the pass launch / latency bound: 42 us for 72 MB; 256: infer +3 %)
  const int ppb_min = gn_ppb_min(dastune::get(dastune::GN_PPB), nseg, maxhw);   // minimum pixels per workgroup
  if (ppb < ppb_min) ppb = ppb_min;
  chunks = (maxhw + ppb - 1) / ppb;
  // LDS of the statistics pass: [pixel lanes][2 x (vectors | channels)] partial sums
  const int vc = C / epv;
  const size_t lds = (size_t)(GN_NT / vc) * 2 * (((C / G) % epv == 0) ? vc : C) * sizeof(float);
  if (lds > 48 * 1024) {
    const void* k = dtype == DAS_BF16 ? (const void*)gn_stats_kernel<bf16_t, GN_NT> : (const void*)gn_stats_kernel<float, GN_NT>;
    if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return DAS_ERR_LAUNCH;
  }
  if (dtype == DAS_BF16) {
    hipLaunchKernelGGL((gn_stats_kernel<bf16_t, GN_NT>), dim3(chunks, nseg), dim3(GN_NT), lds, s,
                       (const bf16_t*)x, *lv, C, pix_stride, G, ppb, stats_ws);
    hipLaunchKernelGGL((gn_apply_kernel<bf16_t, GN_NT>), dim3(chunks, nseg), dim3(GN_NT), 0, s, (const bf16_t*)x, (bf16_t*)y,
                       *lv, C, pix_stride, G, ppb, stats_ws, gamma, beta, eps, relu);
  } else if (dtype == DAS_F32) {
    hipLaunchKernelGGL((gn_stats_kernel<float, GN_NT>), dim3(chunks, nseg), dim3(GN_NT), lds, s,
                       (const float*)x, *lv, C, pix_stride, G, ppb, stats_ws);
    hipLaunchKernelGGL((gn_apply_kernel<float, GN_NT>), dim3(chunks, nseg), dim3(GN_NT), 0, s, (const float*)x, (float*)y,
                       *lv, C, pix_stride, G, ppb, stats_ws, gamma, beta, eps, relu);
  } else {
    return DAS_ERR_ARG;
  }
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}
