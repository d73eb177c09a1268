// Fused per-image decode: sigmoid scores -> threshold -> per-level top-k -> gather + affine
// -> stable descending sort -> greedy OKS-NMS (or soft OKS-NMS) -> first nms_post survivors.
// (das_head.py:690-796 `_get_poses_single`, pose_nms.py:51-126.) One 1024-thread workgroup per
// image; candidate keys live in LDS (128 KiB) whenever they fit. Integer/ordering work is exact:
// key = (score bits << 32) | ~flat_index, so "higher score first, then lower flat index".
// Thresholding before the per-level top-k is equivalent to the reference's order (top-k first,
// threshold after concat) because the threshold is applied to the same score.
// No size limits (round 6; the reference has none, das_head.py:716-723): what is bounded by LDS is the number of
// locations ABOVE THE THRESHOLD, not the frame — a level of any size whose candidates fit is sorted in LDS as before; a
// level with more than LDS_KEYS candidates finds its nms_pre-th key by an 8-pass radix select that re-scores the level
// (no storage); more than SUP_LDS / LDS_KEYS candidates in total (nms_pre <= 0 or huge) keep the suppression flags and
// the globally sorted keys in the per-image workspace instead of LDS (a bitonic network over global memory: slow, exact).
#include "common.h"
#include "prof.h"

namespace {
constexpr int TPB = 1024;
constexpr int LDS_KEYS = 16384;  // candidate keys that fit in LDS (128 KiB of u64 keys)
constexpr int SUP_LDS = 4096;    // suppression flags of the candidates, behind the keys in LDS
// Greedy OKS-NMS, two ways. Up to PAIR_MAX candidates (the usual case: ~150 pass the score threshold) every pair's
// "iou > thr" is evaluated up front, one thread per pair — the f64 exp chain of a pair is the cost, 15-21 of them, and
// evaluating them inside the greedy loop keeps 150 of 1024 threads busy for up to nms_post sequential rounds (250 us of
// a 458 us launch at B = 8). The result is a bit matrix in LDS, and the greedy walk is done by ONE wave that keeps the
// suppressed set in registers (a word per lane) and ORs one matrix row per kept pose: no workgroup barrier per round.
// More candidates: the round-by-round form (a matrix of cap^2 would not pay).
constexpr int PAIR_MAX = 768;

__host__ __device__ inline long long pow2_ceil(long long n) {
  long long p = 1;
  while (p < n) p <<= 1;
  return p;
}
// per image: merged keys [pow2_ceil(cap)] u64 (the power of two: a global-memory bitonic sort pads to it), kx / ky / kz
// [cap * J] f32, area [cap], centers [cap * 3], soft-NMS scores [cap] f32, suppression flags [cap] u8
__host__ __device__ inline long long ws_bytes_per_image(int cap, int J) {
  long long per = pow2_ceil(cap) * 8 + (long long)cap * J * 4 * 3 + (long long)cap * 4 + (long long)cap * 12 + (long long)cap * 4 + cap;
  return (per + 255) / 256 * 256;
}

// pose_nms.py:51-90 for one pair: f64 arithmetic as numpy does, the mean rounded to f32 before the compare
__device__ __forceinline__ float oks_value_f64(const float* kx, const float* ky, const float* area, int a, int c, int J) {
#pragma clang fp contract(off)
  const double denom = (double)((area[a] + area[c]) / 2.f) + 2.220446049250313e-16;
  double acc = 0.0;
  for (int j = 0; j < J; ++j) {
    const float dx = kx[(size_t)c * J + j] - kx[(size_t)a * J + j];
    const float dy = ky[(size_t)c * J + j] - ky[(size_t)a * J + j];
    double var = 0.0256;  // (0.08*2)^2
    if (J == 17) {
      const double sg[17] = {.026, .025, .025, .035, .035, .079, .079, .072, .072, .062, .062, .107, .107, .087,
                             .087, .089, .089};
      var = (sg[j] * 2) * (sg[j] * 2);
    }
    const double e = (double)(dx * dx + dy * dy) / var / denom / 2.0;
    acc += exp(-e);
  }
  return (float)(acc / (double)J);
}
__device__ __forceinline__ bool oks_above_f64(const float* kx, const float* ky, const float* area, int a, int c, int J,
                                              float thr32) {
  return !(oks_value_f64(kx, ky, area, a, c, J) <= thr32);
}
// The decision "oks > thr" only: an f32 evaluation first (an f64 exp costs ~10x an f32 one and there are 15-21 per
// pair); its error is below 1e-5 (each term in (0, 1], a few ulp each), so beyond a margin of 1e-3 around the
// threshold the f32 answer IS the f64 answer; inside the margin the exact f64 form decides.
__device__ __forceinline__ bool oks_above(const float* kx, const float* ky, const float* area, int a, int c, int J,
                                          float thr32) {
  const float denom = (area[a] + area[c]) * 0.5f + 1e-30f;
  float acc = 0.f;
  for (int j = 0; j < J; ++j) {
    const float dx = kx[(size_t)c * J + j] - kx[(size_t)a * J + j];
    const float dy = ky[(size_t)c * J + j] - ky[(size_t)a * J + j];
    float var = 0.0256f;
    if (J == 17) {
      const float sg[17] = {.026f, .025f, .025f, .035f, .035f, .079f, .079f, .072f, .072f, .062f, .062f, .107f, .107f,
                            .087f, .087f, .089f, .089f};
      var = (sg[j] * 2) * (sg[j] * 2);
    }
    acc += expf(-(dx * dx + dy * dy) / var / denom * 0.5f);
  }
  const float iou = acc / (float)J;
  if (fabsf(iou - thr32) > 1e-3f && isfinite(iou)) return iou > thr32;
  return oks_above_f64(kx, ky, area, a, c, J, thr32);
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// (keys: LDS or global memory — a generic pointer; P a power of two)
__device__ void bitonic_sort_desc(unsigned long long* keys, int P) {
  for (int k = 2; k <= P; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < P; i += TPB) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long a = keys[i], b = keys[ixj];
          const bool desc = ((i & k) == 0);
          if (desc ? (a < b) : (a > b)) { keys[i] = b; keys[ixj] = a; }
        }
      }
      __syncthreads();
    }
  }
}

__global__ __launch_bounds__(TPB) void decode_kernel(DasDecodeDesc d, float* __restrict__ out_scores,
                                                     float* __restrict__ out_poses, float* __restrict__ out_centers,
                                                     int* __restrict__ out_index, int* __restrict__ out_count,
                                                     char* __restrict__ ws, int cap) {
#pragma clang fp contract(off)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);
  __shared__ int s_n, s_sel, s_kept;
  const int b = blockIdx.x, tid = threadIdx.x, J = d.J;
  // per-image workspace: merged keys [cap] u64, kx/ky [cap*J] f32, area [cap] f32, center [cap*3], z [cap*J],
  // suppressed [cap] u8, keep [nms_post] int
  char* w = ws + (size_t)b * ws_bytes_per_image(cap, J);
  unsigned long long* mkeys = reinterpret_cast<unsigned long long*>(w);
  float* kx = reinterpret_cast<float*>(mkeys + pow2_ceil(cap));
  float* ky = kx + (size_t)cap * J;
  float* kz = ky + (size_t)cap * J;
  float* area = kz + (size_t)cap * J;
  float* cen = area + cap;
  float* gssc = cen + (size_t)cap * 3;                                                         // soft-NMS scores when cap > SUP_LDS
  // suppression flags: in LDS behind the keys when the candidate capacity allows, else in the workspace
  unsigned char* sup = cap <= SUP_LDS ? reinterpret_cast<unsigned char*>(smem) + (size_t)LDS_KEYS * 8
                                      : reinterpret_cast<unsigned char*>(gssc + cap);

  // ---- candidates above the score threshold. Usual case: no level has more of them than nms_pre (a few hundred
  // pass 0.07), so the per-level top-k keeps everything and ONE sweep over all levels' points collects them (their
  // order is irrelevant: the global sort below defines it). Otherwise: level by level with the top-k sort.
  __shared__ int s_lv[DAS_MAX_LEVELS];
  if (tid < DAS_MAX_LEVELS) s_lv[tid] = 0;
  if (tid == 0) s_n = 0;
  __syncthreads();
  int total = 0;
  {
    int npts_all = 0;
    for (int l = 0; l < d.num_levels; ++l) npts_all += d.H[l] * d.W[l];
    for (int i0 = 0; i0 < npts_all; i0 += 4 * TPB) {
      float sc[4];
      int lv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {     // four points per thread in flight
        const int i = i0 + u * TPB + tid;
        sc[u] = -1.f;
        lv[u] = 0;
        if (i < npts_all) {
          int l = 0, base = 0;
          while (l + 1 < d.num_levels && i >= base + d.H[l] * d.W[l]) { base += d.H[l] * d.W[l]; ++l; }
          const int loc = i - base, npts = d.H[l] * d.W[l];
          const float a = d.cls[l][((size_t)b * npts + loc) * d.cls_ps[l]];
          const float c = d.ctr[l][((size_t)b * npts + loc) * d.ctr_ps[l]];
          sc[u] = sigmoidf_(a) * sigmoidf_(c);
          lv[u] = l;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (sc[u] > d.score_thr) {
          const int pos = atomicAdd(&s_n, 1);
          atomicAdd(&s_lv[lv[u]], 1);
          if (pos < LDS_KEYS)   // (more than fit: counted only — the level loop below re-scores)
            keys[pos] = ((unsigned long long)__float_as_uint(sc[u]) << 32) | (unsigned)(~(unsigned)(i0 + u * TPB + tid));
        }
      }
    }
    __syncthreads();
    bool simple = s_n <= LDS_KEYS;
    for (int l = 0; l < d.num_levels; ++l)
      if (d.nms_pre > 0 && d.H[l] * d.W[l] > d.nms_pre && s_lv[l] > d.nms_pre) simple = false;   // (uniform)
    if (simple) {
      total = s_n;
      for (int i = tid; i < total; i += TPB) mkeys[i] = keys[i];
      __syncthreads();
    } else {
      // level by level. n_lv = the level's candidates above the threshold (counted by the sweep). A level that keeps all
      // of them (no top-k configured for it, or not more than nms_pre) streams its keys straight into the merged list;
      // a level that must be cut to its nms_pre best sorts its candidates in LDS when they fit, and otherwise finds the
      // nms_pre-th largest key by a radix select over re-computed scores (keys are distinct, so "key >= the nms_pre-th"
      // selects exactly nms_pre of them) — no storage proportional to the level.
      __shared__ int s_hist[256];
      __shared__ unsigned long long s_prefix;
      int point_base = 0;
      for (int l = 0; l < d.num_levels; ++l) {
        const int npts = d.H[l] * d.W[l];
        const int n_lv = s_lv[l];
        const bool cut = d.nms_pre > 0 && npts > d.nms_pre && n_lv > d.nms_pre;
        const bool in_lds = cut && n_lv <= LDS_KEYS;
        const float* cls = d.cls[l] + (size_t)b * npts * d.cls_ps[l];
        const float* ctr = d.ctr[l] + (size_t)b * npts * d.ctr_ps[l];
        unsigned long long kth = 0ull;   // keep keys >= kth
        if (cut && !in_lds) {
          // radix select, most significant byte first: after pass p the top (p + 1) bytes of the nms_pre-th largest key
          int want = d.nms_pre;          // rank (1-based, descending) of the key looked for among those matching the prefix
          unsigned long long prefix = 0ull;
          for (int pass = 0; pass < 8; ++pass) {
            const int shift = 56 - 8 * pass;
            __syncthreads();
            for (int i = tid; i < 256; i += TPB) s_hist[i] = 0;
            __syncthreads();
            for (int i = tid; i < npts; i += TPB) {
              const float sv = sigmoidf_(cls[(size_t)i * d.cls_ps[l]]) * sigmoidf_(ctr[(size_t)i * d.ctr_ps[l]]);
              if (sv > d.score_thr) {
                const unsigned long long k = ((unsigned long long)__float_as_uint(sv) << 32) | (unsigned)(~(unsigned)(point_base + i));
                if (pass == 0 || (k >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&s_hist[(int)((k >> shift) & 0xff)], 1);
              }
            }
            __syncthreads();
            if (tid == 0) {
              int acc = 0, bin = 255;
              for (; bin > 0; --bin) {
                if (acc + s_hist[bin] >= want) break;
                acc += s_hist[bin];
              }
              s_sel = want - acc;        // rank inside the chosen bin
              s_prefix = prefix | ((unsigned long long)bin << shift);
            }
            __syncthreads();
            want = s_sel;
            prefix = s_prefix;
          }
          kth = prefix;
        }
        __syncthreads();
        if (tid == 0) s_n = 0;
        __syncthreads();
        unsigned long long* dst = in_lds ? keys : mkeys + total;
        for (int i = tid; i < npts; i += TPB) {
          const float sv = sigmoidf_(cls[(size_t)i * d.cls_ps[l]]) * sigmoidf_(ctr[(size_t)i * d.ctr_ps[l]]);
          if (sv > d.score_thr) {
            const unsigned long long k = ((unsigned long long)__float_as_uint(sv) << 32) | (unsigned)(~(unsigned)(point_base + i));
            if (k >= kth) dst[atomicAdd(&s_n, 1)] = k;
          }
        }
        __syncthreads();
        int n = s_n;
        if (in_lds) {
          int P = 1;
          while (P < n) P <<= 1;
          for (int i = n + tid; i < P; i += TPB) keys[i] = 0ull;
          __syncthreads();
          bitonic_sort_desc(keys, P);
          n = d.nms_pre;
          for (int i = tid; i < n; i += TPB) mkeys[total + i] = keys[i];
        }
        total += n;
        point_base += npts;
        __syncthreads();
      }
    }
  }

  // ---- global order: score desc, flat index asc
  if (total <= TPB) {
    // rank sort: keys are distinct (the location index is part of them), so a key's position is the number of larger
    // keys — one pass over the LDS copy per thread, one barrier (the bitonic network needs log^2 of them)
    unsigned long long mine = 0ull;
    int rank = 0;
    if (tid < total) mine = mkeys[tid];
    __syncthreads();
    for (int i = tid; i < total; i += TPB) keys[i] = mkeys[i];
    __syncthreads();
    if (tid < total)
      for (int j = 0; j < total; ++j) rank += keys[j] > mine ? 1 : 0;
    __syncthreads();
    if (tid < total) keys[rank] = mine;
    __syncthreads();
  } else if (total <= LDS_KEYS) {
    int P = 1;
    while (P < total) P <<= 1;
    for (int i = tid; i < P; i += TPB) keys[i] = i < total ? mkeys[i] : 0ull;
    __syncthreads();
    bitonic_sort_desc(keys, P);
  } else {            // more candidates than LDS holds: the same network over the workspace copy (padded to a power of two)
    int P = 1;
    while (P < total) P <<= 1;
    for (int i = total + tid; i < P; i += TPB) mkeys[i] = 0ull;
    __syncthreads();
    bitonic_sort_desc(mkeys, P);
  }
  const unsigned long long* skeys = total <= LDS_KEYS ? keys : mkeys;   // the candidates in their final order

  // ---- gather + affine per candidate (das_head.py:725-743): 32 lanes per candidate, lane j = joint j (a thread per
  // candidate walking its joints pays one memory round trip per joint: 40 us of the launch at J = 21)
  const float sx = d.scale_factor[b * 2], sy = d.scale_factor[b * 2 + 1];
  const float zs = sqrtf(sx * sy);
  // NMS operands in LDS when they fit behind the keys (the usual ~150 candidates do): keys[0 .. total) | pair bit
  // matrix [n][W] | x [n][J] | y [n][J] | area [n]
  const int nW = (total + 31) >> 5;
  const size_t pm_off = (size_t)PAIR_MAX * 8, xy_off = pm_off + (size_t)total * nW * 4;
  const bool pairwise = !d.nms_soft && total <= PAIR_MAX && J <= 32 &&
                        xy_off + (size_t)total * J * 8 + (size_t)total * 4 <= (size_t)LDS_KEYS * 8;   // (uniform)
  float* skx = reinterpret_cast<float*>(smem + xy_off);
  float* sky = skx + (size_t)total * J;
  float* sarea = sky + (size_t)total * J;
  if (J <= 32) {
    const int sub = tid & 31;
    for (int c = tid >> 5; c < total; c += TPB / 32) {
      const unsigned flat = ~(unsigned)(skeys[c] & 0xffffffffull);
      int l = 0, base = 0;
      while (l + 1 < d.num_levels && (int)flat >= base + d.H[l] * d.W[l]) { base += d.H[l] * d.W[l]; ++l; }
      const int loc = (int)flat - base, Wl = d.W[l];
      const int st = d.stride[l];
      const float ptx = (float)((loc % Wl) * st + st / 2), pty = (float)((loc / Wl) * st + st / 2);
      const float* pp = d.pose[l] + ((size_t)b * d.H[l] * Wl + loc) * d.pose_ps[l];
      const float depth = pp[2] * zs;
      if (sub < 3) cen[c * 3 + sub] = sub == 0 ? (ptx - pp[0]) / sx : sub == 1 ? (pty - pp[1]) / sy : depth;
      float mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY;
      if (sub < J) {
        const float x = (pp[3 + 3 * sub] + ptx) / sx, y = (pp[4 + 3 * sub] + pty) / sy;
        kx[(size_t)c * J + sub] = x;
        ky[(size_t)c * J + sub] = y;
        kz[(size_t)c * J + sub] = pp[5 + 3 * sub] + depth;
        if (pairwise) { skx[(size_t)c * J + sub] = x; sky[(size_t)c * J + sub] = y; }
        mnx = mxx = x;
        mny = mxy = y;
      }
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) {   // min / max are exact: any reduction order gives the reference's value
        mnx = fminf(mnx, __shfl_xor(mnx, o, 64)); mxx = fmaxf(mxx, __shfl_xor(mxx, o, 64));
        mny = fminf(mny, __shfl_xor(mny, o, 64)); mxy = fmaxf(mxy, __shfl_xor(mxy, o, 64));
      }
      if (sub == 0) {
        const float a = (mxx - mnx) * (mxy - mny);
        area[c] = a;
        if (pairwise) sarea[c] = a;
        sup[c] = 0;
      }
    }
  } else {
    for (int c = tid; c < total; c += TPB) {
      const unsigned flat = ~(unsigned)(skeys[c] & 0xffffffffull);
      int l = 0, base = 0;
      while (l + 1 < d.num_levels && (int)flat >= base + d.H[l] * d.W[l]) { base += d.H[l] * d.W[l]; ++l; }
      const int loc = (int)flat - base, Wl = d.W[l];
      const int st = d.stride[l];
      const float ptx = (float)((loc % Wl) * st + st / 2), pty = (float)((loc / Wl) * st + st / 2);
      const float* pp = d.pose[l] + ((size_t)b * d.H[l] * Wl + loc) * d.pose_ps[l];
      const float depth = pp[2] * zs;
      cen[c * 3 + 0] = (ptx - pp[0]) / sx;
      cen[c * 3 + 1] = (pty - pp[1]) / sy;
      cen[c * 3 + 2] = depth;
      float mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY;
      for (int j = 0; j < J; ++j) {
        const float x = (pp[3 + 3 * j] + ptx) / sx, y = (pp[4 + 3 * j] + pty) / sy;
        kx[(size_t)c * J + j] = x;
        ky[(size_t)c * J + j] = y;
        kz[(size_t)c * J + j] = pp[5 + 3 * j] + depth;
        mnx = fminf(mnx, x); mxx = fmaxf(mxx, x);
        mny = fminf(mny, y); mxy = fmaxf(mxy, y);
      }
      area[c] = (mxx - mnx) * (mxy - mny);
      sup[c] = 0;
    }
  }
  if (tid == 0) s_kept = 0;
  __syncthreads();

  // ---- greedy OKS-NMS (pose_nms.py:92-126)
  const float thr32 = d.nms_thr;
  if (d.nms_soft) {
    // Soft OKS-NMS (pose_nms.py:128-194, the `nms_type != 'hard'` branch of das_head.py:784-790): nothing is dropped;
    // each round takes the best remaining candidate and rescales every other remaining score by exp(-oks^2 / thr)
    // (f32 arithmetic on the f32 oks values, as numpy does with these arrays), until nms_post are taken. The
    // reference re-sorts the remaining scores every round; only their maximum matters for the next round, so a
    // workgroup-wide arg-max replaces the sort (ties: the candidate that came first in the initial order).
    float* ssc = cap <= SUP_LDS ? reinterpret_cast<float*>(smem + (size_t)LDS_KEYS * 8 - (size_t)SUP_LDS * 4) : gssc;   // [total]
    __shared__ unsigned long long s_red[TPB / 64];
    for (int c = tid; c < total; c += TPB) ssc[c] = __uint_as_float((unsigned)(skeys[c] >> 32));
    __syncthreads();
    const int maxd = total < d.nms_post ? total : d.nms_post;
    int kept = 0;
    while (kept < maxd) {
      unsigned long long best = 0ull;   // (score bits << 32) | ~slot: the scores are >= 0, so bit order is value order
      for (int c = tid; c < total; c += TPB)
        if (!sup[c]) {
          const unsigned long long k = ((unsigned long long)__float_as_uint(ssc[c]) << 32) | (unsigned)(~(unsigned)c);
          best = k > best ? k : best;
        }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(best, o, 64);
        best = other > best ? other : best;
      }
      if ((tid & 63) == 0) s_red[tid >> 6] = best;
      __syncthreads();
      best = s_red[0];
#pragma unroll
      for (int i = 1; i < TPB / 64; ++i) best = s_red[i] > best ? s_red[i] : best;
      const int sel = (int)~(unsigned)(best & 0xffffffffull);
      if (tid == 0) out_index[(size_t)b * d.nms_post + kept] = sel;   // slot id, remapped below
      ++kept;
      for (int c = tid; c < total; c += TPB) {
        if (c == sel || sup[c]) continue;
        const float iou = oks_value_f64(kx, ky, area, sel, c, J);
        ssc[c] = ssc[c] * (float)exp((double)(-(iou * iou) / thr32));
      }
      __syncthreads();
      if (tid == 0) sup[sel] = 1;
      __syncthreads();
    }
    if (tid == 0) s_kept = kept;
    __syncthreads();
  } else if (pairwise) {
    // "suppresses" bit matrix in LDS: only keys[0 .. total) are still needed, the rest of the key array is free.
    // Row r = candidates c > r with oks(r, c) > thr, W words per row.
    const int n = total, W = nW, half = (n + 1) / 2;
    unsigned* pm = reinterpret_cast<unsigned*>(smem + pm_off);                 // [n][W]
    for (int i = tid; i < n * W; i += TPB) pm[i] = 0u;
    __syncthreads();
    // the upper triangle folded into a (n + 1) / 2 x n rectangle: (r, c > r) as is, (r, c < r) mirrored; one thread per
    // pair evaluates its f64 exp chain (the whole cost of the NMS)
    for (int idx = tid; idx < half * n; idx += TPB) {
      int r = idx / n, c = idx - r * n;
      if (c == r) continue;
      const bool mirrored = c < r;
      if (mirrored) { r = n - 1 - r; c = n - 1 - c; }
      if (mirrored && (n & 1) && r == half - 1) continue;   // (middle row of an odd n: already covered un-mirrored)
      if (oks_above(skx, sky, sarea, r, c, J, thr32)) atomicOr(&pm[r * W + (c >> 5)], 1u << (c & 31));
    }
    __syncthreads();
    // the greedy walk: ONE wave, no workgroup barriers; lane w keeps word w of the suppressed set
    if (tid < 64) {
      unsigned supw = 0u;
      int kept = 0, cur = 0;
      while (kept < d.nms_post) {
        // first candidate >= cur that is not suppressed
        unsigned avail = 0u;
        if (tid < W) {
          avail = ~supw;
          const int lo = tid << 5;
          if (cur > lo) avail &= (cur - lo >= 32) ? 0u : ~((1u << (cur - lo)) - 1u);
          if (lo + 32 > n) avail &= (n - lo <= 0) ? 0u : ((n - lo >= 32) ? ~0u : ((1u << (n - lo)) - 1u));
        }
        const unsigned long long has = __ballot(avail != 0u);
        if (has == 0ull) break;
        const int wsel = __ffsll((long long)has) - 1;
        const unsigned aw = (unsigned)__shfl((int)avail, wsel, 64);
        const int sel = (wsel << 5) + (__ffs((int)aw) - 1);
        if (tid == 0) out_index[(size_t)b * d.nms_post + kept] = sel;  // slot id, remapped below
        ++kept;
        if (tid < W) supw |= pm[sel * W + tid];
        cur = sel + 1;
      }
      if (tid == 0) s_kept = kept;
    }
    __syncthreads();
  } else {
    int cur = 0;
    while (true) {
      if (tid == 0) {
        int s = cur;
        while (s < total && sup[s]) ++s;
        s_sel = (s < total && s_kept < d.nms_post) ? s : -1;
        if (s_sel >= 0) out_index[(size_t)b * d.nms_post + s_kept] = s_sel;  // slot id, remapped below
        if (s_sel >= 0) ++s_kept;
      }
      __syncthreads();
      const int sel = s_sel;
      if (sel < 0) break;
      for (int c = sel + 1 + tid; c < total; c += TPB) {
        if (sup[c]) continue;
        if (oks_above(kx, ky, area, sel, c, J, thr32)) sup[c] = 1;
      }
      cur = sel + 1;
      __syncthreads();
    }
  }

  // ---- emit survivors in kept order
  const int K = s_kept;
  if (tid == 0) out_count[b] = K;
  for (int i = tid; i < K * (J + 1); i += TPB) {
    const int k = i / (J + 1), j = i % (J + 1);
    const int slot = out_index[(size_t)b * d.nms_post + k];
    if (j == J) {
      out_scores[(size_t)b * d.nms_post + k] = __uint_as_float((unsigned)(skeys[slot] >> 32));
      out_centers[((size_t)b * d.nms_post + k) * 3 + 0] = cen[slot * 3 + 0];
      out_centers[((size_t)b * d.nms_post + k) * 3 + 1] = cen[slot * 3 + 1];
      out_centers[((size_t)b * d.nms_post + k) * 3 + 2] = cen[slot * 3 + 2];
    } else {
      float* o = out_poses + (((size_t)b * d.nms_post + k) * J + j) * 3;
      o[0] = kx[(size_t)slot * J + j];
      o[1] = ky[(size_t)slot * J + j];
      o[2] = kz[(size_t)slot * J + j];
    }
  }
  __syncthreads();
  for (int k = tid; k < K; k += TPB) {
    const int slot = out_index[(size_t)b * d.nms_post + k];
    out_index[(size_t)b * d.nms_post + k] = (int)~(unsigned)(skeys[slot] & 0xffffffffull);
  }
}
}  // namespace

extern "C" long long das_decode_ws_bytes(int B, int cap, int J) { return ws_bytes_per_image(cap, J) * B; }

extern "C" int das_decode_cap(const DasDecodeDesc* d) {
  int cap = 0;
  for (int l = 0; l < d->num_levels; ++l) {
    const int n = d->H[l] * d->W[l];
    cap += (d->nms_pre > 0 && n > d->nms_pre) ? d->nms_pre : n;
  }
  return cap;
}

extern "C" int das_decode(const DasDecodeDesc* d, float* out_scores, float* out_poses, float* out_centers,
                          int* out_index, int* out_count, void* ws, void* stream) {
  DAS_PROF(stream);
  if (!d || !out_scores || !out_poses || !out_centers || !out_index || !out_count || !ws) return DAS_ERR_ARG;
  if (d->B < 1 || d->J < 1 || d->num_levels < 1 || d->num_levels > DAS_MAX_LEVELS || d->nms_post < 1) return DAS_ERR_ARG;
  long long locs = 0;
  for (int l = 0; l < d->num_levels; ++l) {
    if (d->H[l] < 1 || d->W[l] < 1 || !d->cls[l] || !d->ctr[l] || !d->pose[l]) return DAS_ERR_ARG;
    locs += (long long)d->H[l] * d->W[l];
  }
  if (locs >= (1ll << 31)) return DAS_ERR_ARG;   // (flat location indices are 32-bit, as the keys store them)
  const int cap = das_decode_cap(d);             // any size: beyond the LDS capacities the kernel works in the workspace
  const int lds = LDS_KEYS * 8 + SUP_LDS;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)decode_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr = true;
  }
  hipLaunchKernelGGL(decode_kernel, dim3(d->B), dim3(TPB), lds, (hipStream_t)stream, *d, out_scores, out_poses,
                     out_centers, out_index, out_count, (char*)ws, cap);
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}
