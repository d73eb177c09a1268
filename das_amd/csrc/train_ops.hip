// Backward kernels of the conv / norm units (training path).
//   conv data-gradient  : das_conv2d_nhwc itself on flipped/transposed weights (+ in_up for strides)
//   conv weight-gradient: conv_wgrad_kernel below — GEMM dW[o][k] = sum_m dY[m][o] * Xcol[m][k] whose
//                         reduction runs over pixel rows, i.e. over the slow axis of both NHWC operands.
//                         bf16: tiles are DMA'd row-major into LDS and transposed for free by
//                         ds_read_b64_tr_b16 while building the MFMA fragments; f32: 16x16x4 MFMA whose
//                         fragments are single dwords, so no transpose is needed.
//   BatchNorm (train) backward, GroupNorm backward, column sums (bias gradients).
#include <algorithm>
#include "prof.h"
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

#include "conv_common.h"
#include "tuning.h"
#include "workspace.h"

using namespace dasconv;

__device__ uint4 g_das_zero_page_train[8];  // this translation unit's zero page (no -fgpu-rdc)

namespace {
constexpr int TPB = 256;
inline int grid_for(long long n, int cap = 8192) {
  long long b = (n + TPB - 1) / TPB;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

__device__ __forceinline__ void dma16(const void* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

typedef short v4i16_t __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------ weight gradient
template <typename T>
struct WG;
template <>
struct WG<bf16_t> {
  static constexpr int ROWB = 256;  // 128 channels x 2 B
  static __device__ __forceinline__ int swz(int slot, int row) { return slot ^ ((row & 7) << 1); }
};
template <>
struct WG<float> {
  static constexpr int ROWB = 512;
  static __device__ __forceinline__ int swz(int slot, int row) { return slot; }
};
// 16-byte slot of a staged row -> slot it is stored in, for rows of ROWB bytes (bf16 tiles of 64 / 128 / 256 channels; f32
// tiles are not swizzled). A transposing fragment read (ds_read_b64_tr_b16) touches 16 rows x 32 bytes; eight consecutive
// rows must land on eight different 32-byte columns of the 256 bytes the banks span: rows of 256 or 512 bytes all start on
// bank 0, so the row's low three bits rotate the 32-byte column; with 128-byte rows two rows share the banks already and
// the remaining two bits do.
template <typename T, int ROWB>
__device__ __forceinline__ int swz_rb(int slot, int row) {
  if constexpr (sizeof(T) != 2) return slot;
  else if constexpr (ROWB == 128) return slot ^ (((row >> 1) & 3) << 1);
  else return slot ^ ((row & 7) << 1);
}
// conv_wgrad_kernel's (Cout x K) tile is four waves of 64 x 64 in one of three arrangements, picked per op on the host
// (wgrad_prepare): 128 x 128 (waves 2 x 2), 64 x 256 (1 x 4: layers with <= 64 output channels — the 128-wide Cout tile
// was half empty) and 256 x 64 (4 x 1: K = 64, the 1x1 convs with 64 input channels). Same 16 384 accumulators per
// workgroup in every arrangement, so partial tiles of all three share one workspace layout and one reduce kernel.
__host__ __device__ inline int shape_to(int shape) { return shape == 1 ? 64 : shape == 2 ? 256 : 128; }
__host__ __device__ inline int shape_tn(int shape) { return shape == 1 ? 256 : shape == 2 ? 64 : 128; }

// The pixel-row reduction advances every staged row by a constant each step, and the rows ONE wave stages per
// step are ROWS consecutive pixel rows. So a single wave-uniform walker (scalar registers, SALU) follows the
// first of them — without divisions and, on the common path, WITHOUT BRANCHES — and each lane derives its own
// row from it with one wrap test; only when the ROWS rows straddle the end of an image plane (or of the tensor)
// do the lanes decode their row in full. What this replaced and why (both measured on the 3x3 256->256 layer):
// a walker per lane costs ~100 VALU instructions per staged instruction pair, more than the MFMAs the data
// feeds; and a walker with loops and early-outs costs ~10 taken branches per step, each an instruction-fetch
// round trip — the kernel ran at the same speed with its MFMAs or its DMAs removed.
struct RowWalk {
  int m;            // first row of the wave (saturates at M)
  int h, w;         // output row / column of row m inside its image plane
  int Ho, Wo;       // output plane of the current level
  int H, W;         // input plane of the current level
  long long pix0;   // first pixel of the image plane in x
  unsigned wmagic;  // 2^32 / Wo + 1: umulhi(t, wmagic) == t / Wo for the t < Wo + 64 a step produces
};
__device__ __forceinline__ void walk_refresh(const ConvP& p, RowWalk& r) {
  const RowGeom g = row_geom(p, r.m);
  if (r.m >= p.M) {  // past the end: every lane takes the full decode, which masks the row
    r.h = r.w = 0; r.Ho = r.Wo = 1; r.H = r.W = 0; r.pix0 = 0; r.wmagic = 0;
    return;
  }
  r.H = g.H; r.W = g.W; r.pix0 = g.pix0;
  if (p.nlev <= 1) {
    const int rem = r.m % p.HoWo;
    r.h = rem / p.Wo; r.w = rem - r.h * p.Wo; r.Ho = p.Ho; r.Wo = p.Wo;
  } else {  // stride 1, "same" padding
    r.Ho = g.H; r.Wo = g.W;
    r.h = g.hi0 + p.pad; r.w = g.wi0 + p.pad;
  }
  r.wmagic = 0xFFFFFFFFu / (unsigned)r.Wo + 1u;
}
__device__ __forceinline__ void walk_advance(const ConvP& p, RowWalk& r, int n) {   // n <= 64
  r.m = min(r.m + n, p.M);
  const int t = r.w + n;
  const int k = r.Wo == 1 ? t : (int)__umulhi((unsigned)t, r.wmagic);   // image rows wrapped
  r.w = t - k * r.Wo;
  r.h += k;
  // next image (or next ragged level, or the end): decode once — every Ho * Wo / n steps
  if (__builtin_expect(r.h >= r.Ho || r.m >= p.M, 0)) walk_refresh(p, r);
}
template <int ROWS>
__device__ __forceinline__ bool rows_in_plane(const ConvP& p, const RowWalk& r) {
  return r.m + ROWS <= p.M && r.Wo >= ROWS && (r.Ho - r.h) * r.Wo - r.w >= ROWS;
}
struct LaneRow {
  int hi0, wi0, H, W;
  long long pix0;
  bool ok;   // row < M
};
template <bool IN_PLANE>
__device__ __forceinline__ LaneRow lane_row(const ConvP& p, const RowWalk& r, int d) {
  LaneRow o;
  if constexpr (IN_PLANE) {
    int w = r.w + d;
    const bool wrap = w >= r.Wo;
    w -= wrap ? r.Wo : 0;
    const int h = r.h + (wrap ? 1 : 0);
    o.hi0 = h * p.stride - p.pad;
    o.wi0 = w * p.stride - p.pad;
    o.H = r.H; o.W = r.W; o.pix0 = r.pix0; o.ok = true;
  } else {
    const RowGeom g = row_geom(p, min(r.m + d, p.M));
    o.hi0 = g.hi0; o.wi0 = g.wi0; o.H = g.H; o.W = g.W; o.pix0 = g.pix0; o.ok = r.m + d < p.M;
  }
  return o;
}

// ---- the pixel-row splits of one (Cout, K) tile are summed in a second pass, not with atomics ----
// f32 atomics retire at ~1 lane per clock per L2 channel: the 12.6 M atomics of a 768-workgroup launch cost
// 38 us whatever their address pattern (probe: tools/dev/probe/atomic_probe.hip), 40 % of the average
// weight-gradient launch of a training step. Plain stores move the same bytes at HBM speed, so every workgroup
// stores its accumulators — in register order, one coalesced float4 per lane — into a per-stream workspace
// [split][tile][slot], and wgrad_reduce_kernel sums the splits of every slot in a fixed order (deterministic
// when one reduce group covers all splits) and adds the result into dW.
template <int WAVES_, int TA_>
struct AccMap {   // register order of a workgroup tile: slot = ((wave * TA + a) * 4 + b) * 64 + lane, float4 = j
  static constexpr int WAVES = WAVES_, TA = TA_, SLOTS = WAVES_ * TA_ * 4 * 64;
};
struct AccMap128 : AccMap<4, 4> {   // conv_wgrad_kernel: four waves of 64 x 64 as 128 x 128 / 64 x 256 / 256 x 64 (shape 0 / 1 / 2)
  static __device__ __forceinline__ int wave_o(int wave, int shape) { return shape == 0 ? (wave >> 1) * 64 : shape == 1 ? 0 : wave * 64; }
  static __device__ __forceinline__ int wave_n(int wave, int shape) { return shape == 0 ? (wave & 1) * 64 : shape == 1 ? wave * 64 : 0; }
};
struct AccMap256 : AccMap<8, 8> {   // conv_wgrad_pp_kernel: 256 x 256, waves 2 x 4 of 128 x 64
  static __device__ __forceinline__ int wave_o(int wave, int) { return (wave & 1) * 128; }
  static __device__ __forceinline__ int wave_n(int wave, int) { return (wave >> 2) * 128 + ((wave >> 1) & 1) * 64; }
};
template <typename MAP, int TA>
__device__ __forceinline__ void store_partial_tile(float* ws, int split, int tiles, int tile, int wave, int lane,
                                                   int o_left, int n_left, const f32x4_t (&acc)[TA][4], int shape = 0) {
  f32x4_t* dst = reinterpret_cast<f32x4_t*>(ws) + ((size_t)split * tiles + tile) * MAP::SLOTS + wave * (TA * 4 * 64) + lane;
#pragma unroll
  for (int a = 0; a < TA; ++a) {
    if (MAP::wave_o(wave, shape) + a * 16 >= o_left) break;   // rows past Cout / columns past K: never read back
#pragma unroll
    for (int b = 0; b < 4; ++b)
      if (MAP::wave_n(wave, shape) + b * 16 < n_left) dst[(a * 4 + b) * 64] = acc[a][b];
  }
}
// ---- many weight gradients in ONE persistent launch, scheduled on the host ----
// A weight gradient has a small result (Cout x K) and a huge reduction (the pixels). Filling 256 CUs with ONE op means
// splitting the pixels, and every split costs a round trip of a partial tile through the workspace (with 256 workgroups
// on a 1024 x 256 result that round trip moves twice the bytes of the operands); sharing a launch between a few ops by
// grid shares (the previous scheme) leaves whole-number effects: an op with 9 tiles and a share of 14 workgroups runs
// 9 of them 1.5x longer than planned while the others idle. Backward defers its weight gradients — they feed nothing
// but the optimizer — so the launcher sees dozens of ops at a time and schedules them like a job shop:
//   * unit = (op, tile, run of pixel steps). A tile whose whole reduction fits under the per-workgroup quota is ONE
//     unit: its accumulators go straight into dW, no workspace, no reduction pass. Longer tiles are cut into equal
//     runs whose partial tiles pass through the workspace (contiguous per tile) and are summed by wgrad_reduce_kernel.
//   * the grid is one resident wave of workgroups; every workgroup walks a LIST of units (longest first). Lists are
//     built by longest-processing-time-first packing per XCD: the units of one (op, run) — all (Cout, K) tiles over
//     the same pixel rows, which share dY and the X rows — go to workgroups of the same XCD (block b runs on XCD
//     b % 8) with near-equal start times, so the rows are fetched into ONE L2.
// The schedule depends on the shapes only: it is cached per op-list signature together with its device copy; operand
// pointers travel in the kernel argument.
constexpr int WG_MAXOPS = 64;
struct WgradOpS {        // geometry of one op (device table)
  int H, W, Cin, xps, Ho, Wo, Cout, rps, KH, KW, stride, pad, M, K, HoWo;
  unsigned xbytes;
  int nlev, B;
  int lvH[MAXLV], lvW[MAXLV], lvStart[MAXLV];
  int tiles;
  int shape;            // conv_wgrad_kernel: wave arrangement of this op's tiles (shape_to / shape_tn)
};
struct WgradUnit {
  int op, tile;
  int step0, nsteps;    // pixel steps [step0, step0 + nsteps) of BKM rows
  int dest;             // >= 0: partial tile number in the workspace; -1: straight into dW
  int pad_[3];
};
struct WgradPtrs { const char* x; const char* dy; float* dw; };
struct WgradSched {
  const WgradUnit* units;
  const int* first;     // [grid + 1]: units of workgroup b = units[first[b] .. first[b + 1])
  const WgradOpS* ops;
  float* ws;
  int accumulate;
  WgradPtrs ptr[WG_MAXOPS];
};
// The argument block lives in HOST memory (every first touch of one of its cache lines is a PCIe round trip, ~2 us):
// a workgroup reads the ONE pointer record it needs through the kernarg segment pointer with a dynamic offset.
// (Indexing the by-value argument dynamically makes the compiler copy all of it to scratch.)
__device__ __forceinline__ void wgrad_load_unit(const WgradSched& g, const WgradUnit& u, ConvP& p, float*& dw) {
  const char* args = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
  const WgradPtrs& pt = *reinterpret_cast<const WgradPtrs*>(args + __builtin_offsetof(WgradSched, ptr) + (size_t)u.op * sizeof(WgradPtrs));
  const WgradOpS& o = g.ops[u.op];
  p.x = pt.x; p.res = pt.dy; p.y = nullptr; p.w = nullptr; dw = pt.dw;
  p.scale = p.shift = nullptr; p.stats = nullptr; p.stat_slots = 1;
  p.H = o.H; p.W = o.W; p.Cin = o.Cin; p.xps = o.xps; p.Ho = o.Ho; p.Wo = o.Wo; p.Cout = o.Cout; p.yps = 0;
  p.KH = o.KH; p.KW = o.KW; p.stride = o.stride; p.pad = o.pad; p.relu_in = p.relu = 0; p.rps = o.rps; p.up_sh = 0;
  p.M = o.M; p.K = o.K; p.HoWo = o.HoWo; p.ntiles = p.nblocks = 0; p.m_base = 0; p.mstep = 0; p.xbytes = o.xbytes; p.nlev = o.nlev; p.B = o.B;
#pragma unroll
  for (int l = 0; l < MAXLV; ++l) { p.lvH[l] = o.lvH[l]; p.lvW[l] = o.lvW[l]; p.lvStart[l] = o.lvStart[l]; }
  p.bnb_raw = p.bnb_y = nullptr; p.bnb_bits = nullptr; p.res_bits = nullptr; p.bnb_mean = p.bnb_invstd = p.bnb_gamma = p.bnb_beta = nullptr; p.bnb_relu = p.bnb_ps = 0;
}
// accumulators of a whole reduction -> dW (row-major Cout x K), same element map as wgrad_reduce_kernel. A wave tile
// inside the result (the common case, wave-uniform test) runs without per-element tests: the old values of an
// accumulating store are fetched sixteen at a time (written with early-outs per element the compiler serialises
// 128 load -> wait -> add -> store round trips per lane); wave tiles that overhang Cout / K take the per-element path.
template <typename MAP, int TA>
__device__ __forceinline__ void store_direct_tile(float* dw, int K, int Cout, int o0, int n0, int wave, int lane,
                                                  int accumulate, const f32x4_t (&acc)[TA][4], int shape = 0) {
  const int q = lane & 15, g4 = lane >> 4;
  const int wo = o0 + MAP::wave_o(wave, shape), wn = n0 + MAP::wave_n(wave, shape);
  if (wo + TA * 16 <= Cout && wn + 64 <= K) {
    float* base = dw + (size_t)(wo + g4 * 4) * K + wn + q;
#pragma unroll
    for (int a = 0; a < TA; ++a) {
      float* r = base + (size_t)a * 16 * K;
      constexpr int JB = TA >= 8 ? 4 : 2;   // rows per batch (the 128 x 128 kernel has no registers to spare)
#pragma unroll
      for (int j0 = 0; j0 < 4; j0 += JB) {
        float old[JB][4];
#pragma unroll
        for (int j = 0; j < JB; ++j)
#pragma unroll
          for (int b = 0; b < 4; ++b) old[j][b] = accumulate ? r[(size_t)(j0 + j) * K + b * 16] : 0.f;
#pragma unroll
        for (int j = 0; j < JB; ++j)
#pragma unroll
          for (int b = 0; b < 4; ++b) r[(size_t)(j0 + j) * K + b * 16] = old[j][b] + acc[a][b][j0 + j];
      }
    }
    return;
  }
#pragma unroll
  for (int a = 0; a < TA; ++a) {
    const int ob = wo + a * 16 + g4 * 4;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int n = wn + b * 16 + q;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (n < K && ob + j < Cout) {
          float* d = dw + (size_t)(ob + j) * K + n;
          *d = accumulate ? *d + acc[a][b][j] : acc[a][b][j];
        }
      }
    }
  }
}
struct WgradRedJob {
  int op, o0, n0;       // tile origin
  int base, splits;     // partial tiles base .. base + splits of the workspace
  int groups;           // reduce groups (> 1: atomics into a zeroed / accumulating dW)
  int shape;            // wave arrangement of the tile (conv_wgrad_kernel: 0 / 1 / 2; 0 for the ping-pong kernel)
  int pad_[1];
};
struct WgradRedArgs {
  const WgradRedJob* jobs;
  const WgradOpS* ops;
  const float* ws;
  int accumulate;
  float* dw[WG_MAXOPS];
};

// grid (SLOTS / 256 * max groups, jobs): every thread owns one float4 slot of one tile and sums it over the splits of
// its group in a fixed order; one group writes (or adds to) dW directly, several groups (one small result, hundreds of
// splits) add atomically into a zeroed / accumulating dW.
template <typename MAP>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(WgradRedArgs g) {
  constexpr int BPT = MAP::SLOTS / 256;
  const WgradRedJob& jb = g.jobs[blockIdx.y];
  const int grp = blockIdx.x / BPT, bx = blockIdx.x % BPT;
  if (grp >= jb.groups) return;
  const char* args = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
  float* dw = *reinterpret_cast<float* const*>(args + __builtin_offsetof(WgradRedArgs, dw) + (size_t)jb.op * sizeof(float*));
  const int Cout = g.ops[jb.op].Cout, K = g.ops[jb.op].K;
  const int slot = bx * 256 + threadIdx.x;
  const int lane = slot & 63, b = (slot >> 6) & 3, a = (slot >> 8) % MAP::TA, wave = slot / (MAP::TA * 256);
  const int o0 = jb.o0 + MAP::wave_o(wave, jb.shape) + a * 16 + (lane >> 4) * 4;
  const int n = jb.n0 + MAP::wave_n(wave, jb.shape) + b * 16 + (lane & 15);
  if (jb.o0 + MAP::wave_o(wave, jb.shape) + a * 16 >= Cout || n >= K) return;
  const int per = (jb.splits + jb.groups - 1) / jb.groups;
  const int s0 = grp * per, s1 = min(jb.splits, s0 + per);
  const f32x4_t* src = reinterpret_cast<const f32x4_t*>(g.ws) + (size_t)jb.base * MAP::SLOTS + slot;
  const size_t stride = MAP::SLOTS;
  f32x4_t sum = {0.f, 0.f, 0.f, 0.f};
  int sp = s0;
  for (; sp + 4 <= s1; sp += 4) {
    const f32x4_t v0 = src[(size_t)sp * stride], v1 = src[(size_t)(sp + 1) * stride];
    const f32x4_t v2 = src[(size_t)(sp + 2) * stride], v3 = src[(size_t)(sp + 3) * stride];
    sum += (v0 + v1) + (v2 + v3);
  }
  for (; sp < s1; ++sp) sum += src[(size_t)sp * stride];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (o0 + j >= Cout) break;
    float* d = dw + (size_t)(o0 + j) * K + n;
    if (jb.groups > 1) atomicAdd(d, sum[j]);
    else *d = g.accumulate ? *d + sum[j] : sum[j];
  }
}

// One unit (tile x run of pixel steps) of conv_wgrad_kernel. p.x = forward input X, p.res = dY (pixel stride p.rps).
// SHAPE: the tile's wave arrangement (shape_to x shape_tn; f32 runs 128 x 128 only). The dY tile (BKM rows x TO channels)
// and the im2col'd X tile (BKM rows x TN columns) are DMA'd row-major, 1 KiB per wave instruction — rows of 128 / 256 /
// 512 bytes, i.e. 8 / 4 / 2 rows per instruction —; wave w stages rows [w BKM/4, (w + 1) BKM/4) of BOTH tiles, so one
// wave-uniform row walker serves both.
template <typename T, int BKM, int SHAPE>   // BKM = pixel rows per step
__device__ __forceinline__ void wgrad_unit(const WgradSched& sch_, const WgradUnit& un, const ConvP& p, float* dw, char* smem) {
  constexpr int EPV = Elem<T>::EPV;
  constexpr int TO = SHAPE == 1 ? 64 : SHAPE == 2 ? 256 : 128, TN = SHAPE == 1 ? 256 : SHAPE == 2 ? 64 : 128;
  constexpr int ROWB_D = TO * (int)sizeof(T), ROWB_X = TN * (int)sizeof(T);
  constexpr int SLOTS_D = ROWB_D / 16, SLOTS_X = ROWB_X / 16;
  constexpr int TILE_D = BKM * ROWB_D, TILE_X = BKM * ROWB_X;   // bytes per operand tile
  constexpr int RPI_D = 1024 / ROWB_D, RPI_X = 1024 / ROWB_X;   // rows per wave-instruction
  constexpr int ROWS = BKM / 4;                                 // rows a wave stages per step
  constexpr int IPW_D = ROWS / RPI_D, IPW_X = ROWS / RPI_X;     // DMA instructions per wave per tile
  static_assert(IPW_D >= 1 && IPW_X >= 1, "tile rows");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntiles = (p.K + TN - 1) / TN;
  const int tile = un.tile;
  const int o0 = (tile / ntiles) * TO, n0 = (tile % ntiles) * TN;
  const int wave_o0 = AccMap128::wave_o(wave, SHAPE), wave_n0 = AccMap128::wave_n(wave, SHAPE);
  const long long m_begin = (long long)un.step0 * BKM;

  const T* xg = reinterpret_cast<const T*>(p.x);
  const T* dyg = reinterpret_cast<const T*>(p.res);
  const T* zero = reinterpret_cast<const T*>(g_das_zero_page_train);

  // per (lane, instruction) constants: dY pointer of the current row, X column -> (tap, ci); plus the row walker
  // (wave-uniform: the first of the ROWS consecutive rows this wave stages)
  bool ook[IPW_D];
  const T* dcur[IPW_D];
  const int d0d = lane / SLOTS_D, d0x = lane / SLOTS_X;
#pragma unroll
  for (int j = 0; j < IPW_D; ++j) {
    const int row = wave * ROWS + j * RPI_D + d0d;
    const int logical = swz_rb<T, ROWB_D>(lane % SLOTS_D, row);
    ook[j] = o0 + logical * EPV < p.Cout;
    dcur[j] = dyg + (m_begin + row) * p.rps + o0 + logical * EPV;
  }
  int xkh[IPW_X], xkw[IPW_X], xci[IPW_X];
  bool nok[IPW_X];
#pragma unroll
  for (int j = 0; j < IPW_X; ++j) {
    const int row = wave * ROWS + j * RPI_X + d0x;
    const int logical = swz_rb<T, ROWB_X>(lane % SLOTS_X, row);
    const int n = n0 + logical * EPV;
    nok[j] = n < p.K;
    const int tap = n / p.Cin;
    xci[j] = n - tap * p.Cin;
    xkh[j] = tap / p.KW;
    xkw[j] = tap - xkh[j] * p.KW;
  }
  RowWalk rw;
  rw.m = (int)min(m_begin + wave * ROWS, (long long)p.M);
  walk_refresh(p, rw);
  const long long dstep = (long long)BKM * p.rps;

  auto issue = [&](int buf) {  // DMA the tiles of the walker's current rows, then advance it one step
    char* sD = smem + buf * (TILE_D + TILE_X);
    char* sX = sD + TILE_D;
    auto stage = [&](auto in_plane) {
      constexpr bool IN = decltype(in_plane)::value;
#pragma unroll
      for (int j = 0; j < IPW_D; ++j) {
        const bool ok = IN ? true : rw.m + j * RPI_D + d0d < p.M;
        dma16((ok && ook[j]) ? dcur[j] : zero, sD + (wave * IPW_D + j) * 1024);
        dcur[j] += dstep;
      }
#pragma unroll
      for (int j = 0; j < IPW_X; ++j) {
        const LaneRow lr = lane_row<IN>(p, rw, j * RPI_X + d0x);
        const int hi = lr.hi0 + xkh[j], wi = lr.wi0 + xkw[j];
        const bool ok = lr.ok && nok[j] && (unsigned)hi < (unsigned)lr.H && (unsigned)wi < (unsigned)lr.W;
        const T* sx = ok ? xg + (lr.pix0 + hi * lr.W + wi) * p.xps + xci[j] : zero;
        dma16(sx, sX + (wave * IPW_X + j) * 1024);
      }
    };
    if (__builtin_expect(rows_in_plane<ROWS>(p, rw), 1)) {
      stage(std::true_type{});
    } else {
      stage(std::false_type{});
    }
    walk_advance(p, rw, BKM);
  };

  f32x4_t acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nsteps = un.nsteps;   // >= 1, inside the tensor (host schedule)
  issue(0);
  __syncthreads();
  const int g4 = lane >> 4, q = lane & 15;
  for (int s = 0; s < nsteps; ++s) {
    const int buf = s & 1;
    if (s + 1 < nsteps) issue(buf ^ 1);
    const char* sD = smem + buf * (TILE_D + TILE_X);
    const char* sX = sD + TILE_D;
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int half = 0; half < BKM / 32; ++half) {
        uint4 fa[4], fb[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          v4i16_t lo[2], hi[2];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int row = half * 32 + h * 16 + g4 * 4 + (q >> 2);
            const int sub = q & 3;
            const int ca = wave_o0 + t * 16, cb = wave_n0 + t * 16;
            const int sa = swz_rb<T, ROWB_D>((ca >> 3) + (sub >> 1), row), sb = swz_rb<T, ROWB_X>((cb >> 3) + (sub >> 1), row);
            lo[h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) v4i16_t*)(sD + row * ROWB_D + sa * 16 + (sub & 1) * 8));
            hi[h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) v4i16_t*)(sX + row * ROWB_X + sb * 16 + (sub & 1) * 8));
          }
          fa[t] = make_uint4(__builtin_bit_cast(uint2, lo[0]).x, __builtin_bit_cast(uint2, lo[0]).y,
                             __builtin_bit_cast(uint2, lo[1]).x, __builtin_bit_cast(uint2, lo[1]).y);
          fb[t] = make_uint4(__builtin_bit_cast(uint2, hi[0]).x, __builtin_bit_cast(uint2, hi[0]).y,
                             __builtin_bit_cast(uint2, hi[1]).x, __builtin_bit_cast(uint2, hi[1]).y);
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa[a]),
                                                                __builtin_bit_cast(bf16x8_t, fb[b]), acc[a][b], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < BKM / 4; ++kk) {
        float fa[4], fb[4];
        const int row = kk * 4 + g4;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          fa[t] = *reinterpret_cast<const float*>(sD + row * ROWB_D + (wave_o0 + t * 16 + q) * 4);
          fb[t] = *reinterpret_cast<const float*>(sX + row * ROWB_X + (wave_n0 + t * 16 + q) * 4);
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[a], fb[b], acc[a][b], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  if (un.dest >= 0) {
    store_partial_tile<AccMap128>(sch_.ws, un.dest, 1, 0, wave, lane, p.Cout - o0, p.K - n0, acc, SHAPE);
  } else {
    store_direct_tile<AccMap128>(dw, p.K, p.Cout, o0, n0, wave, lane, sch_.accumulate, acc, SHAPE);
  }
}

// (second bound: three waves per SIMD = three workgroups per CU, the occupancy the 32-row steps were chosen for: at most 168 registers)
template <typename T, int BKM>
__global__ __launch_bounds__(256, BKM == 32 ? 3 : 2) void conv_wgrad_kernel(WgradSched sch_) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int u_end = sch_.first[blockIdx.x + 1];
  for (int ui = sch_.first[blockIdx.x]; ui < u_end; ++ui) {
    const WgradUnit& un = sch_.units[ui];
    ConvP p;
    float* dw;
    wgrad_load_unit(sch_, un, p, dw);
    if constexpr (sizeof(T) == 2) {
      const int shape = __builtin_amdgcn_readfirstlane(sch_.ops[un.op].shape);
      if (shape == 1) wgrad_unit<T, BKM, 1>(sch_, un, p, dw, smem);
      else if (shape == 2) wgrad_unit<T, BKM, 2>(sch_, un, p, dw, smem);
      else wgrad_unit<T, BKM, 0>(sch_, un, p, dw, smem);
    } else {
      wgrad_unit<T, BKM, 0>(sch_, un, p, dw, smem);
    }
    // next unit (every wave is past the last step's barrier: the LDS buffers are free)
  }
}

// Weight gradient of the 3x3, stride-1, 64 -> 64 channel convs (conv2 of the 128 x 208 stage's bottlenecks, 12 per train
// step): on the 128 x 128 kernel above they were the slowest ops against their floor (105 us for 31 GFLOP / 109 MB: its
// Cout tile is half empty and every one of the five K tiles reads dY and its own taps' pixel rows again). Same idea as
// conv3x3_c64_kernel (conv_igemm.hip): PERSISTENT, one workgroup per CU, tiles are 16 x 16-pixel squares; the square's dY
// (256 pixels x 64 channels) and the 18 x 18 input patch are DMA'd ONCE (two loader waves, one tile ahead, double buffers)
// and NINE waves — one per tap — multiply dY^T by the patch shifted by their tap (transposing LDS reads, 32 pixels per
// MFMA step), each keeping its 64 x 64 slice of dW in registers for the whole launch. Partial dW per workgroup go to the
// workspace [workgroup][tap][o][ci]; wgrad_c64_reduce_kernel sums them in a fixed order (deterministic) into dW.
constexpr int WC64_DY = 32 * 1024, WC64_PATCH = 41 * 1024;
__global__ __launch_bounds__(704) void conv_wgrad_c64_kernel(const char* __restrict__ xg, const char* __restrict__ dyg,
                                                             float* __restrict__ ws, int B, int H, int W, int xps, int rps,
                                                             unsigned xbytes, unsigned dbytes, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][dY tile | patch]
  constexpr int STAGE = WC64_DY + WC64_PATCH;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tw = (W + 15) >> 4, per_img = ((H + 15) >> 4) * tw;
  if (wave >= 9) {   // ---- loader waves: wave 9 stages dY, wave 10 the input patch
    const v4i_t xrs = make_rsrc(xg, xbytes), drs = make_rsrc(dyg, dbytes);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned xps2 = (unsigned)xps * 2u, rps2 = (unsigned)rps * 2u;
    constexpr unsigned OOB = 0xFFFFFFF0u;
    auto issue = [&](int tile, int buf) {
      const int b = tile / per_img, t = tile - b * per_img, th = t / tw;
      const int h0 = th * 16, w0 = (t - th * tw) * 16;
      const unsigned sD = lds0 + buf * STAGE, sX = sD + WC64_DY;
      if (wave == 9) {
#pragma unroll 4
        for (int j = 0; j < 32; ++j) {               // dY: pixel p = r * 16 + c of the square
          const int pix = j * 8 + (lane >> 3);
          const int h = h0 + (pix >> 4), w = w0 + (pix & 15);
          const bool ok = h < H && w < W;
          const unsigned chunk = (unsigned)((lane & 7) ^ (pix & 7));
          dma16_buf(ok ? (unsigned)((b * H + h) * W + w) * rps2 + chunk * 16u : OOB, drs, sD + j * 1024);
        }
      } else {
        int pp = lane >> 3, pr = 0, pc = pp;         // patch pixel of this lane in instruction 0; + 8 per instruction
#pragma unroll 4
        for (int j = 0; j < 41; ++j) {
          const int h = h0 - 1 + pr, w = w0 - 1 + pc;
          const bool ok = pp < 324 && (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
          const unsigned chunk = (unsigned)((lane & 7) ^ (pp & 7));
          dma16_buf(ok ? (unsigned)((b * H + h) * W + w) * xps2 + chunk * 16u : OOB, xrs, sX + j * 1024);
          pp += 8; pc += 8;
          if (pc >= 18) { pc -= 18; ++pr; }
        }
      }
    };
    int i = 0;
    if ((int)blockIdx.x < ntiles) issue(blockIdx.x, 0);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, ++i) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this tile's operands have landed (this wave's half)
      __builtin_amdgcn_s_barrier();                        // A: ... and the tap waves are done with the other stage
      if (tile + (int)gridDim.x < ntiles) issue(tile + gridDim.x, (i + 1) & 1);
      __builtin_amdgcn_s_barrier();                        // B
    }
    return;
  }
  // ---- tap waves: wave = kh * 3 + kw
  const int kh = wave / 3, kw = wave - kh * 3;
  const int g4 = lane >> 4, q = lane & 15, sub = q & 3, col = g4 * 4 + (q >> 2);
  f32x4_t acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  int i = 0;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, ++i) {
    const char* sD = smem + (i & 1) * STAGE;
    const char* sX = sD + WC64_DY;
    __builtin_amdgcn_s_barrier();                          // A
#pragma unroll 2
    for (int ks = 0; ks < 8; ++ks) {                       // 32 pixels per step: rows 2 ks, 2 ks + 1 of the square
      uint4 fa[4], fb[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        v4i16_t lo[2], hi[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int pd = (ks * 2 + h) * 16 + col;                       // dY pixel of the square
          const int px = (ks * 2 + h + kh) * 18 + col + kw;             // input pixel of the patch, shifted by the tap
          const int slot = t * 2 + (sub >> 1);
          lo[h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) v4i16_t*)(sD + pd * 128 + ((slot ^ (pd & 7)) << 4) + (sub & 1) * 8));
          hi[h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) v4i16_t*)(sX + px * 128 + ((slot ^ (px & 7)) << 4) + (sub & 1) * 8));
        }
        fa[t] = make_uint4(__builtin_bit_cast(uint2, lo[0]).x, __builtin_bit_cast(uint2, lo[0]).y,
                           __builtin_bit_cast(uint2, lo[1]).x, __builtin_bit_cast(uint2, lo[1]).y);
        fb[t] = make_uint4(__builtin_bit_cast(uint2, hi[0]).x, __builtin_bit_cast(uint2, hi[0]).y,
                           __builtin_bit_cast(uint2, hi[1]).x, __builtin_bit_cast(uint2, hi[1]).y);
      }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa[a]),
                                                              __builtin_bit_cast(bf16x8_t, fb[b]), acc[a][b], 0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                          // B: every wave is done reading this stage
  }
  // acc[a][b][j]: output channel a * 16 + (lane >> 4) * 4 + j, input channel b * 16 + (lane & 15). Stored straight from
  // the registers that is 4-byte pieces in 64-byte runs (37 MB of partials per launch: 40 us); through LDS (rows padded to
  // 68 floats: conflict-free both ways), 32 output channels at a time, every store instruction writes one contiguous KiB.
  float* dst = ws + ((size_t)blockIdx.x * 9 + wave) * 4096;
  float* stg = reinterpret_cast<float*>(smem) + wave * (32 * 68);
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int j = 0; j < 4; ++j) stg[(a2 * 16 + g4 * 4 + j) * 68 + b * 16 + q] = acc[half * 2 + a2][b][j];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the wave's own staging area: no barrier needed)
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int row = r * 4 + g4;
      const float4 v = *reinterpret_cast<const float4*>(stg + row * 68 + q * 4);
      *reinterpret_cast<float4*>(dst + (half * 32 + row) * 64 + q * 4) = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

// dW[o][tap][ci] (+)= sum over the workgroups' partials in a fixed order: 64 outputs per workgroup, four threads per
// output summing every fourth partial, combined through LDS
__global__ __launch_bounds__(256) void wgrad_c64_reduce_kernel(const float* __restrict__ ws, int nwg, float* __restrict__ dw,
                                                               int accumulate) {
  __shared__ float part[4][64];
  const int k = threadIdx.x >> 6, oi = threadIdx.x & 63;
  const int i = blockIdx.x * 64 + oi;             // = (tap * 64 + o) * 64 + ci of the partial layout
  float s0 = 0.f, s1 = 0.f;
  int g = k;
  for (; g + 4 < nwg; g += 8) { s0 += ws[(size_t)g * 36864 + i]; s1 += ws[(size_t)(g + 4) * 36864 + i]; }
  if (g < nwg) s0 += ws[(size_t)g * 36864 + i];
  part[k][oi] = s0 + s1;
  __syncthreads();
  if (k == 0) {
    const float v = (part[0][oi] + part[1][oi]) + (part[2][oi] + part[3][oi]);
    const int tap = i >> 12, o = (i >> 6) & 63, ci = i & 63;
    float* d = dw + (o * 9 + tap) * 64 + ci;
    *d = accumulate ? *d + v : v;
  }
}

// Ping-pong variant for bf16 with Cout >= 256 and K >= 256: 256 (Cout) x 256 (K columns) tile, 8 waves (2 x 4),
// 128 x 64 per wave — at 64 x 64 per wave the transposing LDS reads take as long as the MFMAs they feed (8 KiB
// per 16 MFMAs, 128 B/clk); 128 x 64 reads 12 KiB per 32. 32 pixel rows per step, four LDS stages of four
// 8 KiB sub-tiles {dY 0..127 | dY 128..255 | X 0..127 | X 128..255} = 128 KiB -> one workgroup per CU.
// Operands come in by buffer_load ... lds (out-of-range offset = zeros: rows past M, channels past Cout, padding
// taps, columns past K), issued three steps ahead. Every step is two barrier intervals, R (issue step s+3, walk
// the row pointers, transposing LDS reads of step s into registers) and M (32 MFMAs); waves 4-7 run one
// interval behind waves 0-3 and wave w / w+4 share a SIMD, so the address walk and the LDS reads of one group
// hide behind the other group's MFMAs (same scheme as conv_glds4_kernel<PP>).
__global__ __launch_bounds__(512) void conv_wgrad_pp_kernel(WgradSched sch_) {
  const int u_end = sch_.first[blockIdx.x + 1];
  for (int ui = sch_.first[blockIdx.x]; ui < u_end; ++ui) {   // (body indented as the single-unit kernel it grew from)
  const WgradUnit& un = sch_.units[ui];
  ConvP p;
  float* dw;
  wgrad_load_unit(sch_, un, p, dw);
  using T = bf16_t;
  constexpr int EPV = 8, ROWB = 256, SLOTS = 16, BKM = 32;
  constexpr int SUB = BKM * ROWB;                // 8 KiB: 32 rows x 128 channels
  constexpr int STAGE = 4 * SUB;
  constexpr int NST = 4;
  constexpr unsigned OOB = 0xFFFFFFF0u;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  const int ntiles = (p.K + 255) / 256;
  const int tile = un.tile;
  const int o0 = (tile / ntiles) * 256, n0 = (tile % ntiles) * 256;
  // wave tiles: Cout half = wave & 1 (dY sub-tile), K-column quarter = wave >> 1 (X sub-tile wave >> 2)
  const int wave_dsub = wave & 1, wave_xsub = wave >> 2, wave_n0 = ((wave >> 1) & 1) * 64;
  const long long m_begin = (long long)un.step0 * BKM;

  const v4i_t xrs = make_rsrc(p.x, p.xbytes);
  // (range up to the end of the last row's last 8-channel vector — it exists: rps >= Cout rounded up to 8 — so that a channel
  // count that is not a multiple of 8 does not cut the vector holding its last channels out of the last pixel row)
  const v4i_t drs = make_rsrc(p.res, (unsigned)(((long long)(p.M - 1) * p.rps + (p.Cout + 7) / 8 * 8) * 2));

  // per sub-tile 8 DMA instructions (4 rows each); wave w stages instructions (w & 3) * 2 + {0, 1} of the dY and
  // of the X sub-tile (w >> 2)
  unsigned dcur[2];
  bool dok[2], nok[2];
  int xkh[2], xkw[2], xci2[2];
  RowWalk rw;   // wave-uniform: the first of this wave's 8 consecutive rows
  rw.m = (int)min(m_begin + (wave & 3) * 8, (long long)p.M);
  walk_refresh(p, rw);
  const int d0 = lane / SLOTS;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = ((wave & 3) * 2 + j) * 4 + d0;
    const int c = (wave >> 2) * 128 + WG<T>::swz(lane % SLOTS, row) * EPV;
    dok[j] = o0 + c < p.Cout;
    dcur[j] = (unsigned)((((long long)m_begin + row) * p.rps + o0 + c) * 2);
    const int n = n0 + c;
    nok[j] = n < p.K;
    const int tap = n / p.Cin;
    xci2[j] = (n - tap * p.Cin) * 2;
    xkh[j] = tap / p.KW;
    xkw[j] = tap - xkh[j] * p.KW;
  }
  const unsigned dstep = (unsigned)BKM * (unsigned)p.rps * 2u;
  const int xps2 = p.xps * 2;

  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  auto issue = [&](int stage) {   // DMA the tiles of the walker's current rows, then advance it one step
    const unsigned sD = lds0 + stage * STAGE + (wave >> 2) * SUB + (wave & 3) * 2048;
    const unsigned sX = sD + 2 * SUB;
    auto stage_rows = [&](auto in_plane) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const LaneRow lr = lane_row<decltype(in_plane)::value>(p, rw, j * 4 + d0);
        dma16_buf(dok[j] && lr.ok ? dcur[j] : OOB, drs, sD + j * 1024);
        dcur[j] += dstep;
        const int hi = lr.hi0 + xkh[j], wi = lr.wi0 + xkw[j];
        const bool ok = lr.ok && nok[j] && (unsigned)hi < (unsigned)lr.H && (unsigned)wi < (unsigned)lr.W;
        // (32-bit: the host only takes this kernel when x is addressable by a buffer descriptor)
        const unsigned off = (unsigned)(((int)lr.pix0 + hi * lr.W + wi) * xps2 + xci2[j]);
        dma16_buf(ok ? off : OOB, xrs, sX + j * 1024);
      }
    };
    if (__builtin_expect(rows_in_plane<8>(p, rw), 1)) {
      stage_rows(std::true_type{});
    } else {
      stage_rows(std::false_type{});
    }
    walk_advance(p, rw, BKM);
  };

  f32x4_t acc[8][4];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nsteps = un.nsteps;   // >= 1, inside the tensor (host schedule)
  // each wave has 4 DMAs in flight per staged step
  auto wait_steps = [&](int ahead) {   // ... until at most `ahead` staged steps of this wave are still in flight
    if (ahead >= 2) {
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else if (ahead == 1) {
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  };
  issue(0);
  if (nsteps > 1) issue(1);
  if (nsteps > 2) issue(2);
  wait_steps(std::min(nsteps, 3) - 1);
  __builtin_amdgcn_s_barrier();                 // step 0 landed
  if (grp == 1) __builtin_amdgcn_s_barrier();   // trailing group: one interval behind
  const int g4 = lane >> 4, q = lane & 15;
  int st = 0, nst = 3;   // stage of step s, stage that step s+3 goes to
  for (int s = 0; s < nsteps; ++s) {
    // ---- R
    if (s + 3 < nsteps) issue(nst);
    const char* sD = smem + st * STAGE + wave_dsub * SUB;
    const char* sX = smem + st * STAGE + (2 + wave_xsub) * SUB;
    uint4 fa[8], fb[4];
    const int sub = q & 3;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      v4i16_t lo[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int row = h * 16 + g4 * 4 + (q >> 2);
        const int sa = WG<T>::swz(t * 2 + (sub >> 1), row);
        lo[h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) v4i16_t*)(sD + row * ROWB + sa * 16 + (sub & 1) * 8));
      }
      fa[t] = make_uint4(__builtin_bit_cast(uint2, lo[0]).x, __builtin_bit_cast(uint2, lo[0]).y,
                         __builtin_bit_cast(uint2, lo[1]).x, __builtin_bit_cast(uint2, lo[1]).y);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      v4i16_t hi[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int row = h * 16 + g4 * 4 + (q >> 2);
        const int sb = WG<T>::swz(((wave_n0 + t * 16) >> 3) + (sub >> 1), row);
        hi[h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) v4i16_t*)(sX + row * ROWB + sb * 16 + (sub & 1) * 8));
      }
      fb[t] = make_uint4(__builtin_bit_cast(uint2, hi[0]).x, __builtin_bit_cast(uint2, hi[0]).y,
                         __builtin_bit_cast(uint2, hi[1]).x, __builtin_bit_cast(uint2, hi[1]).y);
    }
    // the steps issued after step s+1 that may still fly: s+2 and (if it exists) s+3
    if (grp == 1) wait_steps(std::min(nsteps - 1, s + 3) - (s + 1));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // fragment reads retired before the buffers are re-staged
    __builtin_amdgcn_s_barrier();
    // ---- M
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa[a]),
                                                            __builtin_bit_cast(bf16x8_t, fb[b]), acc[a][b], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    if (grp == 0) wait_steps(std::min(nsteps - 1, s + 3) - (s + 1));
    __builtin_amdgcn_s_barrier();
    st = (st + 1) & (NST - 1);
    nst = (nst + 1) & (NST - 1);
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();   // leading group: match the trailing group's extra interval

  if (un.dest >= 0) {
    store_partial_tile<AccMap256>(sch_.ws, un.dest, 1, 0, wave, lane, p.Cout - o0, p.K - n0, acc);
  } else {
    store_direct_tile<AccMap256>(dw, p.K, p.Cout, o0, n0, wave, lane, sch_.accumulate, acc);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the prologue's counted waits assume only its own DMAs are in flight
  }   // next unit (both wave groups are past their last interval: every LDS stage is free)
}

// ------------------------------------------------------------------ column sums (bias gradient)
template <typename T>
__global__ void colsum_kernel(const T* __restrict__ x, long long rows, int C, int ps, float* __restrict__ out, int nout) {
  // (C: the columns walked, a multiple of the vector width; nout <= C: the columns that exist — a layer whose channel
  // count is not a multiple of 8 lives in rows padded to one, and its bias gradient has no padding)
  constexpr int EPV = Elem<T>::EPV;
  extern __shared__ float part[];   // [PL][C] (common.h: lds_put / lds_fold)
  const int VC = C / EPV;
  const int VCB = min(VC, TPB);  // channel vectors handled side by side (wider rows loop over v)
  const int pl = threadIdx.x / VCB, PL = TPB / VCB;
  for (int v = threadIdx.x % VCB; v < VC && pl < PL; v += VCB) {
    float s[EPV];
#pragma unroll
    for (int j = 0; j < EPV; ++j) s[j] = 0.f;
    for (long long r = (long long)blockIdx.x * PL + pl; r < rows; r += (long long)gridDim.x * PL) {
      float f[EPV];
      Elem<T>::unpack(*reinterpret_cast<const uint4*>(x + r * ps + v * EPV), f);
#pragma unroll
      for (int j = 0; j < EPV; ++j) s[j] += f[j];
    }
    lds_put<EPV>(part, C, pl, v * EPV, s);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nout; i += TPB) atomicAdd(out + i, lds_fold(part, C, PL, i));
}

// ------------------------------------------------------------------ BatchNorm (train) backward
// MASK: 0 = no ReLU, 1 = ReLU mask from the saved output y (needed when a residual was added before the
// ReLU), 2 = ReLU mask recomputed from raw (y > 0 <=> bn_affine(raw) > 0): one tensor read less per pass.
template <typename T, int MASK>
__device__ __forceinline__ void bn_masked_grad(float* g, const float* x, const uint4& yv, const float* mu,
                                               const float* is, const float* ga, const float* be) {
  constexpr int EPV = Elem<T>::EPV;
  if (MASK == 1) {
    float o[EPV];
    Elem<T>::unpack(yv, o);
#pragma unroll
    for (int j = 0; j < EPV; ++j) g[j] = o[j] > 0.f ? g[j] : 0.f;
  } else if (MASK == 2) {
#pragma unroll
    for (int j = 0; j < EPV; ++j) g[j] = bn_affine(x[j], mu[j], is[j], ga[j], be[j]) > 0.f ? g[j] : 0.f;
  }
}

// pass 1: s1[c] = sum dZ, s2[c] = sum dZ * xhat with dZ = dY * mask. Four rows in flight per thread.
template <typename T, int MASK, int NT>
__global__ __launch_bounds__(NT) void bn_bwd_reduce_kernel(const T* __restrict__ dy, const T* __restrict__ y,
                                                           const unsigned char* __restrict__ bits,   // (MASK 1: instead of y, common.h relu_bits)
                                                           const T* __restrict__ raw, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, long long rows, int C,
                                                           float* __restrict__ sums) {
  constexpr int EPV = Elem<T>::EPV;
  constexpr int U = NT > 256 ? 2 : 4;   // (8 rows in flight per thread measured no better: 69 / 56 / 45 us against 67 / 49 / 42 in the step, round 4)
  extern __shared__ float part[];  // [PL][2C] (common.h: lds_put / lds_fold)
  const int VC = C / EPV;
  const int VCB = min(VC, NT);
  const int pl = threadIdx.x / VCB, PL = NT / VCB;
  const long long rstride = (long long)gridDim.x * PL;
  for (int v = threadIdx.x % VCB; v < VC && pl < PL; v += VCB) {
    float s1[EPV], s2[EPV], mu[EPV], is[EPV], ga[EPV], be[EPV];
#pragma unroll
    for (int j = 0; j < EPV; ++j) {
      s1[j] = 0.f; s2[j] = 0.f; mu[j] = mean[v * EPV + j]; is[j] = invstd[v * EPV + j];
      ga[j] = MASK == 2 ? gamma[v * EPV + j] : 0.f; be[j] = MASK == 2 ? beta[v * EPV + j] : 0.f;
    }
    for (long long r = (long long)blockIdx.x * PL + pl; r < rows; r += rstride * U) {
      uint4 gv[U], xv[U], yv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long ru = r + u * rstride;
        gv[u] = xv[u] = yv[u] = make_uint4(0, 0, 0, 0);
        if (ru < rows) {
          gv[u] = *reinterpret_cast<const uint4*>(dy + ru * C + v * EPV);
          xv[u] = *reinterpret_cast<const uint4*>(raw + ru * C + v * EPV);
          if (MASK == 1) {   // (the mask byte is only REQUESTED here: expanding it at once would make every load wait for it)
            if (bits) yv[u].x = bits[ru * VC + v];
            else yv[u] = *reinterpret_cast<const uint4*>(y + ru * C + v * EPV);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {  // rows past the end carry dY = 0 and add nothing
        float g[EPV], x[EPV];
        Elem<T>::unpack(gv[u], g);
        Elem<T>::unpack(xv[u], x);
        bn_masked_grad<T, MASK>(g, x, (MASK == 1 && bits) ? mask_vec<T>(yv[u].x) : yv[u], mu, is, ga, be);
#pragma unroll
        for (int j = 0; j < EPV; ++j) { s1[j] += g[j]; s2[j] += g[j] * (x[j] - mu[j]) * is[j]; }
      }
    }
    lds_put<EPV>(part, 2 * C, pl, v * EPV, s1);
    lds_put<EPV>(part, 2 * C, pl, C + v * EPV, s2);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += NT) atomicAdd(sums + i, lds_fold(part, 2 * C, PL, i));
}

// pass 2: dRaw = gamma*invstd*(dZ - s1/N - xhat*s2/N); optionally dRes = dZ. Block 0 also adds the two
// sums into the parameter-gradient accumulators (dbeta += s1, dgamma += s2) when they are given.
template <typename T, int MASK, bool FIXED>
__global__ void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ y,
                                    const unsigned char* __restrict__ bits, const T* __restrict__ raw,
                                    const float* __restrict__ mean, const float* __restrict__ invstd,
                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                    const float* __restrict__ sums, long long rows, int C, T* __restrict__ draw,
                                    T* __restrict__ dres, float* __restrict__ dgamma_acc,
                                    float* __restrict__ dbeta_acc, float inv_n) {
  constexpr int EPV = Elem<T>::EPV;
  const int VC = C / EPV;
  const long long total = rows * VC;
  const long long stride = (long long)gridDim.x * TPB;
  if (blockIdx.x == 0 && dgamma_acc) {
    for (int c = threadIdx.x; c < C; c += TPB) { dbeta_acc[c] += sums[c]; dgamma_acc[c] += sums[C + c]; }
  }
  float mu[EPV], is[EPV], ga[EPV], be[EPV], k1[EPV], k2[EPV], k3[EPV];
  auto load_consts = [&](int c0) {
#pragma unroll
    for (int j = 0; j < EPV; ++j) {
      const int c = c0 + j;
      mu[j] = mean[c]; is[j] = invstd[c]; ga[j] = gamma[c]; be[j] = MASK == 2 ? beta[c] : 0.f;
      k1[j] = ga[j] * is[j]; k2[j] = sums[c] * inv_n; k3[j] = sums[C + c] * inv_n * is[j];
    }
  };
  auto finish = [&](long long i, const uint4& gv, const uint4& xv, const uint4& yv) {
    float g[EPV], x[EPV], o[EPV];
    Elem<T>::unpack(gv, g);
    Elem<T>::unpack(xv, x);
    bn_masked_grad<T, MASK>(g, x, (MASK == 1 && bits) ? mask_vec<T>(yv.x) : yv, mu, is, ga, be);
    if (dres) *reinterpret_cast<uint4*>(dres + i * EPV) = Elem<T>::pack(g);
#pragma unroll
    for (int j = 0; j < EPV; ++j) o[j] = k1[j] * (g[j] - k2[j] - (x[j] - mu[j]) * k3[j]);
    *reinterpret_cast<uint4*>(draw + i * EPV) = Elem<T>::pack(o);
  };
  long long i = (long long)blockIdx.x * TPB + threadIdx.x;
  const uint4 z4 = make_uint4(0, 0, 0, 0);
  if (FIXED) {
    if (i < total) load_consts((int)(i % VC) * EPV);
    for (; i + stride < total; i += 2 * stride) {
      const long long i1 = i + stride;
      const uint4 g0 = *reinterpret_cast<const uint4*>(dy + i * EPV), g1 = *reinterpret_cast<const uint4*>(dy + i1 * EPV);
      const uint4 x0 = *reinterpret_cast<const uint4*>(raw + i * EPV), x1 = *reinterpret_cast<const uint4*>(raw + i1 * EPV);
      uint4 y0 = z4, y1 = z4;
      if (MASK == 1) {   // (with bits: the byte travels in .x and is expanded inside finish)
        if (bits) { y0.x = bits[i]; y1.x = bits[i1]; }
        else { y0 = *reinterpret_cast<const uint4*>(y + i * EPV); y1 = *reinterpret_cast<const uint4*>(y + i1 * EPV); }
      }
      finish(i, g0, x0, y0);
      finish(i1, g1, x1, y1);
    }
  }
  for (; i < total; i += stride) {
    if (!FIXED) load_consts((int)(i % VC) * EPV);
    const uint4 g0 = *reinterpret_cast<const uint4*>(dy + i * EPV);
    const uint4 x0 = *reinterpret_cast<const uint4*>(raw + i * EPV);
    uint4 y0 = z4;
    if (MASK == 1) {
      if (bits) y0.x = bits[i];
      else y0 = *reinterpret_cast<const uint4*>(y + i * EPV);
    }
    finish(i, g0, x0, y0);
  }
}

// The apply pass alone, for a layer whose dZ (already masked) and per-channel sums came out of the epilogue of the
// data-gradient conv that produced dZ (ConvP::bnb_*): dRaw = gamma*invstd*(dZ - s1/N - xhat*s2/N). `sums` is
// [slots][2C] (the conv workgroups' slots): every workgroup folds it into LDS first (no separate fold launch; the
// grid is capped so that this stays a small fraction of the loads). Workgroup 0 also feeds the parameter-gradient
// accumulators (dbeta += s1, dgamma += s2).
template <typename T, bool FIXED>
__global__ void bn_bwd_apply_dz_kernel(const T* __restrict__ dz, const T* __restrict__ raw, const float* __restrict__ mean,
                                       const float* __restrict__ invstd, const float* __restrict__ gamma,
                                       const float* __restrict__ sums, int slots, long long rows, int C,
                                       T* __restrict__ draw, float* __restrict__ dgamma_acc,
                                       float* __restrict__ dbeta_acc, float inv_n) {
  constexpr int EPV = Elem<T>::EPV;
  extern __shared__ float fsum[];   // [2C]
  fold_slots_to_lds(sums, slots, 2 * C, fsum);
  if (blockIdx.x == 0 && dgamma_acc) {
    for (int c = threadIdx.x; c < C; c += TPB) { dbeta_acc[c] += fsum[c]; dgamma_acc[c] += fsum[C + c]; }
  }
  const int VC = C / EPV;
  const long long total = rows * VC;
  const long long stride = (long long)gridDim.x * TPB;
  float mu[EPV], k1[EPV], k2[EPV], k3[EPV];
  auto load_consts = [&](int c0) {
#pragma unroll
    for (int j = 0; j < EPV; ++j) {
      const int c = c0 + j;
      const float is = invstd[c];
      mu[j] = mean[c]; k1[j] = gamma[c] * is; k2[j] = fsum[c] * inv_n; k3[j] = fsum[C + c] * inv_n * is;
    }
  };
  auto finish = [&](long long i, const uint4& gv, const uint4& xv) {
    float g[EPV], x[EPV], o[EPV];
    Elem<T>::unpack(gv, g);
    Elem<T>::unpack(xv, x);
#pragma unroll
    for (int j = 0; j < EPV; ++j) o[j] = k1[j] * (g[j] - k2[j] - (x[j] - mu[j]) * k3[j]);
    *reinterpret_cast<uint4*>(draw + i * EPV) = Elem<T>::pack(o);
  };
  long long i = (long long)blockIdx.x * TPB + threadIdx.x;
  if (FIXED) {
    if (i < total) load_consts((int)(i % VC) * EPV);
    for (; i + stride < total; i += 2 * stride) {
      const long long i1 = i + stride;
      const uint4 g0 = *reinterpret_cast<const uint4*>(dz + i * EPV), g1 = *reinterpret_cast<const uint4*>(dz + i1 * EPV);
      const uint4 x0 = *reinterpret_cast<const uint4*>(raw + i * EPV), x1 = *reinterpret_cast<const uint4*>(raw + i1 * EPV);
      finish(i, g0, x0);
      finish(i1, g1, x1);
    }
  }
  for (; i < total; i += stride) {
    if (!FIXED) load_consts((int)(i % VC) * EPV);
    finish(i, *reinterpret_cast<const uint4*>(dz + i * EPV), *reinterpret_cast<const uint4*>(raw + i * EPV));
  }
}

// bn_bwd_apply_dz_kernel for LARGE tensors (see bn_apply_stream_kernel, norm.hip): the slots were folded by a launch of
// their own (`fsumg`: [2C]), one workgroup takes 256 * VPT vectors, all loads are issued before the first use, the
// per-channel constants are computed once per workgroup into LDS. Arithmetic per element as above.
template <typename T, int VPT>
__global__ __launch_bounds__(TPB) void bn_bwd_apply_dz_stream_kernel(const T* __restrict__ dz, const T* __restrict__ raw,
                                                                     const float* __restrict__ mean,
                                                                     const float* __restrict__ invstd,
                                                                     const float* __restrict__ gamma,
                                                                     const float* __restrict__ fsumg, long long rows, int C,
                                                                     T* __restrict__ draw, float* __restrict__ dgamma_acc,
                                                                     float* __restrict__ dbeta_acc, float inv_n, int ntm) {
  constexpr int EPV = Elem<T>::EPV;
  const bool nt_g = ntm & 1, nt_x = ntm & 2, nt_o = ntm & 4;   // (das_tuning key bn.nt_bwd)
  extern __shared__ float cst[];   // [4][C]: mean, k1, k2, k3
  if (blockIdx.x == 0 && dgamma_acc) {
    for (int c = threadIdx.x; c < C; c += TPB) { dbeta_acc[c] += fsumg[c]; dgamma_acc[c] += fsumg[C + c]; }
  }
  for (int c = threadIdx.x; c < C; c += TPB) {
    const float is = invstd[c];
    cst[c] = mean[c]; cst[C + c] = gamma[c] * is; cst[2 * C + c] = fsumg[c] * inv_n; cst[3 * C + c] = fsumg[C + c] * inv_n * is;
  }
  __syncthreads();
  const int VC = C / EPV;
  const int c0 = (threadIdx.x % VC) * EPV;
  float mu[EPV], k1[EPV], k2[EPV], k3[EPV];
#pragma unroll
  for (int j = 0; j < EPV; ++j) {
    mu[j] = cst[c0 + j]; k1[j] = cst[C + c0 + j]; k2[j] = cst[2 * C + c0 + j]; k3[j] = cst[3 * C + c0 + j];
  }
  const long long total = rows * VC;
  const long long base = (long long)blockIdx.x * (TPB * VPT) + threadIdx.x;
  uint4 gv[VPT], xv[VPT];
#pragma unroll
  for (int u = 0; u < VPT; ++u) {
    const long long i = base + u * TPB;
    gv[u] = make_uint4(0, 0, 0, 0);
    xv[u] = gv[u];
    if (i < total) {
      gv[u] = ld16(dz + i * EPV, nt_g);
      xv[u] = ld16(raw + i * EPV, nt_x);
    }
  }
#pragma unroll
  for (int u = 0; u < VPT; ++u) {
    const long long i = base + u * TPB;
    if (i >= total) break;
    float g[EPV], x[EPV], o[EPV];
    Elem<T>::unpack(gv[u], g);
    Elem<T>::unpack(xv[u], x);
#pragma unroll
    for (int j = 0; j < EPV; ++j) o[j] = k1[j] * (g[j] - k2[j] - (x[j] - mu[j]) * k3[j]);
    st16(draw + i * EPV, Elem<T>::pack(o), nt_o);
  }
}

template <typename T, int MASK>
void launch_bn_backward(const void* dy, const void* y, const unsigned char* bits, const void* raw, long long rows, int C, const float* mean,
                        const float* invstd, const float* gamma, const float* beta, void* draw, void* dres,
                        float* sums, float* dgamma_acc, float* dbeta_acc, int phase, long long stat_rows,
                        hipStream_t s) {
  const int vc = C / Elem<T>::EPV;
  if (phase != 2) {   // pass 1: per-channel sums
    // every block ends in 2C global atomics on the same words: ~20 ns per block of tail
    const int cap = (int)dastune::get(dastune::BN_REDUCE_BLOCKS), nt = (int)dastune::get(dastune::BN_REDUCE_THREADS);
    const int blocks = (int)std::min<long long>(cap, std::max<long long>(1, rows / 64));
    if (nt == 1024) {
      hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, MASK, 1024>), dim3(blocks), dim3(1024), (1024 / std::min(vc, 1024)) * 2 * C * sizeof(float), s,
                         (const T*)dy, (const T*)y, bits, (const T*)raw, mean, invstd, gamma, beta, rows, C, sums);
    } else {
      hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, MASK, 256>), dim3(blocks), dim3(256), (256 / std::min(vc, 256)) * 2 * C * sizeof(float), s,
                         (const T*)dy, (const T*)y, bits, (const T*)raw, mean, invstd, gamma, beta, rows, C, sums);
    }
  }
  if (phase == 1) return;
  const float inv_n = 1.f / (float)stat_rows;   // statistics population (all ranks' rows for SyncBN)
  const int vpt = std::max(1, (int)dastune::get(dastune::BN_VPT));   // vectors per thread (see das_bn_train_apply)
  const int grid = std::max(1, std::min(grid_for(rows * vc), (int)((rows * vc + (long long)TPB * vpt - 1) / ((long long)TPB * vpt))));
  if (((long long)grid * TPB) % vc == 0) {
    hipLaunchKernelGGL((bn_bwd_apply_kernel<T, MASK, true>), dim3(grid), dim3(TPB), 0, s, (const T*)dy, (const T*)y, bits,
                       (const T*)raw, mean, invstd, gamma, beta, sums, rows, C, (T*)draw, (T*)dres, dgamma_acc,
                       dbeta_acc, inv_n);
  } else {
    hipLaunchKernelGGL((bn_bwd_apply_kernel<T, MASK, false>), dim3(grid), dim3(TPB), 0, s, (const T*)dy, (const T*)y, bits,
                       (const T*)raw, mean, invstd, gamma, beta, sums, rows, C, (T*)draw, (T*)dres, dgamma_acc,
                       dbeta_acc, inv_n);
  }
}
}  // namespace

// Workspace of the weight gradient's partial tiles (workspace.h: one buffer per device and stream): 256 MiB covers every
// schedule of the train step (a few hundred cut tiles of 256 KiB per launch) — growing it later means draining the
// stream, a free and an allocation in the middle of a step.
static float* wgrad_workspace(hipStream_t s, size_t bytes) {
  return dasws::get(dasws::WGRAD, s, bytes, (size_t)256 << 20);
}

namespace {
struct HostWgrad {   // one validated op
  WgradOpS o;
  const char* x;
  const char* dy;
  float* dw;
  int dtype, cls;   // cls: 0 = bf16 ping-pong 256 x 256, 1 = bf16 four waves of 64 x 64 (o.shape), 2 = f32 128 x 128
  int tile_o, tile_n;   // tile edges along Cout / K
};

int wgrad_prepare(const void* x, const void* dy, float* dw, const DasConvDesc* d, HostWgrad& h) {
  if (!x || !dy || !dw || !d) return DAS_ERR_ARG;
  // (Cout need not be a multiple of 8 when dY's rows are padded to one: the loaders take whole 8-channel vectors, the stores
  // — direct, partial-tile reduce — keep to the rows below Cout)
  if (d->Cin % 8 || d->Cout < 1 || d->x_pix_stride % 8 || d->y_pix_stride % 8 || d->y_pix_stride < (d->Cout + 7) / 8 * 8)
    return DAS_ERR_ARG;
  if (d->KH < 1 || d->KW < 1 || d->stride < 1 || d->B < 1 || d->in_up > 1) return DAS_ERR_ARG;
  if (d->dtype != DAS_BF16 && d->dtype != DAS_F32) return DAS_ERR_ARG;
  std::memset(&h.o, 0, sizeof(h.o));   // (the record is part of the schedule cache key: no indeterminate padding)
  WgradOpS& o = h.o;
  long long M = (long long)d->B * d->Ho * d->Wo;
  o.nlev = d->num_levels;
  o.B = d->B;
  if (o.nlev > 1) {
    if (o.nlev > MAXLV || d->stride != 1 || d->KH != d->KW || d->pad != d->KH / 2) return DAS_ERR_ARG;
    M = 0;
    for (int l = 0; l < o.nlev; ++l) {
      o.lvH[l] = d->lvl_H[l]; o.lvW[l] = d->lvl_W[l]; o.lvStart[l] = (int)M;
      M += (long long)d->B * d->lvl_H[l] * d->lvl_W[l];
    }
  }
  for (int l = (o.nlev > 1 ? o.nlev : 0); l < MAXLV; ++l) { o.lvH[l] = 0; o.lvW[l] = 0; o.lvStart[l] = 0x7fffffff; }
  if (M <= 0 || M > 0x7fffff00LL) return DAS_ERR_ARG;   // (the row walker adds a step to a row index in 32 bits)
  h.x = (const char*)x; h.dy = (const char*)dy;
  o.H = d->H; o.W = d->W; o.Cin = d->Cin; o.xps = d->x_pix_stride;
  o.Ho = d->Ho; o.Wo = d->Wo; o.Cout = d->Cout; o.rps = d->y_pix_stride;
  o.KH = d->KH; o.KW = d->KW; o.stride = d->stride; o.pad = d->pad;
  o.M = (int)M; o.K = d->KH * d->KW * d->Cin; o.HoWo = d->Ho * d->Wo;
  o.xbytes = 0;
  h.dw = dw; h.dtype = d->dtype;
  // the ping-pong 256 x 256 kernel for the wide layers (K >= 256, Cout >= 256): -5...19 % with cold operands
  // (tools/dev/wgrad_cold_bench.py)
  const int pp_mink = (int)dastune::get(dastune::WGRAD_PP_MINK);   // 0 = off
  h.cls = d->dtype == DAS_F32 ? 2 : 1;
  if (d->dtype == DAS_BF16 && pp_mink > 0 && o.K >= pp_mink && d->Cout >= 256) {
    const long long npix = o.nlev > 1 ? M : (long long)d->B * d->H * d->W;
    const long long xb = ((npix - 1) * d->x_pix_stride + d->Cin) * 2;
    const long long db = ((M - 1) * d->y_pix_stride + (d->Cout + 7) / 8 * 8) * 2;
    if (xb < 0xFFFFFFF0LL && db < 0xFFFFFFE0LL) {
      o.xbytes = (unsigned)xb;
      h.cls = 0;
    }
  }
  // conv_wgrad_c64_kernel: 3x3, stride 1, 64 -> 64 channels on a plain NHWC tensor
  const long long c64_from = dastune::get(dastune::CONV_C64_MINTILES);
  if (d->dtype == DAS_BF16 && c64_from > 0 && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && d->Cin == 64 &&
      d->Cout == 64 && o.nlev <= 1 && d->Ho == d->H && d->Wo == d->W) {
    const long long sq = (long long)((d->H + 15) / 16) * ((d->W + 15) / 16), npix = (long long)d->B * d->H * d->W;
    const long long xb = ((npix - 1) * d->x_pix_stride + 64) * 2, db = ((npix - 1) * d->y_pix_stride + 64) * 2;
    if (sq * d->B >= c64_from && sq * 256 * 3 <= (long long)d->H * d->W * 4 && xb < 0xFFFFFFF0LL && db < 0xFFFFFFF0LL) {
      o.xbytes = (unsigned)xb;
      h.cls = 3;
    }
  }
  // conv_wgrad_kernel<bf16>: the wave arrangement that pads the (Cout x K) result least — 64 x 256 for layers with <= 64
  // output channels, 256 x 64 for K = 64, 128 x 128 otherwise and on ties (wgrad.shapes = 0: always 128 x 128)
  o.shape = 0;
  if (h.cls == 1 && dastune::get(dastune::WGRAD_SHAPES) != 0) {
    long long best = 0;
    for (int sh = 0; sh < 3; ++sh) {
      const long long to = shape_to(sh), tn = shape_tn(sh);
      const long long area = ((d->Cout + to - 1) / to * to) * ((o.K + tn - 1) / tn * tn);
      if (sh == 0 || area < best) { best = area; o.shape = sh; }
    }
  }
  h.tile_o = h.cls == 0 ? 256 : shape_to(o.shape);
  h.tile_n = h.cls == 0 ? 256 : shape_tn(o.shape);
  const int ntiles = (o.K + h.tile_n - 1) / h.tile_n;
  o.tiles = ((d->Cout + h.tile_o - 1) / h.tile_o) * ntiles;
  return DAS_OK;
}

// ---- the schedule of one launch (see the comment above WgradSched) ----
struct WgradPlan {
  std::vector<WgradUnit> units;
  std::vector<int> first;          // [grid + 1]
  std::vector<WgradRedJob> jobs;
  int grid = 0, partials = 0, max_groups = 1;
  std::vector<int> zero_ops;       // ops whose dW must be zeroed first (atomic reduce groups into a fresh result)
  // device copy
  char* dev = nullptr;
  size_t off_units = 0, off_first = 0, off_ops = 0, off_jobs = 0;
};

// Longest-processing-time-first packing of the units onto `grid` workgroups, XCD by XCD. `bkm` pixel rows per step.
// Cost model (in steps): a unit costs its steps plus a constant for prologue / epilogue; a partial tile costs more
// (its workspace round trip competes with the operand stream and feeds the reduction pass).
void wgrad_schedule(const HostWgrad* const* ops, int n, int grid, int bkm, WgradPlan& plan) {
  constexpr double C_UNIT = 8.0, C_PARTIAL = 10.0;   // (a 256 KiB tile stored, or read-modified-written, per unit)
  const int nx = grid >= 8 ? 8 : 1;             // XCDs (block b runs on XCD b % 8)
  std::vector<long long> steps(n);
  double total = 0;
  for (int i = 0; i < n; ++i) {
    steps[i] = ((long long)ops[i]->o.M + bkm - 1) / bkm;
    total += (double)steps[i] * ops[i]->o.tiles;
  }
  struct Group { int op; long long step0; int nsteps; bool partial; };   // the tiles of one op over one run of steps
  std::vector<Group> best_groups;
  std::vector<int> best_splits;
  double best_cost = 1e300;
  std::vector<double> load_x, load_s;
  auto pack = [&](const std::vector<Group>& gs, std::vector<std::vector<int>>* lists, std::vector<int>* unit_group,
                  std::vector<int>* unit_tile) -> double {
    // order: longest units first
    std::vector<int> order(gs.size());
    for (size_t i = 0; i < gs.size(); ++i) order[i] = (int)i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return gs[a].nsteps > gs[b].nsteps; });
    load_x.assign(nx, 0.0);
    load_s.assign(grid, 0.0);
    for (int gi : order) {
      const Group& g = gs[gi];
      const int tiles = ops[g.op]->o.tiles;
      const double c = g.nsteps + (g.partial ? C_PARTIAL : C_UNIT);
      // (a result with more tiles than an XCD has workgroups goes to several XCDs, a run of consecutive tiles each:
      // those share the dY columns of a Cout row of tiles)
      const int per_x = std::max(1, grid / nx);
      for (int t0 = 0; t0 < tiles; t0 += per_x) {
        const int t1 = std::min(tiles, t0 + per_x);
        int x = 0;
        for (int k = 1; k < nx; ++k)
          if (load_x[k] < load_x[x]) x = k;
        load_x[x] += c * (t1 - t0);
        for (int t = t0; t < t1; ++t) {
          int sl = x, sg = 0;                      // least loaded workgroup of this XCD / of the chip
          for (int b = x; b < grid; b += nx)
            if (load_s[b] < load_s[sl]) sl = b;
          for (int b = 1; b < grid; ++b)
            if (load_s[b] < load_s[sg]) sg = b;
          if (load_s[sl] > load_s[sg] + 0.5 * c) sl = sg;   // (balance first: the XCD is a preference)
          load_s[sl] += c;
          if (lists) {
            (*lists)[sl].push_back((int)unit_group->size());
            unit_group->push_back(gi);
            unit_tile->push_back(t);
          }
        }
      }
    }
    double mk = 0;
    for (int b = 0; b < grid; ++b) mk = std::max(mk, load_s[b]);
    return mk;
  };
  // candidate cut lengths lmax (a tile longer than lmax steps is cut into equal runs): fractions of the per-workgroup
  // quota, and for r = 1, 2, ... rounds the shortest lmax whose units still fit r per workgroup (a launch of 261
  // units on 256 workgroups takes two rounds: whole-number effects decide small launches)
  auto units_at = [&](long long lmax) {
    long long u = 0;
    for (int i = 0; i < n; ++i) u += (long long)ops[i]->o.tiles * ((steps[i] + lmax - 1) / lmax);
    return u;
  };
  const double quota = std::max(1.0, total / grid);
  long long max_steps = 8;
  for (int i = 0; i < n; ++i) max_steps = std::max(max_steps, steps[i]);
  std::vector<long long> cands;
  for (double alpha : {1.0, 0.7, 0.5, 0.35, 0.25}) cands.push_back(std::max<long long>(8, (long long)(quota * alpha + 0.5)));
  for (int r : {1, 2, 3, 4, 6, 8}) {
    long long lo = 8, hi = max_steps;          // smallest lmax in [8, max_steps] with units_at(lmax) <= r * grid
    if (units_at(hi) > (long long)r * grid) continue;
    while (lo < hi) {
      const long long mid = (lo + hi) / 2;
      if (units_at(mid) <= (long long)r * grid) hi = mid; else lo = mid + 1;
    }
    cands.push_back(lo);
  }
  std::sort(cands.begin(), cands.end());
  cands.erase(std::unique(cands.begin(), cands.end()), cands.end());
  for (long long lmax : cands) {
    std::vector<Group> gs;
    std::vector<int> splits(n);
    double npart = 0;
    for (int i = 0; i < n; ++i) {
      long long sp = (steps[i] + lmax - 1) / lmax;
      const long long spb = (steps[i] + sp - 1) / sp;
      sp = (steps[i] + spb - 1) / spb;
      splits[i] = (int)sp;
      for (long long c = 0; c < sp; ++c)
        gs.push_back(Group{i, c * spb, (int)std::min<long long>(spb, steps[i] - c * spb), sp > 1});
      if (sp > 1) npart += (double)sp * ops[i]->o.tiles;
    }
    // makespan + the reduction pass (a launch, and every partial tile read once: ~10 steps of chip time per `grid`)
    const double cost = pack(gs, nullptr, nullptr, nullptr) + (npart > 0 ? 4.0 + 10.0 * npart / grid : 0.0);
    if (cost < best_cost) { best_cost = cost; best_groups = gs; best_splits = splits; }
  }
  std::vector<std::vector<int>> lists(grid);
  std::vector<int> unit_group, unit_tile;
  pack(best_groups, &lists, &unit_group, &unit_tile);
  // partial tiles: contiguous per (op, tile)
  std::vector<int> base(n, -1);
  plan.partials = 0;
  plan.jobs.clear();
  plan.zero_ops.clear();
  plan.max_groups = 1;
  for (int i = 0; i < n; ++i) {
    if (best_splits[i] <= 1) continue;
    base[i] = plan.partials;
    const int tile_o = ops[i]->tile_o, tile_n = ops[i]->tile_n, ntiles = (ops[i]->o.K + tile_n - 1) / tile_n;
    // one small result with hundreds of splits: several reduce groups, each adding atomically
    const int groups = std::max(1, std::min(best_splits[i] / 8, 256 / std::max(1, ops[i]->o.tiles * 16)));
    if (groups > 1) plan.zero_ops.push_back(i);
    plan.max_groups = std::max(plan.max_groups, groups);
    for (int t = 0; t < ops[i]->o.tiles; ++t) {
      WgradRedJob jb{};
      jb.op = i; jb.o0 = (t / ntiles) * tile_o; jb.n0 = (t % ntiles) * tile_n; jb.shape = ops[i]->o.shape;
      jb.base = plan.partials + t * best_splits[i]; jb.splits = best_splits[i]; jb.groups = groups;
      plan.jobs.push_back(jb);
    }
    plan.partials += ops[i]->o.tiles * best_splits[i];
  }
  plan.grid = grid;
  plan.units.clear();
  plan.first.assign(grid + 1, 0);
  for (int b = 0; b < grid; ++b) {
    plan.first[b] = (int)plan.units.size();
    for (int ui : lists[b]) {
      const Group& g = best_groups[unit_group[ui]];
      WgradUnit u{};
      u.op = g.op; u.tile = unit_tile[ui]; u.step0 = (int)g.step0; u.nsteps = g.nsteps;
      if (g.partial) {
        const long long spb = (steps[g.op] + best_splits[g.op] - 1) / best_splits[g.op];
        u.dest = base[g.op] + u.tile * best_splits[g.op] + (int)(g.step0 / spb);
      } else {
        u.dest = -1;
      }
      plan.units.push_back(u);
    }
  }
  plan.first[grid] = (int)plan.units.size();
}

thread_local long long g_last_plan[8] = {0, 0, 0, 0, 0, 0, 0, 0};
thread_local long long g_last_shape = 0;   // wave arrangement of the last launch's first op (conv_wgrad_kernel<bf16>)
std::atomic<long long> g_plan_misses{0};   // schedules built so far (process-wide): a steady training loop stops adding to it

// schedules by op-list signature (shapes, class, grid): a training step repeats the same few lists
struct PlanCache {
  std::mutex mu;
  std::map<std::string, std::unique_ptr<WgradPlan>> plans;
};
PlanCache& plan_cache() {
  static PlanCache c;
  return c;
}

// Launch the ops of one class: main kernel (persistent, one resident wave of workgroups), then the reduction of the
// tiles that were cut.
template <typename MAP>
int wgrad_launch_class(HostWgrad** ops, int n, int cls, int accumulate, hipStream_t s) {
  // pixel rows per step: bf16 plain kernel 32 (32 KiB of LDS per workgroup, three workgroups resident per CU, register
  // bound: more independent DMA -> MFMA chains in flight than two workgroups of 64-row steps, +10...14 % measured)
  const int bkm = cls == 1 && dastune::get(dastune::WGRAD_BKM) == 64 ? 64 : 32;
  // grid = ONE resident wave of workgroups (ping-pong: 1 per CU; bf16 128 x 128: 3 per CU; f32: 2 per CU)
  long long grid;
  const long long ucus = dastune::usable_cus();   // (the device's CUs minus comm.reserved_cus)
  if (cls == 0) {
    const long long forced = dastune::get(dastune::WGRAD_PP_BLOCKS);
    // (0: one workgroup per usable CU, or the calling thread's share of them — das_wgrad_pp_share: the launches a side stream
    // runs beside the main stream's kernels)
    grid = forced > 0 ? forced : std::max<long long>(8, ucus / dastune::pp_share_den());
  } else {
    const long long forced = dastune::get(dastune::WGRAD_BLOCKS);
    grid = forced > 0 ? forced : (cls == 1 && bkm == 32 ? 3 : 2) * ucus;
  }
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return DAS_ERR_LAUNCH;
  std::string key;
  key.reserve(32 + (size_t)n * sizeof(WgradOpS));
  const int head[4] = {dev, cls, (int)grid, bkm};
  key.append(reinterpret_cast<const char*>(head), sizeof(head));
  for (int i = 0; i < n; ++i) key.append(reinterpret_cast<const char*>(&ops[i]->o), sizeof(WgradOpS));
  WgradPlan* plan = nullptr;
  {
    PlanCache& pc = plan_cache();
    std::lock_guard<std::mutex> lock(pc.mu);
    auto it = pc.plans.find(key);
    if (it == pc.plans.end()) {
      if (pc.plans.size() >= 512) {   // (shape sweeps: drop everything; launches in flight keep running off their copies only
        if (hipDeviceSynchronize() != hipSuccess) return DAS_ERR_LAUNCH;   //  until they finish, so drain first)
        for (auto& kv : pc.plans)
          if (kv.second->dev) (void)hipFree(kv.second->dev);
        pc.plans.clear();
      }
      g_plan_misses.fetch_add(1, std::memory_order_relaxed);
      std::unique_ptr<WgradPlan> np(new WgradPlan);
      wgrad_schedule(ops, n, (int)grid, bkm, *np);
      auto al = [](size_t v) { return (v + 255) / 256 * 256; };
      np->off_units = 0;
      np->off_first = al(np->units.size() * sizeof(WgradUnit));
      np->off_ops = np->off_first + al(np->first.size() * sizeof(int));
      np->off_jobs = np->off_ops + al((size_t)n * sizeof(WgradOpS));
      const size_t bytes = np->off_jobs + al(np->jobs.size() * sizeof(WgradRedJob)) + 256;
      std::vector<char> host(bytes, 0);
      std::memcpy(host.data() + np->off_units, np->units.data(), np->units.size() * sizeof(WgradUnit));
      std::memcpy(host.data() + np->off_first, np->first.data(), np->first.size() * sizeof(int));
      for (int i = 0; i < n; ++i) std::memcpy(host.data() + np->off_ops + (size_t)i * sizeof(WgradOpS), &ops[i]->o, sizeof(WgradOpS));
      if (!np->jobs.empty())
        std::memcpy(host.data() + np->off_jobs, np->jobs.data(), np->jobs.size() * sizeof(WgradRedJob));
      if (hipMalloc(reinterpret_cast<void**>(&np->dev), bytes) != hipSuccess) return DAS_ERR_LAUNCH;
      if (hipMemcpy(np->dev, host.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return DAS_ERR_LAUNCH;
      it = pc.plans.emplace(std::move(key), std::move(np)).first;
    }
    plan = it->second.get();
  }
  {   // das_wgrad_last_plan
    long long direct = 0, longest = 0;
    for (const WgradUnit& u : plan->units) direct += u.dest < 0;
    for (int b = 0; b < plan->grid; ++b) longest = std::max<long long>(longest, plan->first[b + 1] - plan->first[b]);
    const long long st[8] = {cls, plan->grid, (long long)plan->units.size(), direct, plan->partials, (long long)plan->jobs.size(),
                             longest, plan->max_groups};
    std::memcpy(g_last_plan, st, sizeof(st));
    g_last_shape = ops[0]->o.shape;
  }
  float* ws = nullptr;
  if (plan->partials > 0) {
    ws = wgrad_workspace(s, (size_t)plan->partials * MAP::SLOTS * 4 * sizeof(float));
    if (!ws) return DAS_ERR_LAUNCH;
  }
  WgradSched g;
  g.units = reinterpret_cast<const WgradUnit*>(plan->dev + plan->off_units);
  g.first = reinterpret_cast<const int*>(plan->dev + plan->off_first);
  g.ops = reinterpret_cast<const WgradOpS*>(plan->dev + plan->off_ops);
  g.ws = ws;
  g.accumulate = accumulate;
  for (int i = 0; i < n; ++i) g.ptr[i] = WgradPtrs{ops[i]->x, ops[i]->dy, ops[i]->dw};
  if (!accumulate)
    for (int i : plan->zero_ops)
      if (hipMemsetAsync(ops[i]->dw, 0, sizeof(float) * (size_t)ops[i]->o.Cout * ops[i]->o.K, s) != hipSuccess)
        return DAS_ERR_LAUNCH;
  if (cls == 0) {
    const size_t sm = 4 * 4 * 32 * 256;
    static bool attr_set = false;
    if (!attr_set) {
      if (hipFuncSetAttribute((const void*)conv_wgrad_pp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm) !=
          hipSuccess)
        return DAS_ERR_LAUNCH;
      attr_set = true;
    }
    dastune::note_kernel("conv_wgrad_pp_kernel");
    hipLaunchKernelGGL(conv_wgrad_pp_kernel, dim3((unsigned)grid), dim3(512), sm, s, g);
  } else if (cls == 1) {
    dastune::note_kernel("conv_wgrad_kernel");
    const size_t sm = 2 * (size_t)bkm * 640;   // two buffers of (dY tile + X tile): 512 B of rows per step at 128 x 128, 640 B at 64 x 256 / 256 x 64
    if (bkm == 32) {
      hipLaunchKernelGGL((conv_wgrad_kernel<bf16_t, 32>), dim3((unsigned)grid), dim3(256), sm, s, g);
    } else {
      hipLaunchKernelGGL((conv_wgrad_kernel<bf16_t, 64>), dim3((unsigned)grid), dim3(256), sm, s, g);
    }
  } else {
    dastune::note_kernel("conv_wgrad_kernel");
    const size_t sm = 2 * 2 * 32 * 512;
    hipLaunchKernelGGL((conv_wgrad_kernel<float, 32>), dim3((unsigned)grid), dim3(256), sm, s, g);
  }
  DAS_CHECK_LAUNCH();
  if (!plan->jobs.empty()) {
    WgradRedArgs r;
    r.jobs = reinterpret_cast<const WgradRedJob*>(plan->dev + plan->off_jobs);
    r.ops = g.ops;
    r.ws = ws;
    r.accumulate = accumulate;
    for (int i = 0; i < n; ++i) r.dw[i] = ops[i]->dw;
    hipLaunchKernelGGL(wgrad_reduce_kernel<MAP>, dim3((unsigned)(MAP::SLOTS / 256 * plan->max_groups), (unsigned)plan->jobs.size()),
                       dim3(256), 0, s, r);
    DAS_CHECK_LAUNCH();
  }
  return DAS_OK;
}
}  // namespace

extern "C" int das_conv2d_wgrad_batch(int n, const void* const* xs, const void* const* dys, float* const* dws,
                                      const DasConvDesc* descs, int accumulate, void* stream) {
  DAS_PROF(stream);
  if (n < 1 || n > WG_MAXOPS || !xs || !dys || !dws || !descs) return DAS_ERR_ARG;
  HostWgrad ops[WG_MAXOPS];
  for (int i = 0; i < n; ++i) {
    const int rc = wgrad_prepare(xs[i], dys[i], dws[i], &descs[i], ops[i]);
    if (rc != DAS_OK) return rc;
    for (int j = 0; j < i; ++j)   // two ops adding into one dW inside one launch would race
      if (dws[j] == dws[i]) return DAS_ERR_ARG;
  }
  hipStream_t s = (hipStream_t)stream;
  for (int i = 0; i < n; ++i) {   // class 3 (3x3, 64 -> 64): one persistent launch + its reduction per op
    if (ops[i].cls != 3) continue;
    const WgradOpS& o = ops[i].o;
    const int ntiles = ((o.H + 15) / 16) * ((o.W + 15) / 16) * o.B;
    const int grid = std::min(ntiles, dastune::usable_cus());
    float* ws = wgrad_workspace(s, (size_t)grid * 36864 * sizeof(float));
    if (!ws) return DAS_ERR_LAUNCH;
    const size_t sm = 2 * (size_t)(WC64_DY + WC64_PATCH);
    static bool attr_set = false;
    if (!attr_set) {
      if (hipFuncSetAttribute((const void*)conv_wgrad_c64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm) != hipSuccess)
        return DAS_ERR_LAUNCH;
      attr_set = true;
    }
    const unsigned dbytes = (unsigned)((((long long)o.M - 1) * o.rps + 64) * 2);
    dastune::note_kernel("conv_wgrad_c64_kernel");
    hipLaunchKernelGGL(conv_wgrad_c64_kernel, dim3(grid), dim3(704), sm, s, ops[i].x, ops[i].dy, ws, o.B, o.H, o.W, o.xps, o.rps,
                       o.xbytes, dbytes, ntiles);
    DAS_CHECK_LAUNCH();
    hipLaunchKernelGGL(wgrad_c64_reduce_kernel, dim3(576), dim3(256), 0, s, ws, grid, ops[i].dw, accumulate);
    DAS_CHECK_LAUNCH();
  }
  for (int cls = 0; cls < 3; ++cls) {
    HostWgrad* sel[WG_MAXOPS];
    int k = 0;
    for (int i = 0; i < n; ++i)
      if (ops[i].cls == cls) sel[k++] = &ops[i];
    if (k == 0) continue;
    const int rc = cls == 0 ? wgrad_launch_class<AccMap256>(sel, k, cls, accumulate, s)
                            : wgrad_launch_class<AccMap128>(sel, k, cls, accumulate, s);
    if (rc != DAS_OK) return rc;
  }
  return DAS_OK;
}

extern "C" int das_wgrad_last_plan(long long* out, int n) {
  if (!out || n < 1) return DAS_ERR_ARG;
  for (int i = 0; i < n; ++i)
    out[i] = i < 8 ? g_last_plan[i] : (i == 8 ? g_plan_misses.load(std::memory_order_relaxed) : (i == 9 ? g_last_shape : 0));
  return DAS_OK;
}

extern "C" int das_conv2d_wgrad_nhwc(const void* x, const void* dy, float* dw, const DasConvDesc* d, int accumulate,
                                     void* stream) {
  DAS_PROF(stream);
  return das_conv2d_wgrad_batch(1, &x, &dy, &dw, d, accumulate, stream);
}

static int colsum_impl(const void* x, int dtype, long long rows, int nout, int pix_stride, float* out, bool accumulate, void* stream) {
  const int C = (nout + 7) / 8 * 8;   // columns walked (vectors of 8 / 4): the row holds them (pix_stride >= C)
  if (!x || !out || rows <= 0 || nout < 1 || pix_stride % 8 || pix_stride < C || C > 2048) return DAS_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (!accumulate && hipMemsetAsync(out, 0, sizeof(float) * nout, s) != hipSuccess) return DAS_ERR_LAUNCH;
  const int blocks = (int)std::min<long long>(256, std::max<long long>(1, rows / 64));
  if (dtype == DAS_BF16) {
    hipLaunchKernelGGL(colsum_kernel<bf16_t>, dim3(blocks), dim3(TPB), (TPB / std::min(C / 8, TPB)) * C * sizeof(float), s, (const bf16_t*)x, rows, C,
                       pix_stride, out, nout);
  } else if (dtype == DAS_F32) {
    hipLaunchKernelGGL(colsum_kernel<float>, dim3(blocks), dim3(TPB), (TPB / std::min(C / 4, TPB)) * C * sizeof(float), s, (const float*)x, rows, C,
                       pix_stride, out, nout);
  } else {
    return DAS_ERR_ARG;
  }
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}
extern "C" int das_colsum(const void* x, int dtype, long long rows, int C, int pix_stride, float* out, void* stream) {
  DAS_PROF(stream);
  return colsum_impl(x, dtype, rows, C, pix_stride, out, false, stream);
}
extern "C" int das_colsum_acc(const void* x, int dtype, long long rows, int C, int pix_stride, float* out, void* stream) {
  DAS_PROF(stream);
  return colsum_impl(x, dtype, rows, C, pix_stride, out, true, stream);
}

static int bn_train_backward_impl(const void* dy, const void* y, const unsigned char* bits, const void* raw, int dtype,
                                  long long rows, int C, const float* mean, const float* invstd, const float* gamma,
                                  const float* beta, int relu, void* draw, void* dres, float* sums, int sums_prezeroed,
                                  float* dgamma_acc, float* dbeta_acc, int phase, long long stat_rows, void* stream) {
  if (!dy || !raw || !mean || !invstd || !gamma || !sums || rows <= 0 || C % 8 || C > 2048) return DAS_ERR_ARG;
  if (phase < 0 || phase > 2 || (phase != 1 && !draw) || stat_rows < rows) return DAS_ERR_ARG;
  if (relu && !y && !bits && !beta) return DAS_ERR_ARG;
  if ((dgamma_acc == nullptr) != (dbeta_acc == nullptr)) return DAS_ERR_ARG;
  if (dtype != DAS_BF16 && dtype != DAS_F32) return DAS_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (phase != 2 && !sums_prezeroed && hipMemsetAsync(sums, 0, sizeof(float) * 2 * C, s) != hipSuccess)
    return DAS_ERR_LAUNCH;
  const int mask = !relu ? 0 : ((y || bits) ? 1 : 2);
#define DAS_BN_BWD(T, MASK)                                                                                         \
  launch_bn_backward<T, MASK>(dy, y, bits, raw, rows, C, mean, invstd, gamma, beta, draw, dres, sums, dgamma_acc, dbeta_acc, \
                              phase, stat_rows, s)
  if (dtype == DAS_BF16) {
    if (mask == 0) DAS_BN_BWD(bf16_t, 0); else if (mask == 1) DAS_BN_BWD(bf16_t, 1); else DAS_BN_BWD(bf16_t, 2);
  } else {
    if (mask == 0) DAS_BN_BWD(float, 0); else if (mask == 1) DAS_BN_BWD(float, 1); else DAS_BN_BWD(float, 2);
  }
#undef DAS_BN_BWD
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}

extern "C" int das_bn_train_backward_phase(const void* dy, const void* y, const void* raw, int dtype, long long rows,
                                           int C, const float* mean, const float* invstd, const float* gamma,
                                           const float* beta, int relu, void* draw, void* dres, float* sums,
                                           int sums_prezeroed, float* dgamma_acc, float* dbeta_acc, int phase,
                                           long long stat_rows, void* stream) {
  DAS_PROF(stream);
  return bn_train_backward_impl(dy, y, nullptr, raw, dtype, rows, C, mean, invstd, gamma, beta, relu, draw, dres, sums,
                                sums_prezeroed, dgamma_acc, dbeta_acc, phase, stat_rows, stream);
}

extern "C" int das_bn_train_backward_bits(const void* dy, const void* y_relu_bits, const void* raw, int dtype, long long rows,
                                          int C, const float* mean, const float* invstd, const float* gamma, void* draw,
                                          void* dres, float* sums, int sums_prezeroed, float* dgamma_acc, float* dbeta_acc,
                                          void* stream) {
  DAS_PROF(stream);
  if (!y_relu_bits) return DAS_ERR_ARG;
  return bn_train_backward_impl(dy, nullptr, (const unsigned char*)y_relu_bits, raw, dtype, rows, C, mean, invstd, gamma,
                                nullptr, 1, draw, dres, sums, sums_prezeroed, dgamma_acc, dbeta_acc, 0, rows, stream);
}

extern "C" int das_bn_train_backward_bits_phase(const void* dy, const void* y_relu_bits, const void* raw, int dtype,
                                                long long rows, int C, const float* mean, const float* invstd,
                                                const float* gamma, void* draw, void* dres, float* sums, int sums_prezeroed,
                                                float* dgamma_acc, float* dbeta_acc, int phase, long long stat_rows,
                                                void* stream) {
  DAS_PROF(stream);
  if (!y_relu_bits) return DAS_ERR_ARG;
  return bn_train_backward_impl(dy, nullptr, (const unsigned char*)y_relu_bits, raw, dtype, rows, C, mean, invstd, gamma,
                                nullptr, 1, draw, dres, sums, sums_prezeroed, dgamma_acc, dbeta_acc, phase, stat_rows, stream);
}

extern "C" int das_bn_backward_apply(const void* dz, const void* raw, int dtype, long long rows, int C, const float* mean,
                                     const float* invstd, const float* gamma, const float* sums, int sums_slots,
                                     void* draw, float* dgamma_acc, float* dbeta_acc, long long stat_rows, void* stream) {
  DAS_PROF(stream);
  if (!dz || !raw || !mean || !invstd || !gamma || !sums || !draw || rows <= 0 || C % 8 || C > 2048) return DAS_ERR_ARG;
  if (sums_slots < 1 || sums_slots > 64 || stat_rows < rows) return DAS_ERR_ARG;
  if ((dgamma_acc == nullptr) != (dbeta_acc == nullptr)) return DAS_ERR_ARG;
  if (dtype != DAS_BF16 && dtype != DAS_F32) return DAS_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const float inv_n = 1.f / (float)stat_rows;
  const int vc = C / (dtype == DAS_BF16 ? 8 : 4);
  const int vpt = std::max(1, (int)dastune::get(dastune::BN_VPT));
  const long long stream_from = dastune::get(dastune::BN_STREAM_MINBYTES);
  if (stream_from > 0 && rows * C * (dtype == DAS_BF16 ? 2 : 4) >= stream_from && TPB % vc == 0 &&
      rows * vc / (TPB * 4) < 0x7fffffffLL) {
    const float* folded = sums;
    if (sums_slots > 1) {
      float* ws = dasws::get(dasws::BN_FOLD, s, 2 * (size_t)C * sizeof(float), 64 << 10);
      if (!ws) return DAS_ERR_LAUNCH;
      hipLaunchKernelGGL(fold_slots_kernel, dim3((2 * C + 255) / 256), dim3(256), 0, s, sums, sums_slots, 2 * C, ws);
      DAS_CHECK_LAUNCH();
      folded = ws;
    }
    constexpr int VPT = DAS_BN_STREAM_VPT;
    const int sgrid = (int)((rows * vc + TPB * VPT - 1) / (TPB * VPT));
    const size_t ssm = 4 * (size_t)C * sizeof(float);
#define DAS_BN_DZS(T)                                                                                                   \
  hipLaunchKernelGGL((bn_bwd_apply_dz_stream_kernel<T, VPT>), dim3(sgrid), dim3(TPB), ssm, s, (const T*)dz, (const T*)raw, \
                     mean, invstd, gamma, folded, rows, C, (T*)draw, dgamma_acc, dbeta_acc, inv_n,                       \
                     (int)dastune::get(dastune::BN_NT_BWD))
    if (dtype == DAS_BF16) DAS_BN_DZS(bf16_t); else DAS_BN_DZS(float);
#undef DAS_BN_DZS
    DAS_CHECK_LAUNCH();
    dastune::note_kernel("bn_bwd_apply_dz_stream_kernel");
    return DAS_OK;
  }
  dastune::note_kernel("bn_bwd_apply_dz_kernel");
  // (at most 2048 workgroups: each folds the [slots][2C] sums into LDS before it starts)
  int grid = std::max(1, std::min(std::min(grid_for(rows * vc), 2048),
                                  (int)((rows * vc + (long long)TPB * vpt - 1) / ((long long)TPB * vpt))));
  const bool fixed = ((long long)grid * TPB) % vc == 0;
  const size_t sm = 2 * (size_t)C * sizeof(float);
#define DAS_BN_DZ(T, F)                                                                                                  \
  hipLaunchKernelGGL((bn_bwd_apply_dz_kernel<T, F>), dim3(grid), dim3(TPB), sm, s, (const T*)dz, (const T*)raw, mean,   \
                     invstd, gamma, sums, sums_slots, rows, C, (T*)draw, dgamma_acc, dbeta_acc, inv_n)
  if (dtype == DAS_BF16) {
    if (fixed) DAS_BN_DZ(bf16_t, true); else DAS_BN_DZ(bf16_t, false);
  } else {
    if (fixed) DAS_BN_DZ(float, true); else DAS_BN_DZ(float, false);
  }
#undef DAS_BN_DZ
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}

extern "C" int das_bn_train_backward(const void* dy, const void* y, const void* raw, int dtype, long long rows, int C,
                                     const float* mean, const float* invstd, const float* gamma, const float* beta,
                                     int relu, void* draw, void* dres, float* sums, int sums_prezeroed,
                                     float* dgamma_acc, float* dbeta_acc, void* stream) {
  DAS_PROF(stream);
  return das_bn_train_backward_phase(dy, y, raw, dtype, rows, C, mean, invstd, gamma, beta, relu, draw, dres, sums,
                                     sums_prezeroed, dgamma_acc, dbeta_acc, 0, rows, stream);
}
