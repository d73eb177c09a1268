// Shared pieces of the implicit-GEMM convolution kernels: parameter block, MFMA wrappers and
// the fused epilogue (scale/shift -> LDS-staged C tile -> residual / ReLU / BN statistics ->
// 16-byte coalesced NHWC stores).
#pragma once
#include <type_traits>
#include "common.h"

// Dev builds (make stamps -> libdas_hip_stamps.so, never loaded by the product): thread 0 of every workgroup of the
// tile kernels records the 100 MHz wall clock at up to eight points (entry, main loop, epilogue phases, exit) into a
// host-provided buffer, 8 x u64 per workgroup: tools/dev/conv_stamps.py / t2_bench.py split a launch into prologue /
// K loop / epilogue phases from them.
#ifdef DAS_STAMPS
static __device__ unsigned long long* g_das_stamps;
#define DAS_STAMP(i)                                                                                   \
  do {                                                                                                 \
    if (g_das_stamps && threadIdx.x == 0) g_das_stamps[(size_t)blockIdx.x * 8 + (i)] = wall_clock64(); \
  } while (0)
#else
#define DAS_STAMP(i) do {} while (0)
#endif

namespace dasconv {

constexpr int BM = 128;
constexpr int MAXLV = DAS_MAX_LEVELS;

struct ConvP {
  const char* x;
  const char* w;
  char* y;
  const float* scale;
  const float* shift;
  const char* res;
  float* stats;
  int stat_slots;  // stats is [stat_slots][2][Cout]; workgroup b adds into slot b % stat_slots
  int H, W, Cin, xps;
  int Ho, Wo, Cout, yps;
  int KH, KW, stride, pad;
  int relu_in, relu, rps;
  int up_sh;  // log2 of the input zero-upsampling factor (dgrad of a strided conv): tap coordinate t
              // reads x[t >> up_sh] when t is a multiple of 1 << up_sh, else contributes zero
  int M, K, HoWo, ntiles, nblocks;
  int m_base;       // first output row of this launch's tile 0 (a conv may be covered by two launches: see launch_bn)
  int mstep;        // 0, or the rows a 256-row tile kernel's tile covers (< 256, a multiple of 16): tile t starts at row
                    // m_base + t * mstep and its rows from mstep on are treated as rows past M (balanced_mstep, conv_igemm.hip)
  unsigned xbytes;  // addressable bytes of x from its base (0 if >= 4 GiB): range of the buffer descriptor
  // Fused BatchNorm-backward reduction (data-gradient launches, DasConvDesc.bnb_*): the value about to be stored,
  // g = conv + residual, is the gradient wrt the OUTPUT of a train-mode BatchNorm (+ReLU) layer whose pre-norm
  // tensor is bnb_raw. The epilogue stores dZ = g * mask instead and adds [sum dZ | sum dZ * xhat] per channel into
  // `stats` (same slot scheme as the forward statistics): the separate reduce pass over (dY, raw) disappears.
  const char* bnb_raw;   // (M, Cout) pre-norm tensor, pixel stride bnb_ps; nullptr = feature off
  const char* bnb_y;     // (M, Cout) post-ReLU output: mask = y > 0 (needed when a residual entered before the ReLU);
                         // nullptr with bnb_relu: mask recomputed as bn_affine(raw) > 0
  const unsigned char* bnb_bits;   // instead of bnb_y: its ReLU mask as one byte per 16-byte vector (common.h: relu_bits)
  const unsigned char* res_bits;   // (with bnb_raw) the residual enters masked by these bits: res * mask — a gradient whose
                                   // ReLU mask was recorded as bits, so that the masked copy never has to be written
  const float* bnb_mean;
  const float* bnb_invstd;
  const float* bnb_gamma;
  const float* bnb_beta;
  int bnb_relu, bnb_ps;
  // split-K (conv_glds_kernel on small-M, long-K layers): blockIdx.y = split s covers K steps [nk*s/ksplit, nk*(s+1)/ksplit)
  // and stores its raw f32 accumulators to ws[s][M][Cout]; splitk_finish_kernel sums the slabs and runs the epilogue.
  float* ws;
  int ksplit;
  int* sk_cnt;      // != nullptr: split-K finished INSIDE the tile kernel — one arrival counter per tile (zero between launches): the
                    // workgroup that draws ticket ksplit - 1 sums the slabs (fixed order) and runs the epilogue itself
  // sub-grid output (DasConvDesc.out_sub): output pixel (b, i, j) is row (b * oH + 2 i + oph) * oW + 2 j + opw of y,
  // of the residual and of the bnb_* tensors
  int osub, oph, opw, oH, oW;
  // ragged multi-level input (stride 1, "same" padding): rows of level l start at lvStart[l]
  int nlev, B;
  int lvH[MAXLV], lvW[MAXLV], lvStart[MAXLV];
};

// conv_stem.hip: the 7 x 7 stride-2 stem conv (8 stored input channels -> 64); false when the launch is not its
bool try_launch_stem7x7(const ConvP& p, hipStream_t s);

// ---- LDS-DMA through a buffer descriptor (inline asm: the compiler must not wait for it, cdna guide 5.7)
typedef int v4i_t __attribute__((ext_vector_type(4)));
// Raw buffer descriptor (wave-uniform): base, num_records = bytes, stride 0.
__device__ __forceinline__ v4i_t make_rsrc(const void* base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  v4i_t r;
  r.x = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
  r.y = __builtin_amdgcn_readfirstlane((int)(a >> 32));
  r.z = __builtin_amdgcn_readfirstlane((int)bytes);
  r.w = 0x00020000;
  return r;
}
// Lane l's 16 bytes at byte offset `voff` land at lds_byte_addr + 16*l (M0 = LDS destination of lane 0,
// restored afterwards); offsets >= num_records deliver zeros (probe: tools/dev/probe/buf_lds_probe.hip).
// The caller owns the vmcnt accounting.
__device__ __forceinline__ void dma16_buf(unsigned voff, v4i_t rsrc, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(rsrc), "s"(lds_byte_addr)
               : "memory");
}

// ... the same with the non-temporal cache policy (streamed operands nobody re-reads soon)
__device__ __forceinline__ void dma16_buf_nt(unsigned voff, v4i_t rsrc, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen nt lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(rsrc), "s"(lds_byte_addr)
               : "memory");
}

template <typename T>
__device__ __forceinline__ uint4 relu_vec(uint4 v);
template <>
__device__ __forceinline__ uint4 relu_vec<float>(uint4 v) {
  v.x = (v.x >> 31) ? 0u : v.x; v.y = (v.y >> 31) ? 0u : v.y;
  v.z = (v.z >> 31) ? 0u : v.z; v.w = (v.w >> 31) ? 0u : v.w;
  return v;
}
__device__ __forceinline__ uint32_t relu_bf16x2(uint32_t u) {
  uint32_t lo = (u & 0x8000u) ? 0u : (u & 0xffffu);
  uint32_t hi = (u & 0x80000000u) ? 0u : (u & 0xffff0000u);
  return lo | hi;
}
template <>
__device__ __forceinline__ uint4 relu_vec<bf16_t>(uint4 v) {
  return make_uint4(relu_bf16x2(v.x), relu_bf16x2(v.y), relu_bf16x2(v.z), relu_bf16x2(v.w));
}

// One 16-byte fragment pair -> accumulate. bf16: one 16x16x32 MFMA; f32: four exact-f32 16x16x4 MFMAs
// (lane l holds k-group l>>4 of both operands, so the 4 dwords are 4 consistent k slices).
template <typename T>
__device__ __forceinline__ void mma(const uint4& a, const uint4& b, f32x4_t& c);
template <>
__device__ __forceinline__ void mma<bf16_t>(const uint4& a, const uint4& b, f32x4_t& c) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0,
                                              0, 0);
}
template <>
__device__ __forceinline__ void mma<float>(const uint4& a, const uint4& b, f32x4_t& c) {
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
}

// BMT = pixel rows per workgroup tile: 128 (4 waves) or 256 (8 waves, BN = 128 only: the weight tile is
// then shared by twice as many pixel rows, which is what relieves the L2 -> LDS path on the big layers).
template <int BN, int BMT = 128>
struct Tiling {
  // threads per workgroup: 2 per pixel row, except the 288-row variant of the 256-channel tile (8 waves x 144 x 64)
  static constexpr int NT = (BN == 256) ? 512 : BMT * 2;
  static constexpr int TM = (BN == 256) ? BMT / 32 : (BN == 128) ? 4 : 2;   // 16-row pixel tiles per wave
  static constexpr int TN = (BN >= 64) ? 4 : 2;                             // 16-row channel tiles per wave
  // BN = 256 (BMT = 256 or 288): 2 x 4 waves of BMT / 2 pixels x 64 channels
  static __device__ __forceinline__ int wave_m0(int wave) {
    return BN == 256 ? (wave & 1) * (BMT / 2) : BMT == 256 ? (wave & 3) * 64 : ((BN == 128) ? (wave & 1) * 64 : wave * 32);
  }
  static __device__ __forceinline__ int wave_n0(int wave) {
    return BN == 256 ? (wave >> 1) * 64 : BMT == 256 ? (wave >> 2) * 64 : ((BN == 128) ? (wave >> 1) * 64 : 0);
  }
};

template <typename OT, int BN, int BMT = 128, typename TL = Tiling<BN, BMT>>
constexpr size_t epilogue_smem_bytes() {
  size_t ctile = (size_t)BMT * (BN * sizeof(OT) + 16);
  size_t red = 2 * (size_t)(TL::NT / (BN * sizeof(OT) / 16)) * BN * 4;
  return ctile > red ? ctile : red;
}

// acc[a][b][j]: channel n0 + wave_n0 + a*16 + (lane>>4)*4 + j, pixel m0 + wave_m0 + b*16 + (lane&15).
// Must be entered after a barrier that ends all LDS reads of the main loop.
// T2D: the tile is a 16 x 16 pixel square of one image (conv3x3_c64_kernel): `m0` is then the tile index (images x
// ceil(H / 16) x ceil(W / 16)), and tile row ml = (row of the square) * 16 + column. Pixels of a border square that lie
// outside the image come back as p.M ("no such row": skipped by the callers, in any order).
template <bool T2D>
__device__ __forceinline__ int tile_row_m(const ConvP& p, int m0, int ml) {
  if (!T2D) return m0 + ml;
  const int tw = (p.W + 15) >> 4, per_img = ((p.H + 15) >> 4) * tw;
  const int b = m0 / per_img, t = m0 - b * per_img;
  const int th = t / tw;
  const int h = th * 16 + (ml >> 4), w = (t - th * tw) * 16 + (ml & 15);
  return (h < p.H && w < p.W) ? (b * p.H + h) * p.W + w : p.M;
}

// `carry` (persistent kernels): the thread's per-channel statistic sums [2][16 / sizeof(OT)] live in the CALLER's
// registers across tiles (a thread keeps its channels: vec = tid % VR) and conv_epilogue_flush_stats reduces and adds
// them once per workgroup at the end of the launch, instead of an LDS reduction, two barriers and 2 * BN atomics per tile.
#ifndef DAS_EPI_CH
#define DAS_EPI_CH 4
#endif
#ifndef DAS_EPI_DEEP
#define DAS_EPI_DEEP 1
#endif
struct NoCarry {};
// BITS = false: the ReLU-mask-as-bits operands (ConvP::bnb_bits / res_bits) are compiled out — conv3x3_c64_kernel's epilogue is
// not overlapped with anything (one workgroup owns the CU), and the step's launches of that kernel never carry bits.
// MF = 32: the accumulators are 32 x 32 MFMA tiles (conv_glds4_kernel<.., MF = 32>): acc[a][b][r] = channel n0 + wave_n0 + a*32 +
// 8*(r>>2) + 4*(lane>>5) + (r&3), pixel m0 + wave_m0 + b*32 + (lane&31) — only the staging into the LDS C tile differs.
// BNB = false: the fused-BatchNorm-backward operands (ConvP::bnb_*) are compiled out — for the launches without them (forward,
// plain data gradients) on the kernels that are instantiated both ways: with the code merely branched around, those launches ran
// 2-7 % slower (profiles/r06_epilogue_bnb_template.md). PLAIN = true (with BNB = false): no residual, no ReLU, no sub-grid output
// either — the training forward (statistics only) and the plain data gradients.
template <typename OT, int BN, int BMT = 128, typename TL = Tiling<BN, BMT>, bool T2D = false, bool BITS = true, int MF = 16,
          bool BNB = true, bool PLAIN = false, typename Acc, typename Carry = NoCarry>
__device__ __forceinline__ void conv_epilogue(Acc& acc, const ConvP& p, char* smem, int m0, int n0, Carry&& carry = Carry{}) {
  constexpr bool CARRY = !std::is_same<typename std::decay<Carry>::type, NoCarry>::value;   // (a float[2 * EPVO] otherwise)
  constexpr int TM = TL::TM, TN = TL::TN, NT = TL::NT;
  constexpr int EPVO = 16 / (int)sizeof(OT);
  constexpr int CS = BN * (int)sizeof(OT) + 16;  // padded C-tile row stride in bytes
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_m0 = TL::wave_m0(wave), wave_n0 = TL::wave_n0(wave);
  const int ch4 = (lane >> 4) * 4;
  if constexpr (MF == 32) {
    static_assert(TN == 4 && TM % 2 == 0, "32 x 32 accumulators: 64 channels x a multiple of 32 pixels per wave");
#pragma unroll
    for (int a = 0; a < TN / 2; ++a) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int nl = wave_n0 + a * 32 + g * 8 + (lane >> 5) * 4;
        const int n = n0 + nl;
        float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
        if (n < p.Cout) {
          if (p.scale) { const float4 t = *reinterpret_cast<const float4*>(p.scale + n); sc[0] = t.x; sc[1] = t.y; sc[2] = t.z; sc[3] = t.w; }
          if (p.shift) { const float4 t = *reinterpret_cast<const float4*>(p.shift + n); sh[0] = t.x; sh[1] = t.y; sh[2] = t.z; sh[3] = t.w; }
        }
#pragma unroll
        for (int b = 0; b < TM / 2; ++b) {
          const int ml = wave_m0 + b * 32 + (lane & 31);
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = acc[a][b][g * 4 + j] * sc[j] + sh[j];
          char* dst = smem + ml * CS + nl * (int)sizeof(OT);
          if (sizeof(OT) == 2) {
            *reinterpret_cast<uint2*>(dst) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
          } else {
            *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
          }
        }
      }
    }
  } else {
#pragma unroll
  for (int a = 0; a < TN; ++a) {
    const int nl = wave_n0 + a * 16 + ch4;
    const int n = n0 + nl;
    float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
    if (n < p.Cout) {
      if (p.scale) { const float4 t = *reinterpret_cast<const float4*>(p.scale + n); sc[0] = t.x; sc[1] = t.y; sc[2] = t.z; sc[3] = t.w; }
      if (p.shift) { const float4 t = *reinterpret_cast<const float4*>(p.shift + n); sh[0] = t.x; sh[1] = t.y; sh[2] = t.z; sh[3] = t.w; }
    }
#pragma unroll
    for (int b = 0; b < TM; ++b) {
      const int ml = wave_m0 + b * 16 + (lane & 15);
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = acc[a][b][j] * sc[j] + sh[j];
      char* dst = smem + ml * CS + nl * (int)sizeof(OT);
      if (sizeof(OT) == 2) {
        *reinterpret_cast<uint2*>(dst) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
      } else {
        *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
  }
  }
  __syncthreads();
  DAS_STAMP(5);

  constexpr int VR = BN * (int)sizeof(OT) / 16;  // 16-B vectors per C row
  constexpr int RP = NT / VR;                    // rows per pass
  const int vec = tid % VR, r0 = tid / VR;
  const int n = n0 + vec * EPVO;
  float ssum[EPVO], ssq[EPVO];
#pragma unroll
  for (int j = 0; j < EPVO; ++j) {
    if constexpr (CARRY) { ssum[j] = carry[j]; ssq[j] = carry[EPVO + j]; } else { ssum[j] = 0.f; ssq[j] = 0.f; }
  }
  OT* yg = reinterpret_cast<OT*>(p.y);
  // row of y / residual / bnb tensors that conv output row m goes to
  auto orow = [&](int m) -> long long {
    if (PLAIN || !p.osub) return m;
    const int b = m / p.HoWo, rem = m - b * p.HoWo;
    const int i = rem / p.Wo, j = rem - i * p.Wo;
    return ((long long)b * p.oH + 2 * i + p.oph) * p.oW + 2 * j + p.opw;
  };
  const OT* const rg = PLAIN ? nullptr : reinterpret_cast<const OT*>(p.res);
  const bool relu_ = !PLAIN && p.relu != 0;
  const OT* const bxg = BNB ? reinterpret_cast<const OT*>(p.bnb_raw) : nullptr;   // fused BatchNorm-backward reduction (see ConvP)
  const OT* byg = reinterpret_cast<const OT*>(p.bnb_y);
#ifdef DAS_EPI_NOBITS   // dev build (make nobits): what do the dynamic bits branches cost the launches that carry no bits?
  constexpr bool BITS_ = false;
#else
  constexpr bool BITS_ = BITS;
#endif
  const unsigned char* bbits = BITS_ ? p.bnb_bits : nullptr;
  const unsigned char* rbits = (BITS_ && bxg) ? p.res_bits : nullptr;
  if (n < p.Cout) {
    constexpr int ITERS = BMT / RP;  // rows per thread, processed CH at a time
    // rows in flight per chunk: four, but two on the 256-channel tiles — with 32 vectors of per-channel constants and three operand
    // vectors per row live, four rows made conv_glds4_kernel spill 70 registers around every row (no spills with two: -0.15 ms
    // per step; the 128-channel tiles lose 2-4 % with two, make variant VAR=ch2)
    constexpr int CHMAX = DAS_EPI_CH < 4 ? DAS_EPI_CH : (BN >= 256 ? 2 : 4);
    constexpr int CH0 = (ITERS % 4 == 0 && CHMAX >= 4) ? 4 : (ITERS % 3 == 0 && CHMAX >= 3) ? 3 : ITERS % 2 == 0 ? 2 : 1;
    static_assert(BMT % RP == 0, "tile rows must be a multiple of the rows per pass");
    static_assert(ITERS % CH0 == 0, "tile rows per thread must be a multiple of the chunk");
    float bmu[EPVO], bis[EPVO], bga[EPVO], bbe[EPVO];
    if (bxg) {
#pragma unroll
      for (int j = 0; j < EPVO; ++j) {
        bmu[j] = p.bnb_mean[n + j]; bis[j] = p.bnb_invstd[n + j];
        bga[j] = (p.bnb_relu && !byg && !bbits) ? p.bnb_gamma[n + j] : 0.f;
        bbe[j] = (p.bnb_relu && !byg && !bbits) ? p.bnb_beta[n + j] : 0.f;
      }
    }
    // The rows of a chunk are requested together (their memory latency overlaps instead of adding up row by row). A launch
    // whose only epilogue operand is the pre-norm tensor of the fused BatchNorm backward (mask recomputed: no residual, no y,
    // no bits — most data gradients inside a bottleneck chain) takes ALL its rows in one chunk on the 128-channel tiles: one
    // exposed round trip per tile instead of two.
    auto rows_loop = [&](auto chc) {
      constexpr int CH = decltype(chc)::value;
#pragma unroll 1
    for (int it0 = 0; it0 < ITERS; it0 += CH) {
      // the residual rows of a chunk are requested together, ahead of the LDS reads, so that their
      // memory latency overlaps instead of adding up row by row
      uint4 rv[CH], xv[CH], yv[CH];
      unsigned rmb[CH];
      long long om[CH];
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        const int m = tile_row_m<T2D>(p, m0, r0 + (it0 + u) * RP);
        om[u] = m < p.M ? orow(m) : 0;
      }
      if (rg) {
#pragma unroll
        for (int u = 0; u < CH; ++u) {
          const int m = tile_row_m<T2D>(p, m0, r0 + (it0 + u) * RP);
          rv[u] = make_uint4(0, 0, 0, 0);
          rmb[u] = 0xff;
          if (m < p.M) {
            rv[u] = *reinterpret_cast<const uint4*>(rg + om[u] * p.rps + n);
            if (rbits) rmb[u] = rbits[(om[u] * p.rps + n) / EPVO];
          }
        }
      }
      if (bxg) {
#pragma unroll
        for (int u = 0; u < CH; ++u) {
          const int m = tile_row_m<T2D>(p, m0, r0 + (it0 + u) * RP);
          xv[u] = yv[u] = make_uint4(0, 0, 0, 0);
          if (m < p.M) {
            xv[u] = *reinterpret_cast<const uint4*>(bxg + om[u] * p.bnb_ps + n);
            if (byg) yv[u] = *reinterpret_cast<const uint4*>(byg + om[u] * p.bnb_ps + n);
            else if (bbits) yv[u].x = bbits[(om[u] * p.bnb_ps + n) / EPVO];   // (expanded at use: the load stays in flight)
          }
        }
      }
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        const int ml = r0 + (it0 + u) * RP;
        const int m = tile_row_m<T2D>(p, m0, ml);
        if (m >= p.M) {
          if (T2D) continue;   // (a border square: later rows of the chunk may exist)
          break;
        }
        const uint4 raw = *reinterpret_cast<const uint4*>(smem + ml * CS + vec * 16);
        float f[EPVO];
        Elem<OT>::unpack(raw, f);
        if (p.stats && !bxg) {
#pragma unroll
          for (int j = 0; j < EPVO; ++j) { ssum[j] += f[j]; ssq[j] += f[j] * f[j]; }
        }
        if (rg) {
          float r[EPVO];
          Elem<OT>::unpack(rv[u], r);
          if (rbits) {
#pragma unroll
            for (int j = 0; j < EPVO; ++j) r[j] = (rmb[u] >> j & 1u) ? r[j] : 0.f;
          }
#pragma unroll
          for (int j = 0; j < EPVO; ++j) f[j] += r[j];
        }
        if (relu_) {
#pragma unroll
          for (int j = 0; j < EPVO; ++j) f[j] = fmaxf(f[j], 0.f);
        }
        uint4 outv = (rg || relu_) ? Elem<OT>::pack(f) : raw;
        if (bxg) {
          float x[EPVO];
          Elem<OT>::unpack(xv[u], x);
          if (p.bnb_relu) {
            if (byg || bbits) {
              float o[EPVO];
              Elem<OT>::unpack(byg ? yv[u] : mask_vec<OT>(yv[u].x), o);
#pragma unroll
              for (int j = 0; j < EPVO; ++j) f[j] = o[j] > 0.f ? f[j] : 0.f;
            } else {
#pragma unroll
              for (int j = 0; j < EPVO; ++j) f[j] = bn_affine(x[j], bmu[j], bis[j], bga[j], bbe[j]) > 0.f ? f[j] : 0.f;
            }
          }
          outv = Elem<OT>::pack(f);
          Elem<OT>::unpack(outv, f);   // the sums see dZ as stored (what the apply pass will read)
#pragma unroll
          for (int j = 0; j < EPVO; ++j) { ssum[j] += f[j]; ssq[j] += f[j] * (x[j] - bmu[j]) * bis[j]; }
        }
        *reinterpret_cast<uint4*>(yg + om[u] * p.yps + n) = outv;
      }
    }
    };
    constexpr int CHDEEP = (BN == 128 && BMT >= 256 && !T2D && ITERS % 8 == 0 && DAS_EPI_DEEP) ? 8 : CH0;   // (conv_glds3_kernel: the others have no registers to spare)
    if (CHDEEP != CH0 && bxg && !rg && !byg && !bbits) {
      rows_loop(std::integral_constant<int, CHDEEP>{});
    } else {
      rows_loop(std::integral_constant<int, CH0>{});
    }
  }
  DAS_STAMP(6);
  if constexpr (CARRY) {
#pragma unroll
    for (int j = 0; j < EPVO; ++j) { carry[j] = ssum[j]; carry[EPVO + j] = ssq[j]; }
    return;
  }
  if (p.stats) {
    __syncthreads();
    DAS_STAMP(7);
    float* red = reinterpret_cast<float*>(smem);  // [2][RP][BN]
#pragma unroll
    for (int j = 0; j < EPVO; ++j) {
      red[r0 * BN + vec * EPVO + j] = ssum[j];
      red[(RP + r0) * BN + vec * EPVO + j] = ssq[j];
    }
    __syncthreads();
    for (int i = tid; i < 2 * BN; i += NT) {   // (one pass unless the workgroup has fewer than 2 * BN threads)
      const int which = i / BN, c = i % BN;
      if (n0 + c < p.Cout) {
        float s = 0.f;
        for (int r = 0; r < RP; ++r) s += red[(which * RP + r) * BN + c];
        const int slot = p.stat_slots > 1 ? (int)(blockIdx.x % (unsigned)p.stat_slots) : 0;
        atomicAdd(p.stats + (slot * 2 + which) * p.Cout + n0 + c, s);
      }
    }
  }
}

// The statistics a persistent kernel carried through its tiles (conv_epilogue's `carry`): the same LDS reduction over the
// threads that share a channel vector and one round of atomics, once per workgroup. Threads tid >= TL::NT (loader
// waves) only take part in the two barriers; every wave of the workgroup must call this (after a barrier that ends all
// reads of `smem`).
template <typename OT, int BN, typename TL>
__device__ __forceinline__ void conv_epilogue_flush_stats(const float* carry, const ConvP& p, char* smem, int n0) {
  constexpr int NT = TL::NT, EPVO = 16 / (int)sizeof(OT), VR = BN * (int)sizeof(OT) / 16, RP = NT / VR;
  const int tid = threadIdx.x;
  float* red = reinterpret_cast<float*>(smem);  // [2][RP][BN]
  if (tid < NT) {
    const int vec = tid % VR, r0 = tid / VR;
#pragma unroll
    for (int j = 0; j < EPVO; ++j) {
      red[r0 * BN + vec * EPVO + j] = carry[j];
      red[(RP + r0) * BN + vec * EPVO + j] = carry[EPVO + j];
    }
  }
  __syncthreads();
  if (tid < NT) {
    for (int i = tid; i < 2 * BN; i += NT) {
      const int which = i / BN, c = i % BN;
      if (n0 + c < p.Cout) {
        float s = 0.f;
        for (int r = 0; r < RP; ++r) s += red[(which * RP + r) * BN + c];
        const int slot = p.stat_slots > 1 ? (int)(blockIdx.x % (unsigned)p.stat_slots) : 0;
        atomicAdd(p.stats + (slot * 2 + which) * p.Cout + n0 + c, s);
      }
    }
  }
}

// Decode output row m into (image-plane origin offset in pixels, top-left input coords, plane size).
struct RowGeom {
  long long pix0;  // first pixel row of this image plane in x
  int hi0, wi0, H, W;
};
__device__ __forceinline__ RowGeom row_geom(const ConvP& p, int m) {
  RowGeom g;
  if (m >= p.M) {
    g.pix0 = 0; g.hi0 = -(1 << 28); g.wi0 = 0; g.H = 0; g.W = 0;
    return g;
  }
  if (p.nlev <= 1) {
    const int b = m / p.HoWo, rem = m - b * p.HoWo;
    const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
    g.pix0 = (long long)b * p.H * p.W;
    g.hi0 = ho * p.stride - p.pad;
    g.wi0 = wo * p.stride - p.pad;
    g.H = p.H; g.W = p.W;
    return g;
  }
  // (static indices only: a dynamically indexed p.lvH[l] makes the compiler keep a private copy of p in scratch)
  int Hl = p.lvH[0], Wl = p.lvW[0], start = p.lvStart[0];
#pragma unroll
  for (int i = 1; i < MAXLV; ++i)
    if (i < p.nlev && m >= p.lvStart[i]) { Hl = p.lvH[i]; Wl = p.lvW[i]; start = p.lvStart[i]; }
  const int hw = Hl * Wl;
  const int local = m - start;
  const int b = local / hw, rem = local - b * hw;
  const int ho = rem / Wl, wo = rem - ho * Wl;
  g.pix0 = (long long)start + (long long)b * hw;
  g.hi0 = ho - p.pad;
  g.wi0 = wo - p.pad;
  g.H = Hl; g.W = Wl;
  return g;
}

}  // namespace dasconv
