// Implicit-GEMM convolution for NHWC tensors on gfx950 MFMA, with a fused epilogue.
//
// GEMM view: M = B*Ho*Wo output pixels, N = Cout, K = KH*KW*Cin (tap-major, channel-minor,
// which is exactly the memory order of both the NHWC input row and the packed weights).
// Tile 128(M) x BN(N) per 256-thread workgroup (4 waves). The weight tile feeds the MFMA "A"
// operand and the pixel tile the "B" operand, so every lane ends up with 4 consecutive output
// channels of one pixel.   bf16: v_mfma_f32_16x16x32_bf16   f32: 4 x v_mfma_f32_16x16x4_f32 (exact)
//
// Two main loops share the epilogue in conv_common.h:
//  * conv_glds_kernel  (fast path, Cin % BK == 0): 128 bytes of K per row per step (64 bf16 / 32 f32),
//    both tiles DMA'd global->LDS with global_load_lds_dwordx4 (no VGPR staging), double-buffered,
//    one vmcnt(0)+barrier per step. LDS image is lane-linear, so the bank swizzle is applied to the
//    per-lane SOURCE address and to the fragment read (slot ^= (row>>1)&7); padding taps and tile
//    tails read a 16-byte zero page instead of branching.
//  * conv_reg_kernel   (generic path: the 7x7 stem with Cin=8, tiny test widths): 64 bytes of K per
//    step staged through registers, slot ^= (-(row>>2))&3.
// Both swizzles make ds_read_b128 conflict-free under gfx950's non-contiguous b128 lane groups.
// A "ragged multi-level" mode (nlev > 1) lets one launch cover all FPN levels of the shared-weight
// head convs: rows of level l start at lvStart[l] and carry their own (H, W).
#include <algorithm>
#include "prof.h"
#include <cstdlib>

#include "conv_common.h"
#include "tuning.h"
#include "workspace.h"

using namespace dasconv;

__device__ uint4 g_das_zero_page[8];  // 128 B of zeros: source of every out-of-bounds DMA

#ifdef DAS_STAMPS
extern "C" int das_dev_set_stamps(void* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_das_stamps), &buf, sizeof(buf)) == hipSuccess ? 0 : 2;
}
#endif

namespace {

// =============================================================== generic register-staged kernel
__device__ __forceinline__ int slot64(int row, int kg) { return row * 64 + ((kg ^ ((-(row >> 2)) & 3)) << 4); }

template <typename T, typename OT, int BN, bool ALIGNED, int EPI = 0>   // (EPI: conv_epilogue's variant, as for conv_glds_kernel)
__global__ __launch_bounds__(256) void conv_reg_kernel(ConvP p) {
  constexpr int EPV = Elem<T>::EPV;
  constexpr int BK = 4 * EPV;
  constexpr int TM = Tiling<BN>::TM, TN = Tiling<BN>::TN;
  constexpr int WROWS = (BN >= 64) ? BN / 64 : 1;  // weight rows staged per thread
  constexpr int A_BYTES = BM * 64, W_BYTES = BN * 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int logical = xcd_remap(blockIdx.x, p.nblocks);
  const int n0 = (logical % p.ntiles) * BN;
  const int m0 = p.m_base + (logical / p.ntiles) * BM;
  const int wave_m0 = Tiling<BN>::wave_m0(wave), wave_n0 = Tiling<BN>::wave_n0(wave);

  const int srow = tid >> 2, kg = tid & 3;
  RowGeom rg[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) rg[i] = row_geom(p, m0 + srow + 64 * i);
  const bool w_active = (BN >= 64) || (tid < 128);
  const T* xg = reinterpret_cast<const T*>(p.x);
  const T* wg = reinterpret_cast<const T*>(p.w);

  uint4 ra[2], rw[WROWS];
  int f_kh = 0, f_kw = 0, f_ci = 0;  // incremental tap state when Cin % BK == 0

  auto fetch = [&](int kt) {
    const int k0 = kt * BK + kg * EPV;
    int kh, kw, ci;
    bool kok = true;
    if (ALIGNED) {
      kh = f_kh; kw = f_kw; ci = f_ci + kg * EPV;
    } else {
      kok = k0 < p.K;
      const int tap = k0 / p.Cin;
      ci = k0 - tap * p.Cin;
      kh = tap / p.KW;
      kw = tap - kh * p.KW;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int th = rg[i].hi0 + kh, tw = rg[i].wi0 + kw;
      const int hi = th >> p.up_sh, wi = tw >> p.up_sh;
      const bool on_grid = (((th | tw) & ((1 << p.up_sh) - 1)) == 0);
      uint4 v = make_uint4(0, 0, 0, 0);
      if (kok && on_grid && th >= 0 && hi < rg[i].H && tw >= 0 && wi < rg[i].W) {
        v = *reinterpret_cast<const uint4*>(xg + (rg[i].pix0 + (long long)hi * rg[i].W + wi) * p.xps + ci);
        if (p.relu_in) v = relu_vec<T>(v);
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < WROWS; ++i) {
      const int n = n0 + srow + 64 * i;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (w_active && kok && n < p.Cout) v = *reinterpret_cast<const uint4*>(wg + (long long)n * p.K + k0);
      rw[i] = v;
    }
    if (ALIGNED) {
      f_ci += BK;
      if (f_ci >= p.Cin) {
        f_ci = 0;
        if (++f_kw == p.KW) { f_kw = 0; ++f_kh; }
      }
    }
  };
  auto stash = [&](int buf) {
    char* sA = smem + buf * A_BYTES;
    char* sW = smem + 2 * A_BYTES + buf * W_BYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<uint4*>(sA + slot64(srow + 64 * i, kg)) = ra[i];
    if (w_active) {
#pragma unroll
      for (int i = 0; i < WROWS; ++i) *reinterpret_cast<uint4*>(sW + slot64(srow + 64 * i, kg)) = rw[i];
    }
  };

  f32x4_t acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nk = (p.K + BK - 1) / BK;
  fetch(0);
  stash(0);
  __syncthreads();
  const int frow = lane & 15, fkg = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) fetch(kt + 1);
    const char* sA = smem + buf * A_BYTES;
    const char* sW = smem + 2 * A_BYTES + buf * W_BYTES;
    uint4 fb[TM], fa[TN];
#pragma unroll
    for (int t = 0; t < TM; ++t) fb[t] = *reinterpret_cast<const uint4*>(sA + slot64(wave_m0 + t * 16 + frow, fkg));
#pragma unroll
    for (int t = 0; t < TN; ++t) fa[t] = *reinterpret_cast<const uint4*>(sW + slot64(wave_n0 + t * 16 + frow, fkg));
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int b = 0; b < TM; ++b) mma<T>(fa[a], fb[b], acc[a][b]);
    if (kt + 1 < nk) stash(buf ^ 1);
    __syncthreads();
  }
  conv_epilogue<OT, BN, 128, Tiling<BN>, false, EPI == 0, 16, EPI == 0, EPI == 2>(acc, p, smem, m0, n0);
}

// =============================================================== global_load_lds fast path
__device__ __forceinline__ int slot128(int row, int kg) { return row * 128 + ((kg ^ ((row >> 1) & 7)) << 4); }

__device__ __forceinline__ void dma16(const void* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Same DMA issued from inline asm: hipcc does not track it, so it inserts no `s_waitcnt vmcnt(0)` in front
// of the next ds_read (it does for the builtin: an LDS-DMA is a pending LDS write to its alias analysis).
// The caller owns the vmcnt accounting. M0 = LDS destination of lane 0; restored afterwards.
// (make_rsrc / dma16_buf, the buffer-descriptor form, live in conv_common.h: the weight-gradient kernel uses them too)
__device__ __forceinline__ void dma16_asm(const void* src, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(src), "s"(lds_byte_addr)
               : "memory");
}

// The launch carries no residual, ReLU, sub-grid output or fused-BatchNorm-backward operands: conv_epilogue's PLAIN variants
inline bool epilogue_plain(const ConvP& p) { return !p.res && !p.relu && !p.osub && !p.bnb_raw; }

// ---- split-K (ConvP::ksplit > 1): blockIdx.y = s covers K steps [nk*s/ksplit, nk*(s+1)/ksplit) and stores its raw f32
// accumulators to slab s of the workspace (float4 = 4 consecutive channels of one pixel); splitk_finish_kernel follows.
__device__ __forceinline__ void splitk_range(const ConvP& p, int nk, int& kt0, int& kt1) {
  kt0 = 0; kt1 = nk;
  if (p.ksplit > 1) {
    kt0 = (int)((long long)nk * blockIdx.y / p.ksplit);
    kt1 = (int)((long long)nk * (blockIdx.y + 1) / p.ksplit);
  }
}
// Slab layout = register order: [split][tile][wave][a][b][lane] float4, so that every store / load instruction moves one
// contiguous KiB per wave (a pixel-major layout would scatter 64-byte pieces over 16 rows).
template <int BN, int BMT>
__device__ __forceinline__ size_t splitk_slot(const ConvP& p, int split, int tile) {
  constexpr int TM = Tiling<BN, BMT>::TM, TN = Tiling<BN, BMT>::TN, NW = Tiling<BN, BMT>::NT / 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  return ((((size_t)split * p.nblocks + tile) * NW + wave) * (TN * TM)) * 64 + lane;   // (float4 units; + (a*TM+b)*64)
}
template <int BN, int BMT, typename Acc>
__device__ __forceinline__ void splitk_store(const Acc& acc, const ConvP& p, int tile) {
  constexpr int TM = Tiling<BN, BMT>::TM, TN = Tiling<BN, BMT>::TN;
  float4* dst = reinterpret_cast<float4*>(p.ws) + splitk_slot<BN, BMT>(p, blockIdx.y, tile);
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b)
      dst[(a * TM + b) * 64] = make_float4(acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]);
}

// Split-K finished inside the tile kernel (ConvP::sk_cnt; tuning key conv.splitk_inkernel): after its slab is stored a
// workgroup publishes it and draws a ticket on the tile's arrival counter; the LAST arriver sums all ksplit slabs — its own
// too, re-read, in slab order: the same fixed summation order as splitk_finish_kernel, whoever arrives last — into `acc`
// and returns true (the caller runs the epilogue); the others return false. The hand-off is the guide's counter form
// (cdna guide section 6, guideline 16): plain slab stores -> every wave s_waitcnt vmcnt(0) -> workgroup barrier -> one lane:
// agent-scope release fence, vmcnt(0) again in asm (the compiler may drop the fence's own wait), relaxed agent-scope ticket
// -> last arriver: one agent-scope acquire fence (drops this CU's L1), barrier, plain loads. The counter is reset by the last
// arriver (it was zero when the workspace was allocated), so consecutive launches on the stream find it zero.
template <int BN, int BMT, typename Acc>
__device__ __forceinline__ bool splitk_arrive_and_reduce(Acc& acc, const ConvP& p, int tile, char* smem) {
  constexpr int TM = Tiling<BN, BMT>::TM, TN = Tiling<BN, BMT>::TN;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                   // every wave's slab stores are out; the LDS stages are dead
  volatile int* flag = reinterpret_cast<volatile int*>(smem);
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    *flag = __hip_atomic_fetch_add(p.sk_cnt + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  const int ticket = *flag;
  if (ticket != p.ksplit - 1) return false;
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __hip_atomic_store(p.sk_cnt + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  const float4* src = reinterpret_cast<const float4*>(p.ws) + splitk_slot<BN, BMT>(p, 0, tile);
  const size_t slab = (size_t)p.nblocks * (Tiling<BN, BMT>::NT / 64) * (TN * TM) * 64;
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) {
      const float4 v = src[(a * TM + b) * 64];
      acc[a][b] = f32x4_t{v.x, v.y, v.z, v.w};
    }
  for (int s = 1; s < p.ksplit; ++s) {
    src += slab;
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int b = 0; b < TM; ++b) {
        const float4 v = src[(a * TM + b) * 64];
        acc[a][b][0] += v.x; acc[a][b][1] += v.y; acc[a][b][2] += v.z; acc[a][b][3] += v.w;
      }
  }
  __syncthreads();   // (the flag word is part of the C tile the epilogue is about to write)
  return true;
}

// EPI: conv_epilogue's variant — 0 everything decided at run time, 1 no fused-BatchNorm-backward operands, 2 PLAIN (epilogue_plain)
template <typename T, typename OT, int BN, int BMT, int EPI = 0>
__global__ __launch_bounds__(2 * BMT, 2) void conv_glds_kernel(ConvP p) {   // (two workgroups per CU: at most 256 registers)
  constexpr int EPV = Elem<T>::EPV;
  constexpr int BK = 8 * EPV;  // 128 bytes of K per row
  constexpr int TM = Tiling<BN, BMT>::TM, TN = Tiling<BN, BMT>::TN;
  constexpr int A_BYTES = BMT * 128, W_BYTES = BN * 128, BUF = A_BYTES + W_BYTES;
  constexpr int A_INSTR = 4;        // 1 KiB (8 rows) per wave-instruction, 16 KiB pixel tile / 4 waves
  constexpr int W_INSTR = (BN / 8) / (BMT / 32);  // weight tile chunks per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int logical = xcd_remap(blockIdx.x, p.nblocks);
  const int n0 = (logical % p.ntiles) * BN;
  const int m0 = p.m_base + (logical / p.ntiles) * BMT;
  const int wave_m0 = Tiling<BN, BMT>::wave_m0(wave), wave_n0 = Tiling<BN, BMT>::wave_n0(wave);

  // ---- DMA coordinates: chunk c covers LDS rows c*8..c*8+7; lane -> (row, physical slot)
  const int lrow = lane >> 3, pslot = lane & 7;
  const T* xg = reinterpret_cast<const T*>(p.x);
  const T* wg = reinterpret_cast<const T*>(p.w);
  const T* zero = reinterpret_cast<const T*>(g_das_zero_page);
  RowGeom rg[A_INSTR];
  int akg[A_INSTR];
#pragma unroll
  for (int j = 0; j < A_INSTR; ++j) {
    const int row = (wave * A_INSTR + j) * 8 + lrow;
    rg[j] = row_geom(p, m0 + row);
    akg[j] = (pslot ^ ((row >> 1) & 7)) * EPV;  // logical k offset (elements) this lane fetches
  }
  const T* wrow[W_INSTR];
#pragma unroll
  for (int j = 0; j < W_INSTR; ++j) {
    const int row = (wave * W_INSTR + j) * 8 + lrow;
    const int n = n0 + row;
    wrow[j] = (n < p.Cout) ? wg + (long long)n * p.K + (pslot ^ ((row >> 1) & 7)) * EPV : nullptr;
  }

  // per-row base pointer of tap (0,0), channel group akg: the tap offset is added per step and the
  // pointer is only dereferenced when the tap is in bounds (else the zero page is selected)
  const T* abase[A_INSTR];
#pragma unroll
  for (int j = 0; j < A_INSTR; ++j)
    abase[j] = xg + (rg[j].pix0 + (long long)rg[j].hi0 * rg[j].W + rg[j].wi0) * p.xps + akg[j];

  int kt0, kt1;
  splitk_range(p, p.K / BK, kt0, kt1);   // this workgroup's slice of the K steps
  int f_kh, f_kw, f_ci;
  {
    const int e0 = kt0 * BK, tap = e0 / p.Cin;
    f_ci = e0 - tap * p.Cin;
    f_kh = tap / p.KW;
    f_kw = tap - f_kh * p.KW;
  }
  auto issue = [&](int kt, int buf) {
    char* sA = smem + buf * BUF;
    char* sW = sA + A_BYTES;
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j) {
      const int th = rg[j].hi0 + f_kh, tw = rg[j].wi0 + f_kw;
      const T* cand;
      bool ok;
      if (p.up_sh == 0) {
        ok = (unsigned)th < (unsigned)rg[j].H && (unsigned)tw < (unsigned)rg[j].W;
        cand = abase[j] + (long long)(f_kh * rg[j].W + f_kw) * p.xps + f_ci;
      } else {  // dgrad of a strided conv: only taps landing on the stride grid read dY
        const int hi = th >> p.up_sh, wi = tw >> p.up_sh;
        ok = (((th | tw) & ((1 << p.up_sh) - 1)) == 0) && th >= 0 && tw >= 0 && hi < rg[j].H && wi < rg[j].W;
        cand = xg + (rg[j].pix0 + (long long)hi * rg[j].W + wi) * p.xps + f_ci + akg[j];
      }
      dma16(ok ? cand : zero, sA + (wave * A_INSTR + j) * 1024);
    }
#pragma unroll
    for (int j = 0; j < W_INSTR; ++j) {
      const T* cand = wrow[j] + (long long)kt * BK;
      dma16(wrow[j] ? cand : zero, sW + (wave * W_INSTR + j) * 1024);
    }
    f_ci += BK;
    if (f_ci >= p.Cin) {
      f_ci = 0;
      if (++f_kw == p.KW) { f_kw = 0; ++f_kh; }
    }
  };

  f32x4_t acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  issue(kt0, 0);
  __syncthreads();  // vmcnt(0) + barrier: tile 0 landed
  const int frow = lane & 15, fkg = lane >> 4;
  for (int kt = kt0; kt < kt1; ++kt) {
    const int buf = (kt - kt0) & 1;
    if (kt + 1 < kt1) issue(kt + 1, buf ^ 1);
    const char* sA = smem + buf * BUF;
    const char* sW = sA + A_BYTES;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      uint4 fb[TM], fa[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        fb[i] = *reinterpret_cast<const uint4*>(sA + slot128(wave_m0 + i * 16 + frow, t * 4 + fkg));
      }
#pragma unroll
      for (int i = 0; i < TN; ++i) fa[i] = *reinterpret_cast<const uint4*>(sW + slot128(wave_n0 + i * 16 + frow, t * 4 + fkg));
#pragma unroll
      for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) mma<T>(fa[a], fb[b], acc[a][b]);
    }
    __syncthreads();  // prefetch landed (vmcnt 0) and every wave is done reading `buf`
  }
  if (p.ksplit > 1) { splitk_store<BN, BMT>(acc, p, logical); return; }
  conv_epilogue<OT, BN, BMT, Tiling<BN, BMT>, false, EPI == 0, 16, EPI == 0, EPI == 2>(acc, p, smem, m0, n0);
}

// =============================================================== 3x3, 64 -> 64 channels, stride 1 (stage-1 bottlenecks)
// The 3x3 convs of the 128 x 208 stage (64 channels in and out; forward and data gradient: 24 launches per train step)
// were the slowest layers of the net against their floor: conv_glds_kernel needs 112 us for 31 GFLOP / 109 MB (MFMA
// time ~30 us, HBM time 18 us). A 128 x 64 tile with K = 576 fills 24 KiB of LDS per K step — the pixel rows again for
// every one of the nine taps, and the weights again for every tile — for 0.1 us of MFMA work, and the bytes in flight per
// CU (LDS capacity) over the DMA latency bound the rate at ~6 TB/s of LDS fill. This kernel fills 10x less:
//   * PERSISTENT, one workgroup per CU: the whole weight matrix (64 x 576 bf16, rows padded to 1168 B: conflict-free
//     fragment reads) is loaded into LDS once;
//   * the tile is a 16 x 16 pixel SQUARE and its 18 x 18 x 64-channel input patch (41 KiB, zero-filled outside the image
//     by the buffer descriptor's range check) is DMA'd ONCE; the nine taps read it at shifted positions (pixel-major,
//     the 16-byte chunk index XOR-swizzled with the pixel index);
//   * two more waves only issue the DMA of the NEXT tile's patch into the other buffer while the four MFMA waves work
//     (64 pixels x 64 channels each, 288 MFMAs per tile) and run the shared epilogue (which stages the C tile in the
//     patch buffer it has just finished reading) — the MFMA waves never wait on vmcnt for a DMA, the loader never stores.
struct TilingC64 {
  static constexpr int NT = 256, TM = 4, TN = 4;
  static __device__ __forceinline__ int wave_m0(int wave) { return wave * 64; }
  static __device__ __forceinline__ int wave_n0(int) { return 0; }
};
constexpr int C64_WROW = 1168, C64_WBYTES = 64 * C64_WROW, C64_PATCH = 41 * 1024;   // (324 pixels x 128 B, in 1 KiB DMA units)
template <typename OT, bool BITS, bool BNB = true, bool PLAIN = false>
__global__ __launch_bounds__(384) void conv3x3_c64_kernel(ConvP p, int ntiles) {
  using T = bf16_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sW = smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int c = tid; c < 64 * 72; c += 384) {      // weights: 64 rows of 72 16-byte chunks
    const int row = c / 72, ch = c - row * 72;
    *reinterpret_cast<uint4*>(sW + row * C64_WROW + ch * 16) = *reinterpret_cast<const uint4*>(p.w + (size_t)row * 1152 + ch * 16);
  }
  __syncthreads();
  const int tw = (p.W + 15) >> 4, per_img = ((p.H + 15) >> 4) * tw;
  if (wave >= 4) {   // ---- loader waves (each issues half of a patch's 41 DMA instructions)
    const v4i_t xrs = make_rsrc(p.x, p.xbytes);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + C64_WBYTES;
    const unsigned xps2 = (unsigned)p.xps * 2u;
    constexpr unsigned OOB = 0xFFFFFFF0u;
    auto issue = [&](int tile, int buf) {
      const int b = tile / per_img, t = tile - b * per_img, th = t / tw;
      const int h0 = th * 16 - 1, w0 = (t - th * tw) * 16 - 1;
      const int j0 = (wave - 4) * 21, j1 = wave == 4 ? 21 : 41;
      int pp = j0 * 8 + (lane >> 3);               // patch pixel of this lane in its first instruction; + 8 per instruction
      int pr = pp / 18, pc = pp - pr * 18;
#pragma unroll 1
      for (int j = j0; j < j1; ++j) {
        const int h = h0 + pr, w = w0 + pc;
        const bool ok = pp < 324 && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
        const unsigned chunk = (unsigned)((lane & 7) ^ (pp & 7));
        const unsigned off = (unsigned)((b * p.H + h) * p.W + w) * xps2 + chunk * 16u;
        dma16_buf(ok ? off : OOB, xrs, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + buf * C64_PATCH + j * 1024)));
        pp += 8; pc += 8;
        if (pc >= 18) { pc -= 18; ++pr; }
      }
    };
    int i = 0;
    if ((int)blockIdx.x < ntiles) issue(blockIdx.x, 0);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, ++i) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this tile's patch has landed (this wave's half)
      __builtin_amdgcn_s_barrier();                        // A: ... and the other waves are done with the other buffer
      if (tile + (int)gridDim.x < ntiles) issue(tile + gridDim.x, (i + 1) & 1);
      __builtin_amdgcn_s_barrier();                        // B
      __builtin_amdgcn_s_barrier();                        // (the one inside conv_epilogue: C tile staged)
    }
    if (p.stats) {
      __builtin_amdgcn_s_barrier();                        // (the last tile's C rows have been read back)
      conv_epilogue_flush_stats<OT, 64, TilingC64>(nullptr, p, smem + C64_WBYTES, 0);
    }
    return;
  }
  // ---- MFMA waves: wave w owns rows 4w .. 4w+3 of the square, all 64 output channels
  const int q = lane & 15, kg = lane >> 4;
  float carry[16];     // this thread's statistic sums, carried through the tiles (see conv_epilogue)
#pragma unroll
  for (int j = 0; j < 16; ++j) carry[j] = 0.f;
  int i = 0;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, ++i) {
    char* buf = smem + C64_WBYTES + (i & 1) * C64_PATCH;
    f32x4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_s_barrier();                          // A: the patch is in LDS
#pragma unroll 1
    for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          uint4 fa[4], fb[4];
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const int pp = (wave * 4 + b + kh) * 18 + q + kw;
            fb[b] = *reinterpret_cast<const uint4*>(buf + pp * 128 + (((half * 4 + kg) ^ (pp & 7)) << 4));
          }
#pragma unroll
          for (int a = 0; a < 4; ++a)
            fa[a] = *reinterpret_cast<const uint4*>(sW + (a * 16 + q) * C64_WROW + ((kh * 3 + kw) * 64 + half * 32 + kg * 8) * 2);
#pragma unroll
          for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) mma<T>(fa[a], fb[b], acc[a][b]);
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                          // B: every wave is done reading the patch
    if (p.stats) {   // (wave-uniform)
      conv_epilogue<OT, 64, 256, TilingC64, true, BITS, 16, BNB, PLAIN>(acc, p, buf, tile, 0, carry);
    } else {
      conv_epilogue<OT, 64, 256, TilingC64, true, BITS, 16, BNB, PLAIN>(acc, p, buf, tile, 0);
    }
  }
  if (p.stats) {
    __builtin_amdgcn_s_barrier();
    conv_epilogue_flush_stats<OT, 64, TilingC64>(carry, p, smem + C64_WBYTES, 0);
  }
}

// Second half of a split-K convolution: same grid and tile map as the tile-kernel launch it follows; every lane sums
// its accumulator positions over the slabs (fixed order: deterministic) and the shared epilogue does the rest
// (scale / shift, C tile through LDS, statistics, residual, ReLU, fused BatchNorm-backward sums, 16-byte stores).
template <typename OT, int BN, int BMT, int EPI = 0>   // (EPI as for conv_glds_kernel)
__global__ __launch_bounds__((BN == 256 ? 512 : 2 * BMT)) void splitk_finish_kernel(ConvP p) {
  constexpr int TM = Tiling<BN, BMT>::TM, TN = Tiling<BN, BMT>::TN;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int logical = xcd_remap(blockIdx.x, p.nblocks);
  const int n0 = (logical % p.ntiles) * BN;
  const int m0 = p.m_base + (logical / p.ntiles) * BMT;
  constexpr int NW = Tiling<BN, BMT>::NT / 64;
  const float4* src = reinterpret_cast<const float4*>(p.ws) + splitk_slot<BN, BMT>(p, 0, logical);
  const size_t slab = (size_t)p.nblocks * NW * (TN * TM) * 64;
  f32x4_t acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) {
      const float4 v = src[(a * TM + b) * 64];
      acc[a][b] = f32x4_t{v.x, v.y, v.z, v.w};
    }
  for (int s = 1; s < p.ksplit; ++s) {   // (one slab at a time: all of its loads are in flight together)
    src += slab;
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int b = 0; b < TM; ++b) {
        const float4 v = src[(a * TM + b) * 64];
        acc[a][b][0] += v.x; acc[a][b][1] += v.y; acc[a][b][2] += v.z; acc[a][b][3] += v.w;
      }
  }
  conv_epilogue<OT, BN, BMT, Tiling<BN, BMT>, false, EPI == 0, 16, EPI == 0, EPI == 2>(acc, p, smem, m0, n0);
}

// =============================================================== 3-stage pipeline, 256 x 128 tile, 8 waves
// One workgroup per CU (144 KiB of LDS): occupancy cannot hide DMA latency any more, so the loads run two
// K-steps ahead of the MFMAs: per step  s_waitcnt vmcnt(one tile's DMAs still in flight) -> s_barrier ->
// issue tile k+2 into the buffer freed by step k-1 -> MFMAs on tile k. Counted vmcnt + raw s_barrier
// (a __syncthreads() would drain the DMA queue, cdna guide section 5).
// BITS = false: the epilogue without the mask-bits operands (conv_common.h), for the launches that carry none — the epilogue
// of a one-workgroup-per-CU tile kernel overlaps with nothing, and its dynamic bits branches cost the 256 x 128 kernel 4-8 %
// of a launch (make nobits / tools/dev/train_shapes.py: conv_glds3<pp> 6.2 -> 5.96 ms, conv_glds3 2.8 -> 2.6 ms per step)
// SKONLY: a split-K launch whose sum is finished by splitk_finish_kernel — the epilogue is not compiled in (the partial sums leave in register order)
template <typename T, typename OT, bool PP = false, bool BITS = true, bool BNB = true, bool PLAIN = false, bool SKONLY = false>
__global__ __launch_bounds__(512) void conv_glds3_kernel(ConvP p) {
  constexpr int BN = 128, BMT = 256, NBUF = 3;
  constexpr int EPV = Elem<T>::EPV;
  constexpr int BK = 8 * EPV;
  constexpr int TM = 4, TN = 4;
  constexpr int A_BYTES = BMT * 128, W_BYTES = BN * 128, BUF = A_BYTES + W_BYTES;
  constexpr int A_INSTR = 4, W_INSTR = 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  DAS_STAMP(0);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int logical = xcd_remap(blockIdx.x, p.nblocks);
  const int n0 = (logical % p.ntiles) * BN;
  const int m0 = p.m_base + (logical / p.ntiles) * (p.mstep ? p.mstep : BMT);
  if (p.mstep) p.M = min(p.M, m0 + p.mstep);   // balanced tiles: this tile's rows from mstep on do not exist
  const int wave_m0 = (wave & 3) * 64, wave_n0 = (wave >> 2) * 64;

  // Operands are fetched with `buffer_load_dwordx4 ... offen lds`: 32-bit byte offsets against a buffer
  // descriptor whose hardware range check returns zeros — padding taps, rows past M and weight rows past
  // Cout need no zero page and no 64-bit address arithmetic (probe: tools/dev/probe/buf_lds_probe.hip).
  // Per K-step and DMA instruction the address work is one add and one select; the tap walk is incremental.
  constexpr unsigned E = sizeof(T);
  constexpr unsigned OOB = 0xFFFFFFF0u;
  const int lrow = lane >> 3, pslot = lane & 7;
  const v4i_t xrs = make_rsrc(p.x, p.xbytes), wrs = make_rsrc(p.w, (unsigned)((long long)p.Cout * p.K * E));
  unsigned acur[A_INSTR];     // byte offset of this lane's 16-byte piece at the current tap (valid or not)
  unsigned arowstep[A_INSTR]; // bytes per input row of the lane's level
  int ahi[A_INSTR], awi[A_INSTR], aH[A_INSTR], aW[A_INSTR];
  bool aok[A_INSTR];
#pragma unroll
  for (int j = 0; j < A_INSTR; ++j) {
    const int row = (wave * A_INSTR + j) * 8 + lrow;
    const RowGeom g = row_geom(p, m0 + row);
    const int akg = (pslot ^ ((row >> 1) & 7)) * EPV;
    acur[j] = (unsigned)(((g.pix0 + (long long)g.hi0 * g.W + g.wi0) * p.xps + akg) * (long long)E);
    arowstep[j] = (unsigned)g.W * (unsigned)p.xps * E;
    ahi[j] = g.hi0; awi[j] = g.wi0; aH[j] = g.H; aW[j] = g.W;
    aok[j] = (unsigned)g.hi0 < (unsigned)g.H && (unsigned)g.wi0 < (unsigned)g.W;
  }
  unsigned wcur[W_INSTR];
#pragma unroll
  for (int j = 0; j < W_INSTR; ++j) {
    const int row = (wave * W_INSTR + j) * 8 + lrow;
    wcur[j] = (unsigned)(((long long)(n0 + row) * p.K + (pslot ^ ((row >> 1) & 7)) * EPV) * E);  // rows >= Cout: out of range
  }

  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  int f_kh = 0, f_kw = 0, f_ci = 0;
  const unsigned dA = BK * E, dB = (unsigned)(p.xps - p.Cin + BK) * E;
  const unsigned dC = 0u - (unsigned)(p.KW - 1) * (unsigned)p.xps * E - (unsigned)(p.Cin - BK) * E;  // + rowstep
  auto issue = [&](int kt, int buf) {
    const unsigned sA = lds0 + buf * BUF, sW = sA + A_BYTES;
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j) dma16_buf(aok[j] ? acur[j] : OOB, xrs, sA + (wave * A_INSTR + j) * 1024);
#pragma unroll
    for (int j = 0; j < W_INSTR; ++j) {
      dma16_buf(wcur[j], wrs, sW + (wave * W_INSTR + j) * 1024);
      wcur[j] += BK * E;
    }
    f_ci += BK;
    if (f_ci < p.Cin) {
#pragma unroll
      for (int j = 0; j < A_INSTR; ++j) acur[j] += dA;
    } else {  // next tap (wave-uniform branch): move the pointer, re-evaluate the bounds
      f_ci = 0;
      const bool wrap = ++f_kw == p.KW;
      if (wrap) { f_kw = 0; ++f_kh; }
#pragma unroll
      for (int j = 0; j < A_INSTR; ++j) {
        acur[j] += wrap ? arowstep[j] + dC : dB;
        aok[j] = (unsigned)(ahi[j] + f_kh) < (unsigned)aH[j] && (unsigned)(awi[j] + f_kw) < (unsigned)aW[j];
      }
    }
  };

  f32x4_t acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  int kt0, kt1;
  splitk_range(p, p.K / BK, kt0, kt1);
  const int nk = kt1 - kt0;
  if (kt0) {   // split-K: move the tap walk to this workgroup's first step
    const int e0 = kt0 * BK, tap = e0 / p.Cin;
    f_ci = e0 - tap * p.Cin; f_kh = tap / p.KW; f_kw = tap - f_kh * p.KW;
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j) {
      acur[j] += (unsigned)((((long long)f_kh * aW[j] + f_kw) * p.xps + f_ci) * (long long)E);
      aok[j] = (unsigned)(ahi[j] + f_kh) < (unsigned)aH[j] && (unsigned)(awi[j] + f_kw) < (unsigned)aW[j];
    }
#pragma unroll
    for (int j = 0; j < W_INSTR; ++j) wcur[j] += (unsigned)e0 * E;
  }
  DAS_STAMP(1);
  issue(0, 0);
  if (nk > 1) issue(1, 1);
  const int frow = lane & 15, fkg = lane >> 4;
  int buf = 0, nbuf = 2;  // buffer of tile kt, buffer tile kt+2 goes to
  if constexpr (PP) {
    // Ping-pong schedule (as conv_glds4_kernel<PP>): every K-step is two barrier intervals, R (issue tile kt+2, read
    // the 16 fragments of tile kt) and M (the 32 MFMAs); waves 4-7 run one interval behind waves 0-3 and wave w / w+4
    // share a SIMD, so one of them holds the matrix pipe while the other reads LDS and issues DMA. Tile kt+1 must
    // have landed before the barrier that opens the leading group's R of step kt+1.
    const int grp = wave >> 2;
    auto wait_next = [&](int kt) {   // this wave's DMAs of tile kt+1 (tile kt+2 may still fly: 6 DMAs)
      if (kt + 2 < nk) {
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    };
    if (nk > 1) {
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                 // tile 0 landed
    DAS_STAMP(2);
    if (grp == 1) __builtin_amdgcn_s_barrier();   // trailing group: one interval behind
    for (int kt = 0; kt < nk; ++kt) {
      // ---- R
      if (kt + 2 < nk) issue(kt + 2, nbuf);
      const char* sA = smem + buf * BUF;
      const char* sW = sA + A_BYTES;
      uint4 fb[2][TM], fa[2][TN];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int i = 0; i < TN; ++i) fa[t][i] = *reinterpret_cast<const uint4*>(sW + slot128(wave_n0 + i * 16 + frow, t * 4 + fkg));
#pragma unroll
        for (int i = 0; i < TM; ++i) fb[t][i] = *reinterpret_cast<const uint4*>(sA + slot128(wave_m0 + i * 16 + frow, t * 4 + fkg));
      }
      if (grp == 1) wait_next(kt);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      // ---- M
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
          for (int b = 0; b < TM; ++b) mma<T>(fa[t][a], fb[t][b], acc[a][b]);
      __builtin_amdgcn_s_setprio(0);
      if (grp == 0) wait_next(kt);
      __builtin_amdgcn_s_barrier();
      buf = buf == NBUF - 1 ? 0 : buf + 1;
      nbuf = nbuf == NBUF - 1 ? 0 : nbuf + 1;
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();   // leading group: match the trailing group's extra interval
  } else
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) {
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // tile kt landed; tile kt+1 (6 DMAs per wave) may fly
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (kt + 2 < nk) issue(kt + 2, nbuf);
    const char* sA = smem + buf * BUF;
    const char* sW = sA + A_BYTES;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      uint4 fb[TM], fa[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fb[i] = *reinterpret_cast<const uint4*>(sA + slot128(wave_m0 + i * 16 + frow, t * 4 + fkg));
#pragma unroll
      for (int i = 0; i < TN; ++i) fa[i] = *reinterpret_cast<const uint4*>(sW + slot128(wave_n0 + i * 16 + frow, t * 4 + fkg));
#pragma unroll
      for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) mma<T>(fa[a], fb[b], acc[a][b]);
    }
    buf = buf == NBUF - 1 ? 0 : buf + 1;
    nbuf = nbuf == NBUF - 1 ? 0 : nbuf + 1;
  }
  DAS_STAMP(3);
  if constexpr (SKONLY) {
    splitk_store<BN, BMT>(acc, p, logical);
    DAS_STAMP(4);
    return;
  } else {
  if (p.ksplit > 1) {
    splitk_store<BN, BMT>(acc, p, logical);
    DAS_STAMP(4);
    if (!p.sk_cnt) return;                                                    // (splitk_finish_kernel follows)
    if (!splitk_arrive_and_reduce<BN, BMT>(acc, p, logical, smem)) return;    // not the last split of this tile
  } else {
    __syncthreads();  // all LDS reads done before the C tile reuses the buffers
  }
  conv_epilogue<OT, BN, BMT, Tiling<BN, BMT>, false, BITS, 16, BNB, PLAIN>(acc, p, smem, m0, n0);
  DAS_STAMP(4);
  }
}

// =============================================================== 4-stage pipeline, 256 x 256 tile, 8 waves
// At the 256 x 128 tile the MFMA-bound layers sit on the L2 -> LDS path: one K-step of 64 moves 48 KiB per
// workgroup for 256*128*64 MACs, ~29 TB/s chip-wide at the full MFMA rate. The square tile moves 32 KiB per
// 256*256*32 MACs (1.5x fewer bytes per MAC). K-steps of 32 (64-byte rows) keep a stage at 32 KiB, so four
// stages fit (128 KiB) and the DMAs run three steps ahead; each wave owns 128 pixels x 64 channels (8 x 4
// MFMA tiles, 128 accumulator registers). Wait / barrier scheme as in conv_glds3_kernel with 4 DMAs per wave
// and stage. Needs Cin % 32 == 0; used for Cout >= 256.
// BMT = 288 (nine 16-row MFMA tiles per wave instead of eight): for launches whose tile count at 256 rows spills a few
// tiles into another round of the one-workgroup-per-CU grid (277 tiles on 256 CUs: the B = 8 head convs). The pixel
// tile then needs 18 DMA instructions per stage: every wave issues a third one, waves 0-1 for rows 256..287, the others
// for an out-of-range offset into a scratch KiB (zero fill, no memory traffic), so that the vmcnt arithmetic stays
// wave-uniform.
// MF = 32 (bf16, ping-pong, 256 rows): the wave's 128 pixels x 64 channels as 4 x 2 accumulator tiles of 32 x 32 fed by
// v_mfma_f32_32x32x16_bf16 — the same twelve 16-byte fragment reads per K step (a lane reads row l & 31, k-group l >> 5 of
// each 16-deep half: conflict-free under the same row swizzle), 16 MFMAs of 32 cycles instead of 32 of 16; the matrix pipe's
// ceiling is 15 % higher for the square shape (2382 vs 2075 TF, cdna guide section 3). Tuning key conv.glds4_mfma32.
template <typename T, typename OT, bool PP, int BMT = 256, int MF = 16, bool BNB = true, bool PLAIN = false>
__global__ __launch_bounds__(512) void conv_glds4_kernel(ConvP p) {
  constexpr int BN = 256, NBUF = 4;
  static_assert(BMT == 256 || BMT == 288, "pixel tile: 256 or 288 rows");
  static_assert(MF == 16 || (MF == 32 && PP && BMT == 256 && sizeof(T) == 2), "32 x 32 MFMA tiles: bf16 ping-pong, 256 rows");
  constexpr int EPV = Elem<T>::EPV;
  constexpr int BK = 4 * EPV;  // 64-byte rows
  constexpr int TM = Tiling<BN, BMT>::TM, TN = Tiling<BN, BMT>::TN;
  constexpr int A_BYTES = BMT * 64, W_BYTES = BN * 64, BUF = A_BYTES + W_BYTES;
  constexpr int A_INSTR = BMT == 256 ? 2 : 3, W_INSTR = 2;  // 16 rows x 64 B per wave-instruction
  constexpr int DPS = A_INSTR + W_INSTR;                    // DMAs per wave and stage
  constexpr int SCRATCH = NBUF * BUF;                       // (BMT = 288) landing KiB of the padding DMAs
  constexpr unsigned E = sizeof(T);
  constexpr unsigned OOB = 0xFFFFFFF0u;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  DAS_STAMP(0);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int logical = xcd_remap(blockIdx.x, p.nblocks);
  const int n0 = (logical % p.ntiles) * BN;
  const int m0 = p.m_base + (logical / p.ntiles) * (p.mstep ? p.mstep : BMT);
  if (p.mstep) p.M = min(p.M, m0 + p.mstep);   // balanced tiles (see conv_glds3_kernel)
  const int wave_m0 = Tiling<BN, BMT>::wave_m0(wave), wave_n0 = Tiling<BN, BMT>::wave_n0(wave);

  const int lrow = lane >> 2, pslot = lane & 3;
  const v4i_t xrs = make_rsrc(p.x, p.xbytes), wrs = make_rsrc(p.w, (unsigned)((long long)p.Cout * p.K * E));
  unsigned acur[A_INSTR], arowstep[A_INSTR];
  int ahi[A_INSTR], awi[A_INSTR], aH[A_INSTR], aW[A_INSTR];
  bool aok[A_INSTR];
  // instruction j of wave w stages LDS rows arow(j) .. + 15; the third one (BMT = 288) is real on waves 0-1 only
  auto arow = [&](int j) { return j < 2 ? (wave * 2 + j) * 16 : 256 + wave * 16; };
  const bool a3 = wave < 2;
#pragma unroll
  for (int j = 0; j < A_INSTR; ++j) {
    const int row = arow(j) + lrow;
    const RowGeom g = row_geom(p, (j < 2 || a3) ? m0 + row : p.M);
    const int akg = (pslot ^ ((-(row >> 2)) & 3)) * EPV;   // logical k-group stored at this physical slot
    acur[j] = (unsigned)(((g.pix0 + (long long)g.hi0 * g.W + g.wi0) * p.xps + akg) * (long long)E);
    arowstep[j] = (unsigned)g.W * (unsigned)p.xps * E;
    ahi[j] = g.hi0; awi[j] = g.wi0; aH[j] = g.H; aW[j] = g.W;
    aok[j] = (unsigned)g.hi0 < (unsigned)g.H && (unsigned)g.wi0 < (unsigned)g.W;
  }
  unsigned wcur[W_INSTR];
#pragma unroll
  for (int j = 0; j < W_INSTR; ++j) {
    const int row = (wave * W_INSTR + j) * 16 + lrow;
    wcur[j] = (unsigned)(((long long)(n0 + row) * p.K + (pslot ^ ((-(row >> 2)) & 3)) * EPV) * E);  // rows >= Cout: out of range
  }

  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  int f_kh = 0, f_kw = 0, f_ci = 0;
  const unsigned dA = BK * E, dB = (unsigned)(p.xps - p.Cin + BK) * E;
  const unsigned dC = 0u - (unsigned)(p.KW - 1) * (unsigned)p.xps * E - (unsigned)(p.Cin - BK) * E;  // + rowstep
  auto issue = [&](int buf) {
    const unsigned sA = lds0 + buf * BUF, sW = sA + A_BYTES;
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j)
      dma16_buf(aok[j] ? acur[j] : OOB, xrs, (j < 2 || a3) ? sA + (unsigned)arow(j) * 64u : lds0 + SCRATCH);
#pragma unroll
    for (int j = 0; j < W_INSTR; ++j) {
      dma16_buf(wcur[j], wrs, sW + (wave * W_INSTR + j) * 1024);
      wcur[j] += BK * E;
    }
    f_ci += BK;
    if (f_ci < p.Cin) {
#pragma unroll
      for (int j = 0; j < A_INSTR; ++j) acur[j] += dA;
    } else {  // next tap (wave-uniform branch)
      f_ci = 0;
      const bool wrap = ++f_kw == p.KW;
      if (wrap) { f_kw = 0; ++f_kh; }
#pragma unroll
      for (int j = 0; j < A_INSTR; ++j) {
        acur[j] += wrap ? arowstep[j] + dC : dB;
        aok[j] = (unsigned)(ahi[j] + f_kh) < (unsigned)aH[j] && (unsigned)(awi[j] + f_kw) < (unsigned)aW[j];
      }
    }
  };

  // accumulators: TN x TM tiles of 16 x 16 (4 floats per lane), or (MF = 32) TN/2 x TM/2 tiles of 32 x 32 (16 per lane)
  using AccT = typename std::conditional<MF == 32, f32x16_t, f32x4_t>::type;
  constexpr int AN = MF == 32 ? TN / 2 : TN, AM = MF == 32 ? TM / 2 : TM;
  AccT acc[AN][AM];
#pragma unroll
  for (int a = 0; a < AN; ++a)
#pragma unroll
    for (int b = 0; b < AM; ++b)
#pragma unroll
      for (int j = 0; j < (MF == 32 ? 16 : 4); ++j) acc[a][b][j] = 0.f;

  int kt0, kt1;
  splitk_range(p, p.K / BK, kt0, kt1);
  const int nk = kt1 - kt0;
  if (kt0) {   // split-K: move the tap walk to this workgroup's first step
    const int e0 = kt0 * BK, tap = e0 / p.Cin;
    f_ci = e0 - tap * p.Cin; f_kh = tap / p.KW; f_kw = tap - f_kh * p.KW;
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j) {
      acur[j] += (unsigned)((((long long)f_kh * aW[j] + f_kw) * p.xps + f_ci) * (long long)E);
      aok[j] = (unsigned)(ahi[j] + f_kh) < (unsigned)aH[j] && (unsigned)(awi[j] + f_kw) < (unsigned)aW[j];
    }
#pragma unroll
    for (int j = 0; j < W_INSTR; ++j) wcur[j] += (unsigned)e0 * E;
  }
  DAS_STAMP(1);
  issue(0);
  if (nk > 1) issue(1);
  if (nk > 2) issue(2);
  const int frow = lane & 15, fkg = lane >> 4;
  int buf = 0, nbuf = 3;  // buffer of tile kt, buffer tile kt+3 goes to
  if constexpr (PP && MF == 32) {
    // the ping-pong schedule below, on 32 x 32 tiles
    const int grp = wave >> 2;
    const int frow32 = lane & 31, fk2 = lane >> 5;
    auto wait_next = [&](int kt) {
      if (kt + 3 < nk) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DPS) : "memory");
      } else if (kt + 2 < nk) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPS) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    };
    if (nk > 2) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DPS) : "memory");
    } else if (nk > 1) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPS) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                 // tile 0 landed
    DAS_STAMP(2);
    if (grp == 1) __builtin_amdgcn_s_barrier();   // trailing group: one interval behind
    for (int kt = 0; kt < nk; ++kt) {
      // ---- R
      if (kt + 3 < nk) issue(nbuf);
      const char* sA = smem + buf * BUF;
      const char* sW = sA + A_BYTES;
      uint4 fb[2][AM], fa[2][AN];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i = 0; i < AN; ++i) fa[h][i] = *reinterpret_cast<const uint4*>(sW + slot64(wave_n0 + i * 32 + frow32, h * 2 + fk2));
#pragma unroll
        for (int i = 0; i < AM; ++i) fb[h][i] = *reinterpret_cast<const uint4*>(sA + slot64(wave_m0 + i * 32 + frow32, h * 2 + fk2));
      }
      if (grp == 1) wait_next(kt);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      // ---- M
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int b = 0; b < AM; ++b)
#pragma unroll
          for (int a = 0; a < AN; ++a)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa[h][a]),
                                                                 __builtin_bit_cast(bf16x8_t, fb[h][b]), acc[a][b], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      if (grp == 0) wait_next(kt);
      __builtin_amdgcn_s_barrier();
      buf = buf == NBUF - 1 ? 0 : buf + 1;
      nbuf = nbuf == NBUF - 1 ? 0 : nbuf + 1;
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();   // leading group: match the trailing group's extra interval
  } else if constexpr (PP) {
    // Ping-pong: every K-step is two barrier intervals, R (issue tile kt+3, read the fragments of tile kt into
    // registers) and M (the 32 MFMAs). Waves 4-7 run one interval behind waves 0-3, and wave w / w+4 share a SIMD:
    // while one of them holds the matrix pipe the other one does its LDS reads and DMA issue. Tile kt+1 must have
    // landed (every wave's DMAs waited for) before the barrier that opens the leading group's R of step kt+1: the
    // leading group waits at the end of its M, the trailing group at the end of its R (the same interval).
    const int grp = wave >> 2;
    auto wait_next = [&](int kt) {   // this wave's DMAs of tile kt+1 (tiles kt+2, kt+3 may still fly: 4 DMAs each)
      if (kt + 3 < nk) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DPS) : "memory");
      } else if (kt + 2 < nk) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPS) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    };
    if (nk > 2) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DPS) : "memory");
    } else if (nk > 1) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPS) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                 // tile 0 landed
    DAS_STAMP(2);
    if (grp == 1) __builtin_amdgcn_s_barrier();   // trailing group: one interval behind
    for (int kt = 0; kt < nk; ++kt) {
      // ---- R
      if (kt + 3 < nk) issue(nbuf);
      const char* sA = smem + buf * BUF;
      const char* sW = sA + A_BYTES;
      uint4 fb[TM], fa[TN];
#pragma unroll
      for (int i = 0; i < TN; ++i) fa[i] = *reinterpret_cast<const uint4*>(sW + slot64(wave_n0 + i * 16 + frow, fkg));
#pragma unroll
      for (int i = 0; i < TM; ++i) fb[i] = *reinterpret_cast<const uint4*>(sA + slot64(wave_m0 + i * 16 + frow, fkg));
      if (grp == 1) wait_next(kt);
      // retire the fragment reads before the barrier: the other group's next R re-stages buffers one interval later
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      // ---- M
      __builtin_amdgcn_s_setprio(1);
      if constexpr (MF == 16) {
#pragma unroll
      for (int b = 0; b < TM; ++b)
#pragma unroll
        for (int a = 0; a < TN; ++a) mma<T>(fa[a], fb[b], acc[a][b]);
      }
      __builtin_amdgcn_s_setprio(0);
      if (grp == 0) wait_next(kt);
      __builtin_amdgcn_s_barrier();
      buf = buf == NBUF - 1 ? 0 : buf + 1;
      nbuf = nbuf == NBUF - 1 ? 0 : nbuf + 1;
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();   // leading group: match the trailing group's extra interval
  } else
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt landed; the (up to two) younger tiles, 4 DMAs per wave each, may still fly
    if (kt + 2 < nk) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DPS) : "memory");
    } else if (kt + 1 < nk) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPS) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (kt + 3 < nk) issue(nbuf);
    const char* sA = smem + buf * BUF;
    const char* sW = sA + A_BYTES;
    uint4 fb[TM], fa[TN];
#pragma unroll
    for (int i = 0; i < TN; ++i) fa[i] = *reinterpret_cast<const uint4*>(sW + slot64(wave_n0 + i * 16 + frow, fkg));
#pragma unroll
    for (int i = 0; i < TM; ++i) fb[i] = *reinterpret_cast<const uint4*>(sA + slot64(wave_m0 + i * 16 + frow, fkg));
    if constexpr (MF == 16) {
#pragma unroll
    for (int b = 0; b < TM; ++b)
#pragma unroll
      for (int a = 0; a < TN; ++a) mma<T>(fa[a], fb[b], acc[a][b]);
    }
    buf = buf == NBUF - 1 ? 0 : buf + 1;
    nbuf = nbuf == NBUF - 1 ? 0 : nbuf + 1;
  }
  DAS_STAMP(3);
  if constexpr (MF == 16) {
    if (p.ksplit > 1) { splitk_store<BN, BMT>(acc, p, logical); DAS_STAMP(4); return; }
  }
  __syncthreads();  // all LDS reads done before the C tile reuses the buffers
  conv_epilogue<OT, BN, BMT, Tiling<BN, BMT>, false, BNB, MF, BNB, PLAIN>(acc, p, smem, m0, n0);
  DAS_STAMP(4);
}

// =============================================================== launch
using dastune::device_cus;
using dastune::usable_cus;
// Split-K factor for a tile-kernel launch of `nblocks` workgroups with `nk` K steps (steps of 128 bytes: the 64-byte
// steps of conv_glds4_kernel are counted in pairs): small-M, long-K layers (the 16x26 / 32x52 stages) have too few
// tiles to fill the chip, and every K step of a lone workgroup exposes its DMA latency. The K steps are spread over
// blockIdx.y until there are about conv.splitk_target workgroups per resident slot (`per_cu` of them per CU), each
// keeping at least conv.splitk_minsteps steps; `bit` selects the kernel in conv.splitk_kernels.
inline int pick_ksplit(long long nblocks, int nk, int per_cu, int bit, long long out_elems, bool tail = false) {
  const long long mask = dastune::get(dastune::CONV_SPLITK_KERNELS), target = dastune::get(dastune::CONV_SPLITK_TARGET) * per_cu;
  // (the tail launch of a tail-split conv is pure overhead on top of the full rounds: split it as finely as is sane)
  const long long minsteps = tail ? 4 : std::max<long long>(1, dastune::get(dastune::CONV_SPLITK_MINSTEPS));
  if (!(mask & bit) || target <= 0 || nblocks * 2 > target) return 1;
  long long ks = std::min<long long>({target / nblocks, nk / minsteps, 16});
  while (ks > 1 && ks * out_elems * 4 > ((long long)128 << 20)) --ks;   // slabs: at most 128 MiB
  return ks < 2 ? 1 : (int)ks;
}
// Launch a tile kernel split over blockIdx.y, then the finishing kernel on the same tile map.
// INK: the kernel can finish the sum itself (ConvP::sk_cnt, splitk_arrive_and_reduce) — used when tuning key
// conv.splitk_inkernel is 1: one launch instead of two.
template <typename OT, int BN, int BMT, bool INK = false, typename Kern>
int launch_splitk(Kern kern, ConvP& p, int ks, size_t sm, hipStream_t s) {
  p.ws = dasws::get(dasws::CONV_SPLITK, s, (size_t)ks * p.nblocks * BN * BMT * sizeof(float), (size_t)32 << 20);
  if (!p.ws) return DAS_ERR_LAUNCH;
  p.ksplit = ks;
  if (INK && dastune::get(dastune::CONV_SPLITK_INKERNEL) == 1) {
    p.sk_cnt = reinterpret_cast<int*>(dasws::get(dasws::SPLITK_CNT, s, (size_t)p.nblocks * sizeof(int), (size_t)64 << 10));
    if (!p.sk_cnt) return DAS_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(p.nblocks, ks), dim3(Tiling<BN, BMT>::NT), sm, s, p);
    DAS_CHECK_LAUNCH();
    return DAS_OK;
  }
  static bool fin_attr = false;   // (one per instantiation)
  const size_t sm_fin = epilogue_smem_bytes<OT, BN, BMT>();
  if (!fin_attr) {
    (void)hipFuncSetAttribute((const void*)splitk_finish_kernel<OT, BN, BMT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm_fin);
    (void)hipFuncSetAttribute((const void*)splitk_finish_kernel<OT, BN, BMT, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm_fin);
    (void)hipFuncSetAttribute((const void*)splitk_finish_kernel<OT, BN, BMT, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm_fin);
    fin_attr = true;
  }
  constexpr int NT = Tiling<BN, BMT>::NT;
  hipLaunchKernelGGL(kern, dim3(p.nblocks, ks), dim3(NT), sm, s, p);
  DAS_CHECK_LAUNCH();
  if (p.bnb_raw) hipLaunchKernelGGL((splitk_finish_kernel<OT, BN, BMT>), dim3(p.nblocks), dim3(NT), sm_fin, s, p);
  else if (epilogue_plain(p)) hipLaunchKernelGGL((splitk_finish_kernel<OT, BN, BMT, 2>), dim3(p.nblocks), dim3(NT), sm_fin, s, p);
  else hipLaunchKernelGGL((splitk_finish_kernel<OT, BN, BMT, 1>), dim3(p.nblocks), dim3(NT), sm_fin, s, p);
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}

// The 256-row tile kernels run one workgroup per CU, so a launch is a sequence of rounds of `cus` tiles and a last round
// that is, e.g., 8 % full (277 tiles on 256 CUs: the B = 8 head convs of the inference workload) costs a whole round.
// If the last round would be under half full, its pixel rows are left to a second launch on 128-row tiles
// (4 x the workgroups, two per CU): returns the number of 256-row M tiles the main launch keeps (== mtiles: no split).
inline long long tail_split_mtiles(long long mtiles, int ntiles, bool allowed) {
  const long long cus = device_cus(), nb = mtiles * ntiles, rem = nb % cus;
  if (!allowed || dastune::get(dastune::CONV_TAIL_SPLIT) != 1 || nb <= cus || rem == 0 || rem * 2 > cus) return mtiles;
  const long long keep = (nb - rem) / ntiles;
  return keep >= 1 && keep < mtiles ? keep : mtiles;
}

// Balanced pixel tiles. The same launches, when their last round is more than half full (no tail split): 416 tiles on
// 256 CUs are two rounds for 1.6 rounds of work. The rows are spread over ALL the tiles of the full rounds instead — 512
// tiles of 208 rows: every tile still runs the 256-row K loop (rows from mstep on are rows past M to the loaders and the
// epilogue: zero fill without memory traffic, nothing stored), but moves 19 % fewer bytes, and these launches are bound
// by bytes and latency per CU, not by the matrix pipe. Returns the tile step (a multiple of 16) or 0 (no change) and
// the new number of M tiles.
thread_local int t_last_tile_rows = 0;
inline int balanced_mstep(int rows, int bmt, int ntiles, bool one_by_one, int* mtiles) {
  const long long mode = dastune::get(dastune::CONV_BALANCE_ROWS);
  if (mode <= 0 || (mode == 1 && !one_by_one)) return 0;
  const long long cus = device_cus(), nb = (long long)*mtiles * ntiles;
  const long long rounds = (nb + cus - 1) / cus, cap = rounds * cus / ntiles;   // M tiles the full rounds hold
  if (nb % cus == 0 || cap <= *mtiles || rounds > 8) return 0;
  const int step = (int)(((rows + cap - 1) / cap + 15) / 16 * 16);
  if (step >= bmt || step * 20 > bmt * 19 || step < bmt / 2) return 0;   // (under 5 % to gain; under half a tile: the 128-row kernel's case)
  *mtiles = (rows + step - 1) / step;
  return step;
}

template <typename T, typename OT, int BN>
int launch(const ConvP& p0, bool glds, bool aligned, hipStream_t s, bool may_split = true) {
  ConvP p = p0;
  p.ntiles = (p.Cout + BN - 1) / BN;
  // 256-row tiles (8 waves, weight tile shared by twice the pixels) once they still fill the chip twice over
  // (cold operands, tools/dev/conv_cold_bench.py: 104 tiles of 256 x 128 on 256 CUs still beat 208 of 128 x 128 by
  // 18...22 % on the 3x3 layers of the 32x52 / 16x26 stages — the 3-stage pipeline matters more than the fill)
  const long long mink = dastune::get(dastune::CONV_BIG_MINK), minb = dastune::get(dastune::CONV_BIG_MINBLOCKS);
  const int rows = p.M - p.m_base;   // rows of this launch
  const int nk128 = (int)((long long)p.K * sizeof(T) / 128);
  const long long nb3 = (long long)((rows + 255) / 256) * p.ntiles;
  const bool tail = !may_split;   // this launch covers the rows a 256-row-tile launch left over (see tail_split_mtiles)
  const int ks3 = pick_ksplit(nb3, nk128, 1, 2, (long long)rows * p.Cout, tail);
  const bool big = (may_split || ks3 > 1) && glds && BN == 128 && sizeof(OT) == 2 && p.up_sh == 0 && nb3 * ks3 >= minb &&
                   p.K >= mink &&
                   p.xbytes != 0;  // (0 = more than 4 GiB of input: not addressable by 32-bit buffer offsets)
  const int bm = big ? 256 : BM;
  int mtiles = (rows + bm - 1) / bm;
  if (big && ks3 == 1) {
    const long long keep = tail_split_mtiles(mtiles, p.ntiles, may_split);
    if (keep < mtiles) {   // rows of the under-filled last round: 128-row tiles, second launch
      ConvP tail = p0;
      tail.m_base = p0.m_base + (int)keep * 256;
      const int rc = launch<T, OT, BN>(tail, glds, aligned, s, false);
      if (rc != DAS_OK) return rc;
      p.M = tail.m_base;
      mtiles = (int)keep;
    } else if (may_split) {
      p.mstep = balanced_mstep(rows, 256, p.ntiles, p.KH == 1 && p.KW == 1, &mtiles);
    }
  }
  p.nblocks = p.ntiles * mtiles;
  const size_t sm_reg = std::max<size_t>(2 * (size_t)(BM + BN) * 64, epilogue_smem_bytes<OT, BN>());
  const size_t sm_glds = std::max<size_t>(2 * (size_t)(BM + BN) * 128, epilogue_smem_bytes<OT, BN>());
  constexpr int BIG = (BN == 128) ? 256 : 128;  // only instantiated for BN = 128
  const size_t sm_big = std::max<size_t>(3 * (size_t)(BIG + BN) * 128, epilogue_smem_bytes<OT, BN, BIG>());
  static bool attr_set = false;  // one flag per instantiation; > 64 KiB dynamic LDS needs the opt-in
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv_reg_kernel<T, OT, BN, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm_reg);
    (void)hipFuncSetAttribute((const void*)conv_reg_kernel<T, OT, BN, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm_reg);
    (void)hipFuncSetAttribute((const void*)conv_reg_kernel<T, OT, BN, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm_reg);
    (void)hipFuncSetAttribute((const void*)conv_reg_kernel<T, OT, BN, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm_reg);
    (void)hipFuncSetAttribute((const void*)conv_glds_kernel<T, OT, BN, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm_glds);
    (void)hipFuncSetAttribute((const void*)conv_glds_kernel<T, OT, BN, 128, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm_glds);
    (void)hipFuncSetAttribute((const void*)conv_glds_kernel<T, OT, BN, 128, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm_glds);
    if (BN == 128) {
      (void)hipFuncSetAttribute((const void*)conv_glds3_kernel<T, OT, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm_big);
      (void)hipFuncSetAttribute((const void*)conv_glds3_kernel<T, OT, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm_big);
      (void)hipFuncSetAttribute((const void*)conv_glds3_kernel<T, OT, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm_big);
      (void)hipFuncSetAttribute((const void*)conv_glds3_kernel<T, OT, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm_big);
      (void)hipFuncSetAttribute((const void*)conv_glds3_kernel<T, OT, false, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm_big);
      (void)hipFuncSetAttribute((const void*)conv_glds3_kernel<T, OT, true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm_big);
      (void)hipFuncSetAttribute((const void*)conv_glds3_kernel<T, OT, false, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm_big);
      (void)hipFuncSetAttribute((const void*)conv_glds3_kernel<T, OT, true, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm_big);
      (void)hipFuncSetAttribute((const void*)conv_glds3_kernel<T, OT, false, false, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm_big);
      (void)hipFuncSetAttribute((const void*)conv_glds3_kernel<T, OT, true, false, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm_big);
    }
    attr_set = true;
  }
  if (big) {
    // ping-pong schedule from conv.glds3_pp_mink K on (-1: never)
    const long long ppk = dastune::get(dastune::CONV_GLDS3_PP_MINK);
    const bool pp3 = ppk >= 0 && p.K >= ppk;
    if constexpr (BN == 128 && sizeof(OT) == 2) {
      if (ks3 > 1) {
        dastune::note_kernel("conv_glds3_kernel<splitk>");
        if (dastune::get(dastune::CONV_SPLITK_INKERNEL) == 1)   // (the last arriver of a tile runs the epilogue itself: the full kernel)
          return pp3 ? launch_splitk<OT, 128, 256, true>(conv_glds3_kernel<T, OT, true>, p, ks3, sm_big, s)
                     : launch_splitk<OT, 128, 256, true>(conv_glds3_kernel<T, OT, false>, p, ks3, sm_big, s);
        return pp3 ? launch_splitk<OT, 128, 256>(conv_glds3_kernel<T, OT, true, false, false, false, true>, p, ks3, sm_big, s)
                   : launch_splitk<OT, 128, 256>(conv_glds3_kernel<T, OT, false, false, false, false, true>, p, ks3, sm_big, s);
      }
    }
    dastune::note_kernel(pp3 ? "conv_glds3_kernel<pp>" : "conv_glds3_kernel");
    t_last_tile_rows = p.mstep ? p.mstep : 256;
    const bool bits = p.bnb_bits || p.res_bits, bnb = p.bnb_raw != nullptr;   // (epilogue variants: conv_epilogue's BITS / BNB / PLAIN)
    const bool plain = epilogue_plain(p);
    if (pp3) {
      if (bits) hipLaunchKernelGGL((conv_glds3_kernel<T, OT, true, true>), dim3(p.nblocks), dim3(512), sm_big, s, p);
      else if (bnb) hipLaunchKernelGGL((conv_glds3_kernel<T, OT, true, false>), dim3(p.nblocks), dim3(512), sm_big, s, p);
      else if (plain) hipLaunchKernelGGL((conv_glds3_kernel<T, OT, true, false, false, true>), dim3(p.nblocks), dim3(512), sm_big, s, p);
      else hipLaunchKernelGGL((conv_glds3_kernel<T, OT, true, false, false>), dim3(p.nblocks), dim3(512), sm_big, s, p);
    } else {
      if (bits) hipLaunchKernelGGL((conv_glds3_kernel<T, OT, false, true>), dim3(p.nblocks), dim3(512), sm_big, s, p);
      else if (bnb) hipLaunchKernelGGL((conv_glds3_kernel<T, OT, false, false>), dim3(p.nblocks), dim3(512), sm_big, s, p);
      else if (plain) hipLaunchKernelGGL((conv_glds3_kernel<T, OT, false, false, false, true>), dim3(p.nblocks), dim3(512), sm_big, s, p);
      else hipLaunchKernelGGL((conv_glds3_kernel<T, OT, false, false, false>), dim3(p.nblocks), dim3(512), sm_big, s, p);
    }
  } else if (glds) {
    const int ks = BN >= 64 ? pick_ksplit(p.nblocks, nk128, 2, 1, (long long)rows * p.Cout, tail) : 1;
    if (ks > 1) {
      dastune::note_kernel("conv_glds_kernel<splitk>");
      return launch_splitk<OT, BN, 128>(conv_glds_kernel<T, OT, BN, 128>, p, ks, sm_glds, s);
    }
    dastune::note_kernel("conv_glds_kernel");
    if (p.bnb_raw) hipLaunchKernelGGL((conv_glds_kernel<T, OT, BN, 128>), dim3(p.nblocks), dim3(256), sm_glds, s, p);
    else if (epilogue_plain(p)) hipLaunchKernelGGL((conv_glds_kernel<T, OT, BN, 128, 2>), dim3(p.nblocks), dim3(256), sm_glds, s, p);
    else hipLaunchKernelGGL((conv_glds_kernel<T, OT, BN, 128, 1>), dim3(p.nblocks), dim3(256), sm_glds, s, p);
  } else if (aligned) {
    dastune::note_kernel("conv_reg_kernel");
    if (epilogue_plain(p)) hipLaunchKernelGGL((conv_reg_kernel<T, OT, BN, true, 2>), dim3(p.nblocks), dim3(256), sm_reg, s, p);
    else hipLaunchKernelGGL((conv_reg_kernel<T, OT, BN, true>), dim3(p.nblocks), dim3(256), sm_reg, s, p);
  } else {
    dastune::note_kernel("conv_reg_kernel");
    if (epilogue_plain(p)) hipLaunchKernelGGL((conv_reg_kernel<T, OT, BN, false, 2>), dim3(p.nblocks), dim3(256), sm_reg, s, p);
    else hipLaunchKernelGGL((conv_reg_kernel<T, OT, BN, false>), dim3(p.nblocks), dim3(256), sm_reg, s, p);
  }
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}

// ------------------------------------------------------------------ streaming 1x1 convolution
// The large-M 1x1 layers with K <= 256 (the expand / reduce convs of the 128x208 ... 32x52 stages, forward and data
// gradient) are HBM-bound: 2 * M * (Cin + Cout) bytes against K <= 256 MACs per output. On the tile kernels above
// they run at 2...3.6x their HBM floor, because a workgroup that finishes its K loop in 2...8 steps spends its life in
// the serial load -> MFMA -> store phases of one tile. This kernel is PERSISTENT: the wave's slice of the weights
// (32 channels x K, the MFMA A fragments) stays in registers for the whole launch, the workgroup walks the pixel rows
// in tiles of 64, DMAs them (64 x K, 64-channel sub-tiles with the 128-byte-row swizzle of conv_glds_kernel) three
// tiles ahead into four LDS stages, and stores straight from the accumulators: with the MFMA row
// -> channel permutation below a lane owns EIGHT consecutive channels of a pixel, i.e. one 16-byte store.
// BatchNorm statistics accumulate in registers over the whole launch (one round of atomics per workgroup).
// vmcnt counts loads and stores alike and orders them only within each kind, so a wave that stores cannot count its
// way to "tile i has landed". Hence two kinds of waves: waves 0-7 multiply and store and never wait on memory;
// waves 8-9 only issue the DMA, three tiles ahead into four LDS stages, wait with a counted vmcnt and release the
// others through the workgroup barrier (one barrier per tile).
// MODE (keeps each variant under the 168 registers that ten waves per workgroup allow): 0 = plain, optional BatchNorm
// statistics (training forward); 1 = scale / shift, optional ReLU (eval); 2 = residual add, optional ReLU (data gradient);
// 3 / 4 = data gradient with the fused BatchNorm-backward reduction (ConvP::bnb_*): optional residual, mask from the
// saved output y (3; no y = no mask), from its recorded bits (6) or recomputed from raw with the layer's affine (4),
// per-channel sums in registers;
// 5 = scale / shift, then residual add, optional ReLU (the closing 1x1 conv of an eval-mode bottleneck).
// The per-channel sums of a wave end up in lane 15 of each DPP row: sixteen values (8 channels x {sum, sum of squares})
// for each of the four channel groups. Sixteen atomic instructions with four live lanes each cost the launch 4-5 us of
// tail (tools/dev/stream_fixed.py); the 64 values go through 256 bytes of LDS instead and leave as ONE atomic instruction
// with every lane live: lane = group * 16 + channel * 2 + which.
__device__ __forceinline__ void stream_stat_flush(float* sred, int lane, int n0, const ConvP& p) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (this wave's LDS writes have landed; nothing is reordered across)
  const float v = sred[lane];
  const int ch = n0 + (lane >> 4) * 8 + ((lane & 15) >> 1), w = lane & 1;
  if (ch < p.Cout) {
    const int slot = p.stat_slots > 1 ? (int)(blockIdx.x % (unsigned)p.stat_slots) : 0;
    atomicAdd(p.stats + (slot * 2 + w) * p.Cout + ch, v);
  }
}
template <int KB, int WN, int MODE>   // K = 32 * KB input channels; WN waves across the 32-channel groups, 8 / WN across pixels
__global__ __launch_bounds__(640) void conv1x1_stream_kernel(ConvP p, int ncol, int ntiles, int ntm) {   // ntm: das_tuning key conv.stream_nt
  using T = bf16_t;
  constexpr bool STATS = MODE == 0, AFF = MODE == 1 || MODE == 5, RES = MODE == 2 || MODE == 5, BNB = MODE == 3 || MODE == 4 || MODE == 6;
  constexpr int WM = 8 / WN, TM = 64, PB = TM / WM / 16;    // 16-pixel blocks per wave and tile
  constexpr int K = KB * 32, SUBS = (K + 63) / 64;
  constexpr int SUBB = TM * 128, STAGE = SUBS * SUBB;       // bytes
  // LDS stages / tiles in flight. K = 64: a stage is 8 KiB, so three tiles ahead are 24 KiB per workgroup (two per CU);
  // DAS_STREAM_K64_NS (dev builds, tools/dev/stream_depth_ab.py) deepens that pipeline for an A/B run.
#ifndef DAS_STREAM_K64_NS
#define DAS_STREAM_K64_NS 4
#endif
#ifndef DAS_STREAM_K128_NS
#define DAS_STREAM_K128_NS 4
#endif
  constexpr int NS = KB == 2 ? DAS_STREAM_K64_NS : (KB == 4 ? DAS_STREAM_K128_NS : 4), AHEAD = NS - 1;
  constexpr int IPL = SUBS * 4;                             // DMA instructions per loader wave and tile
  constexpr unsigned OOB = 0xFFFFFFF0u;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int logical0 = xcd_remap(blockIdx.x, gridDim.x);
  if (wave >= 8) {   // ---- loader waves
    const int first = logical0 / ncol, tstride = gridDim.x / ncol;
    const int ldr = wave - 8;
    const v4i_t xrs = make_rsrc(p.x, p.xbytes);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned xrow2 = (unsigned)p.xps * 2u;
    auto issue = [&](int tile, int stage) {
#pragma unroll
      for (int j = 0; j < IPL; ++j) {
        const int ii = ldr * IPL + j, sub = ii >> 3, rr = ii & 7;
        const int row = rr * 8 + (lane >> 3);
        const int kslot = (lane & 7) ^ ((row >> 1) & 7);
        const long long m = (long long)tile * TM + row;
        const bool ok = m < p.M && sub * 64 + kslot * 8 < K;
        const unsigned off = (unsigned)m * xrow2 + (unsigned)(sub * 64 + kslot * 8) * 2u;
        if (ntm & 1) dma16_buf_nt(ok ? off : OOB, xrs, lds0 + stage * STAGE + sub * SUBB + rr * 1024);
        else dma16_buf(ok ? off : OOB, xrs, lds0 + stage * STAGE + sub * SUBB + rr * 1024);
      }
    };
    const int mine = first < ntiles ? (ntiles - first + tstride - 1) / tstride : 0;   // tiles of this workgroup
    for (int i = 0; i < AHEAD && i < mine; ++i) issue(first + i * tstride, i);
    for (int i = 0; i < mine; ++i) {
      const int later = min(mine - 1 - i, AHEAD - 1);   // tiles issued after tile i that may still be in flight
      switch (later) {                                  // (vmcnt takes an immediate)
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPL) : "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * IPL) : "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * IPL < 63 ? 3 * IPL : 63) : "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * IPL < 63 ? 4 * IPL : 63) : "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * IPL < 63 ? 5 * IPL : 63) : "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * IPL < 63 ? 6 * IPL : 63) : "memory"); break;
      }
      __builtin_amdgcn_s_barrier();   // tile i is in LDS; the other waves are done with tile i - 1
      if (i + AHEAD < mine) issue(first + (i + AHEAD) * tstride, (i + AHEAD) % NS);
    }
    return;
  }
  const int wn = wave % WN, wm = wave / WN;
  const int logical = logical0;   // the column blocks of a tile share an XCD (its L2)
  const int col = logical % ncol, first = logical / ncol, tstride = gridDim.x / ncol;
  const int n0 = col * (32 * WN) + wn * 32;                // this wave's 32 output channels
  const int q = lane & 15, g4 = lane >> 4;

  // weights -> MFMA A fragments. MFMA row R of half a holds channel n0 + (R >> 2) * 8 + a * 4 + (R & 3), so that the
  // accumulator lane (pixel q, rows g4 * 4 + j) owns channels n0 + g4 * 8 + a * 4 + j: eight consecutive ones.
  uint4 wf[2][KB];
  {
    const T* wg = reinterpret_cast<const T*>(p.w);
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int ch = n0 + (q >> 2) * 8 + a * 4 + (q & 3);
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
        wf[a][kb] = ch < p.Cout ? *reinterpret_cast<const uint4*>(wg + (long long)ch * K + kb * 32 + g4 * 8)
                                : make_uint4(0, 0, 0, 0);
    }
  }
  const int c8 = n0 + g4 * 8;   // this lane's eight output channels
  const bool cok = c8 < p.Cout;
  float* sred = reinterpret_cast<float*>(smem + NS * STAGE) + wave * 64;   // this wave's 64 reduced sums (see stream_stat_flush)
  float sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    sc[j] = (AFF && p.scale && cok) ? p.scale[c8 + j] : 1.f;
    sh[j] = (AFF && p.shift && cok) ? p.shift[c8 + j] : 0.f;
  }
  float ssum[8], ssq[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { ssum[j] = 0.f; ssq[j] = 0.f; }

  T* yg = reinterpret_cast<T*>(p.y);
  if constexpr (BNB) {
    // The epilogue operands (residual = the other gradient of the same tensor, raw, y) are requested when the tile
    // opens, HB pixel blocks at a time, by hand-issued loads that land while those blocks' MFMAs run; the wait that
    // follows also drains the previous blocks' stores (one counter for both kinds) — the price of not keeping a second
    // register set per operand, which this variant has no room for.
    // (MODE 6 — the mask as bits, no y vectors — has the registers for all of a tile's pixel blocks at once — one exposed round trip per tile instead of two: the
    // small-M launches walk three or four tiles per workgroup with nothing else to hide the latency behind)
#ifndef DAS_STREAM_HB6
#define DAS_STREAM_HB6 PB
#endif
    // (K = 256 with eight column waves keeps 64 weight registers: two blocks at a time there, as in MODE 3)
    constexpr int HB = (MODE == 6 && !(KB == 8 && WN == 8)) ? (DAS_STREAM_HB6 < PB ? DAS_STREAM_HB6 : PB) : (PB > 2 ? 2 : PB);
    const T* rgb = reinterpret_cast<const T*>(p.res);
    const T* bx = reinterpret_cast<const T*>(p.bnb_raw);
    const T* by = MODE == 3 ? reinterpret_cast<const T*>(p.bnb_y) : nullptr;
    const unsigned char* bb = MODE == 6 ? p.bnb_bits : nullptr;   // (the mask as a byte per vector instead of y)
    const unsigned char* rb = rgb ? p.res_bits : nullptr;        // (the residual's own mask: res * mask)
    float mu[8], is[8], ga[8], be[8];
    if constexpr (MODE == 4) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        mu[j] = cok ? p.bnb_mean[c8 + j] : 0.f; is[j] = cok ? p.bnb_invstd[c8 + j] : 0.f;
        ga[j] = cok ? p.bnb_gamma[c8 + j] : 0.f; be[j] = cok ? p.bnb_beta[c8 + j] : 0.f;
      }
    }
    int st = 0;
    for (int t = first; t < ntiles; t += tstride) {
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();    // the loader waves saw tile t land
      asm volatile("" ::: "memory");
      const char* sx = smem + st * STAGE;
#pragma unroll
      for (int h0 = 0; h0 < PB; h0 += HB) {
        v4i_t lr[HB], lx[HB], ly[HB];
        int lb[HB], lrb[HB];
#pragma unroll
        for (int h = 0; h < HB; ++h) {
          long long m = (long long)t * TM + wm * (TM / WM) + (h0 + h) * 16 + q;
          m = m < p.M ? m : p.M - 1;
          const long long eo = m * p.bnb_ps + (cok ? c8 : 0);
          asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(lx[h]) : "v"(bx + eo) : "memory");
          if (by) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ly[h]) : "v"(by + eo) : "memory");
          if (bb) asm volatile("global_load_ubyte %0, %1, off" : "=v"(lb[h]) : "v"(bb + eo / 8) : "memory");
          if (rgb) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(lr[h]) : "v"(rgb + m * p.rps + (cok ? c8 : 0)) : "memory");
          if (rb) asm volatile("global_load_ubyte %0, %1, off" : "=v"(lrb[h]) : "v"(rb + (m * p.rps + (cok ? c8 : 0)) / 8) : "memory");
        }
        f32x4_t acc0[HB], acc1[HB];
#pragma unroll
        for (int h = 0; h < HB; ++h) {
          const int row = wm * (TM / WM) + (h0 + h) * 16 + q;
          acc0[h] = f32x4_t{0.f, 0.f, 0.f, 0.f};
          acc1[h] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kb = 0; kb < KB; ++kb) {
            const int slot = ((kb & 1) * 4 + g4) ^ ((row >> 1) & 7);
            const uint4 bf = *reinterpret_cast<const uint4*>(sx + (kb >> 1) * SUBB + row * 128 + slot * 16);
            acc0[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[0][kb]),
                                                              __builtin_bit_cast(bf16x8_t, bf), acc0[h], 0, 0, 0);
            acc1[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[1][kb]),
                                                              __builtin_bit_cast(bf16x8_t, bf), acc1[h], 0, 0, 0);
          }
        }
#pragma unroll
        for (int h = 0; h < HB; ++h) {   // (the operands name the destination registers: they stay allocated until here)
          if constexpr (MODE == 6) {    // (no y vectors in this variant: their registers are what lets HB cover the whole tile)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(lx[h]), "+v"(lr[h]), "+v"(lb[h]), "+v"(lrb[h])::"memory");
          } else {
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(lx[h]), "+v"(ly[h]), "+v"(lr[h]), "+v"(lb[h]), "+v"(lrb[h])::"memory");
          }
        }
#pragma unroll
        for (int h = 0; h < HB; ++h) {
          const long long m = (long long)t * TM + wm * (TM / WM) + (h0 + h) * 16 + q;
          if (m < p.M && cok) {
            float v[8], x[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = acc0[h][j]; v[4 + j] = acc1[h][j]; }
            uint4 o = Elem<T>::pack(v);
            Elem<T>::unpack(o, v);   // the conv result as a tile kernel would have staged it (bf16)
            if (rgb) {
              float r[8];
              Elem<T>::unpack(__builtin_bit_cast(uint4, lr[h]), r);
              if (rb) {
#pragma unroll
                for (int j = 0; j < 8; ++j) r[j] = ((unsigned)lrb[h] >> j & 1u) ? r[j] : 0.f;
              }
#pragma unroll
              for (int j = 0; j < 8; ++j) v[j] += r[j];
            }
            Elem<T>::unpack(__builtin_bit_cast(uint4, lx[h]), x);
            if (MODE == 4) {
#pragma unroll
              for (int j = 0; j < 8; ++j) v[j] = bn_affine(x[j], mu[j], is[j], ga[j], be[j]) > 0.f ? v[j] : 0.f;
            } else if (by || bb) {
              float yo[8];
              Elem<T>::unpack(by ? __builtin_bit_cast(uint4, ly[h]) : mask_vec<T>((unsigned)lb[h]), yo);
#pragma unroll
              for (int j = 0; j < 8; ++j) v[j] = yo[j] > 0.f ? v[j] : 0.f;
            }
            o = Elem<T>::pack(v);
            Elem<T>::unpack(o, v);   // dZ as stored
#pragma unroll
            for (int j = 0; j < 8; ++j) { ssum[j] += v[j]; ssq[j] += v[j] * x[j]; }   // (sum dZ * raw: centred below)
            st16(yg + m * p.yps + c8, o, ntm & 4);
          }
        }
      }
      st = (st + 1) % NS;
    }
    // sum dZ * xhat = invstd * (sum dZ * raw - mean * sum dZ), per workgroup (linear, so the partials may be centred
    // one by one); then over the 16 pixels (lanes) of a DPP row and one atomic per channel from lane 15 of each row
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float m_ = cok ? p.bnb_mean[c8 + j] : 0.f, i_ = cok ? p.bnb_invstd[c8 + j] : 0.f;
      ssq[j] = i_ * (ssq[j] - m_ * ssum[j]);
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        float v = w ? ssq[j] : ssum[j];
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xF, 0xF, true));
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xF, 0xF, true));
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xF, 0xF, true));
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xF, 0xF, true));
        if (q == 15) sred[g4 * 16 + j * 2 + w] = v;
      }
    }
    stream_stat_flush(sred, lane, n0, p);
    return;
  }
  const T* rg = RES ? reinterpret_cast<const T*>(p.res) : nullptr;
  // Residual rows are fetched one tile ahead by hand-issued loads: if the compiler tracked them it would have to
  // wait with vmcnt(0) (loads and stores mixed in one counter), i.e. for the rows just requested and for every store.
  // Loads return in order among themselves, so "at most PB operations outstanding" right after the PB loads of the
  // NEXT tile were issued proves this tile's rows have landed, whatever the stores do. The instruction count per tile
  // is kept constant (rows past M / past the last tile re-read a valid row).
  v4i_t rv[PB], rnext[PB];
  auto res_issue = [&](int tile, v4i_t (&dst)[PB]) {
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
      long long m = (long long)tile * TM + wm * (TM / WM) + pb * 16 + q;
      m = m < p.M ? m : p.M - 1;
      const T* src = rg + m * p.rps + (cok ? c8 : 0);
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst[pb]) : "v"(src) : "memory");
    }
  };
  if (rg) res_issue(min(first, ntiles - 1), rv);
  int st = 0;
  // (two register sets for the residual rows, used alternately: a copy would read registers with loads in flight)
  auto do_tile = [&](int t, v4i_t (&cur)[PB], v4i_t (&nxt)[PB]) {
    asm volatile("" ::: "memory");   // (this wave's LDS reads and stores of the previous tile stay above the barrier)
    __builtin_amdgcn_s_barrier();    // the loader waves saw tile t land
    asm volatile("" ::: "memory");
    if (rg) res_issue(min(t + tstride, ntiles - 1), nxt);
    const char* sx = smem + st * STAGE;
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
      const int row = wm * (TM / WM) + pb * 16 + q;
      f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        const int slot = ((kb & 1) * 4 + g4) ^ ((row >> 1) & 7);
        const uint4 bf = *reinterpret_cast<const uint4*>(sx + (kb >> 1) * SUBB + row * 128 + slot * 16);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[0][kb]),
                                                       __builtin_bit_cast(bf16x8_t, bf), acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[1][kb]),
                                                       __builtin_bit_cast(bf16x8_t, bf), acc1, 0, 0, 0);
      }
      const long long m = (long long)t * TM + row;
      if (rg && pb == 0) {
#pragma unroll
        for (int i = 0; i < PB; ++i) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(cur[i]) : "n"(PB) : "memory");
      }
      if (m < p.M && cok) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] = AFF ? acc0[j] * sc[j] + sh[j] : acc0[j];
          v[4 + j] = AFF ? acc1[j] * sc[4 + j] + sh[4 + j] : acc1[j];
        }
        uint4 o = Elem<T>::pack(v);
        const bool stats = STATS && p.stats;
        if (stats || rg || p.relu) {
          Elem<T>::unpack(o, v);   // the values as stored: statistics and the residual add see the rounded ones
          if (stats) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { ssum[j] += v[j]; ssq[j] += v[j] * v[j]; }
          }
          if (rg) {
            float r[8];
            Elem<T>::unpack(__builtin_bit_cast(uint4, cur[pb]), r);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += r[j];
          }
          if (p.relu) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
          }
          if (rg || p.relu) o = Elem<T>::pack(v);
        }
        st16(yg + m * p.yps + c8, o, ntm & 4);
      }
    }
    st = (st + 1) % NS;
  };
  for (int t = first; t < ntiles; t += 2 * tstride) {
    do_tile(t, rv, rnext);
    if (t + tstride >= ntiles) break;
    do_tile(t + tstride, rnext, rv);
  }
  if (rg) {   // the trailing dummy loads: their destination registers stay allocated until they have landed
#pragma unroll
    for (int i = 0; i < PB; ++i) asm volatile("s_waitcnt vmcnt(0)" : "+v"(rv[i]), "+v"(rnext[i])::"memory");
  }
  if (STATS && p.stats) {
    // sum over the 16 pixels (lanes) of a DPP row, then one atomic per channel from lane 15 of each row
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        float v = w ? ssq[j] : ssum[j];
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xF, 0xF, true));
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xF, 0xF, true));
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xF, 0xF, true));
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xF, 0xF, true));
        if (q == 15) sred[g4 * 16 + j * 2 + w] = v;
      }
    }
    stream_stat_flush(sred, lane, n0, p);
  }
}

// ------------------------------------------------------------------ weight-stationary 1x1 convolution for K = 512 / 1024 (round 6)
// The long-K "reduce" convs (512 -> 128 at 64 x 104, 1024 -> 256 at 32 x 52: forward with statistics, data gradients with the
// fused BatchNorm backward) run on the tile kernels at 1.4-2.4x their HBM floor: the tile's K loop is bound by LDS-DMA ISSUE
// (a wave's interval carries 6 DMA instructions of ~150 cycles against 512 cycles of MFMAs), and a third of what it issues is
// the weight tile every workgroup re-fetches every step. conv1x1_stream_kernel keeps the weights in registers, but 32 channels
// x K > 256 do not fit beside its ten waves' 168 registers. Here the eight waves of a workgroup (2 per SIMD: 256 registers)
// split K in two: waves 0-3 ("front", one per 32-channel group of the 128-channel column block) hold the lower half of K,
// waves 4-7 ("back") the upper half. The pixel operand streams through LDS in tiles of 16 rows x K (32 KiB at K = 1024, four
// stages, three tiles ahead); the BACK waves issue its DMA — they never store to global memory, so their counted vmcnt does
// say "tile i has landed" (cf. conv1x1_stream_kernel's loader waves) — and hand their partial sums to the front waves through
// 8 KiB of LDS; the front waves add their own, run the epilogue (a lane owns eight consecutive channels of one pixel: one
// 16-byte store) and carry the per-channel sums in registers for the whole launch. ONE workgroup barrier per tile (it says
// "partial sums of tile i written" and "tile i + 1 landed" at once: two alternating partial-sum sets make that legal).
// MODE 0: plain output, optional BatchNorm statistics (training forward). MODE 4: data gradient with the fused BatchNorm-
// backward reduction, ReLU mask recomputed from the pre-norm tensor (ConvP::bnb_raw, no residual: the `bx` launches).
template <int KB, int MODE>   // K = 32 * KB input channels (KB = 16 / 32)
__global__ __launch_bounds__(512) void conv1x1_kstream_kernel(ConvP p, int ncol, int ntiles) {
  using T = bf16_t;
  constexpr bool BNB = MODE == 4 || MODE == 6;
  constexpr int K = KB * 32, KH = KB / 2;                     // K steps of 32 per wave
  constexpr int PB = KB == 16 ? 2 : 1;                        // 16-pixel blocks per tile: 32-row tiles at K = 512, 16 at K = 1024
  constexpr int TM = 16 * PB, SUBS = K / 64, SUBB = TM * 128, STAGE = SUBS * SUBB;   // (32 KiB per stage either way)
  constexpr int NS = 4, AHEAD = NS - 1;
  constexpr int RPS = TM / 8;                                 // DMA instructions per sub-tile (8 rows x 128 B each)
  constexpr int IPW = SUBS * RPS / 4;                         // DMA instructions per back wave and tile
  constexpr unsigned OOB = 0xFFFFFFF0u;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 3, wk = wave >> 2;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);       // the column blocks of a row tile share an XCD (its L2)
  const int col = logical % ncol, first = logical / ncol, tstride = gridDim.x / ncol;
  const int n0 = col * 128 + wn * 32;                         // this wave's 32 output channels
  const int q = lane & 15, g4 = lane >> 4;
  const int mine = first < ntiles ? (ntiles - first + tstride - 1) / tstride : 0;   // row tiles of this workgroup

  // back waves: the first three pixel tiles are requested BEFORE anything else (the weights' 32 KiB per wave then load beside them)
  const v4i_t xrs = make_rsrc(p.x, p.xbytes);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned xrow2 = (unsigned)p.xps * 2u;
  auto issue = [&](int tile, int stage) {
#pragma unroll
    for (int j = 0; j < IPW; ++j) {
      const int ii = wn * IPW + j, sub = ii / RPS, rr = ii % RPS;
      const int row = rr * 8 + (lane >> 3);
      const int kslot = (lane & 7) ^ ((row >> 1) & 7);
      const long long m = (long long)tile * TM + row;
      const unsigned off = (unsigned)m * xrow2 + (unsigned)(sub * 64 + kslot * 8) * 2u;
      dma16_buf(m < p.M ? off : OOB, xrs, lds0 + stage * STAGE + sub * SUBB + rr * 1024);
    }
  };
  if (wk == 1) {
    for (int i = 0; i < AHEAD && i < mine; ++i) issue(first + i * tstride, i);
  }

  // weights -> MFMA A fragments of this wave's K half (row permutation as in conv1x1_stream_kernel: the accumulator lane
  // (pixel q, rows g4 * 4 + j) owns channels n0 + g4 * 8 + a * 4 + j)
  uint4 wf[2][KH];
  {
    const T* wg = reinterpret_cast<const T*>(p.w);
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int ch = n0 + (q >> 2) * 8 + a * 4 + (q & 3);
#pragma unroll
      for (int kb = 0; kb < KH; ++kb)
        wf[a][kb] = ch < p.Cout ? *reinterpret_cast<const uint4*>(wg + (long long)ch * K + (wk * KH + kb) * 32 + g4 * 8)
                                : make_uint4(0, 0, 0, 0);
    }
  }
  // partial sums of the back waves: two sets used alternately (tile i in set i & 1), so that ONE barrier per tile is enough
  f32x4_t* xch = reinterpret_cast<f32x4_t*>(smem + NS * STAGE) + (wn * 64 + lane) * (2 * PB);
  constexpr int XSET = 4 * 64 * 2 * PB;   // f32x4_t per set
  auto partial = [&](int st, int pb, f32x4_t& acc0, f32x4_t& acc1) {   // this wave's K half of pixel block pb of the tile in stage st
    const char* sx = smem + st * STAGE;
    const int row = pb * 16 + q;
    acc0 = f32x4_t{0.f, 0.f, 0.f, 0.f};
    acc1 = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < KH; ++kb) {
      const int kg = wk * KH + kb;
      const int slot = ((kg & 1) * 4 + g4) ^ ((row >> 1) & 7);
      const uint4 bf = *reinterpret_cast<const uint4*>(sx + (kg >> 1) * SUBB + row * 128 + slot * 16);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[0][kb]), __builtin_bit_cast(bf16x8_t, bf), acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[1][kb]), __builtin_bit_cast(bf16x8_t, bf), acc1, 0, 0, 0);
    }
  };
  if (mine == 0) return;   // (wave-uniform and workgroup-uniform: nobody reaches a barrier)

  // One barrier per tile. Barrier i says two things: "the back waves' partial sums of tile i are in LDS (set i & 1)" and "tile
  // i + 1 has landed". Stage of tile i + 3 = stage of tile i - 1, read out before barrier i - 1; partial-sum set i & 1 is
  // written again for tile i + 2, after barrier i + 1, which the front waves only reach after reading set i & 1.
  if (wk == 1) {   // ---- back waves: the upper K half, the DMA of the pixel tiles
    // tile 0 landed: at most tiles 1, 2 (and the weight loads, issued after them: waited for by the compiler at first use) fly
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();     // start: tile 0 is in LDS
    asm volatile("" ::: "memory");
    for (int i = 0; i < mine; ++i) {
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) {
        f32x4_t a0, a1;
        partial(i % NS, pb, a0, a1);
        xch[(i & 1) * XSET + pb * 2] = a0;
        xch[(i & 1) * XSET + pb * 2 + 1] = a1;
      }
      // tile i + 1 landed: the tiles issued after it that may still fly are i + 2 (tile i + 3 goes out below)
      if (i + 2 < mine) {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(IPW) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();   // barrier i
      asm volatile("" ::: "memory");
      if (i + AHEAD < mine) issue(first + (i + AHEAD) * tstride, (i + AHEAD) % NS);   // (into the stage of tile i - 1)
    }
    return;
  }

  // ---- front waves: the lower K half, the sum of the halves, the epilogue
  const int c8 = n0 + g4 * 8;   // this lane's eight output channels
  const bool cok = c8 < p.Cout;
  // this wave's 64 reduced sums: over the start of its OWN partial-sum slots (dead once its loop has ended: the back wave of the
  // same channel group wrote them for the last time before the last barrier) — at K = 512 the stages and the two sets fill all 160 KiB
  float* sred = reinterpret_cast<float*>(reinterpret_cast<f32x4_t*>(smem + NS * STAGE) + wn * 64 * (2 * PB));
  float ssum[8], ssq[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { ssum[j] = 0.f; ssq[j] = 0.f; }
  float mu[8], is[8], ga[8], be[8];
  const T* bx = reinterpret_cast<const T*>(p.bnb_raw);
  // MODE 6: the ReLU mask comes as recorded bits, a second gradient of the same tensor may be added (itself masked by bits)
  const T* rgb = (MODE == 6 || MODE == 2) ? reinterpret_cast<const T*>(p.res) : nullptr;   // (MODE 2: y = conv + residual, nothing else)
  const unsigned char* bb = MODE == 6 ? p.bnb_bits : nullptr;
  const unsigned char* rb = rgb ? p.res_bits : nullptr;
  if constexpr (MODE == 4) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      mu[j] = cok ? p.bnb_mean[c8 + j] : 0.f; is[j] = cok ? p.bnb_invstd[c8 + j] : 0.f;
      ga[j] = cok ? p.bnb_gamma[c8 + j] : 0.f; be[j] = cok ? p.bnb_beta[c8 + j] : 0.f;
    }
  }
  T* yg = reinterpret_cast<T*>(p.y);
  __builtin_amdgcn_s_barrier();       // start: tile 0 is in LDS
  asm volatile("" ::: "memory");
  for (int i = 0; i < mine; ++i) {
    const int t = first + i * tstride;
    v4i_t lx[PB], lr[PB];
    int lb[PB], lrb[PB];
    if constexpr (MODE == 2) {        // the residual rows of this lane's pixels: requested now, used after the MFMAs
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) {
        long long mm = (long long)t * TM + pb * 16 + q;
        mm = mm < p.M ? mm : p.M - 1;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(lr[pb]) : "v"(rgb + mm * p.rps + (cok ? c8 : 0)) : "memory");
      }
    }
    if constexpr (BNB) {              // the pre-norm rows of this lane's pixels: requested now, used after the MFMAs
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) {
        long long mm = (long long)t * TM + pb * 16 + q;
        mm = mm < p.M ? mm : p.M - 1;
        const long long eo = mm * p.bnb_ps + (cok ? c8 : 0);
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(lx[pb]) : "v"(bx + eo) : "memory");
        if constexpr (MODE == 6) {
          asm volatile("global_load_ubyte %0, %1, off" : "=v"(lb[pb]) : "v"(bb + eo / 8) : "memory");
          if (rgb) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(lr[pb]) : "v"(rgb + mm * p.rps + (cok ? c8 : 0)) : "memory");
          if (rb) asm volatile("global_load_ubyte %0, %1, off" : "=v"(lrb[pb]) : "v"(rb + (mm * p.rps + (cok ? c8 : 0)) / 8) : "memory");
        }
      }
    }
    f32x4_t acc0[PB], acc1[PB];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) partial(i % NS, pb, acc0[pb], acc1[pb]);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();     // barrier i: the partial sums of tile i are in set i & 1, tile i + 1 has landed
    asm volatile("" ::: "memory");
    if constexpr (MODE == 6) {
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) asm volatile("s_waitcnt vmcnt(0)" : "+v"(lx[pb]), "+v"(lr[pb]), "+v"(lb[pb]), "+v"(lrb[pb])::"memory");
    } else if constexpr (MODE == 2) {
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) asm volatile("s_waitcnt vmcnt(0)" : "+v"(lr[pb])::"memory");
    } else if constexpr (BNB) {
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) asm volatile("s_waitcnt vmcnt(0)" : "+v"(lx[pb])::"memory");   // (also drains the previous tile's stores)
    }
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
      const long long m = (long long)t * TM + pb * 16 + q;
      const f32x4_t b0 = xch[(i & 1) * XSET + pb * 2], b1 = xch[(i & 1) * XSET + pb * 2 + 1];
      float v[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[j] = acc0[pb][j] + b0[j]; v[4 + j] = acc1[pb][j] + b1[j]; }
      uint4 o = Elem<T>::pack(v);
      if constexpr (BNB) {
        float x[8];
        Elem<T>::unpack(o, v);            // the conv result as a tile kernel would have staged it (bf16)
        Elem<T>::unpack(__builtin_bit_cast(uint4, lx[pb]), x);
        if constexpr (MODE == 6) {
          if (rgb) {
            float r[8];
            Elem<T>::unpack(__builtin_bit_cast(uint4, lr[pb]), r);
            if (rb) {
#pragma unroll
              for (int j = 0; j < 8; ++j) r[j] = ((unsigned)lrb[pb] >> j & 1u) ? r[j] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += r[j];
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = ((unsigned)lb[pb] >> j & 1u) ? v[j] : 0.f;
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = bn_affine(x[j], mu[j], is[j], ga[j], be[j]) > 0.f ? v[j] : 0.f;
        }
        o = Elem<T>::pack(v);
        Elem<T>::unpack(o, v);            // dZ as stored
        if (m < p.M && cok) {
#pragma unroll
          for (int j = 0; j < 8; ++j) { ssum[j] += v[j]; ssq[j] += v[j] * x[j]; }   // (sum dZ * raw: centred below)
        }
      } else if constexpr (MODE == 2) {
        float r[8];
        Elem<T>::unpack(o, v);            // the conv result as a tile kernel would have staged it (bf16)
        Elem<T>::unpack(__builtin_bit_cast(uint4, lr[pb]), r);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += r[j];
        o = Elem<T>::pack(v);
      } else if (p.stats && m < p.M && cok) {
        Elem<T>::unpack(o, v);            // the values as stored
#pragma unroll
        for (int j = 0; j < 8; ++j) { ssum[j] += v[j]; ssq[j] += v[j] * v[j]; }
      }
      if (m < p.M && cok) *reinterpret_cast<uint4*>(yg + m * p.yps + c8) = o;
    }
  }
  if (p.stats) {
    // per-channel sums: over the 16 pixels (lanes) of a DPP row, then one atomic instruction per wave (stream_stat_flush)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if constexpr (MODE == 6) { mu[j] = cok ? p.bnb_mean[c8 + j] : 0.f; is[j] = cok ? p.bnb_invstd[c8 + j] : 0.f; }
      if constexpr (BNB) ssq[j] = is[j] * (ssq[j] - mu[j] * ssum[j]);   // sum dZ * xhat
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        float v = w ? ssq[j] : ssum[j];
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xF, 0xF, true));
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xF, 0xF, true));
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xF, 0xF, true));
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xF, 0xF, true));
        if (q == 15) sred[g4 * 16 + j * 2 + w] = v;
      }
    }
    stream_stat_flush(sred, lane, n0, p);
  }
}

// Takes the 1x1, stride-1, bf16 -> bf16 convs with K = 512 / 1024 and Cout a multiple of 128 in the two modes the step's
// long-K layers use (tuning key conv.kstream = 1). Returns false when the shape is not its.
inline bool try_launch_kstream1x1(const ConvP& p, hipStream_t s) {
  // Which launches take this kernel: a bit mask. 1 = K 512 data gradients with the fused BatchNorm backward, 2 = the same at
  // K 1024, 4 = K 512 forward with Cout 128, 8 = K 512 forward with wider outputs, 16 = K 1024 forward, 32 / 64 = the mask-bits form of
  // the data gradients at K = 512 / 1024 (MODE 6; 512 -> 2048 at 16 x 26: -0.17 ms per step). Default 35 = 1 + 2 + 32: with warm
  // operands (a micro-benchmark that re-reads one tensor: profiles/r06_kstream_ab.md) every K = 512 case wins, but inside the
  // step, on cold operands, only the fused-BatchNorm-backward launches do (512 -> 128: 54.1 -> 34.6 us, 1024 -> 256: 33.2 ->
  // 29.8); the forward launches are level (512 -> 128: 39.3 -> 39.9) or lose (512 -> 256: 54.9 -> 60.9, 512 -> 512: 85 -> 113:
  // four column blocks re-read the pixel tiles) — profiles/r06_train_shapes_kstream_all_k512.txt.
  const long long mask = dastune::get(dastune::CONV_KSTREAM);
  if (mask <= 0) return false;
  const long long min_rows = dastune::get(dastune::CONV_STREAM_MINROWS);
  if (min_rows <= 0 || p.KH != 1 || p.KW != 1 || p.stride != 1 || p.pad != 0 || p.up_sh != 0 || p.relu_in || p.relu ||
      p.xbytes == 0 || p.osub || p.scale || p.shift || p.ksplit > 1)
    return false;
  if ((p.Cin != 512 && p.Cin != 1024) || p.Cout % 128 || p.yps % 8 || p.xps % 8) return false;
  const bool bnb = p.bnb_raw != nullptr;
  // the mask-bits form (`rbm` launches: small-M, wide expand convs' data gradients): bit 32 / 64 of the mask at K = 512 / 1024
  const bool bitsm = bnb && p.bnb_relu && !p.bnb_y && p.bnb_bits && p.stats && p.bnb_ps % 8 == 0 && (!p.res || p.rps % 8 == 0);
  if (bitsm) {
    if (p.M < 4096 || !(mask & (p.Cin == 512 ? 32 : 64))) return false;
  } else if (p.res && !bnb) {   // plain output + residual (MODE 2): bit 128 / 256 at K = 512 / 1024
    if (p.M < min_rows || p.stats || p.rps % 8 || !(mask & (p.Cin == 512 ? 128 : 256))) return false;
  } else {
    if (p.M < min_rows || p.res) return false;
    if (bnb && !(p.bnb_relu && !p.bnb_y && !p.bnb_bits && p.stats && p.bnb_ps % 8 == 0)) return false;
    if (!(mask & (bnb ? (p.Cin == 512 ? 1 : 2) : p.Cin == 512 ? (p.Cout == 128 ? 4 : 8) : 16))) return false;
  }
  const int pbk = p.Cin == 512 ? 2 : 1, tm = 16 * pbk;      // pixel blocks / rows per tile (conv1x1_kstream_kernel: PB, TM)
  const int ncol = p.Cout / 128, ntiles = (p.M + tm - 1) / tm;
  const size_t sm = (size_t)4 * tm * p.Cin * 2 + 2 * 4 * 64 * 32 * pbk;   // stages + two sets of partial sums (K = 512: exactly 160 KiB)
  auto go = [&](auto kern) -> bool {
    static bool attr = false;
    if (!attr) {
      if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm) != hipSuccess) return false;
      attr = true;
    }
    long long grid = std::min<long long>(usable_cus(), (long long)ntiles * ncol);
    grid = std::max<long long>(ncol, grid / ncol * ncol);
    dastune::note_kernel("conv1x1_kstream_kernel");
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), sm, s, p, ncol, ntiles);
    return true;
  };
  const bool resm = p.res && !bnb;
  if (p.Cin == 512)
    return bitsm ? go(conv1x1_kstream_kernel<16, 6>) : resm ? go(conv1x1_kstream_kernel<16, 2>) : bnb ? go(conv1x1_kstream_kernel<16, 4>) : go(conv1x1_kstream_kernel<16, 0>);
  return bitsm ? go(conv1x1_kstream_kernel<32, 6>) : resm ? go(conv1x1_kstream_kernel<32, 2>) : bnb ? go(conv1x1_kstream_kernel<32, 4>) : go(conv1x1_kstream_kernel<32, 0>);
}

// Takes the 1x1, stride-1, bf16 -> bf16 convs with K in {64, 128, 256}, Cout 64 / 128 / a multiple of 256 and enough
// rows to keep a persistent grid busy. Returns false when the shape is not its.
inline bool try_launch_stream1x1(const ConvP& p, hipStream_t s) {
  const long long min_rows = dastune::get(dastune::CONV_STREAM_MINROWS);   // 0 disables
  if (min_rows <= 0 || p.KH != 1 || p.KW != 1 || p.stride != 1 || p.pad != 0 || p.up_sh != 0 || p.relu_in ||
      p.xbytes == 0 || p.M < min_rows || p.osub)
    return false;
  if (p.Cin != 64 && p.Cin != 128 && p.Cin != 256) return false;
  if (p.Cout != 64 && p.Cout != 128 && p.Cout % 256 != 0) return false;
  if (p.yps % 8 || (p.res && p.rps % 8)) return false;
  const int wn = p.Cout >= 256 ? 8 : p.Cout / 32, ncol = p.Cout >= 256 ? p.Cout / 256 : 1;
  const int ntiles = (p.M + 63) / 64;
  const int kb = p.Cin / 32;
  const size_t sm = (size_t)(p.Cin == 64 ? DAS_STREAM_K64_NS : (p.Cin == 128 ? DAS_STREAM_K128_NS : 4)) * ((p.Cin + 63) / 64) * 64 * 128 + 8 * 64 * sizeof(float);   // the stages of 64 pixel rows + the waves' reduced sums
  auto go = [&](auto kern) -> bool {
    static int per_cu = 0;   // (one static per template instance: the lambda is instantiated per kernel type)
    if (per_cu == 0) {
      if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm) != hipSuccess)
        return false;
      int n = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, 640, sm) != hipSuccess || n < 1) return false;
      per_cu = n;
    }
    const int cap = (int)dastune::get(dastune::CONV_STREAM_PERCU);
    const int cus = usable_cus();
    long long grid = (long long)(per_cu > cap && cap > 0 ? cap : per_cu) * cus;
    grid = std::min<long long>(grid, (long long)ntiles * ncol);
    grid = std::max<long long>(ncol, grid / ncol * ncol);
    dastune::note_kernel("conv1x1_stream_kernel");
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(640), sm, s, p, ncol, ntiles, (int)dastune::get(dastune::CONV_STREAM_NT));
    return true;
  };
  const bool aff = p.scale || p.shift;
  const bool bnb = p.bnb_raw != nullptr;
  if (!bnb && p.stats && (aff || p.res)) return false;   // (combinations no caller on the path uses)
  if (bnb && (p.relu || aff)) return false;
  const int mode = bnb ? ((p.bnb_relu && !p.bnb_y && !p.bnb_bits) ? 4 : (p.bnb_bits ? 6 : 3)) : p.res ? (aff ? 5 : 2) : (aff ? 1 : 0);
  if (mode == 4 && p.res) return false;   // (the recomputed mask is only valid when no residual entered before the ReLU)
#define DAS_STREAM_CASE(KBV, WNV)                                         \
  if (kb == KBV && wn == WNV) {                                           \
    if (mode == 0) return go(conv1x1_stream_kernel<KBV, WNV, 0>);         \
    if (mode == 1) return go(conv1x1_stream_kernel<KBV, WNV, 1>);         \
    if (mode == 2) return go(conv1x1_stream_kernel<KBV, WNV, 2>);         \
    if (mode == 3) return go(conv1x1_stream_kernel<KBV, WNV, 3>);         \
    if (mode == 5) return go(conv1x1_stream_kernel<KBV, WNV, 5>);         \
    if (mode == 6) return go(conv1x1_stream_kernel<KBV, WNV, 6>);         \
    if constexpr (WNV == 2) return go(conv1x1_stream_kernel<KBV, WNV, 4>); \
    return false;   /* (mode 4 with wide outputs would spill: the tile kernels take those) */ \
  }
  DAS_STREAM_CASE(2, 2) DAS_STREAM_CASE(2, 4) DAS_STREAM_CASE(2, 8)
  DAS_STREAM_CASE(4, 2) DAS_STREAM_CASE(4, 4) DAS_STREAM_CASE(4, 8)
  DAS_STREAM_CASE(8, 2) DAS_STREAM_CASE(8, 4) DAS_STREAM_CASE(8, 8)
#undef DAS_STREAM_CASE
  return false;
}

// 256 x 256 tile kernel: bf16 in / bf16 out, Cout >= 256, Cin % 32 == 0, enough tiles to cover the chip
template <typename T, typename OT>
bool try_launch4(const ConvP& p0, bool glds, bool aligned, hipStream_t s) {
  if constexpr (sizeof(T) == 2 && sizeof(OT) == 2) {
    const long long minblocks = dastune::get(dastune::CONV_GLDS4_MINBLOCKS);  // default: half a chip of 256 x 256 tiles (measured break-even)
    ConvP p = p0;
    p.ntiles = (p.Cout + 255) / 256;
    const long long mtiles = (p.M + 255) / 256, nb = mtiles * p.ntiles;
    const int ks4 = pick_ksplit(nb, p.K / 64, 1, 4, (long long)p.M * p.Cout);
    if (minblocks <= 0 || p.Cout < 256 || p.Cin % 32 || p.relu_in || p.up_sh != 0 || p.xbytes == 0 || nb * ks4 < minblocks)
      return false;
    // ping-pong schedule for the MFMA-bound shapes (+12...15 % at K >= 512, neutral at 256, a loss for the
    // HBM-bound K = 64 / 128 layers that finish in two or four steps)
    const long long force_pp = dastune::get(dastune::CONV_GLDS4_PP);
    const bool pp = force_pp >= 0 ? force_pp == 1 : p.K >= 256;
    const size_t sm4 = std::max<size_t>(4 * (size_t)(256 + 256) * 64, epilogue_smem_bytes<OT, 256, 256>());
    const size_t sm4x = std::max<size_t>(4 * (size_t)(288 + 256) * 64 + 1024, epilogue_smem_bytes<OT, 256, 288>());
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute((const void*)conv_glds4_kernel<T, OT, false, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm4);
      (void)hipFuncSetAttribute((const void*)conv_glds4_kernel<T, OT, true, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm4);
      (void)hipFuncSetAttribute((const void*)conv_glds4_kernel<T, OT, true, 288>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm4x);
      (void)hipFuncSetAttribute((const void*)conv_glds4_kernel<T, OT, false, 256, 16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm4);
      (void)hipFuncSetAttribute((const void*)conv_glds4_kernel<T, OT, true, 256, 16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm4);
      (void)hipFuncSetAttribute((const void*)conv_glds4_kernel<T, OT, true, 288, 16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm4x);
      (void)hipFuncSetAttribute((const void*)conv_glds4_kernel<T, OT, false, 256, 16, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm4);
      (void)hipFuncSetAttribute((const void*)conv_glds4_kernel<T, OT, true, 256, 16, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm4);
      (void)hipFuncSetAttribute((const void*)conv_glds4_kernel<T, OT, true, 288, 16, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm4x);
      attr_set = true;
    }
    if (ks4 > 1) {
      p.nblocks = (int)nb;
      dastune::note_kernel("conv_glds4_kernel<splitk>");
      const int rc = pp ? launch_splitk<OT, 256, 256>(conv_glds4_kernel<T, OT, true, 256>, p, ks4, sm4, s)
                        : launch_splitk<OT, 256, 256>(conv_glds4_kernel<T, OT, false, 256>, p, ks4, sm4, s);
      return rc == DAS_OK;
    }
    // One workgroup per CU: the launch runs in rounds of `cus` tiles. Three ways to cover M (cost in 32-row slabs per
    // CU): 256-row tiles, all rounds (8 each); 256-row tiles with the under-filled last round handed to a split-K tail
    // launch (measured ~4 on top of the full rounds); 288-row tiles (9 per round, fewer tiles).
    const long long cus = device_cus(), keep = tail_split_mtiles(mtiles, p.ntiles, true);
    const long long nb288 = (long long)((p.M + 287) / 288) * p.ntiles;
    const long long cost256 = keep < mtiles ? (nb / cus) * 8 + 4 : ((nb + cus - 1) / cus) * 8;
    const long long cost288 = ((nb288 + cus - 1) / cus) * 9;
    const long long mf = dastune::get(dastune::CONV_GLDS4_MF);   // 0 = by cost, 8 / 9 force the tile height
    if (pp && (mf == 9 || (mf == 0 && cost288 < cost256))) {
      p.nblocks = (int)nb288;
      t_last_tile_rows = 288;
      dastune::note_kernel("conv_glds4_kernel<pp,288>");
      if (p.bnb_raw) hipLaunchKernelGGL((conv_glds4_kernel<T, OT, true, 288>), dim3(p.nblocks), dim3(512), sm4x, s, p);
      else if (epilogue_plain(p)) hipLaunchKernelGGL((conv_glds4_kernel<T, OT, true, 288, 16, false, true>), dim3(p.nblocks), dim3(512), sm4x, s, p);
      else hipLaunchKernelGGL((conv_glds4_kernel<T, OT, true, 288, 16, false>), dim3(p.nblocks), dim3(512), sm4x, s, p);
      return true;
    }
    int mt = (int)keep;
    if (keep < mtiles) {   // rows of the under-filled last round: second launch (128-row tiles, split-K)
      ConvP tail = p0;
      tail.m_base = (int)keep * 256;
      if (launch<T, OT, 128>(tail, glds, aligned, s, false) != DAS_OK) return false;
      p.M = tail.m_base;
    } else {
      p.mstep = balanced_mstep(p.M, 256, p.ntiles, p.KH == 1 && p.KW == 1, &mt);
    }
    p.nblocks = mt * p.ntiles;
    t_last_tile_rows = p.mstep ? p.mstep : 256;
    dastune::note_kernel(pp ? "conv_glds4_kernel<pp>" : "conv_glds4_kernel");
    if constexpr (sizeof(T) == 2) {
      if (pp && dastune::get(dastune::CONV_GLDS4_MFMA32) == 1) {
        static bool attr32 = false;
        if (!attr32) {
          (void)hipFuncSetAttribute((const void*)conv_glds4_kernel<T, OT, true, 256, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm4);
          attr32 = true;
        }
        dastune::note_kernel("conv_glds4_kernel<pp,mf32>");
        hipLaunchKernelGGL((conv_glds4_kernel<T, OT, true, 256, 32>), dim3(p.nblocks), dim3(512), sm4, s, p);
        return true;
      }
    }
    if (pp) {
      if (p.bnb_raw) hipLaunchKernelGGL((conv_glds4_kernel<T, OT, true, 256>), dim3(p.nblocks), dim3(512), sm4, s, p);
      else if (epilogue_plain(p)) hipLaunchKernelGGL((conv_glds4_kernel<T, OT, true, 256, 16, false, true>), dim3(p.nblocks), dim3(512), sm4, s, p);
      else hipLaunchKernelGGL((conv_glds4_kernel<T, OT, true, 256, 16, false>), dim3(p.nblocks), dim3(512), sm4, s, p);
    } else {
      if (p.bnb_raw) hipLaunchKernelGGL((conv_glds4_kernel<T, OT, false, 256>), dim3(p.nblocks), dim3(512), sm4, s, p);
      else if (epilogue_plain(p)) hipLaunchKernelGGL((conv_glds4_kernel<T, OT, false, 256, 16, false, true>), dim3(p.nblocks), dim3(512), sm4, s, p);
      else hipLaunchKernelGGL((conv_glds4_kernel<T, OT, false, 256, 16, false>), dim3(p.nblocks), dim3(512), sm4, s, p);
    }
    return true;
  } else {
    return false;
  }
}

// Takes the 3x3, stride-1, "same"-padded bf16 convs with 64 input and 64 output channels on plain NHWC tensors (the
// 128 x 208 stage). Returns false when the shape is not its.
template <typename OT>
bool try_launch_c64(const ConvP& p, hipStream_t s) {
  const long long min_tiles = dastune::get(dastune::CONV_C64_MINTILES);   // 0 disables
  if (min_tiles <= 0 || p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad != 1 || p.Cin != 64 || p.Cout != 64 ||
      p.up_sh != 0 || p.relu_in || p.osub || p.nlev > 1 || p.m_base != 0 || p.xbytes == 0 || p.xps % 8 || p.yps % 8 ||
      (p.res && p.rps % 8) || (p.bnb_raw && p.bnb_ps % 8) || p.Ho != p.H || p.Wo != p.W || p.M % (p.H * p.W) || p.K != 576)
    return false;
  // 16 x 16-pixel squares per image; border squares of an image whose height / width is not a multiple of 16 are partly
  // outside (their MFMA work is wasted: not worth it when that is more than a quarter of the launch)
  const long long sq = (long long)((p.H + 15) / 16) * ((p.W + 15) / 16);
  if (sq * 256 * 3 > (long long)p.H * p.W * 4) return false;
  const int ntiles = (int)(sq * (p.M / (p.H * p.W)));
  if (ntiles < min_tiles) return false;
  const size_t sm = C64_WBYTES + 2 * (size_t)C64_PATCH;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)conv3x3_c64_kernel<OT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm) != hipSuccess ||
        hipFuncSetAttribute((const void*)conv3x3_c64_kernel<OT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm) != hipSuccess ||
        hipFuncSetAttribute((const void*)conv3x3_c64_kernel<OT, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm) != hipSuccess ||
        hipFuncSetAttribute((const void*)conv3x3_c64_kernel<OT, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm) != hipSuccess)
      return false;
    attr_set = true;
  }
  const int grid = std::min(ntiles, usable_cus());
  dastune::note_kernel("conv3x3_c64_kernel");
  if (p.bnb_bits || p.res_bits) {
    hipLaunchKernelGGL((conv3x3_c64_kernel<OT, true>), dim3(grid), dim3(384), sm, s, p, ntiles);
  } else if (p.bnb_raw) {   // (the step's launches: conv2 of a bottleneck never sees a mask as bits)
    hipLaunchKernelGGL((conv3x3_c64_kernel<OT, false>), dim3(grid), dim3(384), sm, s, p, ntiles);
  } else if (epilogue_plain(p)) {
    hipLaunchKernelGGL((conv3x3_c64_kernel<OT, false, false, true>), dim3(grid), dim3(384), sm, s, p, ntiles);
  } else {
    hipLaunchKernelGGL((conv3x3_c64_kernel<OT, false, false>), dim3(grid), dim3(384), sm, s, p, ntiles);
  }
  return true;
}

template <typename T, typename OT>
int launch_bn(const ConvP& p, bool glds, bool aligned, hipStream_t s) {
  if constexpr (sizeof(T) == 2 && sizeof(OT) == 2) {
    if (try_launch_stream1x1(p, s) || try_launch_kstream1x1(p, s)) {
      DAS_CHECK_LAUNCH();
      return DAS_OK;
    }
    if (try_launch_c64<OT>(p, s) || try_launch_stem7x7(p, s)) {
      DAS_CHECK_LAUNCH();
      return DAS_OK;
    }
  }
  if (try_launch4<T, OT>(p, glds, aligned, s)) {
    DAS_CHECK_LAUNCH();
    return DAS_OK;
  }
  if (p.Cout > 64) return launch<T, OT, 128>(p, glds, aligned, s);
  if (p.Cout > 32) return launch<T, OT, 64>(p, glds, aligned, s);
  return launch<T, OT, 32>(p, glds, aligned, s);
}

}  // namespace

extern "C" int das_conv_last_tile_rows(void) { return t_last_tile_rows; }

extern "C" int das_conv2d_nhwc(const void* x, const void* w, void* y, const DasConvDesc* d, void* stream) {
  DAS_PROF(stream);
  t_last_tile_rows = 0;
  if (!x || !w || !y || !d) return DAS_ERR_ARG;
  if (d->Cin % 8 || d->Cout % 8 || d->x_pix_stride % 8 || d->y_pix_stride % 8) return DAS_ERR_ARG;
  if (d->x_pix_stride < d->Cin || d->y_pix_stride < d->Cout) return DAS_ERR_ARG;
  if (d->residual && (d->out_dtype != d->dtype || d->res_pix_stride % 8)) return DAS_ERR_ARG;
  if (d->dtype == DAS_F32 && d->out_dtype != DAS_F32) return DAS_ERR_ARG;
  if (d->KH < 1 || d->KW < 1 || d->stride < 1 || d->B < 1) return DAS_ERR_ARG;
  ConvP p;
  long long M = (long long)d->B * d->Ho * d->Wo;
  p.nlev = d->num_levels;
  p.B = d->B;
  if (p.nlev > 1) {
    // ragged multi-level rows: stride 1, "same" padding, output geometry == input geometry
    if (p.nlev > MAXLV || d->stride != 1 || d->KH != d->KW || d->pad != d->KH / 2) return DAS_ERR_ARG;
    M = 0;
    for (int l = 0; l < p.nlev; ++l) {
      if (d->lvl_H[l] < 1 || d->lvl_W[l] < 1) return DAS_ERR_ARG;
      p.lvH[l] = d->lvl_H[l]; p.lvW[l] = d->lvl_W[l]; p.lvStart[l] = (int)M;
      M += (long long)d->B * d->lvl_H[l] * d->lvl_W[l];
    }
  }
  for (int l = (p.nlev > 1 ? p.nlev : 0); l < MAXLV; ++l) { p.lvH[l] = 0; p.lvW[l] = 0; p.lvStart[l] = 0x7fffffff; }
  if (M <= 0 || M > 0x7fffffffLL) return DAS_ERR_ARG;
  p.x = (const char*)x; p.w = (const char*)w; p.y = (char*)y;
  p.scale = d->scale; p.shift = d->shift; p.res = (const char*)d->residual; p.stats = d->stats;
  p.stat_slots = d->stats_slots < 1 ? 1 : d->stats_slots;
  p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.xps = d->x_pix_stride;
  p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout; p.yps = d->y_pix_stride;
  p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad;
  p.relu_in = d->relu_in; p.relu = d->relu; p.rps = d->res_pix_stride;
  if (d->in_up != 0 && d->in_up != 1 && d->in_up != 2) return DAS_ERR_ARG;
  if (d->in_up == 2 && (d->stride != 1 || p.nlev > 1)) return DAS_ERR_ARG;
  p.up_sh = d->in_up == 2 ? 1 : 0;
  p.M = (int)M; p.K = d->KH * d->KW * d->Cin; p.HoWo = d->Ho * d->Wo;
  p.ntiles = p.nblocks = 0; p.m_base = 0; p.mstep = 0;
  p.ws = nullptr; p.ksplit = 1; p.sk_cnt = nullptr;
  p.osub = d->out_sub ? 1 : 0; p.oph = d->out_ph; p.opw = d->out_pw; p.oH = d->out_H; p.oW = d->out_W;
  if (p.osub) {   // sub-grid output: see DasConvDesc
    if (p.nlev > 1 || d->stride != 1 || (unsigned)p.oph > 1u || (unsigned)p.opw > 1u || d->Ho < 1 || d->Wo < 1 ||
        2 * (d->Ho - 1) + p.oph >= p.oH || 2 * (d->Wo - 1) + p.opw >= p.oW)
      return DAS_ERR_ARG;
  }
  p.bnb_raw = (const char*)d->bnb_raw; p.bnb_y = (const char*)d->bnb_y;
  p.bnb_bits = p.bnb_y ? nullptr : (const unsigned char*)d->bnb_mask_bits;
  p.res_bits = (p.bnb_raw && d->residual) ? (const unsigned char*)d->residual_mask_bits : nullptr;
  if (d->residual_mask_bits && !p.res_bits) return DAS_ERR_ARG;   // (only the fused BatchNorm-backward launches mask their residual)
  p.bnb_mean = d->bnb_mean; p.bnb_invstd = d->bnb_invstd; p.bnb_gamma = d->bnb_gamma; p.bnb_beta = d->bnb_beta;
  p.bnb_relu = d->bnb_relu; p.bnb_ps = d->bnb_pix_stride;
  if (p.bnb_raw) {   // fused BatchNorm-backward reduction: see DasConvDesc
    if (!d->stats || d->out_dtype != d->dtype || d->relu || d->scale || d->shift || !p.bnb_mean || !p.bnb_invstd ||
        p.bnb_ps < d->Cout || p.bnb_ps % 8)
      return DAS_ERR_ARG;
    if (p.bnb_relu && !p.bnb_y && !p.bnb_bits && (!p.bnb_gamma || !p.bnb_beta)) return DAS_ERR_ARG;
  }
  {
    const long long npix = p.nlev > 1 ? M : (long long)d->B * d->H * d->W;
    const long long xb = ((npix - 1) * d->x_pix_stride + d->Cin) * (d->dtype == DAS_BF16 ? 2 : 4);
    p.xbytes = xb < 0xFFFFFFF0LL ? (unsigned)xb : 0u;
  }
  hipStream_t s = (hipStream_t)stream;
  if (d->dtype == DAS_BF16) {
    const bool glds = (d->Cin % 64) == 0 && !d->relu_in, aligned = (d->Cin % 32) == 0;
    if (d->out_dtype == DAS_BF16) return launch_bn<bf16_t, bf16_t>(p, glds, aligned, s);
    return launch_bn<bf16_t, float>(p, glds, aligned, s);
  }
  if (d->dtype == DAS_F32) return launch_bn<float, float>(p, (d->Cin % 32) == 0 && !d->relu_in, (d->Cin % 16) == 0, s);
  return DAS_ERR_ARG;
}
