// DCNv2 3x3 (stride 1, pad 1, dilation 1) forward as ONE kernel (round 5): the deformable sampling writes the GEMM's
// pixel operand straight into LDS instead of a `col` tensor in HBM.
//
//   y[m, o] = sum_{k, c} W[o, k, c] * mask[m, k] * bilinear(x, p_m + tap_k + offset[m, k])[c] + bias[o]
//
// Kernel by kernel (das_deform_im2col3x3 + das_conv2d_nhwc on col) the sampling pass is bound by writing col (9 C values
// per pixel: 652 MB per head layer at B = 16) and the GEMM reads it back. Here a 128-pixel x 256-channel tile runs the GEMM's
// K loop over (tap, 64-channel block) steps; per step every thread interpolates two 16-byte vectors of the pixel operand
// from their four corner vectors (requested TWO steps ahead into alternating register sets by hand-issued loads) and stores them
// into the LDS stage the MFMA fragments are read from; the weight tile is DMA'd global -> LDS (inline asm: the compiler
// does not wait for it). The tap geometry (sigmoid, floor, fractions, corner validity, corner row) is computed once per
// (pixel, tap) into an LDS table when the tile opens. Arithmetic of the sampled values = deform_im2col_wave_kernel's,
// operation for operation (`col`, when requested as a side output for the backward pass, is bit-identical).
// mmcv ModulatedDeformConv2dPack.forward -> modulated_deform_conv2d: das_head.py:107-108,
// anchor_free_mono3d_pose_head.py:111-112,131-132, recursive_update.py:177-178.
#include <algorithm>
#include <cstring>
#include <type_traits>

#include "conv_common.h"
#include "prof.h"
#include "tuning.h"

using namespace dasconv;

namespace {
__device__ uint4 g_dcn_zero_page[8];   // 128 B of zeros: source of the weight rows past Cout

__device__ __forceinline__ int slot128(int row, int kg) { return row * 128 + ((kg ^ ((row >> 1) & 7)) << 4); }

// LDS-DMA from inline asm (conv_igemm.hip: dma16_asm): not tracked by the compiler, the caller waits with vmcnt.
__device__ __forceinline__ void dcn_dma16(const void* src, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(src), "s"(lds_byte_addr)
               : "memory");
}

struct DcnP {
  const bf16_t* x;
  const float* om;
  const bf16_t* w;     // (Cout, 9 C): K = tap * C + channel
  bf16_t* col;         // (rows, 9 C) side output or nullptr
  DasLevels lv;
  int C, xps, omps, Cout;
  long long rows;
  int ntiles;
};

struct Geo {           // one (pixel, tap) of the tile
  int row;             // row of the top-left corner (rows of x)
  int self;            // the pixel's own row: a valid address for corners outside the plane (weight 0)
  int W;
  int ok;              // bit c: corner c inside the plane; bit 4: the sample is live (inside (-1, H) x (-1, W))
  float ly, lx, mk;
  int pad_;
};

constexpr int DBM = 128, DBN = 256, DNT = 512;
constexpr int DA_BYTES = DBM * 128, DW_BYTES = DBN * 128, DBUF = DA_BYTES + DW_BYTES;   // one stage: pixels + weights
constexpr int DGEO_BYTES = DBM * 9 * (int)sizeof(Geo);
constexpr int DCN_SMEM = 2 * DBUF + DGEO_BYTES;

template <bool COL>
__global__ __launch_bounds__(DNT) void dcn3x3_fused_kernel(ConvP p, DcnP d) {
#pragma clang fp contract(off)
  using T = bf16_t;
  using TL = Tiling<DBN, DBM>;
  constexpr int TM = TL::TM, TN = TL::TN;
  constexpr int W_INSTR = (DBN / 8) / (DNT / 64);   // 1 KiB (8 rows) per wave instruction
  extern __shared__ __attribute__((aligned(16))) char smem[];
  Geo* geo = reinterpret_cast<Geo*>(smem + 2 * DBUF);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = xcd_remap(blockIdx.x, d.ntiles);
  const long long m0 = (long long)tile * DBM;
  const int wave_m0 = TL::wave_m0(wave), wave_n0 = TL::wave_n0(wave);

  // ---- tap geometry of the tile (deform_im2col_wave_kernel's producer role, one (pixel, tap) per thread and round)
  for (int i = tid; i < DBM * 9; i += DNT) {
    const int pl = i / 9, k = i - pl * 9;
    const bool valid = m0 + pl < d.rows;
    const long long gm = valid ? m0 + pl : d.rows - 1;
    const LvGeom g = lv_geom(d.lv, gm);
    const float* o = d.om + gm * d.omps;
    const float dy = o[2 * k], dx = o[2 * k + 1];
    Geo e;
    e.mk = 1.f / (1.f + expf(-o[18 + k]));
    const float py = (float)(g.h - 1 + k / 3) + dy;
    const float px = (float)(g.w - 1 + k % 3) + dx;
    const bool live = py > -1.f && px > -1.f && py < (float)g.H && px < (float)g.W;
    const float fy = floorf(py), fx = floorf(px);
    const int y0 = live ? (int)fy : 0, x0 = live ? (int)fx : 0;
    e.ly = py - fy; e.lx = px - fx;
    int ok = live ? 16 : 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int yy = y0 + (c >> 1), xx = x0 + (c & 1);
      if (live && yy >= 0 && yy <= g.H - 1 && xx >= 0 && xx <= g.W - 1) ok |= 1 << c;
    }
    e.ok = valid ? ok : 0;                 // (rows past the end: zeros in the operand, never stored)
    e.row = (int)(g.plane0 + (long long)y0 * g.W + x0);   // (row counts < 2^31: checked by the host)
    e.self = (int)gm;
    e.W = g.W;
    e.pad_ = 0;
    geo[i] = e;
  }

  // ---- weight DMA coordinates (conv_glds_kernel's scheme: chunk c covers LDS rows c*8..c*8+7; lane -> (row, physical slot))
  const int lrow = lane >> 3, pslot = lane & 7;
  const T* zero = reinterpret_cast<const T*>(g_dcn_zero_page);
  const T* wrow[W_INSTR];
  const int K = 9 * d.C;
#pragma unroll
  for (int j = 0; j < W_INSTR; ++j) {
    const int row = (wave * W_INSTR + j) * 8 + lrow;
    wrow[j] = row < d.Cout ? d.w + (long long)row * K + (pslot ^ ((row >> 1) & 7)) * 8 : nullptr;
  }
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  auto issue_w = [&](int s, int buf) {
#pragma unroll
    for (int j = 0; j < W_INSTR; ++j)
      dcn_dma16(wrow[j] ? wrow[j] + (long long)s * 64 : zero, lds0 + buf * DBUF + DA_BYTES + (wave * W_INSTR + j) * 1024);
  };

  // ---- the pixel operand: thread -> vectors (pixel pl0 / pl0 + 64, 16-byte slot v of the step's 64 channels).
  // The corner vectors of step s + 2 are requested at the top of step s (hand-issued loads into one of two register sets) and
  // interpolated at the bottom of step s + 1: two K steps of MFMAs between request and use. Every wait is counted by hand
  // (the LDS-DMA of the weights is invisible to the compiler anyway): loads return in order among themselves, stores (the col
  // side output) in any order — the counts below hold with or without them, see the comments at the waits.
  const int pl0 = tid >> 3, v = tid & 7;
  const int CB = d.C >> 6;                      // 64-channel blocks per tap
  const int nk = 9 * CB;
  v4i_t crA[2][4], crB[2][4];
  // (32-bit byte offsets from a scalar base: the host checks that x spans less than 4 GiB; 64-bit address arithmetic for
  // eight loads per step was a third of this kernel's vector instructions)
  const unsigned xrow2 = (unsigned)d.xps * 2u, vb2 = (unsigned)v * 16u;
  auto gather_load = [&](int s, v4i_t (&cr)[2][4]) {
    const int tap = s / CB, cb = s - tap * CB;
    const unsigned cbv = (unsigned)cb * 128u + vb2;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const Geo e = geo[(pl0 + it * 64) * 9 + tap];
#pragma unroll
      for (int c = 0; c < 4; ++c) {   // (a corner outside the plane re-reads the pixel's own row: weight 0)
        const int pix = (e.ok >> c & 1) ? e.row + (c >> 1) * e.W + (c & 1) : e.self;
        const unsigned off = (unsigned)pix * xrow2 + cbv;
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(cr[it][c]) : "v"(off), "s"(d.x) : "memory");
      }
    }
  };
  auto gather_write = [&](int s, int buf, v4i_t (&cr)[2][4]) {
#pragma clang fp contract(off)
    const int tap = s / CB, cb = s - tap * CB;
    char* sA = smem + buf * DBUF;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int pl = pl0 + it * 64;
      const Geo e = geo[pl * 9 + tap];
      const float hy = 1.f - e.ly, hx = 1.f - e.lx;
      const float wts[4] = {hy * hx, hy * e.lx, e.ly * hx, e.ly * e.lx};
      // two-wide vector arithmetic (v_pk_mul_f32 / v_pk_add_f32): the same products and sums, in the same order and with the
      // same roundings as deform_im2col_wave_kernel's scalar loop (contraction stays off), in half the instructions
      das_f32x2_t o2[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) o2[j] = das_f32x2_t{0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        // (branch-free: a corner outside the plane enters with weight 0 — its stand-in row is finite, and adding +-0 to the
        // running sum leaves it bit for bit what skipping the corner leaves)
        const float wc = (e.ok >> c & 1) ? wts[c] : 0.f;
        const das_f32x2_t w2 = {wc, wc};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          // (a dword's two bf16 values become a register PAIR directly: built element by element the compiler spent a third
          // of the step's instructions on moves that put the halves side by side)
          const unsigned xw = (unsigned)cr[it][c][j];
          typedef unsigned das_u32x2_t __attribute__((ext_vector_type(2)));
          const das_u32x2_t u2 = {xw << 16, xw & 0xffff0000u};
          const das_f32x2_t f2 = __builtin_bit_cast(das_f32x2_t, u2);
          const das_f32x2_t pr = w2 * f2;
          o2[j] = o2[j] + pr;
        }
      }
      float out[8];
      const float mke = (e.ok & 16) ? e.mk : 0.f;    // (a dead sample has no corner bits either: its sum is +0)
      const das_f32x2_t m2 = {mke, mke};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const das_f32x2_t r = o2[j] * m2;
        out[2 * j] = r[0];
        out[2 * j + 1] = r[1];
      }
      const uint4 packed = Elem<T>::pack(out);
      *reinterpret_cast<uint4*>(sA + slot128(pl, v)) = packed;
      if (COL) {
        const long long m = m0 + pl;
        if (m < d.rows) *reinterpret_cast<uint4*>(d.col + (m * 9 + tap) * d.C + cb * 64 + v * 8) = packed;
      }
    }
  };
  // all of a register set's loads have landed once at most `N` vector-memory operations are outstanding (the operands name
  // the registers: they stay allocated, and their uses are ordered after the wait)
#define DCN_WAIT(N, cr)                                                                                               \
  asm volatile("s_waitcnt vmcnt(" #N ")"                                                                              \
               : "+v"(cr[0][0]), "+v"(cr[0][1]), "+v"(cr[0][2]), "+v"(cr[0][3]), "+v"(cr[1][0]), "+v"(cr[1][1]),       \
                 "+v"(cr[1][2]), "+v"(cr[1][3])::"memory")

  f32x4_t acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  issue_w(0, 0);
  __syncthreads();          // the geometry table is complete
  gather_load(0, crA);
  DCN_WAIT(0, crA);         // (also the weight tile of step 0)
  gather_write(0, 0, crA);
  if (nk > 1) gather_load(1, crB);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  const int frow = lane & 15, fkg = lane >> 4;
  // one K step; `ld` = the register set the loads of step s + 2 go to (it held step s), `use` = the set holding step s + 1
  // (ONE variant of the step, the same instruction counts in every step: with a run-time choice between two waits the compiler
  // copied the whole register set behind each of them, with several instantiations it spilled 120 registers. The last steps
  // therefore re-request step nk - 1's operands — valid addresses, results unused — instead of requesting nothing.)
  auto step = [&](int s, v4i_t (&ld)[2][4], v4i_t (&use)[2][4]) {
    const int buf = s & 1;
    issue_w(min(s + 1, nk - 1), buf ^ 1);     // 4 DMA instructions per wave
    gather_load(min(s + 2, nk - 1), ld);      // 8 loads
    const char* sA = smem + buf * DBUF;
    const char* sW = sA + DA_BYTES;
    auto multiply = [&]() {
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
    };
    multiply();
    // the 8 loads of step s + 1 (issued one step ago) are older than the 4 DMAs + 8 loads issued at the top of this step; col
    // stores may or may not be done. If one of those 8 loads were still outstanding, so would be the 12 younger loads / DMAs
    // (loads return in order): more than 12 outstanding. Hence "<= 12" proves they have landed, whatever the stores do.
    DCN_WAIT(12, use);
    if (s + 1 < nk) gather_write(s + 1, buf ^ 1, use);
    // (Tried and measured no better: the same work as ONE interleaved instruction stream — wait first, then the MFMAs and the
    // sampling arithmetic in one basic block under __builtin_amdgcn_sched_group_barrier(MFMA 1 / VALU 8): 417 us against 399;
    // two code ORDERS for the two waves of a SIMD, so that one multiplies while the other samples: the compiler spilled 100
    // registers, and scratch traffic breaks the hand-counted waits.)
    // next step's weight tile: the 4 DMAs are older than this step's 8 loads — "<= 8 outstanding" proves they have landed
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's operand vectors are in LDS
    __builtin_amdgcn_s_barrier();                        // ... and every wave is done reading stage `buf`
  };
  for (int s = 0; s < nk; s += 2) {     // (s even: set A held step s)
    step(s, crA, crB);
    if (s + 1 < nk) step(s + 1, crB, crA);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing stand-in loads / DMAs: the epilogue reuses the stages
  __builtin_amdgcn_s_barrier();
#undef DCN_WAIT
  // (bias only: the PLAIN variant of the epilogue — no residual, ReLU, sub-grid output or fused-BatchNorm-backward code)
  conv_epilogue<T, DBN, DBM, Tiling<DBN, DBM>, false, false, 16, false, true>(acc, p, smem, (int)m0, 0);
}
}  // namespace

extern "C" int das_dcn3x3_fused(const void* x, const float* om, const void* w, const float* bias, void* y, void* col, int dtype,
                                const DasLevels* lv, int C, int Cout, int x_pix_stride, int om_pix_stride, int y_pix_stride,
                                void* stream) {
  DAS_PROF(stream);
  if (!x || !om || !w || !y || !lv_valid(lv)) return DAS_ERR_ARG;
  if (dtype != DAS_BF16 || C % 64 || C < 64 || Cout % 8 || Cout < 8 || Cout > DBN) return DAS_ERR_ARG;
  if (x_pix_stride % 8 || x_pix_stride < C || om_pix_stride < 27 || y_pix_stride % 8 || y_pix_stride < Cout) return DAS_ERR_ARG;
  const long long rows = lv_total_rows(*lv);
  if (rows < 1 || rows >= 0x7fffff00LL) return DAS_ERR_ARG;
  if ((rows - 1) * (long long)x_pix_stride * 2 + (long long)C * 2 >= 0xFFFFFFF0LL) return DAS_ERR_ARG;   // (32-bit byte offsets into x)
  ConvP p;
  memset(&p, 0, sizeof(p));
  p.y = (char*)y;
  p.shift = bias;
  p.Cout = Cout;
  p.yps = y_pix_stride;
  p.M = (int)rows;
  p.stat_slots = 1;
  DcnP d;
  d.x = (const bf16_t*)x; d.om = om; d.w = (const bf16_t*)w; d.col = (bf16_t*)col; d.lv = *lv;
  d.C = C; d.xps = x_pix_stride; d.omps = om_pix_stride; d.Cout = Cout; d.rows = rows;
  d.ntiles = (int)((rows + DBM - 1) / DBM);
  hipStream_t s = (hipStream_t)stream;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)dcn3x3_fused_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, DCN_SMEM) != hipSuccess ||
        hipFuncSetAttribute((const void*)dcn3x3_fused_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, DCN_SMEM) != hipSuccess)
      return DAS_ERR_LAUNCH;
    attr_set = true;
  }
  dastune::note_kernel("dcn3x3_fused_kernel");
  if (col) {
    hipLaunchKernelGGL(dcn3x3_fused_kernel<true>, dim3((unsigned)d.ntiles), dim3(DNT), DCN_SMEM, s, p, d);
  } else {
    hipLaunchKernelGGL(dcn3x3_fused_kernel<false>, dim3((unsigned)d.ntiles), dim3(DNT), DCN_SMEM, s, p, d);
  }
  DAS_CHECK_LAUNCH();
  return DAS_OK;
}
