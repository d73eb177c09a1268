// The backbone's stem convolution on gfx950: 7 x 7, stride 2, padding 3, three input channels stored as eight (one pixel =
// 16 bytes), 64 output channels, bf16 -> bf16 (reference: ResNet_top, /root/reference/mmdet3d/models/backbones/mspn_mmpose.py:
// 228-246 — Conv 7x7 s2 + BatchNorm + ReLU in front of the max pool). On the generic register-staged tile kernel
// (conv_reg_kernel) it is a K loop of 13 steps whose every step waits for two dependent 16-byte gathers per thread: 356 us at
// B = 16 x 512 x 832 for 328 MB of traffic and 32 GFLOP (profiles/r06_train_shapes_kstream_all_k512.txt).
//
// Here the weights never leave registers and the pixels of a tile are fetched ONCE:
//   * K is ordered (ky, kx, c) with kx padded 7 -> 8: one K step of the 16 x 16 x 32 MFMA is half a kernel row (4 taps x 8
//     channels), 14 steps; a lane's eight K values are the eight channels of ONE input pixel — one 16-byte LDS read, no
//     im2col arithmetic (the eighth tap's weights are zero);
//   * a wave (one per SIMD: 512 registers) holds all 64 x 448 weights as 56 MFMA A fragments (224 registers) and computes 16
//     consecutive output pixels of a row x 64 channels per block: 14 B-fragment reads (all requested up front, the next
//     block's behind the current block's MFMAs), 56 MFMAs, two 16-byte stores per lane (the A rows are permuted so that a
//     lane's sixteen accumulator values are sixteen consecutive channels);
//   * a workgroup (4 waves) walks tiles of 8 output rows x 32 columns of one image: the 21 x 69-pixel input window (padded
//     to 72 columns, 24 KiB) comes in through LDS-DMA, double buffered — the next tile's window flies while this one is
//     computed; pixels outside the image read a zero page;
//   * the BatchNorm statistics of the training forward (sums over the bf16 values as stored) stay in registers for the whole
//     launch: one round of atomics per workgroup.
// MODE 0: plain output + optional statistics (training).  MODE 1: y = relu?(conv * scale + shift) (eval: folded BatchNorm).
#include <algorithm>
#include <type_traits>

#include "conv_common.h"
#include "tuning.h"

using namespace dasconv;

#ifndef DAS_STEM_VAR
#define DAS_STEM_VAR 0   // dev builds (tools/dev/stem_ab.py): 1 no MFMAs, 2 no stores, 4 no statistics, 8 no window DMA after the first
#endif

namespace {

constexpr int TR = 8, TC = 32;                     // output rows x columns per tile
constexpr int WR = 2 * TR + 5, WC = 72;            // input window: 21 rows x 69 columns, padded to 72
constexpr int WSLOTS = 1536;                       // 16-byte slots per window buffer (21 * 72 = 1512, rounded to 24 wave requests)
constexpr int WBYTES = WSLOTS * 16;

template <int MODE>
__global__ __launch_bounds__(256) void conv_stem7x7_kernel(ConvP p, int tiles_x, int tiles_y, int ntiles) {
  using T = bf16_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane & 15, g4 = lane >> 4;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const T* xg = reinterpret_cast<const T*>(p.x);
  const int tpi = tiles_x * tiles_y;               // tiles per image

  // the window of tile t into buffer `buf`: 24 wave requests of 64 slots, six per wave; a pixel outside the image is an offset
  // past the end of the buffer descriptor (zeros)
  const v4i_t xrs = make_rsrc(p.x, p.xbytes);
  constexpr unsigned OOB = 0xFFFFFFF0u;
  auto issue = [&](int t, int buf) {
    const int b = t / tpi, r = t - b * tpi;
    const int ty = r / tiles_x, tx = r - ty * tiles_x;
    const int iy0 = 2 * ty * TR - 3, ix0 = 2 * tx * TC - 3;
    const unsigned img = (unsigned)b * (unsigned)(p.H * p.W);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int s = (wave * 6 + i) * 64 + lane;
      const int wr = s / WC, wc = s - wr * WC;
      const int iy = iy0 + wr, ix = ix0 + wc;
      const bool ok = wr < WR && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      const unsigned off = (img + (unsigned)(iy * p.W + ix)) * 16u;
      dma16_buf(ok ? off : OOB, xrs, lds0 + buf * WBYTES + (wave * 6 + i) * 1024);
    }
  };
  const int first = blockIdx.x, stride = gridDim.x;
  if (first < ntiles) issue(first, 0);

  // weights -> 56 A fragments. A row r of channel block cb is channel (r / 4) * 16 + cb * 4 + (r % 4): the accumulator lane
  // (pixel q, rows g4 * 4 + j) then owns channels g4 * 16 + cb * 4 + j — sixteen consecutive channels over cb, j.
  uint4 wf[4][14];
  {
    const T* wg = reinterpret_cast<const T*>(p.w);
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      const int ch = (q >> 2) * 16 + cb * 4 + (q & 3);
#pragma unroll
      for (int t = 0; t < 14; ++t) {
        const int ky = t >> 1, kx = (t & 1) * 4 + g4;
        wf[cb][t] = kx < 7 ? *reinterpret_cast<const uint4*>(wg + (long long)ch * 392 + (ky * 7 + kx) * 8) : make_uint4(0, 0, 0, 0);
      }
    }
  }
  const int c16 = g4 * 16;                         // this lane's sixteen output channels
  float sc[16], sh[16];
  if constexpr (MODE == 1) {
#pragma unroll
    for (int j = 0; j < 16; ++j) { sc[j] = p.scale ? p.scale[c16 + j] : 1.f; sh[j] = p.shift ? p.shift[c16 + j] : 0.f; }
  }
  float ssum[16], ssq[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) { ssum[j] = 0.f; ssq[j] = 0.f; }
  T* yg = reinterpret_cast<T*>(p.y);

  // B fragments of block (row lr, column half cx) of the tile in window buffer `buf`: K step t = (ky, half) reads the pixel
  // (2 lr + ky, 2 (cx * 16 + q) + half * 4 + g4) of the window. Inline asm: left to the compiler, every pair of reads ends up right
  // in front of the MFMAs that use it, and with ONE wave per SIMD nobody hides that LDS round trip (4 us per tile instead of 1.5);
  // here the fourteen reads of the NEXT block go out before the current block's MFMAs and `landed` is the counted wait.
  auto frags = [&](int buf, int lr, int cx, v4i_t (&fb)[14]) {
    const unsigned base = lds0 + buf * WBYTES + ((2 * lr) * WC + 2 * (cx * 16 + q) + g4) * 16;
#pragma unroll
    for (int t = 0; t < 14; ++t)
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[t]) : "v"(base), "n"(((t >> 1) * WC + (t & 1) * 4) * 16));
  };
  auto landed = [&](v4i_t (&fb)[14], auto younger) {   // all of fb has arrived; at most `younger` later LDS reads are still out
    asm volatile("s_waitcnt lgkmcnt(%14)"
                 : "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]), "+v"(fb[3]), "+v"(fb[4]), "+v"(fb[5]), "+v"(fb[6]), "+v"(fb[7]), "+v"(fb[8]),
                   "+v"(fb[9]), "+v"(fb[10]), "+v"(fb[11]), "+v"(fb[12]), "+v"(fb[13])
                 : "n"(decltype(younger)::value));
  };
  int it = 0;
  for (int t = first; t < ntiles; t += stride, ++it) {
    const int buf = it & 1;
    // this tile's window has landed (and this wave's stores of the tile before are out); everybody has left the other buffer
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t + stride < ntiles && !(DAS_STEM_VAR & 8)) issue(t + stride, buf ^ 1);
    const int b = t / tpi, r = t - b * tpi;
    const int ty = r / tiles_x, tx = r - ty * tiles_x;
    v4i_t fb[2][14];
    frags(buf, wave * 2, 0, fb[0]);
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {            // this wave's blocks: rows wave * 2 + (blk >> 1), column halves blk & 1
      const int lr = wave * 2 + (blk >> 1), cx = blk & 1;
      if (blk + 1 < 4) {
        frags(buf, wave * 2 + ((blk + 1) >> 1), (blk + 1) & 1, fb[(blk + 1) & 1]);
        landed(fb[blk & 1], std::integral_constant<int, 14>{});
      } else {
        landed(fb[blk & 1], std::integral_constant<int, 0>{});
      }
      f32x4_t acc[4];
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) acc[cb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < ((DAS_STEM_VAR & 1) ? 1 : 14); ++k) {
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
          acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[cb][k]),
                                                            __builtin_bit_cast(bf16x8_t, fb[blk & 1][k]), acc[cb], 0, 0, 0);
      }
      const int oy = ty * TR + lr, ox = tx * TC + cx * 16 + q;
      const bool ok = oy < p.Ho && ox < p.Wo;
      float v[16];
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if constexpr (MODE == 1) v[cb * 4 + j] = acc[cb][j] * sc[cb * 4 + j] + sh[cb * 4 + j];
          else v[cb * 4 + j] = acc[cb][j];
        }
      uint4 o0 = Elem<T>::pack(v), o1 = Elem<T>::pack(v + 8);
      if constexpr (MODE == 1) {
        if (p.relu) { o0 = relu_vec<T>(o0); o1 = relu_vec<T>(o1); }
      } else if (p.stats && ok && !(DAS_STEM_VAR & 4)) {
        float f[16];
        Elem<T>::unpack(o0, f);                    // the values as stored
        Elem<T>::unpack(o1, f + 8);
#pragma unroll
        for (int j = 0; j < 16; ++j) { ssum[j] += f[j]; ssq[j] += f[j] * f[j]; }
      }
      if (ok && !(DAS_STEM_VAR & 2)) {
        T* dst = yg + (((long long)b * p.Ho + oy) * p.Wo + ox) * p.yps + c16;
        *reinterpret_cast<uint4*>(dst) = o0;
        *reinterpret_cast<uint4*>(dst + 8) = o1;
      }
    }
  }
  if (MODE == 0 && p.stats) {
    // per-channel sums: over the 16 pixels (lanes) of a DPP row, over the four waves through LDS, one round of atomics
    __syncthreads();                                   // (the window buffers are re-used: every wave has read its last fragments)
    float* red = reinterpret_cast<float*>(smem);       // [4 waves][2][64]
#pragma unroll
    for (int j = 0; j < 16; ++j) {
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        float v = w ? ssq[j] : ssum[j];
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xF, 0xF, true));
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xF, 0xF, true));
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xF, 0xF, true));
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xF, 0xF, true));
        if (q == 15) red[(wave * 2 + w) * 64 + c16 + j] = v;
      }
    }
    __syncthreads();
    if (tid < 128) {
      const int which = tid >> 6, c = tid & 63;
      const float s = red[(0 * 2 + which) * 64 + c] + red[(1 * 2 + which) * 64 + c] + red[(2 * 2 + which) * 64 + c] + red[(3 * 2 + which) * 64 + c];
      const int slot = p.stat_slots > 1 ? (int)(blockIdx.x % (unsigned)p.stat_slots) : 0;
      atomicAdd(p.stats + (slot * 2 + which) * p.Cout + c, s);
    }
  }
}

}  // namespace

namespace dasconv {

// Takes the 7 x 7, stride-2, pad-3 bf16 convs from 8 stored input channels to 64 output channels on plain NHWC tensors
// (tuning key conv.stem7x7 = 1). Returns false when the launch is not its.
bool try_launch_stem7x7(const ConvP& p, hipStream_t s) {
  if (dastune::get(dastune::CONV_STEM7X7) <= 0) return false;
  if (p.KH != 7 || p.KW != 7 || p.stride != 2 || p.pad != 3 || p.Cin != 8 || p.xps != 8 || p.Cout != 64 || p.yps % 8 ||
      p.up_sh != 0 || p.relu_in || p.osub || p.nlev > 1 || p.m_base != 0 || p.xbytes == 0 || p.res || p.bnb_raw || p.ksplit > 1 || p.K != 392 ||
      p.Ho != (p.H - 1) / 2 + 1 || p.Wo != (p.W - 1) / 2 + 1 || p.M % p.HoWo)
    return false;
  const bool affine = p.scale || p.shift || p.relu;
  if (affine && p.stats) return false;
  const int B = p.M / p.HoWo;
  const int tiles_x = (p.Wo + TC - 1) / TC, tiles_y = (p.Ho + TR - 1) / TR;
  const long long ntiles = (long long)B * tiles_x * tiles_y;
  if (ntiles > 0x7fffffffLL) return false;
  // (border tiles compute outputs nobody stores: not worth it when that is more than half of the launch)
  if ((long long)tiles_x * TC * tiles_y * TR > 2LL * p.Ho * p.Wo) return false;
  const int grid = (int)std::min<long long>(ntiles, dastune::usable_cus());
  dastune::note_kernel("conv_stem7x7_kernel");
  if (affine) hipLaunchKernelGGL((conv_stem7x7_kernel<1>), dim3(grid), dim3(256), 2 * WBYTES, s, p, tiles_x, tiles_y, (int)ntiles);
  else hipLaunchKernelGGL((conv_stem7x7_kernel<0>), dim3(grid), dim3(256), 2 * WBYTES, s, p, tiles_x, tiles_y, (int)ntiles);
  return true;
}

}  // namespace dasconv
