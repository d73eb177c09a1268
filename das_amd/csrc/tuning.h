// Dispatch thresholds of the kernel launchers (host side, process-global) and the record of which kernel a
// launcher picked. Values are set through das_tuning_set (include/das_hip.h): parity tests use it to drive a small
// problem through the kernels that production-size layers dispatch to, benchmarks use it for A/B runs.
// Nothing here reads the environment.
#pragma once

namespace dastune {

enum Key {
  CONV_BIG_MINBLOCKS,    // conv_glds3_kernel (256 x 128 tile) from this many tiles up
  CONV_BIG_MINK,         // ... and K at least this
  CONV_GLDS3_PP_MINK,    // conv_glds3_kernel runs its ping-pong schedule for K >= this (+4...8 % from K = 1024 up,
                         // neutral to -2 % at K = 512 / 576); -1 = never
  CONV_GLDS4_MINBLOCKS,  // conv_glds4_kernel (256 x 256 tile) from this many tiles up; 0 disables the kernel
  CONV_GLDS4_PP,         // -1 = ping-pong schedule for K >= 256, 0 / 1 force the choice
  CONV_GLDS4_MF,         // conv_glds4_kernel pixel-tile height in 32-row units: 0 = by round count, 8 = 256 rows, 9 = 288 rows
  CONV_STREAM_MINROWS,   // conv1x1_stream_kernel from this many pixel rows up; 0 disables the kernel
  CONV_STREAM_PERCU,     // ... workgroups per CU of its persistent grid. ONE: with two (rounds 2-4) the K <= 128 variants ran 512
                         // workgroups of which each walked half the tiles — 18-25 % slower per launch on every shape of the step
                         // (family 12.8 -> 11.4 ms, step -1.45 ms; an A/B that drowned in the step's +-0.5 ms noise until the host
                         // stopped running in lockstep with the GPU, round 5)
  CONV_TAIL_SPLIT,       // 1: a 256-row-tile launch whose last round of workgroups would be under half full hands the
                         // rows of that round to the 128-row tile kernel (second launch)
  CONV_SPLITK_TARGET,    // tile-kernel launches with at most half this many workgroups (x 2 for the 128-row kernel, two
                         // per CU) split their K loop over blockIdx.y up to about this many (+ splitk_finish_kernel);
                         // 0 disables split-K
  CONV_SPLITK_MINSTEPS,  // ... keeping at least this many K steps (of 128 bytes) per split
  CONV_SPLITK_KERNELS,   // bit mask of the kernels that may split: 1 conv_glds_kernel, 2 conv_glds3_kernel, 4 conv_glds4_kernel
  WGRAD_PP_MINK,         // conv_wgrad_pp_kernel for K >= this (and Cout >= 256); 0 disables the kernel
  WGRAD_SHAPES,          // conv_wgrad_kernel<bf16>: 1 = wave arrangement per op (128 x 128 / 64 x 256 / 256 x 64), 0 = always 128 x 128
  WGRAD_BKM,             // pixel rows per step of conv_wgrad_kernel<bf16>: 32 or 64
  WGRAD_BLOCKS,          // target grid of conv_wgrad_kernel; 0 = one resident wave of workgroups (3 or 2 per usable CU)
  WGRAD_PP_BLOCKS,       // target grid of conv_wgrad_pp_kernel; 0 = one workgroup per usable CU (device CUs minus
                         // comm.reserved_cus); lower it when several weight gradients run side by side on different streams
  BN_REDUCE_BLOCKS,      // grid / block size of bn_bwd_reduce_kernel
  BN_REDUCE_THREADS,
  BN_VPT,                // 16-byte vectors per thread of the BatchNorm apply passes
  GN_PPB,                // minimum pixels per workgroup of the GroupNorm passes; 0 = by the launch's size (common.h: gn_ppb_min)
  CONV_C64_MINTILES,     // conv3x3_c64_kernel (3x3, 64 -> 64 channels, 16 x 16-pixel tiles with the input patch and the
                         // whole weight matrix in LDS) from this many tiles up; 0 disables the kernel
  BN_STREAM_MINBYTES,    // BatchNorm apply passes over tensors of at least this many bytes: slot fold as its own launch +
                         // a one-shot pass of small workgroups (bn_apply_stream_kernel); 0 = never
  COMM_RESERVED_CUS,     // CUs the persistent one-workgroup-per-CU grids leave free (for the RCCL kernels of the overlapped
                         // gradient all-reduce when several GPUs train together); 0 on one GPU
  ELEM_UPSTATS_PPB,      // output pixels per workgroup of bilinear_ac_stats_kernel; 0 = by size
  BN_UPMERGE_BLOCKS,     // grid cap of upmerge_bwd_reduce_kernel
  DCN_FUSED_MINROWS,     // das_dcn3x3_fused instead of im2col + GEMM in the eval forward from this many pixel rows up (0 = never):
                         // 15 % ahead at 141 k rows, behind at 71 k (its 128-pixel tiles fill three rounds for 2.2 rounds of work)
  CONV_BALANCE_ROWS,     // one-workgroup-per-CU tile launches whose last round would be partly empty spread their pixel rows evenly
                         // over the tiles of full rounds (tiles of ConvP::mstep < 256 rows): 0 off, 1 the 1x1 convs, 2 every conv
  CONV_GLDS4_MFMA32,     // conv_glds4_kernel<pp> (256 x 256 tiles, bf16) on v_mfma_f32_32x32x16_bf16 instead of 16x16x32: 0 / 1
  CONV_KSTREAM,          // conv1x1_kstream_kernel (weight-stationary, K split over two wave groups) for the 1x1 convs with
                         // K = 512 / 1024, bit mask: 1 / 2 = K 512 / 1024 data gradients with the fused BatchNorm backward,
                         // 4 / 8 = K 512 forward with Cout 128 / wider, 16 = K 1024 forward, 32 / 64 = K 512 / 1024 data gradients with the
                         // fused BatchNorm backward and the mask as recorded bits (+ a second gradient), 128 / 256 = K 512 / 1024 plain output + residual
                         // (measured level with the tile kernels: off); default 35, 0 = never
  CONV_SPLITK_INKERNEL,  // split-K convs on conv_glds3_kernel finish their sum inside the kernel (last arriver per tile) instead of
                         // in splitk_finish_kernel: 0 / 1
  CONV_STREAM_NT,        // conv1x1_stream_kernel: bit 0 the pixel operand's LDS-DMA non-temporal, bit 2 the output stores
  BN_NT_FWD,             // cache policy of bn_apply_stream_kernel's accesses: bit 0 pre-norm tensor loaded non-temporal, bit 1 the
                         // residual, bit 2 the output stored non-temporal (streamed tensors of >= bn.stream_minbytes: larger than
                         // what the 256 MiB Infinity Cache keeps for the consumer anyway)
  BN_NT_BWD,             // ... of bn_bwd_apply_dz_stream_kernel: bit 0 dZ, bit 1 the pre-norm tensor, bit 2 the gradient stored
  CONV_STEM7X7,          // conv_stem7x7_kernel (conv_stem.hip) for the backbone's 7 x 7 stride-2 stem conv instead of conv_reg_kernel: 0 / 1
  N_KEYS
};

long long get(Key k);
// CUs of the current device (cached), and the CUs a persistent one-workgroup-per-CU (or n-per-CU) grid may count on:
// the device's minus comm.reserved_cus (left to the RCCL kernels of an overlapped gradient all-reduce), at least 8.
int device_cus();
int usable_cus();
// The calling thread's share of the chip for conv_wgrad_pp_kernel's grid (das_wgrad_pp_share): grid = usable_cus() / den
// while wgrad.pp_blocks is 0. Thread-local: a side-stream scope in one thread never changes another thread's launches.
int pp_share_den();
// `name` must be a string literal (kept by pointer): the kernel the calling thread's last launcher call picked
void note_kernel(const char* name);
// note_kernel calls of the calling thread so far (prof.hip: did the entry point inside this scope pick a kernel?)
unsigned long long note_count();

}  // namespace dastune
