/*
 * das_hip.h — C ABI of libdas_hip.so, the MI355X (gfx950) device library behind the DAS
 * hot path (MSPN2 backbone -> FPN -> DASHead -> decode).
 *
 * The reference (wangzt-halo/das, a mmdet3d fork) has no native code on this path: every
 * device op it runs is a torch/ATen or mmcv-full CUDA op reached through Python. The entry
 * points below are therefore the ops those call sites bind, restated for NHWC tensors:
 * plain device pointers, sizes, a hipStream_t passed as void*, int status codes. No torch
 * types cross this boundary. All tensors are NHWC ("channels last"), row = one pixel,
 * `*_pix_stride` = elements between consecutive pixels (>= channel count; lets an op read
 * or write a channel slice of a wider tensor). dtype codes: DAS_F32 / DAS_BF16 storage,
 * arithmetic always accumulates in f32.
 *
 * Every function is asynchronous on `stream`, never allocates, never synchronises, and
 * returns DAS_OK or a DAS_ERR_* code (argument errors are detected before any launch).
 */
#ifndef DAS_HIP_H
#define DAS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DAS_OK 0
#define DAS_ERR_ARG 1      /* unsupported shape / alignment / dtype combination */
#define DAS_ERR_LAUNCH 2   /* hipGetLastError() != hipSuccess after the launch */

#define DAS_F32 0
#define DAS_BF16 1

/* Ragged multi-level pixel rows (used by the head ops and their gradients). */
typedef struct {
  int num_levels, B;
  int H[5], W[5];
} DasLevels;

/* Library/ABI version (2: round-2 descriptor layouts) and the code object's target, for the loader's sanity check. */
int das_abi_version(void);
const char* das_target_arch(void);

/* Dispatch thresholds of the launchers (which tile kernel a conv / weight-gradient shape goes to, grid sizes of the
 * norm passes): process-global integers addressed by name, e.g. "conv.glds4_minblocks". The defaults are the
 * measured break-even points; parity tests lower them so that small problems run through the kernels
 * production-size layers dispatch to, A/B benchmarks flip them. No reference counterpart (torch picks its cuDNN /
 * MIOpen algorithm internally: torch.backends.cudnn.benchmark, tools/train.py:115-116). Unknown key: DAS_ERR_ARG.
 * Keys: conv.big_minblocks, conv.big_mink, conv.glds3_pp_mink, conv.glds4_minblocks, conv.glds4_pp (-1 auto / 0 / 1),
 * conv.glds4_mf, conv.stream_minrows, conv.stream_percu, conv.tail_split, conv.splitk_target, conv.splitk_minsteps, conv.splitk_kernels, conv.c64_mintiles, wgrad.pp_mink, wgrad.shapes (1: conv_wgrad_kernel picks 128 x 128 / 64 x 256 / 256 x 64 wave arrangements per op), wgrad.bkm, wgrad.blocks, wgrad.pp_blocks,
 * bn.reduce_blocks, bn.reduce_threads, bn.vpt, gn.ppb, bn.stream_minbytes, comm.reserved_cus (CUs every persistent grid leaves free: wgrad.blocks / wgrad.pp_blocks 0 = one
 * resident wave of workgroups on the remaining CUs), elem.upstats_ppb (output pixels per workgroup of the resampling
 * kernels, 0 = by size), bn.upmerge_blocks (grid cap of the fused reduce passes of upmerge.hip / skipadd.hip),
 * dcn.fused_minrows (das_amd.nn.dcn_v2: das_dcn3x3_fused in the eval forward from this many pixel rows up; 0 = never),
 * conv.balance_rows (0 off / 1 the 1x1 convs / 2 every conv: see das_conv_last_tile_rows),
 * conv.kstream (bit mask: which 1x1 convs with K = 512 / 1024 take the weight-stationary conv1x1_kstream_kernel: 1 / 2 = K 512 / 1024
 * data gradients with the fused BatchNorm backward, 4 / 8 = K 512 forward with Cout 128 / wider, 16 = K 1024 forward, 32 / 64 = K 512 / 1024
 * data gradients with the mask as recorded bits and an optional second gradient, 128 / 256 = K 512 / 1024 plain output + residual; default 35), conv.splitk_inkernel, conv.glds4_mfma32, conv.stream_nt,
 * conv.stem7x7 (1: the 7 x 7 stride-2 stem conv on conv_stem7x7_kernel, 0: on the generic conv_reg_kernel),
 * bn.nt_fwd / bn.nt_bwd (bit masks: which operands of the one-shot BatchNorm passes over large tensors use non-temporal
 * loads / stores; see csrc/tuning.h). */
int das_tuning_set(const char* key, long long value);
int das_tuning_get(const char* key, long long* value);
int das_tuning_reset(void);
/* The CALLING THREAD's share of the chip for the 256 x 256 ping-pong weight-gradient kernel while key wgrad.pp_blocks is 0:
 * its grid becomes (device CUs - comm.reserved_cus) / den workgroups (den = 1: the whole chip, the default). Set to 2 around
 * the weight-gradient launches that run on a side stream beside the main stream's kernels (das_amd.autograd._on_side) and
 * back to 1 afterwards; thread-local, so another thread's launches (another model or device in the process) never see it.
 * den outside 1..16: DAS_ERR_ARG. No reference counterpart (torch's autograd engine has no such control). */
int das_wgrad_pp_share(int den);
/* Name of the kernel the calling thread's last das_conv2d_nhwc / das_conv2d_wgrad_nhwc call launched
 * ("conv_glds4_kernel<pp>", "conv1x1_stream_kernel", ...): lets a parity test assert WHICH kernel it checked. */
const char* das_last_kernel(void);
/* Rows per pixel tile of the calling thread's last das_conv2d_nhwc launch on a 256-row tile kernel: 256 (288), or the
 * smaller balanced step (tuning key conv.balance_rows: a launch whose last round of one-workgroup-per-CU tiles would be
 * partly empty spreads its rows evenly over the tiles of full rounds); 0 if the last launch used another kernel. */
int das_conv_last_tile_rows(void);
/* Measurement aid (no reference counterpart): `blocks` workgroups of `threads` threads and `lds_bytes` of LDS each that
 * do nothing but hold their CU slots for `usec` microseconds on `stream` — stands in for the RCCL kernels of an
 * overlapped gradient all-reduce when only one GPU is at hand (tools/dev/cu_pressure.py, tests/test_ddp_gpu.py):
 * with lds_bytes near 160 KiB a workgroup owns its CU, with a few KiB it shares it as a collective's kernel would.
 * The persistent kernels size their grids for das_tuning key comm.reserved_cus fewer CUs. */
int das_dev_occupy_cus(int blocks, int threads, int lds_bytes, int usec, void* stream);
/* Launch timing inside the library (measurement; the reference's counterpart is the wall clock of
 * tools/analysis_tools/benchmark.py:63-90, which cannot attribute time to kernels). Between das_prof_begin and
 * das_prof_end every entry point that launches kernels records one HIP event on ITS stream right before its first
 * launch and one right after its last: no interpreter / FFI time sits between an event and the launch it brackets. An
 * entry point called by another one does not record again. das_prof_count: records so far (a caller brackets its own
 * call with two counts to learn which records it produced). das_prof_read: waits for the first n records' events and
 * returns their elapsed milliseconds (-1 for a record whose events failed) and, when `names` is given, per record the
 * kernel name its launcher picked (das_last_kernel) or else the entry point's name, name_stride (>= 16) bytes apiece.
 * Event pairs are reused by the next das_prof_begin. Off, a scope costs one load. */
int das_prof_begin(void);
int das_prof_end(void);
long long das_prof_count(void);
int das_prof_read(float* ms, char* names, int name_stride, long long n);

/* ------------------------------------------------------------------------------------
 * Convolution as implicit GEMM on MFMA, fused epilogue.
 * Replaces: every torch `nn.Conv2d` / mmcv `ConvModule` conv on the path
 *   (mspn_mmpose.py:81-105,254,327-379,546; anchor_free_mono3d_pose_head.py:116-165;
 *    das_head.py:112-161; recursive_update.py:171-180,243) together with the eval-mode
 *   BatchNorm affine, bias add, ReLU and residual add that follow them
 *   (mspn_mmpose.py:126-157,381-404).
 *   y = act( (conv(act_in(x), w)) * scale[c] + shift[c] + residual )
 * x: (B,H,W,Cin) dtype, Cin % 8 == 0.  w: (Cout,KH,KW,Cin) dtype, K-contiguous.
 * y: (B,Ho,Wo,Cout) out_dtype. Cout % 8 == 0 (pad weights with zero rows).
 * scale/shift: f32[Cout] or NULL. residual: same dtype/shape as y or NULL (needs
 * out_dtype == dtype). stats: f32[2*Cout] accumulators (sum, sum of squares of the values
 * *as stored in y before residual/relu*), or NULL — used for train-mode BatchNorm.
 */
typedef struct {
  int dtype, out_dtype;
  int B, H, W, Cin, x_pix_stride;
  int Ho, Wo, Cout, y_pix_stride;
  int KH, KW, stride, pad;
  int relu_in, relu;
  const float* scale;
  const float* shift;
  const void* residual;
  int res_pix_stride;
  float* stats;
  /* Ragged multi-level mode (num_levels > 1): x and y hold the rows of all FPN levels back to back,
   * level l = B*lvl_H[l]*lvl_W[l] pixel rows in (b,h,w) order; requires stride 1 and pad = KH/2.
   * One launch then covers every level of the shared-weight head convs (das_head.py:176-178
   * `multi_apply(self.forward_single, feats, ...)`). num_levels <= 1: plain (B,H,W,C) tensor. */
  int num_levels;
  int lvl_H[5], lvl_W[5];
  /* Input zero-upsampling factor, 1 or 2 (0 = 1). With in_up = s the kernel reads x as if s-1 zeros were
   * inserted between its pixels: the data-gradient of a stride-s conv is
   * das_conv2d_nhwc(x = dY, w = flipped/transposed weights, stride 1, pad = KH-1-pad, in_up = s,
   * Ho/Wo = the forward input size). */
  int in_up;
  /* stats is f32[stats_slots][2*Cout] (0 = 1 slot): workgroup b adds into slot b % stats_slots. Thousands of
   * workgroups adding into ONE [2*Cout] array serialise on the same words (~13 ns each: +44 us on the
   * 64-channel convs of the 128 x 208 stage); das_bn_train_apply sums the slots. */
  int stats_slots;
  /* Fused BatchNorm-backward reduction, for data-gradient launches (bnb_raw != NULL switches it on; needs `stats`,
   * out_dtype == dtype, no relu). The value about to be stored, g = conv + residual, is then the gradient with respect
   * to the OUTPUT of a train-mode BatchNorm (+ReLU) layer (mspn_mmpose.py:126-157 under autograd) whose pre-norm
   * tensor is bnb_raw (same rows / channels as y, pixel stride bnb_pix_stride). The kernel stores dZ = g * mask
   * (bnb_relu: mask = bnb_y > 0 if bnb_y is given — required when a residual entered before that ReLU — else
   * bn_affine(raw) > 0 from bnb_mean / invstd / gamma / beta; no bnb_relu: mask = 1) and adds
   * [sum dZ | sum dZ * (raw - mean) * invstd] per channel into stats[slot][2*Cout]: the separate reduction pass
   * of das_bn_train_backward over (dY, raw) is not needed, das_bn_backward_apply finishes the layer. */
  const void* bnb_raw;
  const void* bnb_y;
  const float* bnb_mean;
  const float* bnb_invstd;
  const float* bnb_gamma;
  const float* bnb_beta;
  int bnb_relu, bnb_pix_stride;
  /* Sub-grid output (out_sub = 1; single level, stride 1): conv output pixel (b, i, j), i < Ho, j < Wo, is row
   * (b * out_H + 2 i + out_ph) * out_W + 2 j + out_pw of y — and of residual, bnb_raw, bnb_y, which are tensors of the
   * full B x out_H x out_W grid. The data gradient of a stride-2 conv is four such launches, one per output parity
   * (ph, pw), each over only the taps that land on dY samples — das_conv2d_nhwc with in_up = 2 multiplies three zeros
   * for every sample (torch: conv_transpose / cuDNN dgrad behind mspn_mmpose.py's stride-2 Bottlenecks). */
  int out_sub, out_ph, out_pw, out_H, out_W;
  /* Instead of bnb_y (used when bnb_y == NULL): the ReLU mask of that layer as bits, one byte per 16-byte vector of y
   * (bit j = element j > 0; index = element offset / elements per vector, y contiguous with pixel stride
   * bnb_pix_stride) as das_bn_train_apply / das_bn_dual_apply / das_upmerge_forward record it: 1/16 of y's bytes. */
  const void* bnb_mask_bits;
  /* With bnb_raw and residual: the residual enters as residual * mask, the mask given as bits in the same layout (one byte
   * per 16-byte vector of the residual tensor). For a residual that is the gradient of a BatchNorm + identity + ReLU layer's
   * output: the identity branch receives dY * (y > 0), and with the bits recorded by the forward that masked tensor never
   * has to be written by the layer's backward (das_bn_train_backward with dres == NULL). */
  const void* residual_mask_bits;
} DasConvDesc;
int das_conv2d_nhwc(const void* x, const void* w, void* y, const DasConvDesc* d, void* stream);

/* ------------------------------------------------------------------------------------
 * Training path: gradients of the conv / norm units (torch autograd of the ops above;
 * Fp16OptimizerHook -> loss.backward() in the reference's runner).
 *
 * Data gradient of a conv = das_conv2d_nhwc on (dY, flipped + transposed weights, in_up = stride).
 * Weight gradient: dw f32[Cout][KH][KW][Cin] = sum over output pixels of
 * dY[m][o] * X[m @ tap][ci]; `d` describes the FORWARD conv (x geometry, Ho/Wo, stride, pad,
 * y_pix_stride = pixel stride of dy, ragged levels allowed). accumulate = 0: dw is zeroed by the call;
 * accumulate = 1: the result is added to dw (the optimizer's flat gradient buffer holds conv weights in
 * exactly this layout, so backward adds straight into it — what autograd's AccumulateGrad does in the
 * reference). Cin % 8 == 0. Cout may be any count as long as y_pix_stride is a multiple of 8 and at least Cout rounded up to
 * one (a layer with 45 output channels whose gradient tensor has 48 columns): the rows below Cout are stored, no others.
 */
int das_conv2d_wgrad_nhwc(const void* x, const void* dy, float* dw, const DasConvDesc* d, int accumulate,
                          void* stream);
/* n (<= 64) weight gradients at once: the ops of one kernel class share ONE persistent launch (one resident wave of
 * workgroups) that the host schedules like a job shop. A (Cout, K) tile whose whole pixel reduction fits under the
 * per-workgroup quota is one unit: its accumulators go straight into dw (no workspace, no reduction pass); longer
 * tiles are cut into equal runs whose partial tiles pass through the workspace and a reduction kernel. Every workgroup
 * walks a list of units, lists are packed longest-first per XCD (the tiles of one op over the same pixel rows share
 * an L2). The schedule depends on the shapes only and is cached per op list. xs / dys / dws / descs: arrays of n
 * entries with the meaning of das_conv2d_wgrad_nhwc; the dws must be distinct. Backward may defer its weight
 * gradients to collect such batches: nothing but the optimizer reads them (torch autograd runs them inside each
 * conv's backward node; the reference has no counterpart). */
int das_conv2d_wgrad_batch(int n, const void* const* xs, const void* const* dys, float* const* dws,
                           const DasConvDesc* descs, int accumulate, void* stream);
/* Schedule of the calling thread's last weight-gradient launch (tests, tuning): out[0..7] = kernel class (0 ping-pong
 * 256 x 256, 1 bf16 128 x 128, 2 f32), grid, units, units stored straight into dw, partial tiles through the workspace,
 * reduced tiles, longest unit list of a workgroup, reduce groups; out[8] = schedules built by this process so far (a
 * steady training loop stops adding to it: schedules are cached per op list). */
int das_wgrad_last_plan(long long* out, int n);
/* out f32[C] (zeroed by the call) = column sums of x (rows, C) — bias gradients. C need not be a multiple of 8 when pix_stride
 * is one and at least C rounded up to one (the padding columns are not summed). das_colsum_acc: added to out instead. */
int das_colsum(const void* x, int dtype, long long rows, int C, int pix_stride, float* out, void* stream);
/* The same sums ADDED to out (the optimizer's flat gradient slice of a bias: no temporary, no fill, no separate add). */
int das_colsum_acc(const void* x, int dtype, long long rows, int C, int pix_stride, float* out, void* stream);
/* Train-mode BatchNorm (+ReLU, + residual) backward. dZ = dY * (y > 0) when relu; with y == NULL (allowed
 * when no residual was added before the ReLU) the mask is recomputed from raw, gamma and beta, which saves
 * reading y in both passes. sums f32[2C] (zeroed by the call) receive [sum dZ, sum dZ*xhat] = [dbeta,
 * dgamma]; draw = gamma*invstd*(dZ - s1/N - xhat*s2/N); dres (optional) = dZ, the gradient of the residual
 * input. raw = pre-norm conv output saved by forward. dgamma_acc / dbeta_acc (both or neither): f32[C]
 * parameter-gradient accumulators that the sums are added to. */
int das_bn_train_backward(const void* dy, const void* y, const void* raw, int dtype, long long rows, int C,
                          const float* mean, const float* invstd, const float* gamma, const float* beta, int relu,
                          void* draw, void* dres, float* sums, int sums_prezeroed, float* dgamma_acc,
                          float* dbeta_acc, void* stream);
/* (sums_prezeroed != 0: the caller hands sums already zero-filled — e.g. a slice of a larger scratch cleared
 * once — and the call skips its own memset.) */
/* The same in two phases for SyncBN: phase 1 = only the per-channel sums of this rank's rows; the caller
 * all-reduces `sums`; phase 2 = only the apply pass with 1/N taken from stat_rows (all ranks' rows) — what
 * torch's SyncBatchNorm backward does with its all_reduce of (sum_dy, sum_dy_xmu). phase 0 = both at once.
 * The parameter-gradient accumulators are fed from `sums` in phase 2: hand them only when `sums` is local. */
/* das_bn_train_backward for a ReLU layer whose mask was recorded as bits by the forward (das_bn_train_apply's
 * relu_bits_out): neither pass reads y. */
int das_bn_train_backward_bits(const void* dy, const void* y_relu_bits, const void* raw, int dtype, long long rows, int C,
                               const float* mean, const float* invstd, const float* gamma, void* draw, void* dres,
                               float* sums, int sums_prezeroed, float* dgamma_acc, float* dbeta_acc, void* stream);
/* ... and in the two phases of das_bn_train_backward_phase (SyncBN). */
int das_bn_train_backward_bits_phase(const void* dy, const void* y_relu_bits, const void* raw, int dtype, long long rows, int C,
                                     const float* mean, const float* invstd, const float* gamma, void* draw, void* dres,
                                     float* sums, int sums_prezeroed, float* dgamma_acc, float* dbeta_acc, int phase,
                                     long long stat_rows, void* stream);
int das_bn_train_backward_phase(const void* dy, const void* y, const void* raw, int dtype, long long rows, int C,
                                const float* mean, const float* invstd, const float* gamma, const float* beta,
                                int relu, void* draw, void* dres, float* sums, int sums_prezeroed, float* dgamma_acc,
                                float* dbeta_acc, int phase, long long stat_rows, void* stream);
/* The apply pass alone: dZ is already masked and the per-channel sums [sum dZ | sum dZ*xhat] were accumulated into
 * sums f32[sums_slots][2C] by the data-gradient conv that produced dZ (DasConvDesc.bnb_*), so the layer needs no
 * reduction pass: draw = gamma*invstd*(dZ - s1/N - xhat*s2/N) with N = stat_rows; dgamma_acc / dbeta_acc (both or
 * neither) receive s2 / s1. Same autograd semantics as das_bn_train_backward (mspn_mmpose.py:126-157 backward). */
int das_bn_backward_apply(const void* dz, const void* raw, int dtype, long long rows, int C, const float* mean,
                          const float* invstd, const float* gamma, const float* sums, int sums_slots, void* draw,
                          float* dgamma_acc, float* dbeta_acc, long long stat_rows, void* stream);

/* All conv weights of the network packed in one launch, once per optimizer step. flat_src: the optimizer's
 * f32 master buffer, conv weights stored as (Cout,KH,KW,Cin) (the forward operand layout). For every table
 * entry: fwd_dst[off ...] = the same layout cast to dtype (fwd_dst may be NULL: the f32 path reads the
 * master directly); dgrad_dst[off ...] = (Cin,KH,KW,Cout) with both tap axes flipped — the operand of the
 * data-gradient conv. entries_dev: device copy of the table, tile_start = running sum of
 * KH*KW*ceil(O/64)*ceil(I/64); total_tiles = the final sum. O and I multiples of 4, off a multiple of 4 elements
 * (16-byte accesses). Replaces the per-layer permute / flip / cast
 * that torch (cuDNN/MIOpen) does internally for every conv of `loss.backward()`. */
typedef struct {
  long long off;
  int O, I, KH, KW;
  int tile_start;
  int s2_pad;   /* >= 0: a stride-2 KH x KH conv with this padding — its flipped taps are ALSO written split by output
                 * parity into dgrad_s2_dst[off ...]: the four (Cin, nth, ntw, Cout) operands of the stride-2 data
                 * gradient's sub-grid launches (DasConvDesc.out_sub), classes (0,0), (0,1), (1,0), (1,1) back to back;
                 * -1: no such copies */
} DasPackEntry;
int das_pack_conv_weights(const float* flat_src, void* fwd_dst, void* dgrad_dst, void* dgrad_s2_dst, int dtype,
                          const DasPackEntry* entries_dev, int n_entries, int total_tiles, void* stream);

/* GroupNorm(+ReLU) backward over ragged rows. x = pre-norm input saved by forward, y = forward output
 * (ReLU mask), fwd_stats = the forward's stats workspace (sum, sumsq per level/image/group).
 * gsums_ws f32[num_levels*B*G*2], dgamma/dbeta f32[C]: all zeroed by the call.
 * y == NULL with relu: the mask is recomputed from x with the forward's own arithmetic (needs beta) and neither pass
 * reads y — no residual enters a GroupNorm layer of the head, so y > 0 <=> its affine > 0. */
int das_groupnorm_backward(const void* dy, const void* y, const void* x, void* dx, int dtype, const DasLevels* lv,
                           int C, int pix_stride, int G, const float* fwd_stats, const float* gamma, const float* beta,
                           float eps, int relu, float* gsums_ws, float* dgamma, float* dbeta, void* stream);
/* The same with dgamma / dbeta ACCUMULATED into (parameter-gradient slices of the optimizer's flat buffer); only
 * gsums_ws is zeroed by the call — unless ws_zeroed says the caller hands it over zeroed (a slice of a buffer it fills
 * once for many layers: a fill is a launch of its own, 28 of them per training step before). */
int das_groupnorm_backward_acc(const void* dy, const void* y, const void* x, void* dx, int dtype, const DasLevels* lv,
                               int C, int pix_stride, int G, const float* fwd_stats, const float* gamma, const float* beta,
                               float eps, int relu, float* gsums_ws, float* dgamma, float* dbeta, int ws_zeroed,
                               void* stream);
/* Backward of das_maxpool3x3s2 (gradient goes to the first maximum in scan order, as torch does),
 * das_upsample_bilinear_ac and the upsampled operand of das_add_upsample_nearest. */
int das_maxpool3x3s2_backward(const void* x, const void* dy, void* dx, int dtype, int B, int H, int W, int C,
                              void* stream);
int das_upsample_bilinear_ac_backward(const void* dy, void* dx, int dtype, int B, int H, int W, int C, int Ho, int Wo,
                                      void* stream);
int das_upsample_nearest_backward(const void* dy, void* db, int dtype, int B, int H, int W, int C, int Hb, int Wb,
                                  void* stream);

/* Pack an NCHW f32 image batch into NHWC `dtype` with channels zero-padded to Cpad.
 * Replaces the implicit layout of `img` entering MSPN2.forward (mspn_mmpose.py:657-662). */
int das_pack_nchw_to_nhwc(const float* x, void* y, int dtype, int B, int C, int H, int W, int Cpad, void* stream);
/* NHWC `dtype` (channel slice [c0, c0+C) of rows with pix_stride) -> dense NCHW f32. */
int das_unpack_nhwc_to_nchw(const void* x, float* y, int dtype, int B, int C, int H, int W, int pix_stride,
                            int c0, void* stream);

/* 3x3 stride-2 pad-1 max pooling (mspn_mmpose.py:553 `MaxPool2d`). */
int das_maxpool3x3s2(const void* x, void* y, int dtype, int B, int H, int W, int C, void* stream);
/* The same, also recording the winning tap (0..8 in scan order, the FIRST maximum) of every output element: idx is
 * u8[B*Ho*Wo*C]. das_maxpool3x3s2_backward_argmax(dy, idx) -> dx then gathers from 80 MB instead of re-deriving the maxima
 * from x (das_maxpool3x3s2_backward: 338 us per step at B = 16); both backward routes give the same bits. */
/* das_upsample_bilinear_ac that also reduces the BatchNorm statistics of its OUTPUT: per-channel sum / sum of squares of the
 * stored values added into stats f32[stats_slots][2*C] (zeroed by the caller; same convention as DasConvDesc.stats). For a
 * bias-free 1x1 conv behind a bilinear upsampling (MSPN's `up_conv`, mspn_mmpose.py:385-389): both are linear and
 * commute, so the conv runs on the quarter-size tensor and this call produces the pre-norm tensor with its statistics.
 * y == NULL: statistics only (of the values as they WOULD be stored in `dtype`), nothing is written. */
int das_upsample_bilinear_ac_stats(const void* x, void* y, int dtype, int B, int H, int W, int C, int Ho, int Wo,
                                   float* stats, int stats_slots, void* stream);
/* The merge step of an MSPN upsample unit in train mode, out = relu(BN1(raw1) + BN2(upsample(z))) (mspn_mmpose.py:381-404; z =
 * up_conv applied at LOW resolution, see das_upsample_bilinear_ac_stats), without writing either normalised branch or
 * upsample(z): das_amd/csrc/upmerge.hip has the algebra. raw1 / out (B, Ho, Wo, C), z (B, H, W, C), per-channel f32[C]
 * mean / invstd (published by das_bn_train_apply) and gamma / beta of the two BatchNorm layers.
 * das_upsample_stats_lowres: the batch statistics of upsample(z) — per-channel sum / sum of squares added into
 * stats f32[stats_slots][2C], DasConvDesc.stats' convention — computed from z alone (tables as for
 * das_upmerge_backward_lowres): sum upsample(z) = sum w z, sum upsample(z)^2 = sum z (upsample^T upsample z). */
int das_upsample_stats_lowres(const void* z, int dtype, int B, int H, int W, int C, const float* ah, const float* aw,
                              const float* wh, const float* ww, float* stats, int stats_slots, void* stream);
int das_upmerge_forward(const void* raw1, const void* z, void* out, int dtype, int B, int H, int W, int C, int Ho, int Wo,
                        const float* mean1, const float* invstd1, const float* gamma1, const float* beta1,
                        const float* mean2, const float* invstd2, const float* gamma2, const float* beta2,
                        void* relu_bits_out, void* stream);
/* Backward pass A: dzm = dy * (out > 0) written once (the gradient of both pre-activation branches), and
 * sums f32[3C] = [sum dZ | sum dZ xhat1 | sum dZ xhat2] (zeroed here unless sums_zeroed says the caller did), xhat2 from
 * upsample(z) recomputed on the fly. The first 2C are BatchNorm 1's sums in das_bn_backward_apply's layout (one slot). */
int das_upmerge_backward_reduce(const void* dy, const void* out, const void* out_relu_bits, const void* raw1, const void* z,
                                void* dzm, int dtype,
                                int B, int H, int W, int C, int Ho, int Wo, const float* mean1, const float* invstd1,
                                const float* mean2, const float* invstd2, float* sums, int sums_zeroed, void* stream);
/* Backward pass D, at low resolution: dz = upsample^T(d raw2) from P = upsample^T(dzm) (das_upsample_bilinear_ac_backward), z, the
 * sums of pass A and the tables of upsample^T upsample (ah f32[H][3], aw f32[W][3]: its tridiagonal factors; wh f32[H], ww
 * f32[W]: upsample^T 1). stat_rows = B * Ho * Wo (times the ranks under SyncBN). dgamma2_acc / dbeta2_acc (both or
 * neither): BatchNorm 2's parameter gradients are ADDED there. */
int das_upmerge_backward_lowres(const void* P, const void* z, void* dz, int dtype, int B, int H, int W, int C,
                                const float* ah, const float* aw, const float* wh, const float* ww, const float* sums,
                                const float* gamma2, const float* mean2, const float* invstd2, long long stat_rows,
                                float* dgamma2_acc, float* dbeta2_acc, void* stream);
/* MSPN's cross-stage merge in train mode, out = x + relu(BN1(raw1)) + relu(BN2(raw2)) (mspn_mmpose.py:254-275: the next stage's
 * level feature plus the previous stage's two skip branches, out_skip1 / out_skip2 of mspn_mmpose.py:381-404), without
 * writing the two normalised tensors (das_amd/csrc/skipadd.hip). All tensors (rows, C); bn = 8 pointers to f32[C]:
 * mean1, invstd1, gamma1, beta1, mean2, invstd2, gamma2, beta2.
 * Backward: g = d out (also d x); writes d raw1, d raw2 (the full train-mode BatchNorm backward of each branch, ReLU mask
 * recomputed from raw_i) and sums f32[4C] = [sum g1 | sum g1 xhat1 | sum g2 | sum g2 xhat2] (dbeta_i | dgamma_i; zeroed here
 * unless sums_zeroed); stat_rows = rows. The four accumulators (all or none): the parameter gradients are ADDED there as well.
 * phase: 0 = both passes; SyncBN: 1 = only the sums of this rank's rows (the caller sums them over the ranks), 2 = only the
 * apply pass with `sums` as given and 1/N from stat_rows (all ranks' rows) — hand the accumulators only with LOCAL sums. */
/* out = relu?(BN1(raw1) + BN2(raw2)): a bottleneck's bn3 with the block's projection shortcut `downsample(x)`
 * (mspn_mmpose.py:126-157) normalised on the fly — the shortcut's normalised tensor is never written. bn as below. */
int das_bn_dual_apply(const void* raw1, const void* raw2, void* out, int dtype, long long rows, int C, const float* const* bn,
                      int relu, void* relu_bits_out, void* stream);
int das_bn_relu_add3_forward(const void* x, const void* raw1, const void* raw2, void* out, int dtype, long long rows, int C,
                             const float* const* bn, void* stream);
int das_bn_relu_add3_backward(const void* g, const void* raw1, const void* raw2, void* draw1, void* draw2, int dtype,
                              long long rows, int C, const float* const* bn, float* sums, int sums_zeroed,
                              long long stat_rows, float* dgamma1_acc, float* dbeta1_acc, float* dgamma2_acc, float* dbeta2_acc,
                              int phase, void* stream);
int das_maxpool3x3s2_argmax(const void* x, void* y, void* idx, int dtype, int B, int H, int W, int C, void* stream);
int das_maxpool3x3s2_backward_argmax(const void* dy, const void* idx, void* dx, int dtype, int B, int H, int W, int C,
                                     void* stream);

/* Bilinear upsampling, align_corners=True, to (Ho,Wo) (mspn_mmpose.py:385-389). */
int das_upsample_bilinear_ac(const void* x, void* y, int dtype, int B, int H, int W, int C, int Ho, int Wo,
                             void* stream);

/* y = a + nearest_upsample(b -> (H,W)); FPN top-down path (mmdet FPN.forward). */
int das_add_upsample_nearest(const void* a, const void* b, void* y, int dtype, int B, int H, int W, int C,
                             int Hb, int Wb, void* stream);

/* y = a + b (+ c); c may be NULL (mspn_mmpose.py:284-285 cross-stage skip adds). n elements. */
int das_add3(const void* a, const void* b, const void* c, void* y, int dtype, long long n, int relu, void* stream);

/* Train-mode BatchNorm finalize (torch BatchNorm2d in training mode, mspn_mmpose.py:74-79):
 * from `stats` = [sum(C), sumsq(C)] over `count` pixels compute mean / biased var,
 * y = relu?( (x-mean)*rsqrt(var+eps)*gamma + beta + residual ), and update running stats
 * (momentum, unbiased var). save_mean/save_invstd: f32[C] outputs for backward.
 * num_batches_tracked: optional device int64 scalar (the BatchNorm buffer of that name), incremented by 1.
 * stat_count: the population behind `stats` when it is larger than this tensor's `count` rows — SyncBN
 * (`norm_cfg=dict(type='SyncBN')`, configs/_base_/models/das.py): the caller all-reduces `stats` over the ranks
 * first and passes the global row count; 0 = count. stats_slots: `stats` is f32[stats_slots][2*C] partial sums
 * (DasConvDesc.stats_slots; 0 = 1); with more than one slot the call folds them into slot 0 in place first.
 * y == NULL: finalize only (mean / invstd saved, running statistics and the counter advanced, nothing normalised): a
 * layer whose output has no consumer — the last MSPN stage's finest map under a neck with start_level = 1.
 * relu_bits_out (optional): u8[count * C / (16 / element size)] — the mask y > 0 as one byte per 16-byte vector of y
 * (bit j = element j of the vector), 1/16 of y's bytes: what the backward of a BatchNorm + residual + ReLU layer needs
 * of y (das_bn_train_backward_bits, DasConvDesc.bnb_mask_bits), so that no backward pass reads y for its mask. */
int das_bn_train_apply(const void* x, void* y, int dtype, long long count, int C, const float* stats,
                       const float* gamma, const float* beta, float* running_mean, float* running_var,
                       float momentum, float eps, const void* residual, int relu, float* save_mean,
                       float* save_invstd, long long* num_batches_tracked, long long stat_count, int stats_slots,
                       void* relu_bits_out, void* stream);
/* das_bn_train_apply with y == NULL (finalize only) for up to four layers in ONE launch: the layers whose statistics are
 * complete at the same point of the forward and whose normalisation is left to one fused consumer (a projection shortcut
 * and its conv3 under das_bn_dual_apply; the two 1x1 convs of an upsample unit under das_upmerge_forward). Same fold
 * order and arithmetic per layer as the single call. count = the population behind `stats` (all ranks' rows for SyncBN). */
typedef struct {
  const float* stats;
  int stats_slots, C;
  long long count;
  float* running_mean;
  float* running_var;
  float momentum, eps;
  float* save_mean;
  float* save_invstd;
  long long* num_batches_tracked;
} DasBnFinalize;
int das_bn_finalize_many(const DasBnFinalize* layers, int n, void* stream);

/* Ragged multi-level pixel rows. The DASHead shares its weights across FPN levels
 * (das_head.py:176-178 `multi_apply(self.forward_single, feats, ...)`), so the head ops below take
 * the rows of ALL levels back to back: level l contributes B*H[l]*W[l] rows in (b,h,w) order. A plain
 * (B,H,W,C) tensor is the special case num_levels = 1. */
/* (typedef DasLevels: see the top of this header) */

/* DCNv2 forward as ONE kernel (round 5): y = ModulatedDeformConv2d(x; offsets / mask logits om; w, bias) without a `col`
 * tensor between the sampling and the GEMM — the sampled pixel operand of a 128-pixel x 256-channel tile goes straight into
 * LDS (csrc/dcn_fused.hip). Same call sites as das_deform_im2col3x3 + das_conv2d_nhwc on col (das_head.py:107-108,
 * anchor_free_mono3d_pose_head.py:111-112,131-132, recursive_update.py:177-178). x: rows x C bf16 (C % 64 == 0); om as for
 * das_deform_im2col3x3; w: (Cout, 9 C) bf16, K = tap * C + channel (= the (Cout, 3, 3, C) channels-last weight); bias f32[Cout
 * padded to 8] or NULL; y: rows x Cout bf16, Cout % 8 == 0, Cout <= 256. col: NULL, or (rows, 9 C) bf16 that ALSO receives
 * the sampled values (bit-identical to das_deform_im2col3x3's; the training forward keeps it for the weight gradient). */
int das_dcn3x3_fused(const void* x, const float* om, const void* w, const float* bias, void* y, void* col, int dtype,
                     const DasLevels* lv, int C, int Cout, int x_pix_stride, int om_pix_stride, int y_pix_stride, void* stream);

/* GroupNorm (+ReLU) over NHWC rows (torch GroupNorm, das_head.py:54, recursive_update.py:178,244);
 * statistics per (level, image, group). stats workspace: f32[num_levels*B*G*2], zeroed by the call unless
 * ws_zeroed (the caller hands it over zeroed: see das_groupnorm_backward_acc). */
int das_groupnorm_nhwc(const void* x, void* y, int dtype, const DasLevels* lv, int C, int pix_stride, int G,
                       const float* gamma, const float* beta, float eps, int relu, float* stats_ws, int ws_zeroed,
                       void* stream);

/* DCNv2 deformable im2col (mmcv ModulatedDeformConv2dPack.forward -> modulated_deform_conv2d,
 * das_head.py:107-108, anchor_free_mono3d_pose_head.py:111-112,131-132, recursive_update.py:177-178).
 * x: rows x C dtype with x_pix_stride. om: rows x om_pix_stride f32, channels [0,18) = (dy,dx)
 * per tap k, [18,27) = mask logits (sigmoid applied here). col: (rows, 9*C) dtype, tap-major,
 * so that DCN == das_conv2d_nhwc(1x1, Cin = 9*C) on col. 3x3, stride 1, pad 1, dilation 1. */
int das_deform_im2col3x3(const void* x, const float* om, void* col, int dtype, const DasLevels* lv, int C,
                         int x_pix_stride, int om_pix_stride, void* stream);

/* Recursive-update offset re-sampling (recursive_update.py:9-82 offset_sample/_core), fused.
 * uvd (rows,uvd_ps) f32 [J*3], samp_off (rows,so_ps) f32 [J*heads*2], conf (rows,conf_ps) f32 [J*3]
 * -> out (rows,out_ps) f32 [J*3]. */
int das_offset_sample(const float* uvd, const float* samp_off, const float* conf, float* out, const DasLevels* lv,
                      int J, int heads, int uvd_ps, int so_ps, int conf_ps, int out_ps, void* stream);

/* off = (1-sigmoid(w))*off + sigmoid(w)*nxt  (recursive_update.py:193-195), per pixel over C
 * channels, all f32 with their own pixel strides. */
int das_sigmoid_blend(const float* off, const float* w, const float* nxt, float* out, long long npix, int C,
                      int off_ps, int w_ps, int nxt_ps, int out_ps, void* stream);

/* DASHead.forward_single tail (das_head.py:237-262): per-level Scale, root-joint pinning and,
 * in eval mode, depth/stride/z_norm rescale. raw: (rows, raw_ps) f32 with channel slices
 * off@off_c(2), depth@depth_c(1), uvd@uvd_c(3J), sigma@sigma_c(3J).
 * pose_pred: (rows, 3+6J) f32 = [off(2), depth, uvd(3J), sigma(3J)]; uvd_out: (rows,3J) f32 =
 * scaled+pinned initial uvd (input of the recursive-update branch). scale[l] = the level's four
 * `Scale` values (offset, depth, uv, d); level_stride[l] = head stride of level l.
 * scale_dev (optional): the same values as f32[levels][4] in DEVICE memory, read by the kernels instead of scale[][] — a
 * training step then needs no device-to-host copy of parameters the optimizer has just written (a host synchronisation
 * with the end of the previous step, every step). */
typedef struct {
  int J, root_idx, raw_ps, off_c, depth_c, uvd_c, sigma_c;
  float scale[5][4];
  float level_stride[5];
  float z_norm, depth_factor;
  const float* scale_dev;
} DasHeadDesc;
int das_head_assemble(const float* raw, float* pose_pred, float* uvd_out, const DasLevels* lv,
                      const DasHeadDesc* d, void* stream);
/* Eval-mode overwrite (das_head.py:254-262): uvd := ref_uvd (root z = 0), u,v *= stride,
 * dz *= z_norm, depth /= depth_factor. In train mode only pins ref root z (das_head.py:254). */
int das_head_finalize(float* pose_pred, float* ref_uvd, const DasLevels* lv, const DasHeadDesc* d, int ref_ps,
                      int eval_mode, void* stream);

/* Gradients of the head ops above. Scatter-type gradients accumulate into caller-zeroed f32 buffers:
 * dom (rows, dom_pix_stride) f32 in the om channel order, d_uvd (rows,3J), d_samp_off (rows,8J),
 * d_conf (rows,3J) dense f32. d_raw must be zero-filled by the caller; d_scale f32[5][4] is zeroed by the
 * call. dx (rows, C) dense f32 of das_deform_im2col3x3_backward is WRITTEN by the call (every element; a
 * gather over the (position, tap) pairs around each pixel, atomics only for offsets beyond two pixels). */
int das_deform_im2col3x3_backward(const void* x, const float* om, const void* dcol, float* dx, float* dom, int dtype,
                                  const DasLevels* lv, int C, int x_pix_stride, int om_pix_stride,
                                  int dom_pix_stride, void* stream);
int das_offset_sample_backward(const float* uvd, const float* samp_off, const float* conf, const float* grad_out,
                               float* d_uvd, float* d_samp_off, float* d_conf, const DasLevels* lv, int J, int heads,
                               int uvd_ps, int so_ps, int conf_ps, int gout_ps, void* stream);
int das_sigmoid_blend_backward(const float* off, const float* w, const float* nxt, const float* grad_out, float* d_off,
                               float* d_w, float* d_nxt, long long npix, int C, int off_ps, int w_ps, int nxt_ps,
                               void* stream);
int das_head_assemble_backward(const float* raw, const float* d_pose, const float* d_uvd, float* d_raw,
                               float* d_scale, const DasLevels* lv, const DasHeadDesc* d, void* stream);

/* ------------------------------------------------------------------------------------
 * Dense-head losses and the optimizer side of the train step.
 *
 * das_assign_targets: DASHead._get_target_single (das_head.py:551-651) + the stride normalisation of
 * get_targets (:547) for every row (level-major, image, h, w) in one launch. gt: f32 rows
 * [cx, cy, depth, J x (u,v,dz), J x vis] of all images back to back, gt_start int32[B+1] = first row of
 * each image. Outputs: labels int32 (0 = person, `background` otherwise), targets f32 (rows, 3+4J) =
 * [dx/stride, dy/stride, depth, J x (du,dv,dz), J x vis] of the chosen person, centerness f32.
 * centers (optional, NULL = gt[:, :3]): f32 rows [centers2d.x, centers2d.y, depths] per person — the
 * reference's separate `centers2d` / `depths` arguments (das_head.py:570-577): root offsets, centre box,
 * nearest-centre choice and the depth target come from them, the joint offsets from gt[:, :3].
 */
typedef struct {
  int J, background;
  int stride[5];
  float range_lo[5], range_hi[5];
  float radius, alpha;
} DasTargetDesc;
/* counts (optional, f32[3], zeroed by the caller): += [positives, positives with a depth annotation (some joint's dz != 0),
 * sum of the visibilities of the positives' joints] — the host-side branch conditions of DASHead.loss (das_head.py:385-392,
 * 473-478) without a dozen mask / sum launches. */
int das_assign_targets(const DasLevels* lv, const DasTargetDesc* d, const float* gt, const float* centers,
                       const int* gt_start, int* labels, float* targets, float* centerness, float* counts, void* stream);
/* The positive rows of the pose losses (das_head.py:385-409) from das_assign_targets' outputs: pos = the npos rows with
 * label 0 in ascending order. real f32 (npos, J, 3) = pixel-to-joint targets (image offsets in units of the level's stride,
 * depth in units of z_norm), vis f32 (npos, J), is2d int32 (npos) = no joint carries a depth annotation, slot int32 (npos) =
 * rank of the row among the positives of its kind (its block of J rows in the flows' input), depth_t f32 (npos) = depth
 * target * depth_factor, ctr_pos f32 (npos) = centerness targets of the rows, nvis f32[1] = sum(vis) * nvis_scale. */
int das_positive_rows(const int* pos, int npos, const float* targets, const float* centerness, const DasLevels* lv,
                      const DasTargetDesc* d, float z_norm, float depth_factor, float nvis_scale, float* real, float* vis,
                      int* is2d, int* slot, float* depth_t, float* ctr_pos, float* nvis, void* stream);
/* mmdet FocalLoss(use_sigmoid) / mmcv sigmoid_focal_loss for one class (das_head.py:341-344): per-row
 * gradient + sum of the per-row loss (loss_sum zeroed by the call). logits: row i at logits[i*pix_stride].
 * weight (here and in the two losses below): optional f32 per-element factor, mmdet's `weight` argument
 * (weight_reduce_loss); NULL = 1. */
int das_sigmoid_focal_loss(const float* logits, int pix_stride, const int* labels, const float* weight, long long rows,
                           float gamma, float alpha, float* grad, float* loss_sum, void* stream);
/* mmdet SmoothL1Loss (das_head.py:375-379) and CrossEntropyLoss(use_sigmoid) (:470): elementwise gradient
 * and loss sum over n dense f32 elements (loss_sum zeroed by the call). */
int das_smooth_l1_loss(const float* pred, const float* target, const float* weight, long long n, float beta,
                       float* grad, float* loss_sum, void* stream);
int das_bce_logits_loss(const float* logits, const float* target, const float* weight, long long n, float* grad,
                        float* loss_sum, void* stream);
/* RealNVP log-density of the RLE pose loss (mmdet3d/models/losses/real_nvp.py:60-80 `log_prob`, called from
 * das_head.py:425-446 on (pred - gt) / sigma of every positive x joint). x f32[N][D], D = 3 (or 2); `layers`
 * coupling layers, mask bit (i*D + d) of mask_bits = mask[i][d] (1 = passed through). params f32: per layer
 * [t-net | s-net], net = W1[64][D] b1[64] W2[64][64] b2[64] W3[D][64] b3[D] (nn.Linear layouts; LeakyReLU(0.01)
 * between, Tanh after the s-net). Outputs: logp f32[N] and the final latent z f32[N][D] (all the backward
 * needs: the coupling layers are inverted on the way back).
 * Backward: grad_logp f32[N] -> dx f32[N][D] and the parameter gradients, delivered EITHER into dparams
 * (same layout as params, zeroed by the call) OR added in place through dst_table, a device array of
 * 2*layers*6 pointers (one per tensor in the params order: the optimizer's flat-gradient slices). Exactly one
 * of dparams / dst_table is non-NULL. */
int das_realnvp_log_prob(const float* x, int N, int D, const float* params, int layers, unsigned mask_bits,
                         float* logp, float* z_out, void* stream);
/* Several flows of the same dimension in ONE launch (the head evaluates `flow3d` on the recursive-update
 * prediction and `flow3d_update`-less `flow3d` on the direct one, das_head.py:425-446): x / logp / z / dx hold
 * the jobs' rows back to back, job q owning rows [row_start, row_end) with row_start a multiple of 256 (pad the
 * gap; padded rows are ignored) and jobs[0].row_start == 0. params / dparams / dst_table as in the single-flow
 * calls (backward: exactly one of dparams / dst_table per job). */
#define DAS_FLOW_MAX_JOBS 4
typedef struct {
  const float* params;
  float* dparams;
  float* const* dst_table;
  int row_start, row_end;
} DasFlowJob;
int das_realnvp_log_prob_multi(const float* x, int rows_total, int D, const DasFlowJob* jobs, int njobs, int layers,
                               unsigned mask_bits, float* logp, float* z_out, void* stream);
int das_realnvp_log_prob_multi_backward(const float* z_final, const float* grad_logp, int rows_total, int D,
                                        const DasFlowJob* jobs, int njobs, int layers, unsigned mask_bits, float* dx,
                                        void* stream);
int das_realnvp_log_prob_backward(const float* z_final, const float* grad_logp, int N, int D, const float* params,
                                  int layers, unsigned mask_bits, float* dx, float* dparams,
                                  float* const* dst_table, void* stream);

/* The RLE pose loss around the flows and the depth term, on the positive locations only (replaces the tensor algebra of
 * mmdet3d/models/pose_heads/das_head.py:375-381 (depth: SmoothL1 on the root depth of the 3-D positives) and :385-466
 * (pose) with mmdet3d/models/losses/residual_log_likelihood_loss.py:17-37 (`RLELoss3D.forward`, `logQ`)).
 * pose f32[rows][pose_ps] = [root offset 2 | depth | J x uvd | J x sigma logits] and aux f32[rows][aux_ps] = the refined
 * J x uvd of the recursive-update branch are the head's dense outputs; pos i64[npos] the rows of the positives (distinct);
 * real f32[npos][J][3] the normalised pixel-to-joint targets, vis f32[npos][J], is2d i32[npos] (1 = sample without
 * depth: its z offsets count as 0 and its z sigma logit as 1, :388-391), slot i32[npos] = rank of the positive among
 * the positives of its kind, depth_t f32[npos] the depth targets. Prediction set 0 = aux, set 1 (sets == 2, `prev_loss`)
 * = pose's own uvd. The flow inputs are written in the layout das_realnvp_log_prob_multi reads: x2 f32[sets*stride2][2]
 * for the 2-D samples and x3 f32[sets*stride3][3] for the 3-D ones, row = set * stride + slot * J + j (strides:
 * multiples of 256), with w2 / w3 f32[...] = -3 * vis, the row's weight in the loss (d loss / d log_phi before the
 * upstream scalar); rows no positive owns are left untouched (the caller zeroes the buffers).
 * das_rle_loss: partials f32[das_rle_blocks(d)][2] = per-workgroup sums of vis * (log sigma - log_phi + logQ) over
 * (positive, joint, set, dim) and of the depth term's smooth-L1 values; the caller adds them up in index order and
 * applies code weights / normalisers (the loss value is then independent of the launch).
 * das_rle_backward: g_sums f32[2] (device) = upstream gradients of the two sums; dx2 / dx3 = the flows' input
 * gradients for grad_logp = w * g_sums[0]; writes the positives' rows of dpose / daux (same strides; every element
 * written once, the caller provides zeroed buffers for the other rows). */
typedef struct {
  int J, sets, npos;
  int pose_ps, aux_ps;
  int stride2, stride3;
  float amp;    /* RLELoss3D.amp = 1 / sqrt(2 pi) */
  float beta;   /* SmoothL1Loss.beta of the depth term */
} DasRleDesc;
int das_rle_blocks(const DasRleDesc* d);
int das_rle_prepare(const float* pose, const float* aux, const long long* pos, const float* real, const float* vis,
                    const int* is2d, const int* slot, const DasRleDesc* d, float* x2, float* w2, float* x3, float* w3,
                    void* stream);
int das_rle_loss(const float* pose, const float* aux, const long long* pos, const float* real, const float* vis,
                 const int* is2d, const int* slot, const float* depth_t, const float* logp2, const float* logp3,
                 const DasRleDesc* d, float* partials, void* stream);
int das_rle_backward(const float* pose, const float* aux, const long long* pos, const float* real, const float* vis,
                     const int* is2d, const int* slot, const float* depth_t, const float* dx2, const float* dx3,
                     const float* g_sums, const DasRleDesc* d, float* dpose, float* daux, void* stream);

/* out (+)= sum g^2 over a flat f32 gradient buffer (global-norm clipping, exp_panoptic.py:204-205). */
int das_grad_sumsq(const float* g, long long n, float* out, int zero_first, void* stream);
/* torch.optim.SGD step (momentum, weight decay) on flat f32 buffers with the clip coefficient
 * min(1, max_norm/(sqrt(*grad_sumsq)*grad_scale + 1e-6)) and grad_scale folded in; max_norm <= 0 disables
 * clipping. first_step != 0 initialises the momentum buffer with the gradient (as torch does). */
int das_sgd_momentum_step(float* p, const float* g, float* buf, long long n, float lr, float momentum,
                          float weight_decay, float grad_scale, float max_norm, const float* grad_sumsq,
                          int first_step, void* stream);

/* ------------------------------------------------------------------------------------
 * Decode (das_head.py:690-796 `_get_poses_single`, pose_nms.py:51-126 oks_iou / oks_nms), fused:
 * per image  score = sigmoid(cls)*sigmoid(ctr) -> keep score > score_thr -> per level keep the
 * nms_pre best when the level has more than nms_pre locations -> center = (point - offset)/scale,
 * joints = (uvd + root)/scale, depth *= sqrt(sx*sy) -> order by (score desc, flat location index
 * asc) -> greedy OKS-NMS keeping oks <= nms_thr (f64 arithmetic, f32 compare, as numpy does) ->
 * first nms_post survivors. Inputs are the eval-mode head outputs, f32, NHWC per level:
 * cls/ctr (B,H,W,*_ps) logits in channel 0, pose (B,H,W,pose_ps) = [off2, depth, uvd 3J, ...].
 * Outputs (kept order, row b*nms_post + k): scores, poses (J,3), centers (3), index = flat
 * location index over levels fine->coarse; out_count[b] = number kept.
 */
#define DAS_MAX_LEVELS 5
typedef struct {
  int B, J, num_levels;
  int H[DAS_MAX_LEVELS], W[DAS_MAX_LEVELS], stride[DAS_MAX_LEVELS];
  const float* cls[DAS_MAX_LEVELS];
  const float* ctr[DAS_MAX_LEVELS];
  const float* pose[DAS_MAX_LEVELS];
  int cls_ps[DAS_MAX_LEVELS], ctr_ps[DAS_MAX_LEVELS], pose_ps[DAS_MAX_LEVELS];
  int nms_pre, nms_post;
  float score_thr, nms_thr;
  const float* scale_factor; /* device f32[B*2]: (sx, sy) per image */
  int nms_soft; /* != 0: soft OKS-NMS (pose_nms.py:128-194 soft_oks_nms, the nms_type != 'hard' branch of
                 * das_head.py:784-790): nms_post rounds of "take the best remaining score, multiply every other
                 * remaining score by exp(-oks^2 / nms_thr)"; the scores returned are the original ones */
} DasDecodeDesc;
/* candidate capacity per image = sum over levels of min(H*W, nms_pre) (H*W for nms_pre <= 0).
 * No size limits, as in the reference (das_head.py:716-723 cuts a level to nms_pre only when it has more points): any number of
 * locations per level (what the kernel keeps in LDS are the locations ABOVE score_thr — up to 16 384 per level there, beyond that
 * the level's nms_pre-th key comes from a radix select over re-computed scores) and any capacity (beyond 4096 candidates the
 * suppression flags, beyond 16 384 the globally sorted keys live in the workspace). Only the total number of locations must fit
 * 31 bits (the flat location index). The workspace (das_decode_ws_bytes, device memory, uninitialised) holds per image the merged
 * keys padded to a power of two, the candidates' joints / areas / centres, soft-NMS scores and suppression flags. */
int das_decode_cap(const DasDecodeDesc* d);
long long das_decode_ws_bytes(int B, int cap, int J);
int das_decode(const DasDecodeDesc* d, float* out_scores, float* out_poses, float* out_centers, int* out_index,
               int* out_count, void* ws, void* stream);

/* ---- Image half of the pose data pipeline as a GPU-side augmentation stage: f32 HWC images,
 * BGR as mmcv loads them. Each entry restates the OpenCV / mmcv op the reference pipeline calls
 * (mmdet3d/datasets/pipelines/transforms_3d.py, configs/das/exp_panoptic.py:59-98):
 *   das_img_resize_bilinear     mmcv.imrescale / imresize -> cv2.resize(INTER_LINEAR) on float images (ResizePose :19-61)
 *   das_img_flip_horizontal     mmcv.imflip (RandomFlipPose3D :235-356)
 *   das_img_photometric         mmdet PhotoMetricDistortion (brightness, contrast first / last, saturation and hue
 *                               through OpenCV's float BGR<->HSV, channel permutation), in place, C = 3
 *   das_img_warp_affine         cv2.warpAffine(INTER_LINEAR, BORDER_CONSTANT, borderValue) with the FORWARD 2x3 map M
 *                               (GlobalRotScaleTransPose :973-986), C = 3
 *   das_img_normalize_pad_chw   mmcv.imnormalize (BGR->RGB, cv2.subtract / cv2.multiply with f64 scalars) + Pad(0) +
 *                               HWC->CHW (Normalize, Pad, DefaultFormatBundlePose3D formating.py:383-442) */
typedef struct {
  int use_brightness, use_contrast, contrast_first, use_saturation, use_hue;
  float brightness, contrast, saturation, hue;
  int perm[3]; /* output channel c takes channel perm[c] */
} DasPhotometric;
typedef struct {
  double inv[6];
  float border[3];
} DasAffine; /* (internal) */
typedef struct {
  double mean[3], stdinv[3];
  int to_rgb, mean_f64, std_f64;
} DasNormalize; /* (internal) */
int das_img_resize_bilinear(const float* src, float* dst, int Hs, int Ws, int Hd, int Wd, int C, void* stream);
int das_img_resize_bilinear_u8(const unsigned char* src, unsigned char* dst, int Hs, int Ws, int Hd, int Wd, int C,
                               void* stream); /* 8-bit images: OpenCV's fixed-point INTER_LINEAR (test pipeline) */
int das_img_flip_horizontal(const float* src, float* dst, int H, int W, int C, void* stream);
int das_img_photometric(float* img, int H, int W, const DasPhotometric* p, void* stream);
int das_img_warp_affine(const float* src, float* dst, int Hs, int Ws, int Hd, int Wd, const double* M, const float* border,
                        void* stream);
int das_img_normalize_pad_chw(const float* src, float* dst, int H, int W, int Hp, int Wp, const double* mean,
                              const double* std, int to_rgb, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DAS_HIP_H */
