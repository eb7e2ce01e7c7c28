/*
 * bdvcil_hip.h -- C ABI of libbdvcil_hip.so: MI355X (gfx950) kernels for the TSM
 * class-incremental training hot path of NinV/Background-Debiased-Video-CIL.
 *
 * The reference has NO FFI for this path: its hot path is `mmaction.models` Python executed by
 * PyTorch/cuDNN.  Each entry point below therefore names the reference
 * call site (file:line under /root/reference, or UPSTREAM mmaction2 0.24 / torch op) whose
 * arithmetic it replaces.  INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success, a negative BDV_E* code for argument errors or a positive
 *     hipError_t for launch errors; nothing throws; bdv_last_error() gives a thread-local message.
 *   - the caller owns every buffer (incl. workspaces); pointers are device pointers unless noted;
 *     kernels are enqueued on `stream` (a hipStream_t passed as void*) and never synchronise.
 *   - no global mutable state: re-entrant, thread-safe, hipGraph-capturable.
 *   - activations are fp32 NHWC: [N][H][W][C], N = clips * segments (frames), C % 4 == 0,
 *     16-byte aligned.  Conv weights are [Cout][R][S][Cin] (the storage of a torch
 *     channels_last OIHW tensor).  All arithmetic is fp32 (f32-input MFMA, exact fp32 FMA chains).
 */
#ifndef BDVCIL_HIP_H
#define BDVCIL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BDV_OK 0
#define BDV_EINVAL (-1)   /* bad shape / alignment / null pointer */
#define BDV_EWORKSPACE (-2) /* workspace too small */

/* Activation storage (`act_dtype` arguments, bdv_conv_geom.act_dtype).  BDV_ACT_F32: every activation / gradient tensor is
 * fp32 (the reference's arithmetic, libs/cil/cil.py:744-756 precision 32; the default and the headline metric).
 * BDV_ACT_BF16: activations, conv outputs and their gradients are stored as bf16 (BASELINE config 5, the reference's
 * `precision=16` counterpart; bf16 instead of fp16: same MFMA rate on gfx950, fp32's exponent range, no loss scaler).  Values
 * widen exactly on load, arithmetic and accumulators stay fp32, stores round to nearest-even; BatchNorm statistics, weights,
 * weight gradients, ReLU masks and everything after the average pool stay fp32.  Convolutions then run the single-product bf16
 * MFMA kernels (pieces = 1) only. */
#define BDV_ACT_F32 0
#define BDV_ACT_BF16 1

/* Geometry of one 2-D convolution site (UPSTREAM mmaction ResNet ConvModule.conv). */
typedef struct bdv_conv_geom {
  int32_t N;      /* frames = clips * T */
  int32_t H, W;   /* input spatial size */
  int32_t Cin;    /* input channels, multiple of 4 */
  int32_t Ho, Wo; /* output spatial size */
  int32_t Cout;   /* output channels, multiple of 64 */
  int32_t R, S;   /* filter size */
  int32_t stride; /* 1 or 2 */
  int32_t pad;
  int32_t T;      /* num_segments (frames per clip); used only when fold > 0 */
  int32_t fold;   /* temporal-shift fold = Cin / shift_div, multiple of 4; 0 = no shift */
  int32_t pad_w;  /* column padding when it differs from `pad` (then `pad` pads rows only); < 0: same as `pad`.
                   * A k x 1 x 1 temporal convolution of an I3D block (UPSTREAM mmaction ResNet3d Bottleneck3d.conv1,
                   * configs/_base_/models/i3d_r50.py:13) runs as a k x 1 conv on the [B][T][H*W][C] view of the same NHWC
                   * storage: N = B, H = T, W = H*W, R = k, S = 1, pad = k / 2, pad_w = 0. */
  int32_t Rt;     /* temporal filter taps of the I3D stem (UPSTREAM ResNet3d.conv1, conv1_kernel = (5, 7, 7),
                   * configs/_base_/models/i3d_r50.py:7); 0 or 1 = a 2-D convolution.  Rt > 1: Cin = 4 only, the bf16-piece
                   * kernels only (bdv_conv_fprop_x3, bdv_conv_wgrad_partial_pl).  N = OUTPUT frames, T = output frames per
                   * clip; the input holds N * st_t frames and output frame n reads the input frames n * st_t + dt - Rt / 2 of
                   * its clip (zeros outside).  Weights [Cout][Rt][R][S][4]. */
  int32_t st_t;   /* temporal stride (1 or 2) when Rt > 1 */
  int32_t act_dtype; /* BDV_ACT_F32 | BDV_ACT_BF16: element type of x / y / dy / dx / add_src and of the fused-statistics `y`
                      * (the plane kernels bdv_conv_fprop_pl / _dgrad_pl / _wgrad_partial_pl with pieces = 1 only; Cin % 32 == 0) */
} bdv_conv_geom;

const char* bdv_last_error(void);
int bdv_abi_version(void);
/* sha256 (hex) of the sources this library was built from, in the Makefile's HASHED order; the binding compares it with
 * the sources it finds in-tree and refuses a stale build. */
const char* bdv_source_hash(void);

/* ---- convolution = implicit GEMM on v_mfma_f32_32x32x2_f32 --------------------------------
 * fprop replaces F.conv2d inside ConvModule (+ UPSTREAM TemporalShift.shift when fold > 0:
 * channels [0,fold) read frame t+1, [fold,2fold) read frame t-1, zero at clip ends).
 * x [N,H,W,Cin], w [Cout,R,S,Cin], y [N,Ho,Wo,Cout].
 *
 * Workspaces: bdv_conv_workspace_bytes(g, kind) with kind 0 = fprop, 1 = dgrad, 2 = wgrad.  fprop/dgrad use it
 * for the K-split partial accumulators of the remainder tiles (tile counts that do not fill whole rounds of
 * co-resident workgroups on the 256 CUs); passing NULL disables the split (correct, slower on such shapes). */
size_t bdv_conv_workspace_bytes(const bdv_conv_geom* g, int kind);
/* bn_partial (optional): float[2][rows][Cout], rows = bdv_conv_fprop_stat_rows(g).  When given, the epilogue also
 * writes per-row-tile column sums of y and y*y (the BatchNorm batch statistics, fused: y is not re-read);
 * bdv_bn_train_finalize reduces them in fixed order. */
int bdv_conv_fprop_stat_rows(const bdv_conv_geom* g);
/* affine (optional, excludes bn_partial): eval-mode BatchNorm folded into the epilogue,
 * y = relu?(conv * scale[c] + shift[c] (+ residual)) with scale/shift from bdv_bn_eval_params: the inference and
 * frozen-teacher forwards (UPSTREAM Recognizer2D._do_test, libs/cil/cil.py:520-522) then have no bn_apply pass. */
typedef struct bdv_conv_affine {
  const float* scale;    /* [Cout] */
  const float* shift;    /* [Cout] */
  const void* residual;  /* [N,Ho,Wo,Cout] or NULL (element type: the geometry's act_dtype) */
  int32_t relu;
} bdv_conv_affine;
int bdv_conv_fprop(const float* x, const float* w, float* y, const bdv_conv_geom* g, float* bn_partial,
                   const bdv_conv_affine* affine, void* workspace, size_t workspace_bytes, void* stream);
/* EXPERIMENTAL (off by default in the Python layer, BDVCIL_FPROP_X3=1): the same convolution with every fp32 product formed
 * from bf16 pieces -- each operand value is split in the loader into three round-to-nearest bf16 terms and six
 * v_mfma_f32_32x32x16_bf16 products per K-step are accumulated in fp32 (error of the order of fp32
 * rounding).  Same arguments, workspace and epilogues as bdv_conv_fprop; used for Cout % 128 == 0 and Cin % 32 == 0,
 * other shapes run the fp32-MFMA kernels. */
int bdv_conv_fprop_x3(const float* x, const float* w, float* y, const bdv_conv_geom* g, float* bn_partial,
                      const bdv_conv_affine* affine, void* workspace, size_t workspace_bytes, void* stream);

/* BatchNorm-backward statistics fused into dgrad: dx (the gradient w.r.t. the BN(+ReLU) output of the PREVIOUS conv unit,
 * whose saved conv output is y) is reduced in the dgrad epilogue to partial[0][r][c] = sum(g), partial[1][r][c] =
 * sum(g * xhat) per 128-row tile r (g = dx * mask, xhat = (y - mean) * invstd; rows = bdv_conv_dgrad_stat_rows(g));
 * bdv_bn_backward(stat_partial = ...) then skips its own statistics pass.  Stride 2 needs a filter that
 * reaches every input pixel (R, S >= 2); the partial then has one block of rows per input-parity class
 * (bdv_conv_dgrad_stat_rows / bdv_conv_dgrad_pl_stat_rows give the total) and is zeroed by the call. */
typedef struct bdv_bn_stat_fuse {
  const void* y;             /* [N,H,W,Cin] conv output of the previous unit (element type: the geometry's act_dtype) */
  const uint32_t* relu_mask; /* 1 bit per element of dx, or NULL (no ReLU, or the sign is derived: relu_scale) */
  const float* mean;         /* [Cin] saved batch mean */
  const float* invstd;       /* [Cin] */
  float* partial;            /* float[2][rows][Cin] */
  const float* relu_scale;   /* optional [Cin] pair with relu_mask == NULL: the unit's ReLU sign is y * relu_scale + relu_shift > 0 */
  const float* relu_shift;   /* (units whose activation and mask were never written: bdv_conv_fprop_pl(pre_scale)) */
} bdv_bn_stat_fuse;
int bdv_conv_dgrad_stat_rows(const bdv_conv_geom* g);

/* dgrad: dx[N,H,W,Cin] = unshift(conv_transpose(dy, w)) + (add_src ? add_src * mask : 0),
 * mask = bit e of add_mask_src (the ReLU sign mask written by bdv_bn_apply) when add_mask_src != NULL
 * (fused ReLU-backward of the identity path).
 * Replaces autograd of F.conv2d w.r.t. its input and of TemporalShift.shift. */
int bdv_conv_dgrad(const float* dy, const float* w, float* dx, const float* add_src,
                   const uint32_t* add_mask_src, const bdv_conv_geom* g, const bdv_bn_stat_fuse* bn_stat, void* workspace,
                   size_t workspace_bytes, void* stream);
/* EXPERIMENTAL counterpart of bdv_conv_fprop_x3 for the data gradient.  w_t = the weights transposed per filter tap,
 * (R*S, Cin, Cout) contiguous, so that both operands are contiguous along the contraction; used when Cin % 128 == 0,
 * other shapes run bdv_conv_dgrad's kernels on w. */
int bdv_conv_dgrad_x3(const float* dy, const float* w, const float* w_t, float* dx, const float* add_src,
                      const uint32_t* add_mask_src, const bdv_conv_geom* g, const bdv_bn_stat_fuse* bn_stat, void* workspace,
                      size_t workspace_bytes, void* stream);

/* Weights as bf16 planes for the default conv arithmetic (every fp32 product from three bf16 pieces per operand, six
 * v_mfma_f32_32x32x16_bf16 products, fp32 accumulate; replaces the same F.conv2d / autograd call sites as bdv_conv_fprop and
 * bdv_conv_dgrad).  bdv_conv_split_weights cuts w [Cout,R,S,Cin] once per optimizer step into hi / mid / lo planes
 * (hi = bf16(w), mid = bf16(w - hi), lo = bf16(w - hi - mid), round to nearest), each of
 * bdv_conv_weight_planes_bytes(g) / 3 bytes, in the two layouts the kernels stream:
 *   planes_fprop  [piece][K-step = (ci / 32) * R*S + tap][co][ci % 32]
 *   planes_dgrad  [piece][tap][co / 32][ci][co % 32]
 * (either may be NULL; Cin and Cout multiples of 32).  bdv_conv_fprop_pl / bdv_conv_dgrad_pl take the planes beside w and
 * behave like bdv_conv_fprop / bdv_conv_dgrad (same epilogues, workspace = bdv_conv_workspace_bytes); shapes their
 * 8-wave kernels do not cover (Cout resp. Cin not a multiple of 128, the stem) run the other kernels on w.  The fused
 * statistics partials have bdv_conv_fprop_pl_stat_rows(g, pieces) / bdv_conv_dgrad_pl_stat_rows(g, pieces) rows (one per row
 * tile of the kernel that will run, 128 or 256 rows).
 * pieces = 3 is the arithmetic described above.  pieces = 1 is the REDUCED-PRECISION arithmetic of BASELINE config 5 ("MFMA
 * fp16 tiles"; the reference's trainer is precision 32, libs/cil/cil.py:744-756, so there are no reference numerics for it):
 * each operand value is rounded to bf16 (only the hi plane is used), one v_mfma_f32_32x32x16_bf16 product per step, fp32
 * accumulate, fp32 tensors in HBM -- what torch.autocast(bfloat16) computes for a convolution, without the bf16 output rounding.
 * Error of a result: ~2^-9 relative per product, i.e. ~1e-3 .. 1e-2 of the output scale (tests/test_bf16x1_gpu.py).
 * pieces = 2 is the arithmetic in between ("bf16x2"): the hi and mid planes of each operand (16 significand bits) and the three
 * products hi*hi + hi*mid + mid*hi per step, fp32 accumulate, fp32 tensors; dropped terms <= 2^-15 relative per product (five more
 * bits than TF32, the conv arithmetic torch.backends.cudnn.allow_tf32 gives the reference on its GPUs); never the default, the
 * plane kernels only (other shapes run the pieces = 3 kernels), no bf16 storage (tests/test_bf16x2_gpu.py). */
size_t bdv_conv_weight_planes_bytes(const bdv_conv_geom* g);
int bdv_conv_split_weights(const float* w, const bdv_conv_geom* g, void* planes_fprop, void* planes_dgrad, void* stream);
/* Test / A-B hook, process-wide and not thread-safe: force the tile configuration of the two entry points below
 * (0 = 128x256, 1 = 256x128, 2 = 256x256, 3 = 256x64, 4 = 64x128 (dgrad, stride 1), 5 = 128x128 with four waves and two
 * workgroups per CU, where the tile divides the column count; 6 = none of them; -1 = the planner's rules). */
int bdv_conv_debug_force_tile(int cfg);
/* 1 when bdv_conv_fprop_pl (kind 0) / bdv_conv_dgrad_pl (kind 1) will read the weight planes for this geometry, 0 when it
 * runs a kernel that takes w (the caller then need not build the planes). */
int bdv_conv_uses_planes(const bdv_conv_geom* g, int kind, int pieces);
/* Name of the main kernel that a call with this geometry launches, as a profiler prints it (kind 0 fprop, 1 dgrad, 2 wgrad;
 * arith 0 = the fp32-MFMA entry points, 1 = the *_pl entry points, 2 = the same with pieces = 1, 3 = with pieces = 2).  For profiles and per-kernel
 * accounting. */
int bdv_conv_kernel_name(const bdv_conv_geom* g, int kind, int arith, char* out, size_t n);
int bdv_conv_fprop_pl_stat_rows(const bdv_conv_geom* g, int pieces);
int bdv_conv_dgrad_pl_stat_rows(const bdv_conv_geom* g, int pieces);
/* pre_scale / pre_shift (optional, [Cin] each): x is then the RAW output of the producing conv and the producer's train-mode
 * BatchNorm + ReLU, a = max(x * pre_scale[ci] + pre_shift[ci], 0), is applied in the loader (the same fused multiply-add as
 * bdv_bn_apply, bit-identical activations; halo / ragged lanes stay zero): the apply pass between the two convs, the activation
 * tensor and its ReLU mask are not needed for this consumer (UPSTREAM ConvModule conv -> bn -> relu chains inside Bottleneck /
 * BasicBlock).  Needs pieces = 3, no temporal shift, and bdv_conv_fprop_pre_ok(g); the statistics partial then has
 * bdv_conv_fprop_pre_stat_rows(g) rows. */
/* g->act_dtype = BDV_ACT_BF16: x, y and affine->residual are bf16 tensors (pieces = 1 only; the accumulators, the fused batch
 * statistics and the folded BatchNorm arithmetic stay fp32, y is rounded to nearest-even on store); same for dy / dx / add_src /
 * bn_stat->y of bdv_conv_dgrad_pl and dy / x of bdv_conv_wgrad_partial_pl (weight-gradient slabs stay fp32). */
int bdv_conv_fprop_pl(const void* x, const float* w, const void* planes_fprop, void* y, const bdv_conv_geom* g,
                      float* bn_partial, const bdv_conv_affine* affine, void* workspace, size_t workspace_bytes, int pieces,
                      const float* pre_scale, const float* pre_shift, void* stream);
int bdv_conv_fprop_pre_ok(const bdv_conv_geom* g);
int bdv_conv_fprop_pre_stat_rows(const bdv_conv_geom* g);
int bdv_conv_dgrad_pl(const void* dy, const float* w, const void* planes_dgrad, void* dx, const void* add_src,
                      const uint32_t* add_mask_src, const bdv_conv_geom* g, const bdv_bn_stat_fuse* bn_stat, void* workspace,
                      size_t workspace_bytes, int pieces, void* stream);

/* wgrad: dw[Cout,R,S,Cin] = beta * dw + sum_pixels dy (x) shift(x).  Deterministic split-K:
 * partial slabs go to `workspace`, a second kernel reduces them in fixed order. */
int bdv_conv_wgrad(const float* dy, const float* x, float* dw, float beta, const bdv_conv_geom* g, void* workspace,
                   size_t workspace_bytes, void* stream);
/* The same weight gradient in two steps, so that the split-K reductions of several layers run as ONE launch (a ResNet
 * stage's backward issues one instead of up to 19): bdv_conv_wgrad_partial leaves bdv_conv_wgrad_splits(g) partial
 * products of dw's size in `slab` (caller-owned, alive until reduced); bdv_wgrad_reduce_batched computes
 * dws[k] = beta * dws[k] + sum of the splits[k] slices of slabs[k] for n <= BDV_MAX_REDUCE_ITEMS items (HOST arrays;
 * numels[k] = elements of dws[k], a multiple of 4), with the per-element summation order of bdv_conv_wgrad. */
#define BDV_MAX_REDUCE_ITEMS 32
int bdv_conv_wgrad_splits(const bdv_conv_geom* g);
int bdv_conv_wgrad_partial(const float* dy, const float* x, const bdv_conv_geom* g, void* slab, size_t slab_bytes,
                           void* stream);
int bdv_wgrad_reduce_batched(const float* const* slabs, float* const* dws, const int* splits, const int64_t* numels, int n,
                             float beta, void* stream);
/* The weight gradient's main kernel of the default arithmetic for Cout and Cin multiples of 128 (other shapes run the kernels of
 * bdv_conv_wgrad_partial): 8 waves per workgroup, tiles of 128 / 256 output channels x 128 / 256 input channels of one tap,
 * both operand tiles split into bf16 pieces in the loader and read from LDS with ds_read_b64_tr_b16.  The slab holds
 * bdv_conv_wgrad_pl_splits(g) partial products of dw's size; reduce with bdv_wgrad_reduce_batched. */
int bdv_conv_wgrad_pl_splits(const bdv_conv_geom* g);
/* pre_scale / pre_shift as for bdv_conv_fprop_pl: x is the producer's raw conv output and the activation
 * max(x * pre_scale[ci] + pre_shift[ci], 0) is formed in the loader (needs bdv_conv_wgrad_pre_ok(g)). */
int bdv_conv_wgrad_partial_pl(const void* dy, const void* x, const bdv_conv_geom* g, void* slab, size_t slab_bytes,
                              int pieces, const float* pre_scale, const float* pre_shift, void* stream);
int bdv_conv_wgrad_pre_ok(const bdv_conv_geom* g);
/* EXPERIMENTAL counterpart of bdv_conv_fprop_x3 for the weight gradient's main kernel (128x128 tiles, i.e. Cout and Cin
 * multiples of 128; other shapes run the fp32-MFMA kernels).  Same slab layout and split count as bdv_conv_wgrad_partial. */
int bdv_conv_wgrad_partial_x3(const float* dy, const float* x, const bdv_conv_geom* g, void* slab, size_t slab_bytes,
                              void* stream);

/* ---- BatchNorm2d (train + eval), fused with ReLU / residual add --------------------------
 * Replaces UPSTREAM ConvModule.bn (+ .activate, + block `out + identity`) and their autograd.
 * y is [M][C] (M = N*Ho*Wo). */
size_t bdv_bn_workspace_bytes(int64_t M, int C);
/* train: batch mean / biased var -> scale = gamma*invstd, shift = beta - mean*scale; saves mean and
 * invstd; running_mean/var updated with `momentum` (unbiased var), num_batches_tracked handled by
 * the caller. */
int bdv_bn_train_stats(const float* y, int64_t M, int C, const float* gamma, const float* beta,
                       float eps, float momentum, float* running_mean, float* running_var,
                       float* save_mean, float* save_invstd, float* scale, float* shift,
                       void* workspace, size_t workspace_bytes, void* stream);
/* same outputs as bdv_bn_train_stats, from the partial sums written by bdv_conv_fprop(bn_partial) */
int bdv_bn_train_finalize(const float* partial, int rows, int64_t M, int C, const float* gamma, const float* beta,
                          float eps, float momentum, float* running_mean, float* running_var, float* save_mean,
                          float* save_invstd, float* scale, float* shift, void* stream);
/* eval: scale/shift from running statistics. */
int bdv_bn_eval_params(int C, const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, float eps, float* scale, float* shift, void* stream);
/* out = act(y*scale[c] + shift[c] + r), act = ReLU if relu != 0; r = 0 without res, res itself, or -- with
 * res_scale/res_shift -- res*res_scale[c] + res_shift[c] (the block's downsample conv output gets its own BatchNorm
 * here instead of in a separate pass).  relu_mask (optional, needs C % 32 == 0): bit e of relu_mask[] = (out[e] > 0)
 * for flat element index e -- the 1-bit-per-element ReLU sign mask the backward kernels read instead of the fp32
 * activation. */
/* act_dtype (BDV_ACT_F32 | BDV_ACT_BF16): element type of y, res and out. */
int bdv_bn_apply(const void* y, const float* scale, const float* shift, const void* res, const float* res_scale,
                 const float* res_shift, void* out, uint32_t* relu_mask, int64_t M, int C, int relu, int act_dtype, void* stream);
/* backward of (BN-train -> +res -> ReLU): g = dout * (relu_mask bit if relu), dgamma = sum g*xhat,
 * dbeta = sum g, dy = gamma*invstd*(g - dbeta/M - xhat*dgamma/M).  dgamma/dbeta are written as
 * beta_acc*old + new.  The residual-path gradient is g itself; consumers re-derive it from
 * (dout, relu_mask) -- see bdv_conv_dgrad(add_src, add_mask_src) and bdv_relu_bwd. */
/* stat_partial (optional): float[2][stat_rows][C] written by bdv_conv_dgrad(bn_stat); the statistics pass is skipped. */
/* relu_scale / relu_shift (optional [C] pair, with relu != 0 and relu_mask == NULL): the ReLU sign is derived from y as
 * y * relu_scale + relu_shift > 0, the forward's own expression, for units whose mask was never written. */
/* act_dtype: element type of dout, y and dy (statistics, dgamma / dbeta and the coefficients stay fp32). */
int bdv_bn_backward(const void* dout, const uint32_t* relu_mask, const void* y, const float* gamma,
                    const float* save_mean, const float* save_invstd, void* dy, float* dgamma,
                    float* dbeta, float beta_acc, int64_t M, int C, int relu, const float* stat_partial, int stat_rows,
                    const float* relu_scale, const float* relu_shift, void* workspace, size_t workspace_bytes, int act_dtype,
                    void* stream);
/* g = dout * relu_mask (+ add) : masked gradient for an identity path that has no conv behind it */
int bdv_relu_bwd(const void* dout, const uint32_t* relu_mask, const void* add, void* g, int64_t numel, int act_dtype, void* stream);
/* out = a + b (gradient junctions) */
int bdv_add(const void* a, const void* b, void* out, int64_t numel, int act_dtype, void* stream);

/* ---- stem helpers ------------------------------------------------------------------------ */
/* (N,3,H,W) fp32 -> (N,H,W,4) fp32, 4th channel 0: boundary layout change for the NCHW batch
 * the reference hands over (libs/cil/cil.py:514-516). */
int bdv_nchw3_to_nhwc4(const float* x, float* out, int N, int H, int W, void* stream);
/* UPSTREAM ResNet.maxpool: MaxPool2d(3, stride 2, pad 1) on NHWC; idx holds the winning tap (0..8)
 * per output element (first maximum in scan order, as torch). */
/* out_dtype: element type of `out` (x is the fp32 stem activation): the pooled tensor is the first one stored in the activation
 * storage type.  bdv_maxpool_bwd: dout_dtype is the type of dout, dx is fp32. */
int bdv_maxpool_fwd(const float* x, void* out, uint8_t* idx, int N, int H, int W, int C, int out_dtype, void* stream);
int bdv_maxpool_bwd(const void* dout, const uint8_t* idx, float* dx, int N, int H, int W, int C, int dout_dtype, void* stream);
/* MaxPool3d((2,1,1), stride (2,1,1)) -- pool2 of UPSTREAM mmaction ResNet3d (I3D, configs/_base_/models/i3d_r50.py:1-27): the
 * element-wise larger of frames 2t and 2t+1 of x [2*frames_out][frame_elems] (frame_elems % 32 == 0; an even number of frames
 * per clip keeps pairs inside a clip).  sel: 1 bit per output element (second frame won); the backward fills dx completely. */
int bdv_maxpool_t2_fwd(const float* x, float* out, uint32_t* sel, int64_t frames_out, int64_t frame_elems, void* stream);
int bdv_maxpool_t2_bwd(const float* dout, const uint32_t* sel, float* dx, int64_t frames_out, int64_t frame_elems, void* stream);
/* Stem tail of a training forward in one pass (UPSTREAM ConvModule norm+act, then ResNet.maxpool): a = relu(y * scale +
 * shift), out = maxpool(a), idx as bdv_maxpool_fwd, relu_mask = 1 bit per element of a (a > 0) as bdv_bn_apply writes it.
 * The activation a itself is not materialised. */
/* Backward of the same stem tail: BatchNorm(+ReLU) backward whose incoming gradient is the MaxPool2d(3,2,1) backward of
 * dpool [N,Ho,Wo,C]; that gradient is gathered on the fly in both passes (statistics, apply) and never materialised.
 * dy [N,H,W,C]; dgamma/dbeta as bdv_bn_backward; workspace = bdv_bn_workspace_bytes(N*H*W, C). */
/* The stem conv output y and its gradient dy are fp32 in every storage mode (the stem reads the fp32 NHWC4 frames); the pooled
 * tensor `out` and its gradient `dpool` carry the activation storage type (out_dtype / dpool_dtype). */
int bdv_bn_backward_maxpool(const void* dpool, const uint8_t* pool_idx, const uint32_t* relu_mask, const float* y,
                            const float* gamma, const float* save_mean, const float* save_invstd, float* dy, float* dgamma,
                            float* dbeta, float beta_acc, int N, int H, int W, int C, void* workspace,
                            size_t workspace_bytes, int dpool_dtype, void* stream);
int bdv_bn_relu_maxpool_fwd(const float* y, const float* scale, const float* shift, void* out, uint8_t* idx,
                            uint32_t* relu_mask, int N, int H, int W, int C, int out_dtype, void* stream);
/* UPSTREAM TSMHead.avg_pool = AdaptiveAvgPool2d(1): [N,HW,C] -> [N,C] */
/* act_dtype: element type of x / dx; the pooled features and their gradient are fp32. */
int bdv_avgpool_fwd(const void* x, float* out, int N, int HW, int C, int act_dtype, void* stream);
int bdv_avgpool_bwd(const float* dout, void* dx, int N, int HW, int C, int act_dtype, void* stream);

/* ---- fused background-mix / normalize front-end -------------------------------------------
 * libs/loader/comix_loader.py:72-75,138-145 + UPSTREAM Normalize (img_norm_cfg, config :121-122).
 * frames (B,T,H,W,3) uint8 RGB, bg (B,H,W,3) uint8 -- or fp32 pixel values in [0,255] when bg_f32 != 0 (the
 * output of bdv_bg_resize_crop_u8) --, mix (B) uint8 {0,1};
 * out_nhwc4 (B*T,H,W,4) and/or out_nchw (B,T,3,H,W) (either may be NULL).
 * frame: (x-mean)*inv_std ; bg: (x-mean)/std ; blend x*(1-alpha)+bg*alpha where mix[b] != 0. */
int bdv_bgmix_normalize_u8(const uint8_t* frames, const void* bg, int bg_f32, const uint8_t* mix, float alpha,
                           const float mean[3], const float std[3], const float inv_std[3],
                           float* out_nhwc4, float* out_nchw, int B, int T, int H, int W, void* stream);
/* The stages of BackgroundMixDataset.bg_pipeline before Normalize (libs/loader/comix_loader.py:72-73): torchvision
 * Resize(bg_resize) of the float image (smaller edge -> bg_resize; bilinear, align_corners=False, no antialias filter:
 * the same values as the antialiased form when the image is enlarged) and RandomCrop(bg_crop_size) at the offsets the
 * caller drew.  src: B uint8 RGB images (Hs,Ws,3) of one size; (Hr,Wr): the resized size; top / left: (B) int32 on the
 * device; out: (B,crop_h,crop_w,3) fp32 pixel values in [0,255] (not rounded, as the reference's float image). */
int bdv_bg_resize_crop_u8(const uint8_t* src, int B, int Hs, int Ws, int Hr, int Wr, const int32_t* top,
                          const int32_t* left, int crop_h, int crop_w, float* out, void* stream);
/* Test-time crops + normalize (val/test pipelines, configs/...bgmix_plus_randAug.py:140-171): CenterCrop / ThreeCrop /
 * TenCrop as emitted by the crop transforms (the reference's own libs/pipelines/five_crops.py:77-100 shows the offset
 * rule and the crop-major order; TenCrop adds the horizontally flipped copy of each crop right after it).
 * frames (B,T,H,W,3) uint8; crops = ncrops x (x_offset, y_offset, flip) HOST table, ncrops <= BDV_MAX_CROPS;
 * out_nhwc4 (B*ncrops*T, crop_h, crop_w, 4) and/or out_nchw (B, ncrops*T, 3, crop_h, crop_w): frame (b, k, t) is
 * frame t of sample b cropped by entry k, values (x-mean)*inv_std. */
#define BDV_MAX_CROPS 12
int bdv_crop_normalize_u8(const uint8_t* frames, const int32_t* crops, int ncrops, int crop_h, int crop_w,
                          const float mean[3], const float inv_std[3], float* out_nhwc4, float* out_nchw, int B, int T,
                          int H, int W, void* stream);

/* ---- classifier heads ---------------------------------------------------------------------- */
/* libs/models/cil_heads/cosine_linear.py:27-43 (LSC): sim[n,k] = sum_p softmax_p(c)[p]*c[p],
 * c[p] = cos(x_n, w_{k,p}) with torch>=2 eps semantics (each norm clamped at 1e-8).
 * x (N,D), w (K,P*D), sim (N,K); xnorm (N), wnorm (K*P), cosbuf (N,K*P) are saved for backward. */
int bdv_lsc_fwd(const float* x, const float* w, float* sim, float* xnorm, float* wnorm, float* cosbuf,
                int N, int D, int K, int P, void* stream);
int bdv_lsc_bwd(const float* dsim, const float* x, const float* w, const float* xnorm, const float* wnorm,
                const float* cosbuf, float* dx, float* dw, float beta_w, float* dcos_ws,
                int N, int D, int K, int P, void* stream);
/* libs/models/cil_heads/inc_net.py:36-37 (IncrementalNet): out = x W^T + b */
int bdv_linear_fwd(const float* x, const float* w, const float* b, float* out, int N, int D, int K, void* stream);
int bdv_linear_bwd(const float* dout, const float* x, const float* w, float* dx, float* dw, float* db,
                   float beta_w, int N, int D, int K, void* stream);
/* UPSTREAM AvgConsensus(dim=1): (B*T,K) -> (B,K) mean over T, and its backward */
int bdv_consensus_fwd(const float* s, float* out, int B, int T, int K, void* stream);
int bdv_consensus_bwd(const float* dout, float* ds, int B, int T, int K, void* stream);
/* UPSTREAM TSMHead.dropout: mask from a counter-based hash of (seed, element index); same call
 * with the same seed reproduces the mask, so backward = forward applied to the gradient. */
int bdv_dropout(const float* x, float* out, int64_t numel, float p, uint64_t seed, void* stream);

/* ---- fused losses -------------------------------------------------------------------------- */
/* libs/losses/lsc_loss.py:30-58 (exclude_pos_denominator, optional hinge).  Writes loss (1),
 * dsim (B,K) and deta (1) for upstream gradient 1; callers scale.  `class_weights` (K) or NULL:
 * lsc_loss.py:50-51, the row's term is scaled by class_weights[target] before the negation and the hinge. */
int bdv_lsc_loss(const float* sim, const int64_t* targets, const float* eta, float margin, int hinge,
                 const float* class_weights, float* loss, float* dsim, float* deta, int B, int K, void* stream);
/* libs/cil/icarl.py:123-125 soft-target CE: loss = mean_b(-sum_k tgt*log_softmax(score)).
 * `soft_targets` (B,K) or NULL with integer `labels` (B) (= plain mean cross-entropy). */
int bdv_softce_loss(const float* score, const float* soft_targets, const int64_t* labels, float* loss,
                    float* dscore, int B, int K, void* stream);
/* icarl.py:101-120: targets = base_targets (the foreground-ratio soft labels of :103-111, built by bdv_acm_targets with
 * alpha = 4) or, when base_targets is NULL, onehot(label); rows with label < prev_K <- softmax(prev_logits). */
int bdv_icarl_targets(const int64_t* labels, const float* prev_logits, int prev_K, const float* base_targets,
                      float* targets, int B, int K, void* stream);
/* libs/losses/acm_smooth_ce.py:18-28 (ACMSmoothCE smooth labels): targets = onehot(label) * lam + (1 - lam) *
 * onehot(background_label), lam = 1 - (1 - foreground_ratio)^alpha; background label -1 counts as class 0. */
int bdv_acm_targets(const int64_t* labels, const int64_t* background_labels, const float* foreground_ratio, float alpha,
                    float* targets, int B, int K, void* stream);
/* UPSTREAM Recognizer2D average_clip('prob'): (B*n,K) -> softmax over K, mean over n -> (B,K) */
int bdv_softmax_mean(const float* s, float* out, int B, int n, int K, int apply_softmax, void* stream);
/* UPSTREAM top_k_accuracy: acc[0]=top-1, acc[1]=top-5 hit rates, computed on device (no D2H sync) */
int bdv_topk_acc(const float* score, const int64_t* labels, float* acc, int B, int K, void* stream);
/* libs/cil/cil.py:519-541 feature distillation: mse = mean((cur-prev)^2); dcur = gscale*2*(cur-prev)/numel */
size_t bdv_reduce_workspace_bytes(void);
/* act_dtype: element type of cur / prev / dcur (the hooked stage outputs carry the activation storage type; the avg_pool
 * features are fp32 in every mode). */
int bdv_kd_mse_fwd(const void* cur, const void* prev, float* mse, int64_t numel, void* workspace,
                   size_t workspace_bytes, int act_dtype, void* stream);
int bdv_kd_mse_bwd(const void* cur, const void* prev, const float* gscale_dev, float gscale_host,
                   void* dcur, int64_t numel, int act_dtype, void* stream);

/* ---- representation path: clip representations, NME classifier, class means, herding --------- */
/* libs/cil/cil.py:501-506 (_extract_repr) + :564-571 (predict_step, extract_repr): feat = the hooked
 * cls_head.avg_pool output flattened to (B*crops*T, D); repr (B*crops, D) = F.normalize(mean over the T segments);
 * mean_crops (B, D) = mean over crops of repr. */
int bdv_repr_from_features(const float* feat, float* repr, float* mean_crops, int B, int crops, int T, int D,
                           void* stream);
/* libs/cil/cil.py:945-960 (NME): similarity (S,K) = mean over crops of F.cosine_similarity(repr, class_means)
 * (each norm clamped at 1e-8), pred (S) = first arg-max over K.  repr (S*crops, D), class_means (K, D). */
size_t bdv_nme_workspace_bytes(int K, int D);
int bdv_nme_classify(const float* repr, const float* class_means, float* similarity, int64_t* pred, int S,
                     int crops, int D, int K, void* workspace, size_t workspace_bytes, void* stream);
/* libs/cil/cil.py:1079-1083: means[k] = mean of the rows of repr (n, D) whose label is k (NaN for an empty class,
 * like torch.mean of an empty selection). */
int bdv_class_means(const float* repr, const int64_t* labels, float* means, int n, int D, int K, void* stream);
/* libs/cil/memory_selection.py:70-92 + :150-164 (Herding, one class): greedy selection of num_exemplars rows of
 * features (n, D) whose running mean stays closest to the class mean (cosine: 1 - cos, else pairwise L2 distance with
 * torch's 1e-6).  class_mean (D) is what calc_mean_features returns; indices are positions in `features` in
 * selection order (ties -> lowest index, like argmin on the shrinking tensor); dist are the winning distances. */
size_t bdv_herding_workspace_bytes(int n, int D);
int bdv_herding_select(const float* features, int n, int D, int num_exemplars, int cosine_distance,
                       float* class_mean, int64_t* indices, float* dist, void* workspace, size_t workspace_bytes,
                       void* stream);

/* ---- data path: RandAugment on uint8 frames ---------------------
 * libs/pipelines/rand_augment.py:17-160 as applied by RandAugment._rand_aug (:237-264): ONE operation per clip (the
 * same for each of its T frames), bit-identical to the Pillow routines the reference calls.  in / out: (B, T, H, W, 3)
 * uint8 RGB, distinct buffers.  op_i (B, 8) int32 and op_d (B, 4) float64 are device tables, one row per clip:
 *   op_i[0] = operation: 0 Identity | 1 AutoContrast | 2 Equalize | 3 Solarize (op_d[0] = threshold) |
 *             4 Posterize (op_i[1] = bits) | 5 Color | 6 Contrast | 7 Brightness | 8 Sharpness (op_d[0] = factor in [0,1]) |
 *             9 affine, nearest, 16.16 fixed point: ShearX / ShearY / Rotate (op_i[1..6] = FIX(a0) FIX(a1)
 *               FIX(a2 + a0/2 + a1/2) FIX(a3) FIX(a4) FIX(a5 + a3/2 + a4/2), FIX(v) = floor(v * 65536 + 0.5)) |
 *            10 affine without cross terms: TranslateX / TranslateY (op_d[0..3] = a0 a2 a4 a5) |
 *            11 CutoutAbs (op_i[1..4] = x0 y0 x1 y1, inclusive)
 *   op_i[7] = fill colour 0xRRGGBB for operations 9-11 (FILL_COLOR (124,116,104), rand_augment.py:15).
 * Three launches per call (histograms, per-frame tables, apply); call once per operation slot (n = 2 in every config). */
size_t bdv_randaug_workspace_bytes(int B, int T, int H, int W);
int bdv_randaug_apply(const uint8_t* in, uint8_t* out, const int32_t* op_i, const double* op_d, int B, int T, int H,
                      int W, void* workspace, size_t workspace_bytes, void* stream);

/* Resize / MultiScaleCrop + Resize of the frame pipeline (configs/ucf101/bgmix_plus_randAug/...py:127-136; UPSTREAM mmaction2
 * Resize -> mmcv.imresize('bilinear') -> cv2.resize(INTER_LINEAR) on uint8 images): OpenCV's published fixed-point arithmetic
 * (11-bit weights, horizontal then vertical pass, the 2x2 box mean for an exact 2x shrink), PARITY UNPINNED -- cv2 is in neither
 * the reference tree nor this image (oracle/resize_oracle.py).  src (N,Hs,Ws,3) uint8 -> dst (N,Hd,Wd,3) uint8.  boxes: NULL (the
 * whole frame) or one (x0, y0, w, h) int32 quadruple per group of frames_per_box consecutive frames (a clip shares its crop), given
 * on the device (read by the kernel) AND on the host (validated against the frame before the launch). */
int bdv_resize_linear_u8(const uint8_t* src, int N, int Hs, int Ws, const int32_t* boxes, int frames_per_box,
                         const int32_t* boxes_host, uint8_t* dst, int Hd, int Wd, void* stream);

/* ---- JPEG frame decode ----------------------------------------------
 * The first stage of every pipeline of the configs: RawFrameDecode (configs/ucf101/bgmix_plus_randAug/
 * bgmix_seed_1000_inc_10_stages_bgmix_plus_randAug.py:126, :144, :160; UPSTREAM mmaction2 RawFrameDecode ->
 * mmcv.imfrombytes(channel_order='rgb') -> cv2.imdecode), i.e. libjpeg(-turbo) with its default ISLOW inverse DCT, fancy
 * chroma upsampling and fixed-point YCbCr -> RGB: reproduced bit for bit (oracle/jpeg_oracle.py, pinned by Pillow's
 * libjpeg-turbo).  Baseline / extended-sequential Huffman streams, 8-bit, grey or YCbCr 4:4:4 / 4:2:2 / 4:2:0, restart
 * intervals, interleaved or per-component scans; anything else returns BDV_EINVAL with the reason in bdv_last_error().
 *   1. bdv_jpeg_parse          (host) header -> bdv_jpeg_info: sizes, sampling, block grids, the components' quantisation tables
 *   2. bdv_jpeg_entropy_decode (host, thread-safe: call it from one thread per image) Huffman-decodes the stream into
 *      QUANTISED coefficients, info->coef_count shorts: per component a raster of 64-short blocks in natural (row-major)
 *      order over the MCU-padded block grid, component c at coef_offset[c]
 *   3. bdv_jpeg_reconstruct_u8 (device) a batch of B images of ONE geometry (everything in `info` but the tables): coefs
 *      (B, coef_count) int16 and qts (B, 3, 64) uint16 on the device -> rgb (B, height, width, 3) uint8; workspace =
 *      bdv_jpeg_workspace_bytes(info, B) bytes (the component planes).  Two launches: dequantise + inverse DCT, then
 *      upsample + colour conversion. */
typedef struct {
  int32_t width, height, ncomp;          /* ncomp: 1 (grey, written to all three output channels) or 3 (YCbCr) */
  int32_t h[3], v[3];                    /* sampling factors */
  int32_t blocks_w[3], blocks_h[3];      /* block grid of each component, padded to whole MCUs */
  int32_t down_w[3], down_h[3];          /* real sample size of each component: ceil(width * h / hmax), ceil(height * v / vmax) */
  uint16_t qt[3][64];                    /* each component's quantisation table, natural order */
  int64_t coef_offset[3], coef_count;    /* in shorts, inside one image's coefficient buffer */
} bdv_jpeg_info;
int bdv_jpeg_parse(const unsigned char* data, size_t n, bdv_jpeg_info* info);
int bdv_jpeg_entropy_decode(const unsigned char* data, size_t n, const bdv_jpeg_info* info, short* coefs);
/* Step 2 for n streams of ONE geometry on `threads` host threads (std::thread; the caller's thread is one of them): coefs
 * (n, coef_count) int16 and qts (n, 3, 64) uint16 are HOST buffers (pinned, for the upload); returns the first failing image's
 * error.  What decode.JpegDecoder calls per batch. */
int bdv_jpeg_entropy_decode_batch(const unsigned char* const* data, const size_t* sizes, int n, const bdv_jpeg_info* info,
                                  short* coefs, unsigned short* qts, int threads);
size_t bdv_jpeg_workspace_bytes(const bdv_jpeg_info* info, int B);
int bdv_jpeg_reconstruct_u8(const short* coefs, const unsigned short* qts, const bdv_jpeg_info* info, int B, void* workspace,
                            size_t workspace_bytes, unsigned char* rgb, void* stream);

/* ---- optimizer: multi-tensor global-norm clip + SGD(momentum, wd) ---------------------------
 * torch.optim.SGD built at libs/cil/cil.py:467 with the groups of libs/models/cil_heads/tsm.py:273-303
 * and PL gradient_clip_val (cil.py:743).  Tables are device arrays, one entry per tensor. */
int bdv_multi_sqnorm(const float* const* grads, const int64_t* numels, int ntensors, float* out_sqnorm,
                     void* workspace, size_t workspace_bytes, void* stream);
/* clip_coef (device scalar) = min(1, max_norm / (sqrt(sqnorm)*grad_scale + 1e-6)); max_norm <= 0 -> 1 */
int bdv_clip_coef(const float* sqnorm, float grad_scale, float max_norm, float* clip_coef, void* stream);
/* g = grad*grad_scale*clip_coef + wd*p ; buf = momentum*buf + g ; p -= lr*buf */
int bdv_multi_sgd(float* const* params, const float* const* grads, float* const* bufs,
                  const int64_t* numels, const float* lrs, const float* wds, int ntensors,
                  float momentum, float grad_scale, const float* clip_coef, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BDVCIL_HIP_H */
