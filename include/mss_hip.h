/* mss_hip.h -- C ABI of libmss_hip.so, the MI355X (gfx950) kernels behind the dense-segmentation +
 * OOD-scoring hot path of gaozhitong/MultiShiftSeg.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream); every call is asynchronous
 *     on that stream and performs no host synchronisation, allocation or free;
 *   - return value: 0 = launched, MSS_ERR_* (>= 1001) = rejected precondition, other = hipError_t.
 *     The reference only printf()s launch failures (ops/src/cuda/ms_deform_im2col_cuda.cuh:953-957);
 *     callers of this ABI must raise on non-zero.
 *   - activations inside the DeepLab path are NHWC fp32 with an explicit pixel stride (`ld*`,
 *     in floats) so that kernels read/write channel slices of concat buffers in place.
 *
 * Each entry point cites the reference interface it replaces (paths relative to the reference root).
 */
#ifndef MSS_HIP_H
#define MSS_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSS_ABI_VERSION 8      /* 2: round-2 struct / workspace changes; 3: mss_msda_backward_binned_f32; 4: mss_add_layernorm_bwd_sum_f32, mss_stem_conv_pool_f32,
                                  mss_wino_input_transform_bnbwd_f32, mss_wino_input_transform_upcat_f32,
                                  mss_bn_fold_train_from_partials_f32; 5 (round 4): mss_adam_step_f32 takes double hyper-parameters, mss_env_reset,
                                  mss_wino_input_transform_aspp3_f32, mss_msda_prepare_backward_ld_f32, mss_rcl_pairs_device2_f32, mss_rcl_loss_device_f32, mss_m2f_fused_score_ws_f32, mss_oodm_compact_packed_f32,
                                  6 (late round 4, additive): mss_msda_forward_fused_ld_f32, mss_msda_prepare_ld_f32, mss_add_layernorm_q_f32, mss_add_layernorm_bwd_sum2_f32, mss_msda_forward_fused_save_f32, mss_msda_backward_binned_proj_f32, mss_gap_from_partials_f32;
                                  7 (round 5): MssConvArgs.w_split + mss_gemm_split_weights_bf16x3 (the split-bf16 GEMM route);
                                  8 (round 6): mss_msda_forward_window_f32 removed (the measured-slower LDS-window forward left the product); additive: the mss_oodm_*lanes* entry points,
                                  mss_gemm_split_last_mfma */
int mss_abi_version(void);

/* The MSS_* environment switches (A/B experiments, test routes; none is needed in production) are read once per call site and
 * cached. A process that changes one after its first call into the library calls this to have them re-read. */
int mss_env_reset(void);        /* returns the new generation */
int mss_env_generation(void);

/* ---------------------------------------------------------------------------------------------
 * B1 -- MultiScaleDeformableAttention extension
 * replaces ms_deform_attn_forward / ms_deform_attn_backward
 *   lib/network/mask2former/modeling/pixel_decoder/ops/src/vision.cpp:18-21
 *   .../ops/src/ms_deform_attn.h:25-66, .../ops/src/cuda/ms_deform_attn_cuda.cu:25-157
 *   kernels: .../ops/src/cuda/ms_deform_im2col_cuda.cuh:242-304 (fwd), :306-925 (bwd variants)
 * value [N,S,M,D]; spatial_shapes int64 [L,2]=(H_l,W_l); level_start_index int64 [L];
 * sampling_loc [N,Lq,M,L,P,2] (x,y) in [0,1]; attn_weight [N,Lq,M,L,P]; out [N,Lq,M*D].
 * forward: `out` may be uninitialised. backward: the callee zero-fills grad_value, grad_loc,
 * grad_attn itself (the reference allocates them with at::zeros, ms_deform_attn_cuda.cu:126-128).
 * There is no im2col_step argument: the whole batch is one launch (the reference's chunk loop,
 * ms_deform_attn_cuda.cu:66-80, only bounds a temporary it needs and we do not). */
int mss_msda_forward_f32(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                         const float* sampling_loc, const float* attn_weight, int N, int S, int M, int D, int L,
                         int Lq, int P, float* out, void* stream);
int mss_msda_forward_f64(const double* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                         const double* sampling_loc, const double* attn_weight, int N, int S, int M, int D, int L,
                         int Lq, int P, double* out, void* stream);
int mss_msda_backward_f32(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                          const float* sampling_loc, const float* attn_weight, const float* grad_out, int N,
                          int S, int M, int D, int L, int Lq, int P, float* grad_value, float* grad_loc,
                          float* grad_attn, void* stream);
/* The same ms_deform_attn_backward (vision.cpp:18-21 -> ms_deform_attn_cuda.cu:88-157, kernel family
 * ms_deform_im2col_cuda.cuh:306-925) with grad_value on the BINNED owner-computes path (round 3): every sample is filed
 * once under the tile of its level that holds its top-left corner (counting sort on the device), a workgroup per tile sums
 * its records in 64-bit fixed-point LDS words and stores the tile -- no floating-point atomic, no re-scan, bit-reproducible.
 * `host_shapes`: HOST copy of spatial_shapes (tile geometry and launch grids depend on it); `workspace`: 256-byte aligned
 * device scratch of >= mss_msda_backward_workspace_bytes(...) bytes. fp32, D == 32, L <= 8, 16-byte aligned value / grad_out;
 * otherwise MSS_ERR_UNSUPPORTED (workspace_bytes query: 0) and the caller uses mss_msda_backward_f32. */
long long mss_msda_backward_workspace_bytes(const int64_t* host_shapes, int N, int M, int D, int L, int Lq, int P);
int mss_msda_backward_binned_f32(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                 const int64_t* host_shapes, const float* sampling_loc, const float* attn_weight,
                                 const float* grad_out, int N, int S, int M, int D, int L, int Lq, int P, float* grad_value,
                                 float* grad_loc, float* grad_attn, void* workspace, long long workspace_bytes, void* stream);
int mss_msda_backward_f64(const double* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                          const double* sampling_loc, const double* attn_weight, const double* grad_out, int N,
                          int S, int M, int D, int L, int Lq, int P, double* grad_value, double* grad_loc,
                          double* grad_attn, void* stream);

/* forward straight from the module's raw projections -- offsets [N,Lq,M,L,P,2] (sampling_offsets Linear),
 * logits [N,Lq,M,L*P] (attention_weights Linear), reference_points [N,Lq,L,2] -- with the softmax and the location
 * arithmetic of ops/modules/ms_deform_attn.py:100-109 inside the sampling kernel: sampling_loc / attn_weight are never
 * materialised. fp32, D in {16, 32, 64}; MSS_ERR_UNSUPPORTED otherwise (run mss_msda_prepare_f32 + mss_msda_forward_f32).
 * Equal to that two-kernel route up to the summation order of the L*P exponentials; L*P <= 20. */
int mss_msda_forward_fused_f32(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                               const float* offsets, const float* logits, const float* reference_points, int N, int S,
                               int M, int D, int L, int Lq, int P, float* out, void* stream);

/* Operand preparation of the MSDeformAttn module in one pass (ops/modules/ms_deform_attn.py:100-109, the
 * reference_points[..., 2] branch): attn = softmax over the L*P logits of each (query, head), sampling_loc =
 * reference_points[n,q,l] + offsets / (W_l, H_l). offsets [N,Lq,M,L,P,2], logits [N,Lq,M,L*P], reference_points
 * [N,Lq,L,2]; L*P <= 20 (MSS_ERR_UNSUPPORTED above). The backward returns the gradients w.r.t. offsets and logits. */
int mss_msda_prepare_f32(const float* offsets, const float* logits, const float* reference_points,
                         const int64_t* spatial_shapes, int N, int Lq, int M, int L, int P, float* sampling_loc,
                         float* attn_weight, void* stream);
int mss_msda_prepare_backward_f32(const float* attn_weight, const float* grad_attn, const float* grad_loc,
                                  const int64_t* spatial_shapes, int N, int Lq, int M, int L, int P, float* grad_offsets,
                                  float* grad_logits, void* stream);
/* The same with row strides: grad_offsets row (n, q) at + (n*Lq + q) * ld_offsets (>= M*2*L*P floats), grad_logits likewise with
 * ld_logits (>= M*L*P). Both gradients in ONE [N*Lq, M*3*L*P] buffer (offsets | logits) make the weight gradient and the data
 * gradient of `sampling_offsets` and `attention_weights` -- two Linears on the same query (ops/modules/ms_deform_attn.py:98-100)
 * -- one GEMM each. */
int mss_msda_prepare_backward_ld_f32(const float* attn_weight, const float* grad_attn, const float* grad_loc,
                                     const int64_t* spatial_shapes, int N, int Lq, int M, int L, int P, float* grad_offsets,
                                     long long ld_offsets, float* grad_logits, long long ld_logits, void* stream);
/* mss_msda_backward_binned_f32 with that module backward folded into its gather pass: d(offsets) / d(logits) leave instead of
 * grad_sampling_loc / grad_attn_weight (never materialised; no prepare-backward launch). Bit-identical to the two-call sequence. */
int mss_msda_backward_binned_proj_f32(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                      const int64_t* host_shapes, const float* sampling_loc, const float* attn_weight,
                                      const float* grad_out, int N, int S, int M, int D, int L, int Lq, int P, float* grad_value,
                                      float* grad_offsets, long long ld_offsets, float* grad_logits, long long ld_logits,
                                      void* workspace, long long workspace_bytes, void* stream);
/* Forward side of the same idea (ops/modules/ms_deform_attn.py:98-101: `sampling_offsets(query)` and `attention_weights(query)`):
 * offsets row (n, q) at + (n*Lq + q) * ld_offsets, logits likewise with ld_logits (0 = dense), so that both may be column ranges of
 * the [N*Lq, M*3*L*P] output of ONE product query [Woff ; Watt]^T. Outputs of the prepare form stay dense. */
int mss_msda_forward_fused_ld_f32(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                  const float* offsets, long long ld_offsets, const float* logits, long long ld_logits,
                                  const float* reference_points, int N, int S, int M, int D, int L, int Lq, int P, float* out,
                                  void* stream);
/* ... and with the sampling locations / attention weights the kernel formed written out (both or neither), for a backward that
 * reads what the forward used (mss_msda_backward_*_f32 take exactly these two tensors) instead of re-deriving them. */
int mss_msda_forward_fused_save_f32(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                    const float* offsets, long long ld_offsets, const float* logits, long long ld_logits,
                                    const float* reference_points, int N, int S, int M, int D, int L, int Lq, int P, float* out,
                                    float* sampling_loc_out, float* attn_weight_out, void* stream);
int mss_msda_prepare_ld_f32(const float* offsets, long long ld_offsets, const float* logits, long long ld_logits,
                            const float* reference_points, const int64_t* spatial_shapes, int N, int Lq, int M, int L, int P,
                            float* sampling_loc, float* attn_weight, void* stream);

/* ---------------------------------------------------------------------------------------------
 * B2 -- DeepWV3Plus operator set (replaces the cuDNN/ATen ops under
 *        lib/network/deepv3/deepv3.py:258-285 and lib/network/deepv3/wider_resnet.py:169-182)      */

/* One bias-free 2-D convolution as an implicit GEMM on fp32 MFMA.
 * replaces nn.Conv2d forward at deepv3.py:62-72,79-81,235-250 and wider_resnet.py:104-137,303-305 */
typedef struct MssConvArgs {
  const float* x;          /* input  NHWC [N,H,W,>=C], pixel stride ldx                         */
  const float* w;          /* packed weights [R*S][Kpad][C] from mss_conv2d_pack_weights_f32     */
  float* y;                /* output NHWC [N,OH,OW,>=K], pixel stride ldy                        */
  const float* in_scale;   /* optional prologue  a = x*in_scale[c] + in_shift[c] (BatchNorm of   */
  const float* in_shift;   /*   the producer, folded); sample n uses row n*in_ss_stride          */
  const float* out_scale;  /* optional epilogue  y = acc*out_scale[k] + out_shift[k]             */
  const float* out_shift;
  const float* res;        /* optional residual  y += res[m*ldres + k]  (wider_resnet.py:181)    */
  int N, H, W, C, ldx;
  int OH, OW, K, Kpad, ldy;
  int R, S, stride, dil, pad;
  int in_ss_stride;        /* 0: one affine for all samples; C: one per sample (Dropout2d fold)  */
  int in_relu, out_relu;   /* ReLU after the prologue affine / at the very end of the epilogue   */
  int ldres;
  int M, mtiles, ntiles;   /* filled in by the launcher                                          */
  int batch;               /* > 1: that many independent GEMMs in one launch (1x1 only, no res);  */
  long long x_bs, w_bs, y_bs; /* element strides between their x / w / y (Winograd positions)    */
  float* stats;            /* optional: per-channel partial sums of the OUTPUT for the next layer's train-mode    */
                           /*   BatchNorm, [ceil(M/64)][2][K] floats (row group g: sum, sum of squares of rows      */
                           /*   64g..64g+63); reduce with mss_bn_stats_partials_f32. Not with batch > 1.            */
  int res_mask;            /* 1: `res` gates instead of adds -- y = res[m*ldres + k] > 0 ? y : 0 -- the ReLU backward of a  */
                           /*   producer whose stored OUTPUT is `res` (FFN of msdeformattn.py:122-131) fused in the dgrad   */
  const void* w_split;     /* optional: the SAME weights as `w`, split into three bf16 planes by mss_gemm_split_weights_bf16x3.     */
                           /*   Non-NULL selects the split-bf16 route (6 bf16 MFMAs with fp32 accumulation per product block,       */
                           /*   fp32 accuracy) for every shape the persistent GEMM kernel takes (1x1, > 64 output channels,         */
                           /*   >= 48 input channels); other shapes run the native fp32 kernels on `w` (which must always be set).   */
  int route;               /* 0: native fp32 MFMA. 1: mss_conv2d_wgrad_f32 evaluates the TN product the same split-bf16 way (both    */
                           /*   operands split in the loader) where K % 128 == 0 and C % 256 == 0; other shapes stay native.         */
} MssConvArgs;

int mss_conv2d_forward_f32(MssConvArgs* args, void* stream);
int mss_conv2d_kpad(int K);
/* 1 if mss_conv2d_forward_f32 runs these arguments on the persistent GEMM kernel (csrc/gemm.hip: 1x1, stride 1,
 * more than 64 output channels, or 33..64 of them over >= 16 384 rows), 2 for the few-rows kernel (1x1 over <= 8 pixels, nothing fused), 3 for the
 * split-bf16 form of the persistent GEMM kernel (args->w_split set and the shape eligible), 4 for its implicit-GEMM form, 0 for the native implicit-GEMM kernel. Profiling label only. */
int mss_conv2d_forward_route(const MssConvArgs* args);
/* The split-bf16 form of packed weights for MssConvArgs.w_split: w [batch][Kpad][C] fp32 (batch stride w_bs floats; what
 * mss_conv2d_pack_weights_f32 with R = S = 1 or mss_wino_pack_weights_f32 produce, Kpad % 128 == 0, C % 16 == 0) -> `planes`,
 * mss_gemm_split_weights_bytes(batch, Kpad, C) = batch * Kpad * C * 6 bytes: every value as three round-to-nearest bf16 terms
 * hi + mid + lo (exact), stored per 128-row block and 16-deep K-step in the order the kernel stages them. A fp32 product is then
 * a_hi b_hi + a_hi b_mid + a_mid b_hi + a_hi b_lo + a_lo b_hi + a_mid b_mid on the bf16 matrix cores with fp32 accumulation
 * (the three dropped terms are below 2^-26 |a b|): same contraction as nn.Conv2d / F.linear in fp32
 * (deepv3.py:47-92,258-285, wider_resnet.py:169-182), at fp32 accuracy. mss_conv2d_forward_route answers 3 when a call runs it. */
long long mss_gemm_split_weights_bytes(int batch, int Kpad, int C);
int mss_gemm_split_weights_bf16x3(const float* w, void* planes, int batch, int Kpad, int C, long long w_bs, void* stream);
/* The same for an implicit-GEMM layer (3x3 with stride 2 or few input channels, 1x1 with stride 2: what conv_igemm_kernel takes, with
 * > 64 output channels and one prologue affine): w [taps][Kpad][C] from mss_conv2d_pack_weights_f32 (taps = R*S <= 9) -> planes of
 * mss_gemm_split_weights_bytes(taps, Kpad, C) bytes with the taps folded into ONE reduction of taps*C; mss_conv2d_forward_route answers 4. */
int mss_conv_split_weights_bf16x3(const float* w, void* planes, int taps, int Kpad, int C, void* stream);
/* Which matrix instruction the LAST split-bf16 GEMM launched by the calling thread used: 16 = v_mfma_f32_16x16x32_bf16 on concatenated
 * planes with the weights brought in by LDS-DMA (products without a prologue whose output start, pitch and batch stride sit on the
 * 16-byte grid), 32 = v_mfma_f32_32x32x16_bf16 (everything else), 0 = none yet. Observability for tests and profiling labels only. */
int mss_gemm_split_last_mfma(void);
/* w [K][C][R][S] (nn.Conv2d.weight) -> packed [R*S][Kpad][Cp] (zero padded).
 * flip=1 packs the data-gradient filter instead (K<->C swapped, taps rotated 180 degrees);
 * then Kpad/Cp refer to the swapped roles. */
int mss_conv2d_pack_weights_f32(const float* w, float* packed, int K, int C, int R, int S, int Kpad, int Cp,
                                int flip, void* stream);
/* weight gradient: dwp[tap][k][c] = sum_m dy[m][k]*act(x[m@tap][c]); dwp [taps][Kpad][Cp] is fully overwritten.
 * Deterministic (no atomics; pixel-range partials are summed in a fixed order: the reference pins
 * cudnn.deterministic, lib/utils/utils.py:10-13). ws: scratch of mss_conv2d_wgrad_workspace_bytes(args, Cp) bytes
 * (NULL allowed when that is 0). Replaces the autograd wgrad of nn.Conv2d for aspp/bot_aspp/bot_fine/ood_head
 * (exps/DeepLab.yaml:10-11, train_deeplab.py:113-132) */
long long mss_conv2d_wgrad_workspace_bytes(const MssConvArgs* args, int Cp);
int mss_conv2d_wgrad_f32(MssConvArgs* args, const float* dy, int lddy, float* dwp, int Cp, float* ws,
                         long long ws_bytes, void* stream);
int mss_conv2d_unpack_wgrad_f32(const float* packed, float* grad, int K, int C, int R, int S, int Kpad,
                                int Cp, int accumulate, void* stream);
/* 1 when mss_conv2d_wgrad_f32 runs these arguments on the split-bf16 TN kernel (args->route == 1, K % 128 == 0, C % 256 == 0, enough
 * tiles to fill half the chip), else 0 (native fp32 MFMA kernels). Profiling label only. */
int mss_conv2d_wgrad_route(const MssConvArgs* args, int lddy);

/* Winograd F(m x m, 3x3) path, m = `tile` = 2 or 4, P = (m+2)^2 positions, for stride-1 3x3 convolutions with
 * many channels (csrc/winograd.hip): weights [K][C][3][3] -> U [P][Kpad][Cp]; x -> X' [P][T][C]
 * (T = mss_wino_num_tiles, BatchNorm/ReLU prologue and zero padding fused); then ONE mss_conv2d_forward_f32
 * call in batched 1x1 mode (batch = P, x_bs = T*C, w_bs = Kpad*Cp, y_bs = T*K) gives Y' [P][T][K];
 * Y' -> NHWC y (+ residual). Dilation is exact (per-residue sub-grids). MFMA work is 9*m^2/P = 2.25x (m = 2)
 * or 4x (m = 4) below the direct form; fp32 rounding error ~1e-6 (m = 2) / ~1e-5 (m = 4) relative per layer. */
long long mss_wino_num_tiles(int N, int H, int W, int dil, int tile);
int mss_wino_pack_weights_f32(const float* w, float* u, int K, int C, int Kpad, int Cp, int tile, void* stream);
/* The same U straight into the split-bf16 planes MssConvArgs.w_split takes ((tile + 2)^2 batch entries of [Kpad][C]; bit-identical to
 * mss_gemm_split_weights_bf16x3 of the function above with Cp == C) without writing and re-reading U in fp32. Kpad % 128 == 0,
 * C % 16 == 0, both pointers 16-byte aligned; planes: mss_gemm_split_weights_bytes((tile + 2)^2, Kpad, C) bytes. */
int mss_wino_pack_split_bf16x3(const float* w, void* planes, int K, int C, int Kpad, int tile, void* stream);
int mss_wino_input_transform_f32(const float* x, int ldx, int N, int H, int W, int C, int dil, int tile,
                                 const float* scale, const float* shift, int relu, float* xt, void* stream);
/* The same X', but of dx = the train-mode BatchNorm+ReLU backward of the gradient dy w.r.t. the layer input x2, computed on the fly
 * (the arithmetic of mss_bn_relu_bwd_apply_f32): the data-gradient convolution behind a BatchNorm backward never materialises dx.
 * scale / shift: the forward's folded affine; mean / invstd: its saved batch statistics; accum: [2C] doubles of
 * mss_bn_relu_bwd_reduce_f32 over the same (dy, x2). MSS_ERR_UNSUPPORTED where the LDS-staged transform is not taken. */
int mss_wino_input_transform_bnbwd_f32(const float* dy, int lddy, const float* x2, int ldx2, int N, int H, int W, int C, int dil,
                                       int tile, const float* scale, const float* shift, const float* mean, const float* invstd,
                                       const double* accum, int relu, float* xt, void* stream);
/* The same X' for the decoder's first 3x3 layer, whose input is a concat (deepv3.py:269-275): channels [0, c_split) from `a`, the
 * rest = the align_corners=True bilinear upsample of `small` [N][IH][IW][C - c_split] to H x W, interpolated inside the transform
 * (the full-resolution upsampled map is never stored). MSS_ERR_UNSUPPORTED where the LDS-staged transform is not taken. */
int mss_wino_input_transform_upcat_f32(const float* a, int lda, int c_split, const float* small, int ld_small, int IH, int IW, int N,
                                       int H, int W, int C, int tile, float* xt, void* stream);
/* ASPP (deepv3.py:84-92: three 3x3 branches of rates d, 2d, 3d on the SAME 4096-channel map): X' for all three dilations from
 * ONE read of x. xt_m = exactly what mss_wino_input_transform_f32(x, ..., dil = (m+1)*d, tile = tiles[m], no prologue) writes.
 * tiles: a HOST pointer (the one exception to this header's "every pointer is a device pointer" rule; read before the launch, so
 * the call may be captured into a hipGraph) to 3 tile edges, each 4 or 6. MSS_ERR_UNSUPPORTED when a base residue sub-grid does not fit in LDS or a
 * tile edge is 2 (the caller then runs the three transforms separately). */
int mss_wino_input_transform_aspp3_f32(const float* x, int ldx, int N, int H, int W, int C, int d, const int* tiles, float* xt0,
                                       float* xt1, float* xt2, void* stream);
int mss_wino_output_transform_f32(const float* yt, int N, int H, int W, int K, int dil, int tile, const float* res,
                                  int ldres, float* y, int ldy, float* stats, void* stream);
/* stats (optional): [mss_wino_output_stats_parts(...)][2][K] partial sums / sums of squares of y, as MssConvArgs.stats */
int mss_wino_output_stats_parts(int N, int H, int W, int K, int dil, int tile);

/* weight gradient in the Winograd domain: dY' = A dY A^T per tile, then ONE mss_conv2d_wgrad_f32 call in
 * batched mode (batch = P, R = S = 1, x = X', x_bs = T*C, dy = dY', y_bs = T*K) accumulates
 * dU [P][Kpad][Cp], and dg = G^T dU G gives the [K][C][3][3] gradient. */
int mss_wino_grad_output_transform_f32(const float* dy, int lddy, int N, int H, int W, int K, int dil, int tile,
                                       float* dyt, void* stream);
int mss_wino_weight_grad_transform_f32(const float* du, float* dw, int K, int C, int Kpad, int Cp, int tile,
                                       void* stream);

/* image NCHW [N,C,H,W] -> NHWC [N,H,W,Cp] with channels C..Cp-1 zero (feeds mod1.conv1). */
int mss_nchw_to_nhwc_pad_f32(const float* x, float* y, int N, int C, int H, int W, int Cp, void* stream);
/* stem (mod1.conv1, 3 -> 64 channels, 3x3, padding 1; wider_resnet.py:303) as a dense GEMM: img [N,3,H,W] NCHW ->
 * out [N,H,W,32] NHWC with channel j = c*9 + r*3 + s holding img[n][c][y+r-1][x+s-1] (zero padding), 27..31 zero: the
 * column order of weight.reshape(64, 27). The 1x1 convolution of that tensor with the reshaped weight IS conv1. */
int mss_im2col3x3_c3_f32(const float* img, float* out, int N, int H, int W, void* stream);
/* The stem in ONE kernel (csrc/stem.hip): y [N][OH][OW][64] NHWC (pixel stride ldy >= 64) = MaxPool2d(3, stride 2, padding 1) of
 * conv3x3(img [N][3][H][W] NCHW, w [64][3][3][3], padding 1, no bias) -- mod1.conv1 + pool2 of the trunk (wider_resnet.py:343-345,
 * 353-355; no BatchNorm between them). OH = (H-1)/2 + 1, OW = (W-1)/2 + 1. The full-resolution 64-channel map is never stored. */
int mss_stem_conv_pool_f32(const float* img, const float* w, float* y, int ldy, int N, int H, int W, void* stream);

/* BatchNorm2d pieces (mynn.py:8-12 Norm2d = nn.BatchNorm2d, eps 1e-5, momentum 0.1).
 * stats: per-channel batch mean and biased variance of an NHWC tensor (M pixels). `accum` is a scratch of
 * mss_col_reduce_accum_doubles(M, C) doubles (contents irrelevant on entry): its first 2*C entries receive the
 * sums (sum | sum of squares), the rest holds the first stage's per-workgroup partials, which a second kernel adds
 * in a fixed order -- no atomics, bit-reproducible (the reference pins cudnn.deterministic, lib/utils/utils.py:10-13). */
long long mss_col_reduce_accum_doubles(long long M, int C);
/* accum[0:2C] = column sums of a partial-sum matrix [nparts][2][C] written by a producer (MssConvArgs.stats,
 * mss_wino_output_transform_f32): the statistics pass then never re-reads the activation. accum as above with
 * M = nparts. */
int mss_bn_stats_partials_f32(const float* partials, long long nparts, int C, double* accum, void* stream);
int mss_bn_stats_nhwc_f32(const float* x, long long M, int C, int ldx, double* accum, void* stream);
/* finalise: from accum -> (mean, var) -> scale = gamma*rsqrt(var+eps), shift = beta-mean*scale;
 * if running_mean != NULL also running = (1-mom)*running + mom*{mean, var*M/(M-1)}.
 * save_mean / save_invstd (optional) are kept for the backward. */
int mss_bn_finalize_train_f32(const double* accum, long long M, int C, const float* gamma, const float* beta,
                              float eps, float momentum, float* running_mean, float* running_var, float* scale,
                              float* shift, float* save_mean, float* save_invstd, void* stream);
/* mss_bn_stats_partials_f32 + mss_bn_finalize_train_f32 in two launches instead of three (bit-identical results): the train-mode fold
 * of a BatchNorm whose producer left partial sums. M = rows covered by the partial sums. */
int mss_bn_fold_train_from_partials_f32(const float* partials, long long nparts, int C, double* accum, long long M,
                                        const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                                        float* running_var, float* scale, float* shift, float* save_mean, float* save_invstd,
                                        void* stream);
/* eval mode: scale/shift from the running statistics. */
int mss_bn_fold_eval_f32(const float* gamma, const float* beta, const float* running_mean,
                         const float* running_var, float eps, int C, float* scale, float* shift, void* stream);
/* y = relu?(x*scale[c]+shift[c]) on NHWC rows (used where a fused prologue is not possible). */
int mss_affine_relu_nhwc_f32(const float* x, int ldx, float* y, int ldy, long long M, int C, const float* scale,
                             const float* shift, int relu, void* stream);
/* backward of y = relu(bn_train(x)): given dy and x (pre-BN), scale/save_mean/save_invstd/gamma,
 * per-channel sums into accum[0:2C] (scratch of mss_col_reduce_accum_doubles(M, C) doubles, as above):
 * sum(dz), sum(dz*xhat) where dz = dy * (y>0). */
int mss_bn_relu_bwd_reduce_f32(const float* dy, int lddy, const float* x, int ldx, long long M, int C,
                               const float* scale, const float* shift, const float* save_mean,
                               const float* save_invstd, int relu, double* accum, void* stream);
/* dx = gamma*invstd*(dz - sum_dz/M - xhat*sum_dzxhat/M) (train) or dz*scale (eval, accum==NULL);
 * optional dgamma/dbeta outputs (added to, may be NULL). */
int mss_bn_relu_bwd_apply_f32(const float* dy, int lddy, const float* x, int ldx, float* dx, int lddx,
                              long long M, int C, const float* gamma, const float* scale, const float* shift,
                              const float* save_mean, const float* save_invstd, int relu, const double* accum,
                              float* dgamma, float* dbeta, void* stream);

/* MaxPool2d(3, stride 2, padding 1) on NHWC (wider_resnet.py:353-355). */
int mss_maxpool3s2_nhwc_f32(const float* x, int ldx, float* y, int ldy, int N, int H, int W, int C, int OH,
                            int OW, void* stream);
/* AdaptiveAvgPool2d(1) (deepv3.py:77,87): y[n][c] = mean over HW. ws: scratch of mss_colsum_workspace_floats(N, HW, C)
 * floats (pixel-range partials, added in a fixed order: no atomics). */
long long mss_colsum_workspace_floats(int N, int HW, int C);
int mss_gap_nhwc_f32(const float* x, int ldx, float* y, int N, int HW, int C, float* ws, void* stream);
/* the same from the per-64-row column sums a producing conv / GEMM left in MssConvArgs.stats ([ceil(N*HW/64)][2][C]; HW % 64 == 0,
 * MSS_ERR_UNSUPPORTED otherwise): the map itself is not read again (r04). float64 accumulation in a fixed order. */
int mss_gap_from_partials_f32(const float* partials, int N, int HW, int C, float* y, void* stream);
/* broadcast y[n][p][c] = relu?(v[n][c]*scale[c]+shift[c]) over HW pixels: the "Upsample" of the
 * 1x1 image-pooling map (deepv3.py:86). */
int mss_broadcast_rows_nhwc_f32(const float* v, float* y, int ldy, int N, int HW, int C, const float* scale,
                                const float* shift, int relu, void* stream);
/* backward of the broadcast: dv[n][c] = sum_p dy[n][p][c] */
int mss_colsum_nhwc_f32(const float* dy, int lddy, float* dv, int N, int HW, int C, float* ws, void* stream);

/* F.interpolate(mode='bilinear', align_corners=True) (mynn.py:28-33) on NHWC, forward and its
 * transpose (gather form, deterministic). Optional prologue affine+relu on the input. */
int mss_upsample_ac_nhwc_f32(const float* x, int ldx, float* y, int ldy, int N, int IH, int IW, int OH, int OW,
                             int C, void* stream);
int mss_upsample_ac_nhwc_bwd_f32(const float* dy, int lddy, float* dx, int lddx, int N, int IH, int IW, int OH,
                                 int OW, int C, void* stream);

/* The OOD-score tail (deepv3.py:251-253,279-283):
 *   score[n,oy,ox] = bilinear_ac( -logsumexp_c dec2[n,:,:,c] )    dec2 NHWC [N,IH,IW,C] (ld)
 *   logit[n,c,oy,ox] = bilinear_ac( dec1[n,:,:,c] )               written NCHW
 *   label[n,oy,ox]  = argmax_c logit (first max wins, as torch.argmax), optional (uint8)
 * Any of score/logit/label may be NULL. */
int mss_ood_score_f32(const float* dec2, int ld2, const float* dec1, int ld1, int N, int IH, int IW, int C,
                      int OH, int OW, float* score, float* logit_nchw, uint8_t* label, void* stream);
/* backward: given dscore [N,OH,OW] and dlogit NCHW [N,C,OH,OW] (either may be NULL) produce
 * ddec2 / ddec1 (NHWC, assigned). */
int mss_ood_score_bwd_f32(const float* dec2, int ld2, const float* dscore, const float* dlogit_nchw, int N,
                          int IH, int IW, int C, int OH, int OW, float* ddec2, int ldd2, float* ddec1, int ldd1,
                          void* stream);

/* Mask2Former anomaly score (train_m2f.py:387-407):
 *   score[b,h,w] = 1 - max_{c<C} sum_q softmax(cls[b,q,:])[c] * sigmoid(mask[b,q,h,w])
 * cls [B,Q,C+1], mask [B,Q,H,W] (row stride Wm >= W so a padded mask can be cropped in place). */
int mss_m2f_score_f32(const float* cls, const float* mask, int B, int Q, int C, int H, int W, int Hm, int Wm,
                      float* score, void* stream);

/* Fused RelContrastiveLoss (lib/loss.py:34-156): value AND both gradients in a few streaming passes.
 * The host side (multishiftseg_amd/loss.py) owns the workspaces and the call order:
 *   pass1 -> select -> pass2 -> compact -> cin_bwd -> pairs x2 -> finalize
 * Nothing here synchronises with the host; every normaliser is read from device counters. */
typedef struct MssRclArgs {
  const float* logit;      /* [B,C,H,W] NCHW                                          */
  const float* score;      /* [B,H,W]                                                 */
  int64_t* target;         /* [B,H,W] int64; mutated exactly like loss.py:110-111,115 */
  int B, C, H, W;
  float w_ce_orig, w_ce_aug, w_contras;   /* ce_weights[0], ce_weights[1], contras_weight */
  float m0, m1, m2;                        /* inoutaug_contras_margins_tri                 */
  int select;                              /* conduct_pixel_selection && 0 < ratio < 1     */
  float selection_ratio;
} MssRclArgs;
/* counters: double[16]; slots 0 sum_ce_orig, 1 n_in_orig, 2 n_in_aug, 3 n_ood, 4 sum_c_in,
 * 5 n_same_in, 6 sum_ce_aug_all, 7 sum_selected_ce, 8 n_selected, 9 sum_c_orig, 10 sum_c_aug,
 * 11 n_pairs, 12 labels in [C, 99) or < 0 (F.nll_loss raises on those, loss.py:59: here the loss turns NaN and
 * out[6] of finalize reports the count). pass1 zeroes them.
 * pass1 (loss.py:46-69,90-96): lse[B*H*W] (only the augmented half is guaranteed to be written), ce_aug[(B/2)*H*W]
 * (+inf where ignored), kind[B*H*W] (0 void / 1 in-distribution / 2 OOD, from the UNmutated targets); B must be even
 * ([orig...; aug...] pairs). dlogit (optional [B,C,H,W]): pass1 also writes the gradient of every pixel whose weight
 * does not depend on the selection: the original half always, the augmented half when a->select == 0. */
int mss_rcl_pass1_f32(const MssRclArgs* a, float* lse, float* ce_aug, uint8_t* kind, double* counters, float* dlogit,
                      void* stream);
/* exact k-th smallest of ce_aug, k = int(float32(ratio) * float32(n_in_aug)) as torch computes
 * it (loss.py:98-99); replaces torch.topk (loss.py:102). hist_ws: uint32[256]; sel: uint32[8]
 * = {threshold key, n_less, k, n_equal_to_take, tie tickets, ...}. */
int mss_rcl_select_f32(const float* ce_aug, long long n, const double* counters, float selection_ratio,
                       uint32_t* hist_ws, uint32_t* sel, void* stream);
/* the same selection, bit-identical sel[0..3], in 5 launches instead of 9 (each pick rides in front of the next byte's histogram
 * pass): what mss_rcl_loss_device_f32 runs. scratch: MSS_RCL_SELECT_SCRATCH_WORDS uint32 (four histograms + two state buffers),
 * cleared by the call unless scratch_zeroed != 0. */
#define MSS_RCL_SELECT_SCRATCH_WORDS (4 * 256 + 16)
int mss_rcl_select_merged_f32(const float* ce_aug, long long n, const double* counters, float selection_ratio,
                              uint32_t* scratch, int scratch_zeroed, uint32_t* sel, void* stream);
/* the same selection one radix pass at a time (shift = 24, 16, 8, 0): a data-parallel caller
 * all-reduces the 256-bin histogram between hist and pick -> exact GLOBAL k-th smallest. */
int mss_rcl_select_init_f32(const double* counters, float selection_ratio, uint32_t* hist_ws, uint32_t* sel,
                            void* stream);
int mss_rcl_select_hist_f32(const float* ce_aug, long long n, const uint32_t* sel, int shift, uint32_t* hist_ws,
                            void* stream);
int mss_rcl_select_pick_f32(uint32_t* sel, uint32_t* hist_ws, int shift, void* stream);
/* pass2 (selection mode only; a no-op when a->select == 0): the AUGMENTED half of dlogit (NCHW, assigned; may be
 * NULL) = grad_scale * d loss / d logit for the selected pixels and 0 for the others, target mutation, counters[7..8].
 * The original half was written by pass1, unscaled: grad_scale must be 1 (MSS_ERR_BAD_ARG otherwise; scale the loss
 * gradient downstream). */
int mss_rcl_pass2_f32(const MssRclArgs* a, const float* lse, const float* ce_aug, const uint8_t* kind,
                      uint32_t* sel, double* counters, float grad_scale, float* dlogit, void* stream);
/* row-major ordered compaction of pixel indices into the three sets of loss.py:122-124
 * (in-dist original half, in-dist augmented half, OOD); sizes to n_out (uint32[3]);
 * block_counts: uint32[3 * mss_rcl_num_compact_blocks(B,H,W)]. */
int mss_rcl_num_compact_blocks(int B, int H, int W);
int mss_rcl_compact_f32(const uint8_t* kind, int B, int H, int W, int32_t* idx_orig, int32_t* idx_aug,
                        int32_t* idx_ood, uint32_t* block_counts, uint32_t* n_out, void* stream);
/* dscore of the consistency term (loss.py:141-145); ASSIGNS all of dscore, call before pairs. */
int mss_rcl_cin_bwd_f32(const MssRclArgs* a, const uint8_t* kind, const double* counters, float grad_w,
                        float* dscore, void* stream);
/* hinge over n pairs (loss.py:129-137): sum_i relu(score[idx_a[perm_a[i]]] + margin -
 * score[idx_o[perm_o[i]]]) -> counters[9 + slot]; dscore += -+ grad_w/n (atomics; may be NULL).
 * perm_*: int64 permutations as torch.randperm returns them (parity mode: the caller injects
 * the reference's own permutations). */
int mss_rcl_pairs_f32(const float* score, const int32_t* idx_a, const int64_t* perm_a, const int32_t* idx_o,
                      const int64_t* perm_o, long long n, float margin, double* counters, int slot, float grad_w,
                      float* dscore, void* stream);
/* same, but n = min(max_samples, n_out[0..2]) and the two permutations are keyed Feistel
 * bijections evaluated on the fly: no host round trip, no materialised randperm. */
int mss_rcl_pairs_device_f32(const float* score, const int32_t* idx_a, const int32_t* idx_o, const uint32_t* n_out,
                             int set_a, long long max_samples, uint32_t seed_a, uint32_t seed_o, float margin,
                             double* counters, int slot, float grad_w, float* dscore, void* stream);
/* Both hinge terms of the device-pairing mode in one launch: pair i of set `orig` (slot 0, margin_orig) and of set `aug` (slot 1,
 * margin_aug) against OOD element feistel(i, n_ood, seed_ood); same sums, same gradients as two mss_rcl_pairs_device_f32 calls
 * with (set 0, seed_orig) and (set 1, seed_aug). */
int mss_rcl_pairs_device2_f32(const float* score, const int32_t* idx_orig, const int32_t* idx_aug, const int32_t* idx_ood,
                              const uint32_t* n_out, long long max_samples, uint32_t seed_orig, uint32_t seed_aug, uint32_t seed_ood,
                              float margin_orig, float margin_aug, double* counters, float grad_w, float* dscore, void* stream);
/* data-parallel pairing over the rank-major concatenation of all ranks' sets: this rank owns the
 * slice [a_off, a_off+a_cnt_local) of global set A (a_cnt_global elements); ood_all is the
 * all-gathered [W][cap] OOD score vector with exclusive global offsets ood_off[W+1]; gradients w.r.t.
 * OOD scores are accumulated into g_ood [W][cap] (the caller all-reduces it and scatters its row). */
int mss_rcl_pairs_global_f32(const float* score, const int32_t* idx_a, uint32_t a_off, uint32_t a_cnt_local,
                             uint32_t a_cnt_global, const float* ood_all, const uint32_t* ood_off, int W,
                             uint32_t cap, uint32_t n_pairs, uint32_t seed_a, uint32_t seed_o, float margin,
                             double* counters, int slot, float coef, float* dscore, float* g_ood, void* stream);
int mss_rcl_gather_f32(const float* src, const int32_t* idx, uint32_t n, float* dst, void* stream);
int mss_rcl_scatter_add_f32(const float* g, const int32_t* idx, uint32_t n, float* dst, void* stream);
/* The whole loss of the device-pairing mode in ONE call (pass 1, radix select, pass 2, compaction, the in-distribution hinge, both
 * paired hinges, finalize -- the launches of the entry points above, issued back to back): workspace of
 * mss_rcl_workspace_bytes(B, H, W) bytes (256-byte aligned, contents irrelevant), max_samples = int(B*H*W * sample_ratio),
 * seed = the caller's step counter (pair i of step s couples Feistel(i; s) elements), dlogit / dscore nullable, out float[8] as
 * mss_rcl_finalize_f32. No host synchronisation. */
long long mss_rcl_workspace_bytes(int B, int H, int W);
int mss_rcl_loss_device_f32(const MssRclArgs* a, void* workspace, long long workspace_bytes, long long max_samples, uint32_t seed,
                            float* dlogit, float* dscore, float* out, void* stream);
/* out: float[8] = {loss, ce_orig, ce_aug, c_orig, c_aug, c_in, -, -} (loss.py:73-88,147). */
int mss_rcl_finalize_f32(const MssRclArgs* a, const double* counters, const uint32_t* sel, float* out,
                         void* stream);

/* Adam with L2-coupled weight decay (torch.optim.Adam semantics, train_deeplab.py:134-149), one parameter tensor per
 * call, updated in place together with its two moment buffers. Hyper-parameters are doubles, as Python holds them: the
 * bias corrections 1 - beta^step, the step size lr / bias1 and sqrt(bias2) are formed in double on the host and only the
 * derived scalars are rounded to float (torch/optim/adam.py, single-tensor path); step counts from 1. */
int mss_adam_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n, double lr,
                      double beta1, double beta2, double eps, double weight_decay, int step, void* stream);

/* Mask2Former anomaly score fused with the mask upsample (csrc/m2f.hip): logit = pixel-major
 * low-resolution mask logits [B, hm, wm, ldq] (queries contiguous; produced by mss_conv2d_forward_f32 in batched 1x1
 * mode from mask_features and mask_embed = einsum("bqc,bchw->bqhw"), mask2former_transformer_decoder.py:544-548);
 * the kernel applies F.interpolate(size=(Hi,Wi), bilinear, align_corners=False) (maskformer_model.py:264-277), the
 * sigmoid, the class mix and 1 - max (train_m2f.py:387-407) and writes the crop [B,H,W]. Q % 4 == 0, C <= 20. */
int mss_m2f_fused_score_f32(const float* cls, const float* logit, int B, int Q, int C, int hm, int wm, int ldq, int Hi,
                            int Wi, int H, int W, float* score, void* stream);
/* The same with a scratch of B * Q * 32 floats: the class mix then runs on the matrix cores (v_mfma_f32_32x32x2_f32: [classes x Q] x
 * [Q x pixels] per 32-pixel strip, one interpolation + sigmoid per lane and MFMA) instead of 20 multiply-adds per (pixel, query)
 * on the vector ALUs; results agree with the form above to fp32 summation order. prob_ws NULL = the form above. */
int mss_m2f_fused_score_ws_f32(const float* cls, const float* logit, int B, int Q, int C, int hm, int wm, int ldq, int Hi,
                               int Wi, int H, int W, float* score, float* prob_ws, void* stream);

/* Pixel-level OOD metrics on the device (csrc/metric.hip): exact AUROC / average precision / FPR at `recall_level`
 * over all pixels with label id_out (positives) and id_in (negatives). Replaces eval_ood_measure, get_measures and
 * fpr_and_fdr_at_recall (lib/utils/metric.py:170-180, 130-153, 87-127; callers test_deeplab.py:94-102,
 * train_deeplab.py:236-241), which run sklearn on host copies of every score map.
 *   compact:  one pass over a batch of n pixels: order-preserving u32 keys of the id_in pixels' scores packed
 *             at the front of keys[0..n), those of the id_out pixels at the back; counts (device u64[2], zero on
 *             entry) = {#id_in, #id_out}. No host synchronisation per batch.
 *   sort:     ascending key sort (own 4-pass LSD radix sort, csrc/metric.hip; keys 4-byte aligned, n < 2^32), temp sized
 *             by mss_oodm_sort_temp_bytes
 *   measures: out[0..2] = {AUROC, AUPRC, FPR@recall_level} (device f64[3]); u2_part / ap_part: device scratch of
 *             mss_oodm_rank_blocks(P) elements each. P, N >= 1 (the host mirror returns None otherwise, as
 *             metric.py:176-180 does). */
int mss_oodm_compact_f32(const float* score, const long long* label, long long n, long long id_in, long long id_out,
                         unsigned int* keys, unsigned long long* counts, void* stream);
/* The same with both totals in ONE counter: *packed_count (zero on entry) ends as #id_in | (#id_out << 32); n < 2^32. Half the
 * same-address atomics of the form above (they were most of the kernel's time on a 1024 x 2048 map). */
int mss_oodm_compact_packed_f32(const float* score, const long long* label, long long n, long long id_in, long long id_out,
                                unsigned int* keys, unsigned long long* packed_count, void* stream);
/* Round 6: the same on EIGHT counters (the single counter is one address 512 workgroups of a 1024 x 2048 map add to one after the other:
 * the kernel's 20 us). The 4096-pixel chunks are dealt round-robin onto 8 lanes; lane L owns keys[L * cap, (L + 1) * cap) with
 * cap = mss_oodm_compact_lanes_cap(n) (keys: 8 * cap slots), its id_in keys from the front of the segment, its id_out keys from the
 * back; lane_counts (device u64[8], zero on entry)[L] = #id_in | (#id_out << 32) of the lane. n < 2^32. */
long long mss_oodm_compact_lanes_cap(long long n);
/* ... and for up to MSS_OODM_BATCH maps in one launch (an evaluation sweep that holds its score maps: test_deeplab.py:84-102 appends
 * every batch and evaluates at the end): entry m as the arguments of mss_oodm_compact_lanes_f32; entries >= count are ignored. */
#define MSS_OODM_BATCH 16
typedef struct MssOodmBatch {
  const float* score[MSS_OODM_BATCH];
  const long long* label[MSS_OODM_BATCH];
  unsigned int* keys[MSS_OODM_BATCH];
  unsigned long long* lane_counts[MSS_OODM_BATCH];
  long long n[MSS_OODM_BATCH];
} MssOodmBatch;
int mss_oodm_compact_lanes_batch_f32(const MssOodmBatch* batch, int count, long long id_in, long long id_out, void* stream);
/* One map's eight lane segments (keys, cap, lane_counts as mss_oodm_compact_lanes_f32 left them) copied to neg_out[0 .. #id_in) and
 * pos_out[0 .. #id_out) of that map: the sweep's contiguous sort inputs, the caller advancing the two pointers by the map's totals. */
int mss_oodm_gather_lanes_u32(const unsigned int* keys, long long cap, const unsigned long long* lane_counts, unsigned int* neg_out,
                              unsigned int* pos_out, void* stream);
int mss_oodm_compact_lanes_f32(const float* score, const long long* label, long long n, long long id_in, long long id_out,
                               unsigned int* keys, unsigned long long* lane_counts, void* stream);
long long mss_oodm_sort_temp_bytes(long long n);
int mss_oodm_sort_u32(const unsigned int* keys_in, unsigned int* keys_out, long long n, void* temp, long long temp_bytes,
                      void* stream);
int mss_oodm_rank_blocks(long long P);
int mss_oodm_measures_f64(const unsigned int* pos_sorted, long long P, const unsigned int* neg_sorted, long long N,
                          double recall_level, unsigned long long* u2_part, double* ap_part, double* out, void* stream);

/* ---- Mask2Former pixel decoder glue (csrc/norm.hip; msdeformattn.py:116-131,215-219,262-281,314-358) ----
 * y = LayerNorm(x + res) over the last dimension C (multiple of 256, <= 1024; res may be NULL): the post-norm residual
 * sites of MSDeformAttnTransformerEncoderLayer in one pass. stat (optional) [rows][2] = (mean, rstd) for the backward. */
int mss_add_layernorm_f32(const float* x, const float* res, long long rows, int C, const float* gamma, const float* beta,
                          float eps, float* y, float* stat, void* stream);
/* backward: dz [rows][C] (= gradient w.r.t. x and w.r.t. res), dgamma / dbeta [C] (optional). ws: scratch of
 * mss_add_layernorm_bwd_workspace_floats floats; per-workgroup partials added in a fixed order (no atomics). */
long long mss_add_layernorm_bwd_workspace_floats(long long rows, int C);
int mss_add_layernorm_bwd_f32(const float* gy, const float* x, const float* res, const float* stat, long long rows, int C,
                              const float* gamma, float* dz, float* dgamma, float* dbeta, float* ws, void* stream);
/* the same, and dzsum [C] = per-channel sums of dz: the bias gradient of the Linear whose output was `res`
 * (msdeformattn.py:116-131: output_proj / linear2) without another pass over [rows][C]. ws: 3/2 of the floats above. */
int mss_add_layernorm_bwd_sum_f32(const float* gy, const float* x, const float* res, const float* stat, long long rows, int C,
                                  const float* gamma, float* dz, float* dgamma, float* dbeta, float* dzsum, float* ws,
                                  void* stream);
/* r04: the output of an encoder layer's second LayerNorm has two consumers in the next layer, the layer input `src` and the
 * query q = src + pos (msdeformattn.py:116-118 with_pos_embed). mss_add_layernorm_q_f32 also writes q = y + pos[row % pos_rows]
 * (pos: [pos_rows][C], one image's tokens when the batch shares them); mss_add_layernorm_bwd_sum2_f32 takes the two gradients
 * (gy2 may be NULL) and adds them while loading. Same bits as the separate elementwise passes they replace. */
int mss_add_layernorm_q_f32(const float* x, const float* res, long long rows, int C, const float* gamma, const float* beta,
                            float eps, float* y, float* stat, const float* pos, long long pos_rows, float* q, void* stream);
int mss_add_layernorm_bwd_sum2_f32(const float* gy, const float* gy2, const float* x, const float* res, const float* stat, long long rows,
                                   int C, const float* gamma, float* dz, float* dgamma, float* dbeta, float* dzsum, float* ws,
                                   void* stream);
/* nn.GroupNorm(groups, C) on NHWC x [N][HW][C] (pixel stride ldx, sample stride x_sample_stride floats), optional ReLU,
 * output with its own pixel / sample strides (e.g. straight into the encoder's token buffer [N][sum HW][C]).
 * C/groups a multiple of 4, C <= 1024. ws: scratch of mss_groupnorm_workspace_floats floats. Deterministic. */
long long mss_groupnorm_workspace_floats(int N, int HW, int C, int groups);
int mss_groupnorm_nhwc_f32(const float* x, int ldx, long long x_sample_stride, int N, int HW, int C, int groups,
                           const float* gamma, const float* beta, float eps, int relu, float* y, int ldy,
                           long long y_sample_stride, float* ws, void* stream);
/* backward of mss_groupnorm_nhwc_f32 (the pixel decoder is trainable in Mask2Former's second stage): stat = the forward's
 * [N*groups][2] (mean, rstd), left by the forward at ws + mss_groupnorm_stat_offset(N, HW, C) floats; relu = the forward
 * fused a ReLU. dx contiguous NHWC (pixel stride lddx), dgamma / dbeta [C] optional. Deterministic (fixed-order sums). */
long long mss_groupnorm_stat_offset(int N, int HW, int C);
long long mss_groupnorm_bwd_workspace_floats(int N, int HW, int C, int groups);
int mss_groupnorm_nhwc_bwd_f32(const float* gy, int ldg, long long g_sample_stride, const float* x, int ldx,
                               long long x_sample_stride, int N, int HW, int C, int groups, const float* stat,
                               const float* gamma, const float* beta, int relu, float* dx, int lddx, float* dgamma,
                               float* dbeta, float* ws, void* stream);
/* transpose of the bilinear part of mss_upsample_bilinear_add_nhwc_f32: dtop (+)= B^T dy (the lateral gradient is dy). */
int mss_upsample_bilinear_bwd_nhwc_f32(const float* dy, int lddy, int N, int OH, int OW, float* dtop, int ldt,
                                       long long top_sample_stride, int IH, int IW, int C, int accumulate, void* stream);
/* NCHW gradient -> rows of an NHWC / token buffer (pixel stride ldd, sample stride d_sample_stride floats), optional +=. */
int mss_nchw_to_nhwc_strided_f32(const float* g, int N, int C, int HW, float* dst, int ldd, long long d_sample_stride,
                                 int accumulate, void* stream);
/* FPN top-down step (msdeformattn.py:344): y = lat + F.interpolate(top, size=(OH, OW), mode="bilinear",
 * align_corners=False); NHWC with pixel strides (top also with a sample stride: a level inside the token buffer). */
int mss_upsample_bilinear_add_nhwc_f32(const float* top, int ldt, long long top_sample_stride, int N, int IH, int IW,
                                       const float* lat, int ldl, float* y, int ldy, int OH, int OW, int C, void* stream);
/* NHWC (pixel stride ldx, sample stride x_sample_stride floats) -> contiguous NCHW: the decoder returns NCHW maps like
 * the reference. */
int mss_nhwc_to_nchw_f32(const float* x, int ldx, long long x_sample_stride, int N, int HW, int C, float* y, void* stream);

/* ---- on-device data path of the DeepLab trainer (csrc/data.hip) ----
 * One kernel for what DiverseCityscapes.__getitem__ + its transforms + the trainer's batch concat do per step
 * (lib/dataset/cityscapes.py:153-171, lib/utils/img_utils.py:110-153,246-259,398-435, train_deeplab.py:190-195):
 * img / gen [B,H,W,3] uint8 (original and generated image, pre-decoded, device), tgt / gen_tgt [B,H,W] uint8;
 * mix_p [B] double (device; NULL = no mixup): gen <- uint8(p*img + (1-p)*gen) in float64; crop [B][2] int (device): top,
 * left of the common h x w window; flip [B] int (device, NULL = none): horizontal flip of the window; mean3 / std3:
 * HOST double[3] (the Python floats of opt.data.mean / std; Normalize uses their float32 roundings, the pasted object the
 * doubles, as the reference does); obj_img [B][OHmax][OWmax][3] float32 (0..255, already rescaled), obj_mask
 * [B][OHmax][OWmax] uint8, obj_geom [B][6] int = (y1, x1, bh, bw, h0, w0): the mask's bounding box inside the object and
 * the paste corner in the window (bh = 0: nothing pasted; all three NULL: no anomaly mix). Outputs: out_img [2B,3,h,w]
 * float32 and out_tgt [2B,h,w] int64, originals first, then the augmented images. */
int mss_data_pair_f32(const uint8_t* img, const uint8_t* gen, const uint8_t* tgt, const uint8_t* gen_tgt, int B, int H, int W,
                      int h, int w, const double* mix_p, const int* crop, const int* flip, const double* mean3,
                      const double* std3, const float* obj_img, const uint8_t* obj_mask, const int* obj_geom, int OHmax,
                      int OWmax, float* out_img, int64_t* out_tgt, void* stream);

/* Calibration kernels (bench.py: achievable peaks of this device next to the datasheet ones).
 * mss_peak_mfma_f32: blocks x 4 waves x iters x 16 back-to-back v_mfma_f32_32x32x2_f32 (4096 FLOP
 * each), out >= blocks*256 floats. mss_peak_stream_f32: float4 copy of n floats (8*n bytes moved);
 * variant 0..3 = plain / 4 loads in flight / + nontemporal / + contiguous 16-KB chunks per workgroup; round 6: 4 = write-only (4*n bytes),
 * 5 = read-only (4*n bytes), 6 = first half of src read, all of dst written (6*n bytes: the 1 : 2 mix of the Winograd input transforms); n % 8 == 0. */
int mss_peak_mfma_f32(float* out, int blocks, int iters, void* stream);
/* The same for the bf16 matrix cores under load (the split-bf16 GEMM route's ceiling): register-only loops on pseudo-random operands,
 * blocks x 4 waves x iters x 1 572 864 FLOP; shape 0 = 48 x v_mfma_f32_32x32x16_bf16 per iteration, 1 = 96 x v_mfma_f32_16x16x32_bf16. */
int mss_peak_mfma_bf16(float* out, int blocks, int iters, int shape, void* stream);
/* Clock calibration: one wave spins for `ticks` of s_memrealtime; out[0] = s_memrealtime ticks, out[1] = s_memtime ticks of the span
 * (out: 2 x uint64 on the device). Timed from the host it gives the frequency of both counters. */
int mss_peak_clock(unsigned long long* out, unsigned long long ticks, void* stream);
int mss_peak_stream_f32(const float* src, float* dst, long long n, int variant, void* stream);
/* layout experiment behind the Winograd-domain layout: one coalesced float4 read, ns (16|36) float4 writes
 * into ns slabs that are n floats apart (blocked = 0) or adjacent per block of blk_floats floats (blocked = 1). */
int mss_peak_scatter_f32(const float* src, float* dst, long long n, int ns, int blocked, long long blk_floats,
                         void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MSS_HIP_H */
