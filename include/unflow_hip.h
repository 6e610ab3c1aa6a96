/*
 * libunflow_hip.so — C ABI of the MI355X-native UnFlow training-step kernels.
 *
 * This is the drop-in boundary for the reference's operator layer
 * (src/e2eflow/ops.py:69-107 and the host launchers declared in
 * ops/correlation_op.cc:22-36, ops/backward_warp_op.cc:22-31,
 * ops/forward_warp_op.cc:23-30, ops/downsample_op.cc:21-23), widened — as the
 * north star asks — to the conv/deconv stacks (slim.conv2d / conv2d_transpose in
 * src/e2eflow/core/flownet.py), image_warp (core/image_warp.py) and the loss
 * terms (core/losses.py) that the reference leaves to TensorFlow.
 *
 * Conventions
 *  - All tensors are device pointers to fp32 (the reference's ops are
 *    `float`-only: REGISTER_OP(... ": float") in every ops/[name]_op.cc).
 *  - The caller allocates every output and workspace; the library never
 *    allocates, frees or synchronises (TF's allocate_output replaced by
 *    caller-owned buffers).  Every entry takes a hipStream_t (as void*) and is
 *    fully asynchronous and re-entrant.  The only process-wide state is the
 *    option table below (kernel selection / planning switches), written solely
 *    through unflow_set_option; the library never reads the environment.
 *  - Return value: 0 on success, a negative UNFLOW_ERR_* otherwise; on error
 *    nothing is launched (mirrors OP_REQUIRES -> InvalidArgument).
 *  - Layouts at the op boundary are the reference's: correlation NCHW
 *    (correlation_op.cc:53-56,64), warps / downsample NHWC with flow channel
 *    0 = x/u, 1 = y/v (backward_warp_op.cu.cc:26-28).  The *_nhwc entry points
 *    and the conv/loss entry points are channels-last with an explicit channel
 *    stride ("ld") so that producers write straight into channel slices of a
 *    consumer's concat buffer.
 */
#ifndef UNFLOW_HIP_H_
#define UNFLOW_HIP_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* unflow_stream_t; /* hipStream_t */
typedef struct unflow_planes unflow_planes; /* 16-bit operand planes of a tensor (defined with the conv entry points) */

#define UNFLOW_OK 0
#define UNFLOW_ERR_NULL (-1)          /* null pointer argument                                   */
#define UNFLOW_ERR_EMPTY_OUTPUT (-2)  /* correlation_op.cc:60-61 "Invalid correlation settings"   */
#define UNFLOW_ERR_EVEN_KERNEL (-3)   /* correlation_op.h:16-17  "kernel_size must be odd"         */
#define UNFLOW_ERR_NOT_DIVISIBLE (-4) /* downsample_op.cc:37-40                                    */
#define UNFLOW_ERR_SHAPE (-5)         /* correlation_op.cc:47-48 "Input shapes have to be the same" and other shape errors */
#define UNFLOW_ERR_UNSUPPORTED (-7)   /* configuration outside what the kernels implement          */
#define UNFLOW_ERR_LAUNCH (-8)        /* hipGetLastError() after launch                            */
#define UNFLOW_ERR_WORKSPACE (-9)     /* workspace too small                                       */

const char* unflow_status_string(int status);
int unflow_version(void);

/* Library options (csrc/options.h lists them with their defaults): integer switches that select kernels and split
 * planning — e.g. "conv_math_fp32", "halo", "gather_max_split".  They replace what the reference fixes at build time
 * through its JIT compile flags (src/e2eflow/ops.py:21-48); set them once after loading the library, before the first
 * launch (a later change applies to the launches planned after it).  UNFLOW_ERR_UNSUPPORTED: no such option.
 * unflow_option_names: the names, '\n'-separated. */
int unflow_set_option(const char* name, int value);
int unflow_get_option(const char* name, int* value);
const char* unflow_option_names(void);

/* ===================================================================== */
/* Correlation — replaces Correlation()/CorrelationGrad()                 */
/* (ops/correlation_op.cc:22-36; kernels ops/correlation_op.cu.cc:30-248) */
/* ===================================================================== */

/* out3 = {out_channels, out_height, out_width}; geometry of correlation_op.h:36-51. */
int unflow_correlation_out_shape(int H, int W, int kernel_size, int max_displacement, int pad,
                                 int stride_1, int stride_2, int* out3);

/* Bytes of scratch the NCHW entry points ask for: channels-last fp32 staging copies (in0, in1, g0, g1, dout: the minimum they
 * accept) + room for the two inputs' bf16 x 3 operand planes where the matrix-core kernels take the shape (kernel_size 1,
 * stride_1 1, pad >= max_displacement, C % 16 == 0).  With the full size the entry points build the planes and run the
 * kernels of the training step (81-channel +-4 volume, 8 x 256 x 96 x 128: 271 / 652 us forward / backward instead of
 * 912 / 1298); with the minimum they run the fp32-input kernels. */
size_t unflow_correlation_workspace_bytes(int B, int C, int H, int W, int kernel_size,
                                          int max_displacement, int pad, int stride_1, int stride_2);

/* NCHW in, NCHW out (output 0 of the reference op; padded_0/1 are not materialised). */
int unflow_correlation_fwd(const float* in0, const float* in1, float* out, int B, int C, int H, int W,
                           int kernel_size, int max_displacement, int pad, int stride_1, int stride_2,
                           void* workspace, size_t workspace_bytes, unflow_stream_t stream);

/* CorrelationGrad (ops.py:94-104): (dOut, in0, in1) -> (grad0, grad1), all NCHW. */
int unflow_correlation_bwd(const float* dout, const float* in0, const float* in1, float* grad0, float* grad1,
                           int B, int C, int H, int W, int kernel_size, int max_displacement, int pad,
                           int stride_1, int stride_2, void* workspace, size_t workspace_bytes,
                           unflow_stream_t stream);

/* Channels-last form used inside the step.  in0/in1: [B,H,W,ld_in] (C channels used);
 * sample n of in0 is paired with sample (n + pair_shift) % B of in1 (pair_shift = B/2 with
 * in0 == in1 runs both flow directions of a [im1-features; im2-features] batch in one launch).
 * out: [B,oh,ow,ld_out], oc channels written. */
int unflow_correlation_nhwc_fwd(const float* in0, const float* in1, int ld_in, int pair_shift, float* out,
                                int ld_out, int B, int C, int H, int W, int kernel_size, int max_displacement,
                                int pad, int stride_1, int stride_2, unflow_stream_t stream);

/* grad0[n] = d/d in0[n]; grad1[m] = d/d in1[m] with m = (n + pair_shift) % B.
 * If accumulate_g1_into_g0 != 0 (requires in0 == in1 storage), both are summed into grad0
 * (grad1 may be NULL): the gradient wrt the shared feature tensor. */
int unflow_correlation_nhwc_bwd(const float* dout, int ld_dout, const float* in0, const float* in1, int ld_in,
                                int pair_shift, float* grad0, float* grad1, int ld_grad,
                                int accumulate_g1_into_g0, int B, int C, int H, int W, int kernel_size,
                                int max_displacement, int pad, int stride_1, int stride_2, unflow_stream_t stream);

/* ===================================================================== */
/* Warps / downsample — replace BackwardWarp(), BackwardWarpGrad(),        */
/* ForwardWarp(), ForwardWarpGrad(), Downsample()                          */
/* ===================================================================== */

/* ops/backward_warp_op.cu.cc:14-68: zero outside the image, floorf(float(x)+u). */
int unflow_backward_warp_fwd(const float* images, const float* flows, float* out, int B, int H, int W, int C,
                             unflow_stream_t stream);
/* ops/backward_warp_op.cu.cc:70-138: gradient wrt flows only (ops.py:80-84). */
int unflow_backward_warp_bwd(const float* dout, const float* images, const float* flows, float* dflows,
                             int B, int H, int W, int C, unflow_stream_t stream);
/* (x0,y0) integer tap corner per pixel, [B,H,W,2] int32 — for bit-exact index checks. */
int unflow_backward_warp_indices(const float* flows, int* xy0, int B, int H, int W, unflow_stream_t stream);

/* src/e2eflow/core/image_warp.py:4-76: clamp-to-edge, x + int(floor(u)); what the training graph uses.
 * im: [B_im,H,W,ld_im] (C channels used); output sample n reads image sample (n + pair_shift) % B. */
int unflow_image_warp_fwd(const float* im, int ld_im, const float* flow, float flow_scale, float* out,
                          int* idx4 /* optional [B,H,W,4] gather indices, may be NULL */, int pair_shift,
                          int B, int H, int W, int C, unflow_stream_t stream);
/* TF autodiff of that graph: d_im (scatter-add; may be NULL; must be zero-filled by the caller or
 * hold a running sum) and d_flow (multiplied by flow_scale, the derivative of flow*flow_scale). */
int unflow_image_warp_bwd(const float* dout, const float* im, int ld_im, const float* flow, float flow_scale,
                          float* d_im, float* d_flow, int accumulate_d_flow, int pair_shift, int B, int H, int W,
                          int C, unflow_stream_t stream);

/* ops/forward_warp_op.cu.cc:16-65.  deterministic == 0: float-atomic scatter like the reference
 * (summation order varies run to run).  deterministic != 0: the same scatter accumulated in 2^31-scaled
 * 64-bit integers (order-independent, bit-reproducible); needs workspace >= 8*B*H*W bytes.
 * With workspace >= unflow_forward_warp_workspace_bytes (either mode) sources whose 9 x 9 footprint leaves their
 * tile's LDS window are binned by target tile and gathered without global atomics (flows of tens of pixels: 10x faster);
 * with less, the same sums are formed with global atomics. */
size_t unflow_forward_warp_workspace_bytes(int B, int H, int W, int deterministic);
int unflow_forward_warp_fwd(const float* flows, float* out, int B, int H, int W, int deterministic,
                            void* workspace, size_t workspace_bytes, unflow_stream_t stream);
int unflow_forward_warp_bwd(const float* dout, const float* flows, float* dflows, int B, int H, int W,
                            unflow_stream_t stream);
/* {x_lo,x_hi,y_lo,y_hi} splat footprint per pixel ([B,H,W,4] int32; -1 when rejected). */
int unflow_forward_warp_ranges(const float* flows, int* ranges, int B, int H, int W, unflow_stream_t stream);

/* ops/downsample_op.cu.cc:15-49 (box mean); H,W must be divisible by scale (downsample_op.cc:37-40). */
int unflow_downsample_fwd(const float* images, float* out, int B, int H, int W, int C, int scale,
                          unflow_stream_t stream);
/* tf.image.resize_area(images, [out_h, out_w]) (core/util.py:12-14 and the odd-size branch of util.downsample, :26): area-weighted
 * mean of the source rectangle of every output pixel, NHWC. */
int unflow_resize_area(const float* images, float* out, int B, int H, int W, int C, int out_h, int out_w,
                       unflow_stream_t stream);
/* The loss pyramid's image chain downsample(., 4), downsample(., 2) x 4 (unsupervised.py:99-100,145-146) on 3-channel images in
 * one launch; levels[k] = [N, H / (4 << k), W / (4 << k), 3], k = 0..4, each bit-identical to the chained unflow_downsample_fwd
 * calls.  H, W multiples of 64 (UNFLOW_ERR_NOT_DIVISIBLE otherwise). */
int unflow_image_pyramid5(const float* images, float* const* levels, int N, int H, int W, unflow_stream_t stream);

/* ===================================================================== */
/* conv / deconv stacks — slim.conv2d / slim.conv2d_transpose of           */
/* src/e2eflow/core/flownet.py:89-237, channels-last, TF 'SAME' padding.   */
/* Channel counts of the gathered operand must be multiples of 4 (pad the  */
/* buffers; padded weights stay zero).                                     */
/* ===================================================================== */

/* y[b,oy,ox,0:Cout] = act(bias + sum_{ky,kx,ci} x[b,oy*s-pt+ky, ox*s-pl+kx, ci] * w[ky,kx,ci,co])
 * x: [B,H,W,ldx] (Cin used), w: HWIO [k,k,Cin,Cout], y: [B,ceil(H/s),ceil(W/s),ldy].
 * leaky != 0 applies max(0.1*v, v) (flownet.py:84-86). bias may be NULL. */
int unflow_conv2d_fwd(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy, int B,
                      int H, int W, int Cin, int Cout, int k, int stride, int leaky, void* workspace,
                      size_t workspace_bytes, unflow_stream_t stream);

/* dx[.,0:Cin] (+)= conv-transpose of dz with w.  Epilogue: if accumulate, adds the existing dx first;
 * then, for channels c in [act_lo, act_hi), multiplies by leaky'(act_src[., c]) (1 if >0 else 0.1),
 * turning d(output) of the producing layer into d(pre-activation).  act_src may be NULL. */
int unflow_conv2d_bwd_data(const float* dz, int lddz, const float* w, float* dx, int lddx, int B, int H, int W,
                           int Cin, int Cout, int k, int stride, int accumulate, const float* act_src,
                           int ld_act, int act_lo, int act_hi, void* workspace, size_t workspace_bytes,
                           unflow_stream_t stream);

/* dw[k,k,Cin,Cout] = sum over sites; dbias[Cout] (may be NULL) = column sums of dz.  Deterministic
 * (split partials in workspace, fixed-order reduce). */
int unflow_conv2d_bwd_filter(const float* x, int ldx, const float* dz, int lddz, float* dw, float* dbias, int B,
                             int H, int W, int Cin, int Cout, int k, int stride, void* workspace,
                             size_t workspace_bytes, unflow_stream_t stream);

/* slim.conv2d_transpose(k=4, stride=2, SAME): y is [B,2H,2W,ldy]; w: [4,4,Cout,Cin] (TF layout). */
int unflow_conv2d_transpose_fwd(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy,
                                int B, int H, int W, int Cin, int Cout, int leaky, void* workspace,
                                size_t workspace_bytes, unflow_stream_t stream);
int unflow_conv2d_transpose_bwd_data(const float* dz, int lddz, const float* w, float* dx, int lddx, int B, int H,
                                     int W, int Cin, int Cout, int accumulate, const float* act_src, int ld_act,
                                     int act_lo, int act_hi, void* workspace, size_t workspace_bytes,
                                     unflow_stream_t stream);
int unflow_conv2d_transpose_bwd_filter(const float* x, int ldx, const float* dz, int lddz, float* dw, float* dbias,
                                       int B, int H, int W, int Cin, int Cout, void* workspace,
                                       size_t workspace_bytes, unflow_stream_t stream);

/* All bias gradients of a step in two launches: out[i][c] = sum over the npix[i] rows of x[i][., c]
 * (x[i]: [npix[i], ld[i]] slice with C[i] columns).  The pointer/size arrays are HOST arrays, n <= 32.
 * Deterministic (fixed-order chunked reduction). */
size_t unflow_colsum_batched_workspace_bytes(int n, const int* C);
int unflow_colsum_batched(int n, const float* const* x, const int* ld, const long* npix, const int* C,
                          float* const* out, void* workspace, size_t workspace_bytes, unflow_stream_t stream);

/* Exact workspace requirement of the conv/deconv entry points above for these dimensions. */
size_t unflow_conv_workspace_bytes(int B, int H, int W, int Cin, int Cout, int k, int stride);

/* dz[., c] = dy[., c] * leaky'(y[., c]) in place over a [npix, C] slice (stand-alone form of the epilogue). */
int unflow_leaky_bwd_inplace(float* dy, int lddy, const float* y, int ldy, long npix, int C, unflow_stream_t stream);

/* ===================================================================== */
/* loss terms — src/e2eflow/core/losses.py, fused per pyramid level.        */
/* A "directed batch" holds both flow directions: samples [0,B) forward,    */
/* [B,2B) backward; image sample n is paired with (n + B) % 2B.             */
/* ===================================================================== */

/* gray[b,y,x] = 255 * (0.2989 R + 0.5870 G + 0.1140 B)  (losses.py:94, tf.image.rgb_to_grayscale). */
int unflow_rgb_to_gray255(const float* im, int ld_im, float* gray, long npix, unflow_stream_t stream);

/* image_warp of a 3-channel image followed by the grayscale above, fused (losses.py:22-23,94);
 * taps are re-derived in the backward pass. out_gray: [N,H,W]. */
int unflow_warp_gray_fwd(const float* im, int ld_im, const float* flow, float flow_scale, float* out_gray,
                         int pair_shift, int N, int H, int W, unflow_stream_t stream);
/* d_flow (+)= flow_scale * d(gray_warped)/d(flow) * d_gray. */
int unflow_warp_gray_bwd(const float* d_gray, const float* im, int ld_im, const float* flow, float flow_scale,
                         float* d_flow, int accumulate, int pair_shift, int N, int H, int W,
                         unflow_stream_t stream);

/* ternary_loss (losses.py:90-122) on gray images: per pixel soft-Hamming distance of the census
 * transforms, Charbonnier (alpha .45, eps 1e-3), masked by mask*interior(max_distance), summed into
 * loss_acc[0] scaled by `weight`/(normalizer).  dist_out [N,H,W] receives d(weighted loss)/d(distance) per
 * pixel (mask, interior and weight folded in) — the only thing the backward pass needs.  mask: [N_mask,H,W] with
 * sample n using mask[n % N_mask].  max_distance <= 4 (the reference uses 1..3, unsupervised.py:88). */
int unflow_ternary_fwd(const float* gray1, const float* gray2w, const float* mask, int n_mask, float* dist_out,
                       float* loss_acc, float weight, float normalizer, int max_distance, int N, int H, int W,
                       unflow_stream_t stream);
/* d_gray2w = d(weight * loss)/d(gray2w)  (gather form, no atomics); `dist` is dist_out of unflow_ternary_fwd (the
 * mask / weight arguments are kept for signature symmetry and ignored). */
int unflow_ternary_bwd(const float* gray1, const float* gray2w, const float* mask, int n_mask,
                       const float* dist, float* d_gray2w, float weight, float normalizer, int max_distance,
                       int N, int H, int W, unflow_stream_t stream);

/* Fused forms the step driver uses (same arithmetic, fewer launches per pyramid level):
 * unflow_gray_pair        = unflow_rgb_to_gray255(im) + unflow_warp_gray_fwd(im, flow) in one launch;
 * unflow_ternary_warp_bwd = unflow_ternary_bwd followed by unflow_warp_gray_bwd, without materialising d_gray2w
 *                           (`dist` = the per-pixel weights left by unflow_ternary_fwd). */
int unflow_gray_pair(const float* im, int ld_im, const float* flow, float flow_scale, float* gray1, float* gray2w,
                     int pair_shift, int N, int H, int W, unflow_stream_t stream);
int unflow_ternary_warp_bwd(const float* gray1, const float* gray2w, const float* dist, const float* im, int ld_im,
                            const float* flow, float flow_scale, float* d_flow, int accumulate, int pair_shift,
                            int max_distance, int N, int H, int W, unflow_stream_t stream);

/* second_order_loss (losses.py:258-295) on flow*flow_scale; loss_acc[0] += weight * sum/normalizer;
 * d_flow (+)= gradient wrt the raw flow (includes flow_scale).  Either output may be NULL. */
int unflow_second_order_fwd_bwd(const float* flow, float flow_scale, float* loss_acc, float* d_flow,
                                int accumulate, float weight, float normalizer, int N, int H, int W,
                                unflow_stream_t stream);

/* Masks and the non-data terms of compute_losses (losses.py:25-73) in one pass over the directed batch:
 *   mask = base_mask (broadcast over n_base samples; the down-sampled border mask) or, when base_mask is NULL,
 *          create_outgoing_mask(flow*flow_scale) (losses.py:347-366);
 *   fb_occ = |f + w|^2 > 0.01(|f|^2 + |w|^2) + 0.5 with w = warped_other*flow_scale (losses.py:43-49);
 *   disocc_other = fwarp[(n+pair_shift)%N] < 0.8 (forward_warp of the partner's flow, losses.py:28-29);
 *   occlusion_mode 0: none, 1: 'fb' (mask *= 1-fb_occ), 2: 'disocc' (mask *= 1-disocc_other) (losses.py:51-56);
 *   loss_acc += occ_weight*charb(1-mask) + sym_weight*charb((1-mask)-disocc_other) + fb_weight*charb(f+w, mask),
 *   each with the reference's normaliser (batch_per_direction*H*W*channels).
 * mask_out [N,H,W] (may be NULL) receives the final mask for the data terms.  Only 'fb' is differentiable:
 * d_flow (+)= its gradient wrt the raw flow, d_warped = its gradient wrt warped_other (feed to
 * unflow_image_warp_bwd).  warped_other / fwarp may be NULL when no term needs them. */
int unflow_mask_terms(const float* flow, const float* warped_other, const float* fwarp, const float* base_mask,
                      int n_base, float flow_scale, int occlusion_mode, float* mask_out, float* loss_acc, float* d_flow,
                      float* d_warped, int accumulate, float fb_weight, float occ_weight, float sym_weight,
                      int batch_per_direction, int pair_shift, int N, int H, int W, unflow_stream_t stream);

/* compute_losses with the DEFAULT terms (ternary/census + second-order smoothness, config.ini [train]) over the whole
 * loss pyramid (unsupervised.py:85-147) in four launches instead of four per level.  Same arithmetic as
 * unflow_second_order_fwd_bwd + unflow_gray_pair + unflow_ternary_fwd + unflow_ternary_warp_bwd per level.
 * ternary_scale / smooth_scale are the complete factors layer_weight * term_weight / normaliser of the level.
 * loss_acc[0] += the weighted loss; with_grad: d_flow of every level is OVERWRITTEN with d loss / d flow. */
#define UNFLOW_MAX_PYR_LEVELS 7
typedef struct unflow_pyr_level {
  const float* im;      /* [N,H,W,3] images in [0,1] of this level */
  const float* flow;    /* [N,H,W,2] raw network flow of this level */
  float* gray1;         /* [N,H,W] scratch */
  float* gray2w;        /* [N,H,W] scratch */
  const float* mask;    /* [n_mask,H,W] */
  float* dist;          /* [N,H,W] scratch (per-pixel census weights) */
  float* d_flow;        /* [N,H,W,2] out (may be NULL when with_grad == 0) */
  int H, W, n_mask, max_distance;
  float flow_scale, ternary_scale, smooth_scale;
} unflow_pyr_level;
/* sizeof(unflow_pyr_level) as compiled into the library (bindings check their struct layout against it) */
int unflow_sizeof_pyr_level(void);
int unflow_loss_pyramid_default(const unflow_pyr_level* levels, int n_levels, int N, int pair_shift, float* loss_acc,
                                int with_grad, unflow_stream_t stream);

/* photometric_loss (losses.py:198-199): charbonnier(im1 - image_warp(im2, flow), mask, beta=255), fused with the warp;
 * loss_acc[0] += weight*sum/normalizer; d_flow (+)= gradient wrt the raw flow.  mask: [n_mask,H,W]. */
int unflow_photometric_fwd_bwd(const float* im, int ld_im, const float* flow, float flow_scale, const float* mask,
                               int n_mask, float* loss_acc, float* d_flow, int accumulate, float weight, float normalizer,
                               int pair_shift, int N, int H, int W, unflow_stream_t stream);

/* charbonnier_loss (losses.py:298-322), the stand-alone form: loss_acc[0] += weight * sum(min(mask * ((x*beta)^2 +
 * epsilon^2)^alpha, truncate)) / (npix*C) over x [npix, C]; mask [npix, mask_channels], mask_channels 1 or C, or NULL;
 * truncate < 0: none.  length_sq (losses.py:12-13): out[i] = sum_c x[i,c]^2. */
int unflow_charbonnier_loss(const float* x, const float* mask, int mask_channels, float truncate, float alpha, float beta,
                            float epsilon, float* loss_acc, float weight, long npix, int C, unflow_stream_t stream);
int unflow_length_sq(const float* x, float* out, long npix, int C, unflow_stream_t stream);

/* smoothness_loss (losses.py:206-255): first-order forward differences of flow*flow_scale, Charbonnier. */
int unflow_smooth_1st_fwd_bwd(const float* flow, float flow_scale, float* loss_acc, float* d_flow, int accumulate,
                              float weight, float normalizer, int N, int H, int W, unflow_stream_t stream);

/* gradient_loss (losses.py:225-247): Sobel-gradient constancy between im1 and the warped second image [N,H,W,3].
 * fwd writes gdiff [N,H,W,6] = d(weighted loss)/d(diff); bwd turns it into d/d(im2_warped) [N,H,W,3]
 * (then unflow_image_warp_bwd gives the flow gradient). */
int unflow_gradient_loss_fwd(const float* im1, int ld_im1, const float* im2_warped, const float* mask, int n_mask,
                             float* gdiff, float* loss_acc, float weight, float normalizer, int N, int H, int W,
                             unflow_stream_t stream);
int unflow_gradient_loss_bwd(const float* gdiff, float* d_im2_warped, int N, int H, int W, unflow_stream_t stream);

/* ===================================================================== */
/* step plumbing                                                           */
/* ===================================================================== */

/* net input: out[n,y,x,0:3] = im[n,y,x,:]/255 - mean[c]/255, out[...,3] = 0 ([N,H,W,4]);
 * loss image: out01[n,y,x,0:3] = im/255 (unsupervised.py:29-32,69-70). out01 may be NULL. */
int unflow_prepare_images(const float* im_u8range, float* net_in4, float* out01, const float* mean3, long npix,
                          unflow_stream_t stream);

/* Both frames in one launch: im1 -> samples [0, B), im2 -> samples [B, 2B) of the directed batch (npix_each = B*H*W each);
 * net_pl (may be NULL): also the 16-bit operand planes of net_in4 (row length net_pl->ld = 4 or 8). */
int unflow_prepare_image_pair(const float* im1, const float* im2, long npix_each, float* net_in4, float* out01,
                              const float* mean3, const unflow_planes* net_pl, unflow_stream_t stream);

/* Elementwise helpers of the step driver (what the TF graph does with tf.multiply / tf.add_n / tf.zeros between ops):
 * y = s*x; y *= x; y += x (n floats); stream-ordered zero fill and device-to-device copy. */
int unflow_scale(const float* x, float s, float* y, long n, unflow_stream_t stream);
int unflow_mul_inplace(float* y, const float* x, long n, unflow_stream_t stream);
int unflow_add_inplace(float* y, const float* x, long n, unflow_stream_t stream);
int unflow_zero(void* p, size_t bytes, unflow_stream_t stream);
int unflow_copy(void* dst, const void* src, size_t bytes, unflow_stream_t stream);

/* ---- augmentation ---------------------------------------------------------------------- */

/* Spatial-transformer resampling of core/spatial_transformer.py:56-175 as used by random_affine
 * (core/augment.py:50-55): out[b,i,j,:] = bilinear sample of U[b % n_u] at theta[b % n_theta] @ (x_t, y_t, 1),
 * x_t/y_t = linspace(-1,1,out_w/out_h), source coords (x_s+1)*W/2, indices clipped to the image BEFORE the
 * weights are formed (:84-87,113-120).  theta: DEVICE [n_theta,2,3] row-major.  U channels-last with channel
 * stride ld_u; out with ld_out.  No gradient (the reference wraps the result in stop_gradient, augment.py:54). */
int unflow_stn_affine_fwd(const float* U, int n_u, int ld_u, const float* theta, int n_theta, float* out, int ld_out,
                          int B, int H, int W, int C, int out_h, int out_w, unflow_stream_t stream);

/* random_photometric (core/augment.py:78-108) given its draws, fused with the mean subtraction of
 * core/unsupervised.py:67-68: out[n,y,x,c] = pow(clamp((im*(contrast+1)+brightness)*colour[c], 0, 1), 1/gamma)
 * + noise - mean3[c]/255, c < 3; channels 3..ld_out-1 are written as 0.  contrast/brightness/gamma/noise: DEVICE
 * [n_par], colour3: DEVICE [n_par,3]; sample n uses draw n % n_par.  mean3: HOST [3] in [0,255], or NULL. */
int unflow_photometric_augment(const float* im, int ld_in, float* out, int ld_out, const float* contrast,
                               const float* brightness, const float* colour3, const float* gamma, const float* noise,
                               int n_par, const float* mean3, int N, int H, int W, unflow_stream_t stream);

/* Input tensor of a FlowNetS stage (flownet.py:46-59), channels-last with stride ld_out (pad channels untouched):
 * [first, second] (6 ch) when prev_flow2 == NULL, else [first, second, flow, warp(second, flow), |warp - first|]
 * (14 ch) with flow = resize_bilinear(prev_flow2 [N,h,w,2]) * flow_scale (= 4 * FLOW_SCALE).  Forward only: the
 * reference stops the gradient here unless train_all (flownet.py:51-54). */
int unflow_stack_input(const float* net_in4, const float* prev_flow2, float* out, int ld_out, int pair_shift, int N,
                       int H, int W, int h, int w, float flow_scale, unflow_stream_t stream);

/* Gradient of unflow_stack_input wrt prev_flow2 for train_all (flownet.py:51-54 without the stop_gradient): d_out is
 * the gradient wrt the 14-channel stage input (channel stride ld_out); d_prev_flow2 [N,h,w,2] is ACCUMULATED with
 * float atomics (zero it first).  The images receive no gradient (they are data). */
int unflow_stack_input_bwd(const float* d_out, int ld_out, const float* net_in4, const float* prev_flow2,
                           float* d_prev_flow2, int pair_shift, int N, int H, int W, int h, int w, float flow_scale,
                           unflow_stream_t stream);

/* ===================================================================== */
/* 16-bit operand planes (csrc/conv_planes.hip)                            */
/* The conv / conv_transpose layers above can take their operands from,     */
/* and write their results to, "operand planes": the tensor once more as    */
/* n_planes arrays of 16-bit values with the tensor's [pixel][channel]      */
/* layout (channel stride ld, a multiple of 4; plane p at base +            */
/* p*plane_stride elements):                                                */
/*   n_planes == 3: bf16 hi/mid/lo with x = hi + mid + lo exactly — the     */
/*     products are summed from six bf16 MFMA terms with fp32 accumulation, */
/*     fp32-class accuracy (the default training mode);                     */
/*   n_planes == 1: fp16 (fp16 activations/weights in, fp32 accumulate).    */
/* `base` points at channel 0 of the slice the fp32 pointer addresses.  A   */
/* consumed slice of C channels is walked in groups of 8: channels          */
/* C .. round_up_8(C)-1 of the plane rows must exist (ld >= that) and be 0. */
/* Every *_pl entry point falls back to the fp32-operand kernel of the      */
/* same op when the planes are NULL / unusable, and still writes the        */
/* output planes (y_pl / dx_pl; may be NULL).                               */
/* ===================================================================== */
typedef struct unflow_planes {
  void* base;
  long plane_stride;
  int ld;
  int n_planes;
  float scale; /* the planes hold scale * value (0 = 1).  fp16 planes of GRADIENT tensors carry a power-of-two scale so that
                  small gradients stay in fp16's normal range; producers multiply, consumers divide their sums.  Must be 1
                  (or 0) for bf16 x 3 planes, which have fp32's exponent range. */
} unflow_planes;

/* unflow_correlation_nhwc_fwd with the features' operand planes (n_planes == 3, kernel_size 1, stride_1 1, C % 16 == 0):
 * the products run on the bf16 matrix cores (six terms, fp32 accumulation).  Falls back to the fp32 entry point when the
 * planes are NULL / unusable (then in0 / in1 must be given). */
int unflow_correlation_nhwc_fwd_pl(const float* in0, const float* in1, int ld_in, const unflow_planes* in0_pl,
                                   const unflow_planes* in1_pl, int pair_shift, float* out, int ld_out, int B, int C,
                                   int H, int W, int kernel_size, int max_displacement, int pad, int stride_1,
                                   int stride_2, unflow_stream_t stream);

/* unflow_correlation_nhwc_bwd with the features' operand planes (n_planes == 3, kernel_size 1, stride_1 1, C % 64 == 0): the
 * feature operand of the banded products comes from the planes (LDS-DMA + transposing reads), the band operand is split in
 * registers; six terms on the bf16 matrix cores, fp32 accumulation.  Falls back to the fp32 entry point otherwise. */
int unflow_correlation_nhwc_bwd_pl(const float* dout, int ld_dout, const float* in0, const float* in1, int ld_in,
                                   const unflow_planes* in0_pl, const unflow_planes* in1_pl, int pair_shift, float* grad0,
                                   float* grad1, int ld_grad, int accumulate_g1_into_g0, int B, int C, int H, int W,
                                   int kernel_size, int max_displacement, int pad, int stride_1, int stride_2,
                                   unflow_stream_t stream);

/* fp32 [npix][ldx] (C channels) -> planes; plane channels C .. C_fill-1 are zero-filled (C <= C_fill <= round_up_8(C),
 * C_fill a multiple of 4: a slice that ends the buffer row before the next multiple of 8 passes the row's end). */
int unflow_planes_from_f32(const float* x, int ldx, long npix, int C, int C_fill, const unflow_planes* out,
                           unflow_stream_t stream);

/* Planes of weight tensors W[taps][R][Cc] (conv: HWIO, R = Cin, Cc = Cout; conv_transpose: R = Cout, Cc = Cin):
 *   direct[i]     [p][tap][R][round_up_8(Cc)]  — operand of conv2d_bwd_data_pl and conv2d_transpose_fwd_pl
 *   transposed[i] [p][tap][Cc][round_up_8(R)]  — operand of conv2d_fwd_pl and conv2d_transpose_bwd_data_pl
 * (either may be NULL), plane stride = unflow_weight_planes_elems(...).  One launch for the whole table. */
size_t unflow_weight_planes_elems(int taps, int R, int Cc, int transposed);
int unflow_weight_planes_batched(int n, const float* const* w, const int* taps, const int* R, const int* Cc,
                                 void* const* direct, void* const* transposed, int n_planes, unflow_stream_t stream);

/* The optimizer update of train.py:151-152 (unflow_adam_step's arithmetic, bit-identical parameters) and the re-split of the
 * updated weights into their planes in ONE pass over a table of tensors W[taps][R][Cc] that live inside the flat buffers P / G /
 * M / V (identical layouts; w[i] points into P): 64 x 64 tiles, the new values go from the tile to both plane copies without a
 * second read.  direct[i] / transposed[i] may be NULL (a tensor without planes — the Cout = 2 layers, or the bias block as one
 * pseudo-tensor with taps = R = 1); regularized[i] != 0 adds l2_scale * p to the gradient (and, with loss_acc, l2_scale * 0.5 * p^2
 * of the PRE-update values to loss_acc[0], like unflow_adam_step_regloss).  n_planes 0: no tensor of the table has planes. */
int unflow_adam_planes_batched(int n, float* const* w, const int* taps, const int* R, const int* Cc, void* const* direct,
                               void* const* transposed, const int* regularized, int n_planes, float* P, const float* G, float* M,
                               float* V, float grad_scale, float l2_scale, float lr_t, float beta1, float beta2, float eps,
                               float* loss_acc, unflow_stream_t stream);

size_t unflow_conv_pl_workspace_bytes(int B, int H, int W, int Cin, int Cout, int k, int stride, int n_planes);

/* Every Cout = 2 filter gradient of a refinement decoder (flownet.py:89-131: flowN = conv k3 -> 2, flowN_upM =
 * conv2d_transpose 2 -> 2) in one batch of two launches.  kind[i] 0: flowN head — x [B,H,W,Cin] (ldx), dz [B,H,W,2] (lddz),
 * dw [3,3,Cin,2]; kind[i] 1: flowN_upM — x [B,H,W,2], dz [B,2H,2W,2], dw [4,4,2,2] (Cin[i] = 2).  n <= 16.
 * UNFLOW_ERR_UNSUPPORTED when a layer has no strip form (W % 4 != 0): use the per-layer entry points then. */
size_t unflow_flow_wgrad_batched_workspace_bytes(int n, const int* kind, const int* B, const int* H, const int* W, const int* Cin);
int unflow_flow_wgrad_batched(int n, const int* kind, const float* const* x, const int* ldx, const float* const* dz,
                              const int* lddz, float* const* dw, const int* B, const int* H, const int* W, const int* Cin,
                              void* workspace, size_t workspace_bytes, unflow_stream_t stream);

/* Test hook (host only, no GPU): the (M tile, N tile, parity class, K split) each workgroup of a gather / halo launch decodes from
 * its linear id — XCD-contiguous remap + work order 0 / 1 / 2 of csrc/conv_planes.hip — 4 ints per workgroup, M tile = -1 for
 * the padding workgroups of order 2.  Returns the grid size (out == NULL: query only). */
int unflow_debug_work_order(int mt, int nt, int ncls, int nsplit, int order, int xcd, int* out, int out_blocks);

/* ---- gradient exchange (csrc/comm_rccl.hip): replaces average_gradients, src/e2eflow/core/train.py:388-422 ----------------
 * One process per GPU; the flat fp32 gradient buffer is summed over the ranks by RCCL (ncclAllReduce over xGMI) on the
 * caller's stream, the 1 / world factor rides in unflow_adam_step's grad_scale.  RCCL is resolved at run time:
 * unflow_comm_available() returns its version code (> 0) or 0, and without it the other calls return UNFLOW_ERR_UNSUPPORTED.
 * Bootstrap: rank 0 calls unflow_comm_unique_id (128 bytes) and passes the id to every rank out of band; every rank then
 * calls unflow_comm_init(id, nranks, rank, &comm) with its GPU current (collective: returns when all ranks have joined). */
int unflow_comm_available(void);
int unflow_comm_unique_id(void* id128);
int unflow_comm_init(const void* id128, int nranks, int rank, void** comm);
int unflow_comm_info(void* comm, int* nranks, int* rank);
/* buf[0 .. n) <- sum over the ranks, in place, enqueued on `stream`; same n and call order on every rank. */
int unflow_allreduce_sum_f32(float* buf, long n, void* comm, unflow_stream_t stream);
int unflow_comm_destroy(void* comm);

/* Persistent stream-K halo kernel (csrc/conv_streamk.hip).  Test hook (host only, no GPU) for a launch of `ncls` tap classes x
 * nt N tiles x mtp M tile pairs, every item nchunk chunks x ntaps[class] K tiles, the item list ordered M group (ngroups of
 * them) > class > N tile > pair and laid end to end in K tiles: range_pos[w] (G + 1 ints) = the first K-tile position of
 * workgroup w's range; decoded[4 i .. 4 i + 3] = (class, N tile, M pair, K tile inside the item) of position pos[i].  Returns 0,
 * or UNFLOW_ERR_SHAPE for arguments outside the kernel's limits. */
int unflow_debug_streamk_plan(int G, int mtp, int nt, int nchunk, int ncls, const int* ntaps, int ngroups, int* range_pos, int npos,
                              const int* pos, int* decoded);
/* ... and the number of bounded spins of those kernels that gave up since the last call (reads and clears a device counter:
 * must be 0; a result computed past a timeout is wrong). */
int unflow_debug_streamk_timeouts(void);

/* w_pl: the `transposed` planes of w (ld = round_up_8(Cin)). */
int unflow_conv2d_fwd_pl(const float* x, int ldx, const unflow_planes* x_pl, const float* w, const unflow_planes* w_pl,
                         const float* bias, float* y, int ldy, const unflow_planes* y_pl, int B, int H, int W, int Cin,
                         int Cout, int k, int stride, int leaky, void* workspace, size_t workspace_bytes,
                         unflow_stream_t stream);
/* w_pl: the `direct` planes of w (ld = round_up_8(Cout)); dx_pl receives channels [pl_lo, pl_hi) of dx (the range whose
 * values are final after this call, i.e. [act_lo, act_hi)).  The leaky-ReLU derivative of [act_lo, act_hi) is taken from
 * act_src (the fp32 activation) or, when act_src is NULL, from the first plane of act_pl (the sign of the activation is the
 * sign of its bf16 / fp16 leading plane).  Planes-only tensors: y (forward) / dx (data gradient, accumulate == 0) may be NULL
 * when the corresponding planes are given — a tensor that only convolutions read need not exist in fp32. */
int unflow_conv2d_bwd_data_pl(const float* dz, int lddz, const unflow_planes* dz_pl, const float* w,
                              const unflow_planes* w_pl, float* dx, int lddx, const unflow_planes* dx_pl, int pl_lo,
                              int pl_hi, int B, int H, int W, int Cin, int Cout, int k, int stride, int accumulate,
                              const float* act_src, int ld_act, const unflow_planes* act_pl, int act_lo, int act_hi,
                              void* workspace, size_t workspace_bytes, unflow_stream_t stream);
int unflow_conv2d_bwd_filter_pl(const float* x, int ldx, const unflow_planes* x_pl, const float* dz, int lddz,
                                const unflow_planes* dz_pl, float* dw, int B, int H, int W, int Cin, int Cout, int k,
                                int stride, void* workspace, size_t workspace_bytes, unflow_stream_t stream);
/* conv_transpose k4 s2: w [4,4,Cout,Cin]; fwd takes the `direct` planes (ld = round_up_8(Cin)), bwd_data the
 * `transposed` ones (ld = round_up_8(Cout)). */
int unflow_conv2d_transpose_fwd_pl(const float* x, int ldx, const unflow_planes* x_pl, const float* w,
                                   const unflow_planes* w_pl, const float* bias, float* y, int ldy,
                                   const unflow_planes* y_pl, int B, int H, int W, int Cin, int Cout, int leaky,
                                   void* workspace, size_t workspace_bytes, unflow_stream_t stream);
int unflow_conv2d_transpose_bwd_data_pl(const float* dz, int lddz, const unflow_planes* dz_pl, const float* w,
                                        const unflow_planes* w_pl, float* dx, int lddx, const unflow_planes* dx_pl,
                                        int pl_lo, int pl_hi, int B, int H, int W, int Cin, int Cout, int accumulate,
                                        const float* act_src, int ld_act, const unflow_planes* act_pl, int act_lo,
                                        int act_hi, void* workspace, size_t workspace_bytes, unflow_stream_t stream);
int unflow_conv2d_transpose_bwd_filter_pl(const float* x, int ldx, const unflow_planes* x_pl, const float* dz, int lddz,
                                          const unflow_planes* dz_pl, float* dw, int B, int H, int W, int Cin, int Cout,
                                          void* workspace, size_t workspace_bytes, unflow_stream_t stream);

/* tf.image.resize_bilinear (TF1 legacy, align_corners=False) * scale (unsupervised.py:103-104). */
int unflow_resize_bilinear_tf1(const float* in, float* out, int B, int H, int W, int C, int out_h, int out_w,
                               float scale, unflow_stream_t stream);

/* Fused L2-regularised TF-form Adam over a flat parameter vector (train.py:151-152; flownet.py:176):
 * g = grad*grad_scale + (i < n_regularized ? l2_scale * p : 0);
 * m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= lr_t * m / (sqrt(v) + eps). */
int unflow_adam_step(float* p, const float* grad, float* m, float* v, long n, long n_regularized, float grad_scale,
                     float l2_scale, float lr_t, float beta1, float beta2, float eps, unflow_stream_t stream);

/* Same update, and additionally loss_acc[0] += l2_scale * 0.5 * sum(p[0:n_regularized]^2) of the PRE-update
 * parameters: the regularisation term of this step's loss (unsupervised.py:149) rides on the pass Adam makes over
 * the parameters anyway, instead of a separate unflow_l2_loss launch. */
int unflow_adam_step_regloss(float* p, const float* grad, float* m, float* v, long n, long n_regularized,
                             float grad_scale, float l2_scale, float lr_t, float beta1, float beta2, float eps,
                             float* loss_acc, unflow_stream_t stream);

/* loss_acc[0] += scale * 0.5 * sum(p[0:n]^2)  (tf.nn.l2_loss via slim.l2_regularizer). */
int unflow_l2_loss(const float* p, long n, float scale, float* loss_acc, unflow_stream_t stream);

/* sum/mean helpers for metrics: out[0] = sum(|f1-f2|_2 * mask), out[1] = sum(mask)  (flow_util.py:98-103). */
int unflow_flow_error_sums(const float* f1, const float* f2, const float* mask /* may be NULL = ones */,
                           float* out2, long npix, unflow_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* UNFLOW_HIP_H_ */
