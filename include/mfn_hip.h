/*
 * mfn_hip.h -- C ABI of libmfn_hip.so: the MI355X (gfx950) implementation of MaskFlownet's
 * per-pyramid-level matching hot path.  This is the drop-in boundary: every entry point
 * replaces one MXNet operator call made by /root/reference (file:line cited per function)
 * and is what a ctypes / MXNet-CustomOp binding on the reference side binds to
 * (INTEGRATION.md shows the stubs).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no C++/torch/MXNet types cross the boundary.
 *   - every tensor is a contiguous fp32 NCHW buffer in DEVICE memory, owned by the caller.
 *     Outputs are fully overwritten unless a `req` argument says otherwise.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls only
 *     enqueue work; they never synchronise and never allocate device memory.
 *   - return value: 0 = ok; < 0 = bad argument (MFN_E_*); > 0 = a hipError_t from the
 *     launch.  mfn_last_error() returns a thread-local description of the last failure.
 *     Nothing throws across the boundary.
 *   - the operator entry points are re-entrant and keep no per-call state: any number of host threads may call
 *     them concurrently on their own streams.  What arithmetic a call uses is the process's setting
 *     (mfn_set_arithmetic: one atomic value per operator, read by whichever thread makes the call).  The process-global MEASUREMENT state is the exception and is not
 *     part of the drop-in surface: mfn_set_tuning (tilings / code paths only) / mfn_profile_* / mfn_debug_set_timeline
 *     write plain globals that every launch reads -- call them only while no other thread is inside the library.
 *   - flow tensors use the network's channel order: channel 0 = dy (vertical),
 *     channel 1 = dx (horizontal)   (/root/reference/network/pipeline.py:105,
 *     /root/reference/network/layer.py:17).
 */
#ifndef MFN_HIP_H
#define MFN_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MFN_ABI_VERSION 1

/* status codes (< 0); > 0 are hipError_t values */
#define MFN_OK 0
#define MFN_E_NULL (-1)        /* a required pointer is NULL */
#define MFN_E_SHAPE (-2)       /* a dimension is non-positive or inconsistent */
#define MFN_E_PARAM (-3)       /* an operator parameter is outside what MXNet accepts */
#define MFN_E_UNSUPPORTED (-4) /* valid for MXNet, not implemented by this library */
#define MFN_E_WORKSPACE (-5)   /* workspace missing or too small */
#define MFN_E_ALIGN (-6)       /* a pointer is not 4-byte aligned */

/* OpReqType of MXNet (include/mxnet/op_attr_types.h) for gradient outputs */
#define MFN_REQ_NULL 0
#define MFN_REQ_WRITE 1
#define MFN_REQ_ADD 3

int mfn_abi_version(void);
const char *mfn_version_string(void);
/* Thread-local text for the last non-zero status returned on this thread ("" if none). */
const char *mfn_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Correlation  -- replaces F.Correlation(data1, data2, kernel_size, max_displacement, stride1,
 * stride2, pad_size, is_multiply) at /root/reference/network/MaskFlownet.py:193-195 (md=4) and
 * :440-441 (md=2).  Semantics: MXNet src/operator/correlation.cc:
 *   out[n, (dy/s2 + r)*D + (dx/s2 + r), i, j] =
 *       1/(K*K*C) * sum_{h,w<K} sum_c pad(data1)[n,c,y1+h,x1+w] * pad(data2)[n,c,y1+dy+h,x1+dx+w]
 *   r = max_displacement/stride2, D = 2r+1, y1 = i*stride1 + max_displacement (padded coords).
 * data1,data2: (N,C,H,W); out: (N, D*D, top_h, top_w) from mfn_correlation_out_shape.
 * The reference configuration (kernel_size=1, strides 1, pad_size == max_displacement) runs
 * the LDS-tiled kernel; any other valid configuration runs a generic kernel.
 * ------------------------------------------------------------------------------------------- */
int mfn_correlation_out_shape(int H, int W, int max_displacement, int kernel_size, int stride1,
                              int stride2, int pad_size, int *top_channels, int *top_h, int *top_w);
int mfn_correlation_fwd(const float *data1, const float *data2, float *out, int N, int C, int H,
                        int W, int max_displacement, int kernel_size, int stride1, int stride2,
                        int pad_size, int is_multiply, void *stream);
/* Same operator with a scratch buffer: levels with few pixels and many channels (L6..L3 of the
 * pyramid) are split into channel slices across workgroups, partial sums go to `workspace` and a
 * second kernel reduces them in a fixed order (deterministic).  workspace may be NULL / too small:
 * the call then runs the single-pass kernel.  mfn_correlation_workspace_bytes returns the size the
 * heuristic wants (0 when slicing would not be used). */
size_t mfn_correlation_workspace_bytes(int N, int C, int H, int W, int max_displacement, int kernel_size,
                                       int stride1, int stride2, int pad_size, int is_multiply);
int mfn_correlation_fwd_ws(const float *data1, const float *data2, float *out, int N, int C, int H,
                           int W, int max_displacement, int kernel_size, int stride1, int stride2,
                           int pad_size, int is_multiply, void *workspace, size_t workspace_bytes,
                           void *stream);
/* Fused epilogue: the reference applies LeakyReLU(0.1) to every cost volume right
 * away (/root/reference/network/MaskFlownet.py:217,231,249,267,285: self.leakyRELU(self.corr(...))).
 * activation: MFN_ACT_NONE or MFN_ACT_LEAKY_0_1 (max(v, 0.1 v), applied after the 1/C normalisation,
 * bit-identical to the separate elementwise op).  Otherwise identical to mfn_correlation_fwd_ws. */
#define MFN_ACT_NONE 0
#define MFN_ACT_LEAKY_0_1 1
int mfn_correlation_fwd_act(const float *data1, const float *data2, float *out, int N, int C, int H,
                            int W, int max_displacement, int kernel_size, int stride1, int stride2,
                            int pad_size, int is_multiply, int activation, void *workspace,
                            size_t workspace_bytes, void *stream);
/* Rest of f-1: the cost volume written straight into its channel slice of the decoder's concat buffer
 * (/root/reference/network/MaskFlownet.py:235,253,271,289: x = concat(corr, c1, feat, flow)) -- no concat copy.
 * out points at channel c0 of image 0 of a contiguous (N, Ctot, h, w) buffer; out_batch_stride = Ctot*h*w elements
 * (0 = dense, i.e. D*D*h*w).  Must be >= one output image; the 16-byte-store kernels need out 16-byte aligned and
 * the stride a multiple of 4 (otherwise the generic kernel runs).  Otherwise identical to mfn_correlation_fwd_act. */
int mfn_correlation_fwd_into(const float *data1, const float *data2, float *out, long long out_batch_stride,
                             int N, int C, int H, int W, int max_displacement, int kernel_size, int stride1,
                             int stride2, int pad_size, int is_multiply, int activation, void *workspace,
                             size_t workspace_bytes, void *stream);
/* Backward of the same call site (training, /root/reference/network/pipeline.py:112-113).
 * g1/g2: (N,C,H,W); req1/req2 in {MFN_REQ_NULL, MFN_REQ_WRITE, MFN_REQ_ADD}. */
int mfn_correlation_bwd(const float *gout, const float *data1, const float *data2, float *g1,
                        float *g2, int N, int C, int H, int W, int max_displacement,
                        int kernel_size, int stride1, int stride2, int pad_size, int is_multiply,
                        int req1, int req2, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Warp -- replaces the operator pair of /root/reference/network/layer.py:14-18
 * (Reconstruction2D) and :26-30 (Reconstruction2DSmooth):
 *     grid = GridGenerator(flow.flip(axis=1), 'warp') [.clip(-1, 1)];  out = BilinearSampler(x, grid)
 * fused into one kernel.  x,out: (N,C,H,W); flow_yx: (N,2,H,W).  clip_grid=1 is the Smooth
 * variant (border replicate).  Semantics: MXNet grid_generator-inl.h (kWarp) +
 * bilinear_sampler.cc, including the fp32 normalise/denormalise
 * round trip and per-tap zero padding.
 * ------------------------------------------------------------------------------------------- */
int mfn_warp_fwd(const float *x, const float *flow_yx, float *out, int N, int C, int H, int W,
                 int clip_grid, void *stream);
/* gx: (N,C,H,W) data gradient, gflow_yx: (N,2,H,W) flow gradient (NULL / MFN_REQ_NULL to skip,
 * which is what block_grad=True in layer.py:15-16 does). */
int mfn_warp_bwd(const float *gout, const float *x, const float *flow_yx, float *gx,
                 float *gflow_yx, int N, int C, int H, int W, int clip_grid, int req_x,
                 int req_flow, void *stream);
/* The two MXNet operators on their own (also used by /root/reference/augmentation.py:60-64,
 * 306-321,333).  flow_xy/grid: (N,2,H,W) channel 0 = x.  theta: (N,6) row-major 2x3. */
int mfn_grid_generator_warp(const float *flow_xy, float *grid, int N, int H, int W, void *stream);
int mfn_grid_generator_affine(const float *theta, float *grid, int N, int H, int W, void *stream);
int mfn_bilinear_sampler_fwd(const float *data, const float *grid, float *out, int N, int C,
                             int iH, int iW, int oH, int oW, void *stream);
/* Backward of the operator pair (what MXNet's autograd reaches where the reference differentiates a warp it built
 * from the two operators: c40 of /root/reference/network/MaskFlownet.py:311, block_grad=False).  gdata: (N,C,iH,iW),
 * ggrid: (N,2,oH,oW) channel 0 = x; gflow_xy = ggrid / ((size-1)/2).  req in {MFN_REQ_NULL, _WRITE, _ADD}. */
int mfn_bilinear_sampler_bwd(const float *gout, const float *data, const float *grid, float *gdata,
                             float *ggrid, int N, int C, int iH, int iW, int oH, int oW,
                             int req_data, int req_grid, void *stream);
int mfn_grid_generator_warp_bwd(const float *ggrid, float *gflow_xy, int N, int H, int W, int req,
                                void *stream);

/* ---------------------------------------------------------------------------------------------
 * DeformableConvolution -- replaces F.contrib.DeformableConvolution(x, offset, weight[, bias],
 * kernel, stride, dilate, pad, num_filter, num_group, num_deformable_group, no_bias) at
 * /root/reference/network/layer.py:117-124 (kwargs :91-95).  Semantics: MXNet
 * contrib/nn/deformable_im2col.* + contrib/deformable_convolution-inl.h:
 *   out[n,o,y,x] = bias[o] + sum_{c,i,j} w[o,c,i,j] * S(x[n,c], y*sh-ph+i*dh + off[n,2k,y,x],
 *                                                        x*sw-pw+j*dw + off[n,2k+1,y,x]),  k=i*kw+j
 *   S = 0 when a coordinate is < 0 or >= dim; bilinear with clamp-to-last inside [dim-1, dim).
 * x: (N,Cin,H,W); offset: (N, 2*kh*kw*deform_groups, Ho, Wo); w: (Cout, Cin/groups, kh, kw);
 * bias: (Cout) or NULL (no_bias); out: (N,Cout,Ho,Wo).
 * The im2col buffer is never materialised: the gather feeds fp32 MFMA tiles directly.
 * `workspace` holds the re-laid-out weights and, for coarse levels whose reduction dimension is
 * split across workgroups, the partial sums (mfn_deform_conv_workspace_bytes); it is scratch,
 * valid only for the duration of the call's stream work.
 * ------------------------------------------------------------------------------------------- */
int mfn_deform_conv_out_shape(int H, int W, int kh, int kw, int sh, int sw, int ph, int pw, int dh,
                              int dw, int *Ho, int *Wo);
size_t mfn_deform_conv_workspace_bytes(int N, int Cin, int H, int W, int Cout, int kh, int kw, int sh,
                                       int sw, int ph, int pw, int dh, int dw, int groups,
                                       int deform_groups);
int mfn_deform_conv_fwd(const float *x, const float *offset, const float *w,
                        const float *bias_or_null, float *out, int N, int Cin, int H, int W,
                        int Cout, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                        int groups, int deform_groups, void *workspace, size_t workspace_bytes,
                        void *stream);
/* Fused form of the reference's call pattern /root/reference/network/MaskFlownet.py:230,248,266,
 * 284: offset = repeat9(flow * flow_scale / flow_stride) is never built; every tap of pixel
 * (y,x) uses (flow[n,0,y,x], flow[n,1,y,x]) * flow_scale / flow_stride.  Same result as
 * mfn_offsets_from_flow + mfn_deform_conv_fwd.  Requires stride 1 (Ho=H, Wo=W). */
int mfn_deform_conv_shared_fwd(const float *x, const float *flow_yx, float flow_scale,
                               float flow_stride, const float *w, const float *bias_or_null,
                               float *out, int N, int Cin, int H, int W, int Cout, int kh, int kw,
                               int ph, int pw, int dh, int dw, int groups, void *workspace,
                               size_t workspace_bytes, void *stream);
/* Inference: a Gluon block (/root/reference/network/layer.py:97-124) owns constant weights, so
 * their re-layout can be done once instead of on every call.  mfn_deform_conv_pack_weights writes
 * the layout the kernels stream (an opaque function of the 15 shape ints and of mfn_set_tuning;
 * mfn_deform_conv_packed_weight_bytes gives its size) and the *_packed entry points take it in
 * place of `w`, with the layout tag pack_weights returned; their workspace then only holds split-K
 * partial sums (the value returned by mfn_deform_conv_workspace_bytes is always enough).  Results
 * are bit-identical to the unpacked calls.  A tag that does not match the layout the current
 * shape/tuning needs is refused with MFN_E_WORKSPACE, never silently used. */
size_t mfn_deform_conv_packed_weight_bytes(int N, int Cin, int H, int W, int Cout, int kh, int kw,
                                           int sh, int sw, int ph, int pw, int dh, int dw,
                                           int groups, int deform_groups);
int mfn_deform_conv_pack_weights(const float *w, int N, int Cin, int H, int W, int Cout, int kh,
                                 int kw, int sh, int sw, int ph, int pw, int dh, int dw, int groups,
                                 int deform_groups, void *packed, size_t packed_bytes,
                                 unsigned long long *layout_tag, void *stream);
int mfn_deform_conv_fwd_packed(const float *x, const float *offset, const void *packed,
                               size_t packed_bytes, unsigned long long layout_tag,
                               const float *bias_or_null, float *out, int N,
                               int Cin, int H, int W, int Cout, int kh, int kw, int sh, int sw,
                               int ph, int pw, int dh, int dw, int groups, int deform_groups,
                               void *workspace, size_t workspace_bytes, void *stream);
int mfn_deform_conv_shared_fwd_packed(const float *x, const float *flow_yx, float flow_scale,
                                      float flow_stride, const void *packed, size_t packed_bytes,
                                      unsigned long long layout_tag, const float *bias_or_null, float *out, int N, int Cin, int H,
                                      int W, int Cout, int kh, int kw, int ph, int pw, int dh,
                                      int dw, int groups, void *workspace, size_t workspace_bytes,
                                      void *stream);
/* The whole warp step of the matching module in one launch, for
 * /root/reference/network/MaskFlownet.py:230-233 (and :248-251, :266-269, :284-287):
 *   warp = deform(c2, repeat9(flow*scale/stride)); warp = warp * sigmoid(mask) + tradeoff; warp = LeakyReLU(0.1)(warp)
 * mask: (N,1,H,W) or NULL; tradeoff: (N,Cout,H,W) (the conv5f(feat) term) or NULL; activation: MFN_ACT_*.
 * Give either `w` (re-packed on every call into workspace) or `packed` + layout tag (see above).
 * Elementwise results equal the separate ops' (sigmoid = 1/(1+expf(-m)) in fp32). */
int mfn_deform_conv_matching_fwd(const float *x, const float *flow_yx, float flow_scale, float flow_stride,
                                 const float *w_or_null, const void *packed_or_null, size_t packed_bytes,
                                 unsigned long long layout_tag, const float *bias_or_null,
                                 const float *mask_or_null, const float *tradeoff_or_null, int activation,
                                 float *out, int N, int Cin, int H, int W, int Cout, int kh, int kw, int ph,
                                 int pw, int dh, int dw, int groups, void *workspace, size_t workspace_bytes,
                                 void *stream);
/* Backward (training).  gx,goffset,gw,gbias as the forward's x,offset,w,bias; req_* per output
 * (MFN_REQ_NULL skips it and the pointer may be NULL).  The column gradient is formed on the fly.
 * Scratch (mfn_deform_conv_bwd_workspace_bytes; 0 for shapes other than 3x3 / stride 1 / dilation 1
 * / one group): the per-block partial sums of the weight / bias gradient, which a second kernel
 * adds in a fixed order.  workspace may be NULL or smaller (parameter gradients through atomics:
 * same results up to summation order).  Tiles whose nine taps share one offset
 * (MaskFlownet.py:230) take all taps in one pass, the others tap by tap, in the same launch.  gx and goffset are accumulated with fp32 atomics (as MXNet's GPU kernels do): their
 * bit-level results can differ from run to run by summation order; with the workspace gw and gbias
 * are deterministic. */
size_t mfn_deform_conv_bwd_workspace_bytes(int N, int Cin, int H, int W, int Cout, int kh, int kw,
                                           int sh, int sw, int ph, int pw, int dh, int dw,
                                           int groups, int deform_groups);
int mfn_deform_conv_bwd(const float *gout, const float *x, const float *offset, const float *w,
                        float *gx, float *goffset, float *gw, float *gbias_or_null, int N, int Cin,
                        int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph, int pw,
                        int dh, int dw, int groups, int deform_groups, int req_x, int req_offset,
                        int req_w, int req_bias, void *workspace, size_t workspace_bytes,
                        void *stream);
/* Backward of the fused call mfn_deform_conv_shared_fwd (training with the offsets never built by the caller):
 * gflow (N,2,H,W) = d loss / d flow_yx = flow_scale / flow_stride * sum over the taps of the offset gradient; gx, gw, gbias
 * and the req_* as mfn_deform_conv_bwd.  The workspace (mfn_deform_conv_shared_bwd_workspace_bytes, required, 16-byte
 * aligned) has room for the offsets and their gradient: shapes outside the lane = pixel kernels (3x3, pad 1, one group,
 * W % 4 == 0, Cin % 4 == 0, Cout <= 96) are composed as mfn_offsets_from_flow -> mfn_deform_conv_bwd ->
 * mfn_offsets_from_flow_bwd; inside, the kernels read the flow field and write gflow themselves.  mfn_offsets_from_flow_bwd
 * is the gradient of mfn_offsets_from_flow on its own (req: MFN_REQ_WRITE or MFN_REQ_ADD). */
size_t mfn_deform_conv_shared_bwd_workspace_bytes(int N, int Cin, int H, int W, int Cout, int kh, int kw, int ph,
                                                  int pw, int dh, int dw, int groups);
int mfn_deform_conv_shared_bwd(const float *gout, const float *x, const float *flow_yx, float flow_scale,
                               float flow_stride, const float *w, float *gx, float *gflow, float *gw,
                               float *gbias, int N, int Cin, int H, int W, int Cout, int kh, int kw, int ph,
                               int pw, int dh, int dw, int groups, int req_x, int req_flow, int req_w,
                               int req_bias, void *workspace, size_t workspace_bytes, void *stream);
int mfn_offsets_from_flow_bwd(const float *goffset, float *gflow_yx, int N, int H, int W, int taps,
                              float flow_scale, float flow_stride, int req, void *stream);

/* Upsample(factor) of flow / mask between pyramid levels -- replaces the Gluon block
 * /root/reference/network/MaskFlownet.py:35-62 (edge pad + Deconvolution with the triangle kernel
 * 1-|f-1-a|/f, kernel 2f-1, stride f, pad f-1, last row/column dropped; call sites :228-229 ... :311).
 * x: (N,C,H,W) -> out: (N,C,H*factor,W*factor); factor 1 copies.  Bit-identical to the fp32 oracle. */
int mfn_upsample_fwd(const float *x, float *out, int N, int C, int H, int W, int factor, void *stream);
/* Its backward (the block is linear; on the gradient path of every level: flow5 = Upsample(2)(flow6), MaskFlownet.py:228):
 * gout: (N,C,H*factor,W*factor) -> gx: (N,C,H,W); req: MFN_REQ_*.  What MXNet's autograd computes through the
 * pad / Deconvolution / slice chain of :51-62. */
int mfn_upsample_bwd(const float *gout, float *gx, int N, int C, int H, int W, int factor, int req, void *stream);
/* LeakyReLU(slope) backward from the forward OUTPUT y: gin = gout * (y > 0 ? 1 : slope) -- the gradient side of the fused
 * activations (mfn_correlation_fwd_act, mfn_conv2d_fwd activation = MFN_ACT_LEAKY_0_1; nn.LeakyReLU(0.1) of
 * /root/reference/network/MaskFlownet.py:76,217).  gin may alias gout. */
int mfn_leaky_relu_bwd(const float *gout, const float *y, float *gin, size_t n, float slope, void *stream);
/* Offset builder of /root/reference/network/MaskFlownet.py:230:
 *   offset[n, 2k+t, y, x] = flow_yx[n, t, y, x] * scale / stride   for k < taps. */
int mfn_offsets_from_flow(const float *flow_yx, float *offset, int N, int H, int W, int taps,
                          float scale, float stride, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Convolution / Deconvolution -- replace the Gluon blocks of
 * /root/reference/network/MaskFlownet.py:79-163: nn.Conv2D(channels, kernel_size=3, strides, padding, dilation)
 * [+ LeakyReLU(0.1)] of conv() / predict_flow() / predict_mask() (:165-191) and nn.Conv2DTranspose(channels, 4, 2, 1)
 * [+ LeakyReLU(0.1)] of deconv() (:175-183), i.e. MXNet's Convolution / Deconvolution operators:
 *   Convolution:   out[n,o,y,x] = b[o] + sum_{c,i,j} w[o,c,i,j] * x[n,c, y*sh-ph+i*dh, x*sw-pw+j*dw]      (zero outside)
 *   Deconvolution: out[n,o,y,x] = b[o] + sum_{c,i,j} w[c,o,i,j] * x[n,c,(y+ph-i*dh)/sh,(x+pw-j*dw)/sw]   (exact divisions only)
 * x: (N,Cin,H,W); w: (Cout,Cin/groups,kh,kw), transposed: (Cin,Cout/groups,kh,kw); out: (N,Cout,Ho,Wo) from
 * mfn_conv2d_out_shape.  3x3 convolutions and 4x4 transposed convolutions with groups == 1 run fused gather + matrix-core
 * implicit GEMM kernels (no im2col buffer; arithmetic per mfn_set_arithmetic("convolution", ..): the default is fp32-EQUIVALENT --
 * operands as three bf16 terms, see "Arithmetic" --, MFN_ARITH_FP32 the bit-exact fp32 MFMA chain); every other parameter set
 * runs a generic kernel.
 * activation: MFN_ACT_NONE | MFN_ACT_LEAKY_0_1 (fused, bit-identical to the separate elementwise op).
 * out_batch_stride / in_batch_stride: elements between consecutive output / input images (0 = dense): a layer writes
 * straight into its channel slice of the decoder's concat buffer and the next layer reads the buffer's channel suffix
 * (x = concat(conv(x), x), MaskFlownet.py:219-223) -- no concat copies.
 * Weights: give `w` (re-laid-out into `workspace` on every call, mfn_conv2d_workspace_bytes) or a buffer made once by
 * mfn_conv2d_pack_weights + its layout tag (mfn_conv2d_packed_weight_bytes; a tag that does not match the plan of the
 * current shape / tuning is refused, never silently used).  3x3 / stride 1 / pad 1 layers with >= 32 filters on large images are
 * packed for the deformable convolution's matrix-core kernel (kernels/deform_conv_mma.h, CONV form): a call with such a buffer needs
 * x and out 16-byte aligned and batch strides that are multiples of 4 elements (MFN_E_ALIGN otherwise; a call that gives `w` picks
 * the other kernel by itself).
 * ------------------------------------------------------------------------------------------- */
int mfn_conv2d_out_shape(int H, int W, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int transposed,
                         int adj_h, int adj_w, int *Ho, int *Wo);
size_t mfn_conv2d_workspace_bytes(int N, int Cin, int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph,
                                  int pw, int dh, int dw, int groups, int transposed);
size_t mfn_conv2d_packed_weight_bytes(int N, int Cin, int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph,
                                      int pw, int dh, int dw, int groups, int transposed);
int mfn_conv2d_pack_weights(const float *w, int N, int Cin, int H, int W, int Cout, int kh, int kw, int sh, int sw,
                            int ph, int pw, int dh, int dw, int groups, int transposed, void *packed,
                            size_t packed_bytes, unsigned long long *layout_tag, void *stream);
int mfn_conv2d_fwd(const float *x, long long in_batch_stride, const float *w_or_null, const void *packed_or_null,
                   size_t packed_bytes, unsigned long long layout_tag, const float *bias_or_null, float *out,
                   long long out_batch_stride, int N, int Cin, int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph, int pw, int dh,
                   int dw, int groups, int transposed, int adj_h, int adj_w, int activation, void *workspace,
                   size_t workspace_bytes, void *stream);

/* Backward of mfn_conv2d_fwd (num_group == 1): what autograd runs for the Conv2D / Conv2DTranspose blocks of
 * /root/reference/network/MaskFlownet.py:79-163 in network/pipeline.py:112-113.  gout: (N,Cout,Ho,Wo); y: the forward output
 * (only read when activation = MFN_ACT_LEAKY_0_1, NULL otherwise); gx: (N,Cin,H,W); gw: the weight's own layout; gbias: (Cout).
 * req_*: MFN_REQ_NULL / WRITE / ADD per gradient.  workspace: mfn_conv2d_bwd_workspace_bytes (16-byte aligned).
 * The weight gradient runs on the deformable convolution's weight-gradient kernels with zero offsets; where those sum
 * through fp32 atomics (filters > 96 per call or widths that are no multiple of 4) its last bits vary from run to run. */
size_t mfn_conv2d_bwd_workspace_bytes(int N, int Cin, int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph,
                                      int pw, int dh, int dw, int groups, int transposed, int adj_h, int adj_w,
                                      int activation);
int mfn_conv2d_bwd(const float *gout, const float *x, const float *w, const float *y_or_null, float *gx, float *gw,
                   float *gbias, int N, int Cin, int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph, int pw,
                   int dh, int dw, int groups, int transposed, int adj_h, int adj_w, int activation, int req_x, int req_w,
                   int req_bias, void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Stream plumbing for the host side (no reference equivalent: MXNet's engine does this).
 * A hot-path pass is ~10 short launches; capturing it once into a hipGraph and replaying it
 * removes the per-launch host cost.  Capture is plain hipStreamBeginCapture on `stream`.
 * ------------------------------------------------------------------------------------------- */
int mfn_graph_begin_capture(void *stream);
int mfn_graph_end_capture(void *stream, void **graph_exec_out);
int mfn_graph_launch(void *graph_exec, void *stream);
int mfn_graph_destroy(void *graph_exec);

/* Built-in kernel timer: while enabled, every launch made by this library on this thread is
 * bracketed by HIP events on ITS OWN stream (hipExtLaunchKernelGGL start/stop events) and
 * accumulated per kernel name.  Used by bench.py for the roofline line. */
int mfn_profile_enable(int on);
int mfn_profile_reset(void);
/* Records taken until the next call carry the suffix "@tag" (NULL / "" clears it): tells the launches of one operator
 * call apart inside a fully profiled pass. */
int mfn_profile_tag(const char *tag);
/* Synchronises the recorded events, then reports launches and summed milliseconds of every
 * kernel whose name contains `name_substr`.  Returns the number of matching launches. */
int mfn_profile_query(const char *name_substr, int *launches, double *total_ms);
/* Writes up to `cap` bytes of "name launches total_ms\n" lines into buf; returns bytes needed. */
int mfn_profile_dump(char *buf, int cap);

/* ---------------------------------------------------------------------------------------------
 * Arithmetic.  The three GEMM-shaped operators (Correlation, DeformableConvolution, Convolution / Deconvolution) exist in
 * two arithmetics; which one a call uses is ONE setting per process (atomic; default compiled in) that every thread's calls read --
 * the thread that launches a kernel is often not the one that chose: torch runs backward on its autograd thread, MXNet runs every
 * CustomOp on worker threads.  Set it before the calls it should govern; it is not one of the measurement knobs below:
 *   MFN_ARITH_DEFAULT  the library's choice: the bf16 x 3 matrix-core kernels wherever one exists for the call's shape
 *                      (level shapes of the network), the fp32 kernels elsewhere.
 *   MFN_ARITH_FP32     fp32 FMA chains everywhere (v_fma_f32 / v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32): the accumulation order of an fp32
 *                      inner product, bit-reproducible against itself across tilings of K only within one kernel.
 *   MFN_ARITH_BF16X3   as the default.
 * bf16 x 3: every fp32 operand is written exactly as hi + mid + lo with three bf16 terms (24 significant bits) and SIX of the
 * nine partial products -- those of weight >= 2^-16 -- are accumulated in fp32 by v_mfma_f32_*_bf16.  I/O stays fp32.  The
 * dropped products are <= 2^-24 of a product each (~1 ulp per product, not per sum); the acceptance rule, asserted per kernel in
 * tests/test_gpu_parity.py, is "maximum error against the fp64 oracle within 1.25 x (deformable convolution) / 1.5 x (cost
 * volumes) / 2 x (convolutions) of the fp32 kernel's on the same input, and <= 1e-5 of max|ref|" (observed: 0.7-1.1 x).  It is
 * fp32-EQUIVALENT, not the bit pattern of an FMA chain.  Divergence on non-finite inputs: an +-inf operand splits into
 * inf + NaN (inf - bf16(inf)), so outputs that an FMA chain would make +-inf come back NaN; NaN inputs give NaN either way;
 * values beyond bf16's range do not exist (bf16 has fp32's exponent); fp32 denormal operands lose their low terms (flushed), an
 * absolute error below 2^-126 per product.  Which OUTPUTS a non-finite input pixel reaches: under the default arithmetic the
 * deformable convolution's taps outside the image read zeros, so exactly the outputs whose valid taps touch the pixel (MXNet's
 * set); the fp32 kernel (MFN_ARITH_FP32, dc_lds_kernel) multiplies border-clamped reads by zero weights, so a non-finite pixel
 * within three rows / columns of the border also poisons neighbouring outputs whose taps fall outside the image
 * (tests/test_gpu_parity.py::test_deform_numeric_range_edge_cases).
 * Layouts packed by mfn_deform_conv_pack_weights / mfn_conv2d_pack_weights depend on the arithmetic in force when they were packed;
 * a call under another arithmetic refuses them (layout tag) instead of misreading them.
 * op: "correlation" | "deformable_convolution" | "convolution" | "all".  Unknown op / mode: MFN_E_PARAM. */
#define MFN_ARITH_DEFAULT (-1)
#define MFN_ARITH_FP32 0
#define MFN_ARITH_BF16X3 1
int mfn_set_arithmetic(const char *op, int mode);
int mfn_get_arithmetic(const char *op, int *mode);

/* Kernel-selection knobs for tuning sweeps: tilings and code paths only, never arithmetic (process-global, NOT thread-safe,
 * not part of the drop-in surface: set them before the first operator call or while no other thread is inside the library).
 * Unknown keys return MFN_E_PARAM.  Keys: see maskflownet_amd/csrc/tuning.h. */
/* Measurement only: when non-NULL, instrumented kernels write 4 x uint64 wall-clock stamps (100 MHz)
 * per workgroup into this device buffer (caller sizes it: 32 bytes x workgroups). */
int mfn_debug_set_timeline(void *device_buffer);
int mfn_set_tuning(const char *key, int value);
int mfn_get_tuning(const char *key, int *value);

#ifdef __cplusplus
}
#endif
#endif /* MFN_HIP_H */
