/*
 * mfn_ref.h -- CPU ORACLE for the MaskFlownet matching hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may call it, and only as the
 * checker.  The product path (maskflownet_amd/) never links, imports or falls
 * back to anything in oracle/.
 *
 * PARITY UNPINNED (against MXNet itself): the arithmetic of this path lives in
 * Apache MXNet 1.5.x ("tested with Python 3.6 and MXNet 1.5",
 * /root/reference/README.md:27), a third-party dependency that is neither
 * vendored under /root/reference nor installable here (no wheel, no network,
 * Python 3.10).  The reference ships no tests or golden vectors for the path
 * (SURVEY.md section 4 / 8c).  This file therefore RESTATES the published
 * algorithms of the MXNet CPU operators, loop for loop:
 *   src/operator/correlation.cc           CorrelationForward / CorrelationBackward, AddPad
 *   src/operator/grid_generator-inl.h     GridGeneratorOp (kWarp, kAffine)
 *   src/operator/bilinear_sampler.cc      BilinearSamplerForward / BilinearSamplerBackward
 *   src/operator/contrib/nn/deformable_im2col.h (.cuh)   deformable_im2col / col2im / col2im_coord
 *   src/operator/contrib/deformable_convolution-inl.h    im2col + GEMM + bias
 * anchored on the reference's own call sites:
 *   network/MaskFlownet.py:193-195, 440-441   F.Correlation(pad_size=md, kernel_size=1,
 *                                             max_displacement=md, stride1=1, stride2=1, is_multiply=1)
 *   network/layer.py:14-18, 26-30             flow.flip(axis=1) -> GridGenerator('warp') [-> clip(-1,1)]
 *                                             -> BilinearSampler
 *   network/layer.py:117-124                  contrib.DeformableConvolution(kernel=(3,3), stride=1,
 *                                             dilate=1, pad=1, num_group=1, num_deformable_group=1)
 *   network/MaskFlownet.py:230,248,266,284    offset = repeat9(flow * scale / stride)
 * What pins it instead (tests/test_oracle_*.py): analytic known-answer tests,
 * an independent fp64 numpy restatement (oracle/ref_numpy.py), and the
 * independent implementations that do exist here: torch.nn.functional
 * grid_sample(align_corners=True) for the sampler and conv2d for
 * deformable convolution at zero / integer offsets.
 *
 * Two symbol sets are exported from one body (mfn_ref_body.inc):
 *   mfn_ref_*    REAL = float   -- loop- and precision-faithful restatement
 *   mfn_ref64_*  REAL = double  -- same loops in fp64: the "who is wrong" arbiter
 * All tensors are contiguous NCHW host arrays.  Functions return 0 on success,
 * <0 on a bad argument (same convention as include/mfn_hip.h).
 */
#ifndef MFN_REF_H
#define MFN_REF_H

#ifdef __cplusplus
extern "C" {
#endif

#define MFN_REF_DECLARE(SFX, REAL)                                                                  \
  /* MXNet Correlation shape inference (correlation-inl.h CorrelationProp::InferShape). */          \
  int mfn_ref##SFX##_correlation_out_shape(int H, int W, int max_disp, int kernel, int stride1,     \
                                           int stride2, int pad, int *top_c, int *top_h,            \
                                           int *top_w);                                             \
  int mfn_ref##SFX##_correlation_fwd(const REAL *data1, const REAL *data2, REAL *out, int N, int C, \
                                     int H, int W, int max_disp, int kernel, int stride1,           \
                                     int stride2, int pad, int is_multiply);                        \
  int mfn_ref##SFX##_correlation_bwd(const REAL *gout, const REAL *data1, const REAL *data2,        \
                                     REAL *g1, REAL *g2, int N, int C, int H, int W, int max_disp,  \
                                     int kernel, int stride1, int stride2, int pad,                 \
                                     int is_multiply);                                              \
  /* GridGenerator(transform_type='warp'): flow_xy (N,2,H,W) ch0 = x -> grid (N,2,H,W). */          \
  int mfn_ref##SFX##_grid_generator_warp(const REAL *flow_xy, REAL *grid, int N, int H, int W);     \
  /* backward of the above: gflow_xy = ggrid / ((size - 1) / 2) per channel. */                     \
  int mfn_ref##SFX##_grid_generator_warp_bwd(const REAL *ggrid, REAL *gflow_xy, int N, int H,       \
                                             int W);                                                \
  /* GridGenerator(transform_type='affine'): theta (N,6) -> grid (N,2,H,W). */                      \
  int mfn_ref##SFX##_grid_generator_affine(const REAL *theta, REAL *grid, int N, int H, int W);     \
  int mfn_ref##SFX##_bilinear_sampler_fwd(const REAL *data, const REAL *grid, REAL *out, int N,     \
                                          int C, int iH, int iW, int oH, int oW);                   \
  int mfn_ref##SFX##_bilinear_sampler_bwd(const REAL *gout, const REAL *data, const REAL *grid,     \
                                          REAL *gdata, REAL *ggrid, int N, int C, int iH, int iW,   \
                                          int oH, int oW);                                          \
  /* layer.py Reconstruction2D (clip_grid=0) / Reconstruction2DSmooth (clip_grid=1):                \
     flow_yx (N,2,H,W) with channel 0 = dy, channel 1 = dx (the network's convention). */           \
  int mfn_ref##SFX##_warp_fwd(const REAL *x, const REAL *flow_yx, REAL *out, int N, int C, int H,   \
                              int W, int clip_grid);                                                \
  int mfn_ref##SFX##_warp_bwd(const REAL *gout, const REAL *x, const REAL *flow_yx, REAL *gx,       \
                              REAL *gflow_yx, int N, int C, int H, int W, int clip_grid);           \
  int mfn_ref##SFX##_deform_conv_out_shape(int H, int W, int kh, int kw, int sh, int sw, int ph,    \
                                           int pw, int dh, int dw, int *Ho, int *Wo);               \
  int mfn_ref##SFX##_deform_conv_fwd(const REAL *x, const REAL *offset, const REAL *w,              \
                                     const REAL *bias_or_null, REAL *out, int N, int Cin, int H,    \
                                     int W, int Cout, int kh, int kw, int sh, int sw, int ph,       \
                                     int pw, int dh, int dw, int groups, int deform_groups);        \
  int mfn_ref##SFX##_deform_conv_bwd(const REAL *gout, const REAL *x, const REAL *offset,           \
                                     const REAL *w, REAL *gx, REAL *goffset, REAL *gw,              \
                                     REAL *gbias_or_null, int N, int Cin, int H, int W, int Cout,   \
                                     int kh, int kw, int sh, int sw, int ph, int pw, int dh,        \
                                     int dw, int groups, int deform_groups);                        \
  /* MaskFlownet.py:230 offset builder: off[n,2k+t,y,x] = flow[n,t,y,x]*scale/stride, k<taps. */    \
  int mfn_ref##SFX##_offsets_from_flow(const REAL *flow_yx, REAL *offset, int N, int H, int W,      \
                                       int taps, REAL scale, REAL stride);                          \
  /* MaskFlownet.py:35-62 Upsample(factor): edge-pad + transposed conv with a triangle kernel. */   \
  int mfn_ref##SFX##_upsample(const REAL *img, REAL *out, int N, int C, int H, int W, int factor);   \
  /* MXNet Convolution / Deconvolution behind nn.Conv2D / nn.Conv2DTranspose of MaskFlownet.py:79-163 */ \
  int mfn_ref##SFX##_conv2d_out_shape(int H, int W, int kh, int kw, int sh, int sw, int ph, int pw,  \
                                      int dh, int dw, int transposed, int adj_h, int adj_w, int *Ho, \
                                      int *Wo);                                                      \
  int mfn_ref##SFX##_conv2d_fwd(const REAL *x, const REAL *w, const REAL *bias_or_null, REAL *out,  \
                                int N, int Cin, int H, int W, int Cout, int kh, int kw, int sh,      \
                                int sw, int ph, int pw, int dh, int dw, int groups);                 \
  int mfn_ref##SFX##_conv2d_transpose_fwd(const REAL *x, const REAL *w, const REAL *bias_or_null,   \
                                          REAL *out, int N, int Cin, int H, int W, int Cout, int kh, \
                                          int kw, int sh, int sw, int ph, int pw, int dh, int dw,    \
                                          int groups, int adj_h, int adj_w);

MFN_REF_DECLARE(, float)
MFN_REF_DECLARE(64, double)

const char *mfn_ref_version(void);
/* DeformableConvolution forward: 0 = bilinear fractions as MXNet's deformable_im2col.h computes them (default),
 * 1 = from absolute coordinates as SURVEY.md A.3 states them (mfn_ref.c). */
void mfn_ref_set_dc_fraction_mode(int mode);
int mfn_ref_get_dc_fraction_mode(void);

#ifdef __cplusplus
}
#endif
#endif /* MFN_REF_H */
