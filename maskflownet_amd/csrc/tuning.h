// tuning.h -- kernel-selection knobs behind mfn_set_tuning()/mfn_get_tuning() (include/mfn_hip.h).
// 0 (or any value a key does not list) means "let the library choose".  Round 3 cut the list from 47 keys to the 13 that
// select between code paths the library ships (tests force every path through them); the measurement knobs of rounds 1 / 2
// (tilings, ring depths, cache policies per kernel family, staggering, ablation masks) are gone with the variants they chose
// between -- DESIGN.md records what each of them measured.  Round 5: the three keys that selected ARITHMETIC (corr.gram, dc.mma,
// conv.mma) are gone from here: that is mfn_set_arithmetic (include/mfn_hip.h), per thread; these keys only choose tilings / paths.
//   corr.variant  6: corr_tiled_kernel (images narrower than 16 columns), 16 / 20 / 22: corr_dma_kernel with 1 / 2 / 3 channel
//                 groups, 26 / 31: the same with a tile's displacement rows spread over 5 / 3 blocks (coarse levels); -1 = the
//                 plan (api_impl.inc corr_plan)
//                 48 / 46: corr_gram_kernel (32-channel levels, 48 also 64-channel ones: the band of the Gram matrix on the matrix cores) -- 48
//                 the plan's form (three bf16 terms split ON the matrix cores, results leaving one step behind the chains), 46 raw operands
//                 on the fp32 matrix instruction (an fmaf chain over the channels: the plan's form under MFN_ARITH_FP32)
//   corr.form     the form the PLAN gives a 32-channel level (0 = 48 / 46 by arithmetic; 46 / 48; 16: corr_dma_kernel under MFN_ARITH_FP32) -- unlike corr.variant it leaves the other levels' plan alone --;
//                 20: 64-channel levels stay on corr_dma_kernel (two channel groups) instead of the two-chunk Gram band
//                 44 / 45: corr_gramk_kernel (coarse levels: the same band, a block = an 8 x 2 pixel block of f1 and two / half of the
//                 f2 rows it meets, one wave per 32 channels)
//   corr.rows     output rows per work item of corr_gram_kernel (6 or 8; 64-channel levels always 2; 0 = the plan)
//   corr.direct   LDS-free kernel of the tiniest levels: 0 auto (< 180 px images), 1 always, 2 never
//   store.policy  cache policy of the kernels' output stores: -1 auto (outputs >= 4 MB: sc0 sc1 for cost volumes and offsets,
//                 nt for warp; deformable conv and everything smaller plain), 0 plain, 1 nt, 2 sc0 sc1 (write-through), 3 sc0 sc1 nt
//   dc.mt         dc_mma_kernel (kernels/deform_conv_mma.h) only, together with dc.pt and dc.nw: 32-filter tiles per wave (the K slices
//                 per pixel tile are nw / pt); all three > 0 replace the plan's tiling
//   dc.pt         pixel tiles per block: 1 | 2 | 4   (the block's 4 waves split K 4/pt ways)
//   dc.ksb        K split across blocks (partial sums + reduce kernel); 0 = heuristic
//   dc.nw         waves per block: 0 auto, 4, 8 (8 only with pt = 1)
//   dc.off        bits that switch tiers of the shared-offset forward path off: 1 = the LDS source-window staging (every tile on
//                 the global-gather tier), 2 = the shared-offset 4x4-neighbourhood gather altogether (per-tap path)
//   path.generic  bits that force the generic one-thread-per-output kernels: 1 = correlation, 2 = deformable convolution
//                 (forward and backward), 4 = convolution
//   bwd.off       bits that switch backward kernels off: 1 = the lane = pixel shared-offset input / offset gradient (tap by tap
//                 only), 2 = flow mode (mfn_deform_conv_shared_bwd then always composes: offsets into the workspace ->
//                 mfn_deform_conv_bwd -> sum of the taps' offset gradients), 4 = corr_bwd_lds_kernel (corr_bwd_block_kernel at
//                 every level)
//   conv.mt / conv.pt  32-filter tiles per wave (1..4) / pixel tiles per block (4, or 1 = four in-block K slices); 0 = plan (the plan
//                 only picks pt = 4 and mt > 2 for images of >= 1024 pixel tiles: the CPU emulation tests reach those kernels through these)
//   conv.dcm      plain 3x3 / stride 1 / pad 1 convolutions on dc_mma_kernel<.., CONV> (kernels/deform_conv_mma.h): 0 = the plan (filter counts
//                 that are multiples of 32 on images of >= 384 pixel tiles, default arithmetic), 1 = never, 2 = whatever the tile count
//                 (the CPU emulation tests reach the kernel through this)
#pragma once
#include <string.h>
namespace mfn {
struct Tuning {
  int corr_variant = -1, corr_direct = 0, corr_rows = 0, corr_form = 0;
  int store_policy = -1;
  int dc_pt = 0, dc_ksb = 0, dc_nw = 0, dc_off = 0, dc_mt = 0;
  int path_generic = 0, bwd_off = 0;
  int conv_mt = 0, conv_pt = 0, conv_dcm = 0;
  int *slot(const char *key) {
    if (!strcmp(key, "corr.variant")) return &corr_variant;
    if (!strcmp(key, "corr.direct")) return &corr_direct;
    if (!strcmp(key, "corr.rows")) return &corr_rows;
    if (!strcmp(key, "corr.form")) return &corr_form;
    if (!strcmp(key, "store.policy")) return &store_policy;
    if (!strcmp(key, "dc.pt")) return &dc_pt;
    if (!strcmp(key, "dc.mt")) return &dc_mt;
    if (!strcmp(key, "dc.ksb")) return &dc_ksb;
    if (!strcmp(key, "dc.nw")) return &dc_nw;
    if (!strcmp(key, "dc.off")) return &dc_off;
    if (!strcmp(key, "path.generic")) return &path_generic;
    if (!strcmp(key, "bwd.off")) return &bwd_off;
    if (!strcmp(key, "conv.mt")) return &conv_mt;
    if (!strcmp(key, "conv.pt")) return &conv_pt;
    if (!strcmp(key, "conv.dcm")) return &conv_dcm;
    return nullptr;
  }
};
}  // namespace mfn
