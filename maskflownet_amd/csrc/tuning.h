// tuning.h -- kernel-selection knobs behind mfn_set_tuning()/mfn_get_tuning() (include/mfn_hip.h).
// 0 (or any value a key does not list) means "let the library choose".
//   corr.tw      tile width of the tiled correlation kernel: 64 | 32 | 16 | 8
//   corr.variant named (NCH, CK, DYW, PF, WPE) point, see kernels/correlation.h; -1 = default
//   corr.slices  channel slices per tile (partial sums + reduce kernel); 0 = heuristic
//   corr.lanemap 0: ds_read_b128 service-group lane order, 1: natural lane order
//   corr.band    one-launch row-band kernel of the coarse levels: 0 auto (<= 2048 px images), 1 always, 2 never
//   corr.direct  LDS-free kernel of the tiniest levels: 0 auto (< 180 px images), 1 always, 2 never
//   corr.xcd     1: XCD-aware block remap (neighbouring tiles share an L2)
//   corr.generic 1: force the generic one-thread-per-output kernel
//   corr.bwdsplit 1: the correlation backward computes g1 and g2 in separate blocks of one launch, 2: in separate launches,
//                 0: one thread computes both
//   corr.stagger shader cycles by which the LDS-DMA correlation kernel delays the k-th dispatch round of blocks (0 = off)
//   store.corr / store.dc / store.warp / store.off   the same per kernel family (override store.policy)
//   store.policy cache policy of the kernels' output stores: -1 auto (outputs >= 4 MB: sc0 sc1 for cost volumes and offsets,
//                nt for warp; deformable conv and everything smaller plain), 0 plain, 1 nt, 2 sc0 sc1 (write-through), 3 sc0 sc1 nt
//   corr.ablate  measurement only, bit mask (LDS-DMA kernel): 1 no stores, 2 no global loads, 4 no LDS reads / FMAs
//   warp.vec     pixels per thread of the warp kernel: 0 auto (fast kernel) | 1 general | 4 adjacent px, 16-byte stores | 2, 8: 2 / 4 strided px
//   dc.mt        32-filter MFMA tiles per wave: 1 | 2 | 3 | 4
//   dc.pt        pixel tiles per block: 1 | 2 | 4   (the block's 4 waves split K 4/pt ways)
//   dc.ksb       K split across blocks (partial sums + reduce kernel); 0 = heuristic
//   dc.nw        waves per block: 0 auto, 4, 8 (8 only with mt = pt = 1)
//   dc.xcd       1: every XCD works on one contiguous range of pixel tiles (mfn_xcd_remap), 0: dispatch order
//   dc.tile      pixel-tile shape: 0 auto (4x8), 16 force 2x16, 1 force 32 flattened pixels
//   dc.stage     0: disable the LDS source-window staging of the shared-offset path
//   dc.fast      0: disable the shared-offset 4x4-neighbourhood gather
//   dc.generic   1: force the generic one-thread-per-output kernel
//   dc.bwdshared 0: input/offset gradient tap by tap only (no shared-offset kernel)
//   dc.bwdpix    0: shared-offset backward with lane = channel (dc_bwd_input_shared_kernel) instead of lane = pixel (dc_backward.h)
//   corr.bwdlds  0: corr_bwd_block_kernel at every level; 1 (default): corr_bwd_lds_kernel where W is 8, 16, ... 256 (the other
//                feature map's rows copied to LDS once per block, g2's gout quads as unaligned loads, gout requested a row ahead)
//   dc.bwdflow   0: mfn_deform_conv_shared_bwd always composes (offsets into the workspace -> mfn_deform_conv_bwd -> sum of
//                the taps' offset gradients); 1 (default): where the lane = pixel kernels apply they read the flow field and
//                write d/dflow themselves -- no offset tensor, no goffset
//   dc.bwdsplit2 1: input and offset gradient in separate launches of the lane = pixel kernel, two blocks per CU each (measured
//                r02, levels 5..2: 71 / 97 / 116 / 167 us against 54 / 69 / 106 / 161: the column-gradient GEMM and the setup
//                are done twice and the gx flush's atomic instructions do not get faster); 0 (default): one launch forms
//                both, one block per CU
//   dc.bwdksplit filter slices (blockIdx.z) of the lane = pixel input / offset gradient: 0 auto (256 / blocks), 1, 2, ...
//   dc.bwdwpix   weight gradient: 1 (default) the forward's column producer + slab reduce (dc_backward.h) up to 96 filters, 2: up
//                to 128 filters (deterministic sums at every level), 0: per-tap gathers + atomics (dc_bwd_weight_mfma_kernel)
//   dc.bwdwpc    1 (default): the slab kernel with producer and consumer waves (eight per block), 0: four waves that do both
//   dc.bwdstrips 2x16-pixel strips per block of the shared-offset backward kernel: 0 auto, 2, 4
//   dc.bwdscratch 1: the shared-offset backward hands its gx windows over through the workspace and a gather pass adds them
//                 (no atomics; measured r02: dc_bwd_input_shared 547 -> 492 us per cfg5 pass + 50 us of gather = no gain, the
//                 kernel is bound by its per-pixel instructions, not by the atomic flush), 0 (default): atomic flush
//   dc.bwdwblocks target number of blocks of the weight-gradient kernel (pixel slices x combo groups x filter groups); 0 auto
//   conv.generic 1: force the generic one-thread-per-output convolution kernel
//   conv.shuffle 0: run 4x4 / stride-2 / pad-1 transposed convolutions with the masked-tap kernel instead of as a 3x3 convolution
//                with four pseudo-filters per filter (one per output parity)
//   conv.row3    one dwordx3 load per kernel row instead of three dword gathers (3x3, column dilation 1): 1 = for strided
//                convolutions (default), 2 = always, 0 = never
//   conv.mt / conv.pt  32-filter tiles per wave (1..4) / pixel tiles per block (4, or 1 = four in-block K slices); 0 = plan
#pragma once
#include <string.h>
namespace mfn {
struct Tuning {
  int corr_tw = 0, corr_variant = -1, corr_xcd = 1, corr_generic = 0, corr_ablate = 0, corr_slices = 0, corr_lanemap = 0, corr_band = 0, corr_direct = 0, corr_bwdsplit = 1, corr_stagger = 0, corr_bwdlds = 1;
  int store_policy = -1, store_corr = -1, store_dc = -1, store_warp = -1, store_off = -1;
  int warp_vec = 0;
  int conv_generic = 0, conv_mt = 0, conv_pt = 0, conv_shuffle = 1, conv_row3 = 1;
  int dc_mt = 0, dc_pt = 0, dc_ksb = 0, dc_fast = 1, dc_generic = 0, dc_stage = 1, dc_tile = 0, dc_nw = 0, dc_xcd = 1, dc_bwdshared = 1, dc_bwdwblocks = 0, dc_bwdstrips = 0, dc_bwdscratch = 0, dc_bwdpix = 1, dc_bwdwpix = 1, dc_bwdksplit = 0, dc_bwdwpc = 1, dc_bwdsplit2 = 0, dc_bwdflow = 1;
  int *slot(const char *key) {
    if (!strcmp(key, "corr.tw")) return &corr_tw;
    if (!strcmp(key, "corr.variant")) return &corr_variant;
    if (!strcmp(key, "corr.bwdlds")) return &corr_bwdlds;
    if (!strcmp(key, "corr.xcd")) return &corr_xcd;
    if (!strcmp(key, "corr.generic")) return &corr_generic;
    if (!strcmp(key, "corr.ablate")) return &corr_ablate;
    if (!strcmp(key, "store.policy")) return &store_policy;
    if (!strcmp(key, "store.corr")) return &store_corr;
    if (!strcmp(key, "store.dc")) return &store_dc;
    if (!strcmp(key, "store.warp")) return &store_warp;
    if (!strcmp(key, "store.off")) return &store_off;
    if (!strcmp(key, "corr.slices")) return &corr_slices;
    if (!strcmp(key, "corr.lanemap")) return &corr_lanemap;
    if (!strcmp(key, "corr.band")) return &corr_band;
    if (!strcmp(key, "corr.direct")) return &corr_direct;
    if (!strcmp(key, "corr.bwdsplit")) return &corr_bwdsplit;
    if (!strcmp(key, "corr.stagger")) return &corr_stagger;
    if (!strcmp(key, "warp.vec")) return &warp_vec;
    if (!strcmp(key, "conv.generic")) return &conv_generic;
    if (!strcmp(key, "conv.mt")) return &conv_mt;
    if (!strcmp(key, "conv.pt")) return &conv_pt;
    if (!strcmp(key, "conv.shuffle")) return &conv_shuffle;
    if (!strcmp(key, "conv.row3")) return &conv_row3;
    if (!strcmp(key, "dc.mt")) return &dc_mt;
    if (!strcmp(key, "dc.pt")) return &dc_pt;
    if (!strcmp(key, "dc.ksb")) return &dc_ksb;
    if (!strcmp(key, "dc.fast")) return &dc_fast;
    if (!strcmp(key, "dc.stage")) return &dc_stage;
    if (!strcmp(key, "dc.tile")) return &dc_tile;
    if (!strcmp(key, "dc.nw")) return &dc_nw;
    if (!strcmp(key, "dc.xcd")) return &dc_xcd;
    if (!strcmp(key, "dc.generic")) return &dc_generic;
    if (!strcmp(key, "dc.bwdshared")) return &dc_bwdshared;
    if (!strcmp(key, "dc.bwdwblocks")) return &dc_bwdwblocks;
    if (!strcmp(key, "dc.bwdstrips")) return &dc_bwdstrips;
    if (!strcmp(key, "dc.bwdscratch")) return &dc_bwdscratch;
    if (!strcmp(key, "dc.bwdpix")) return &dc_bwdpix;
    if (!strcmp(key, "dc.bwdwpix")) return &dc_bwdwpix;
    if (!strcmp(key, "dc.bwdksplit")) return &dc_bwdksplit;
    if (!strcmp(key, "dc.bwdwpc")) return &dc_bwdwpc;
    if (!strcmp(key, "dc.bwdsplit2")) return &dc_bwdsplit2;
    if (!strcmp(key, "dc.bwdflow")) return &dc_bwdflow;
    return nullptr;
  }
};
}  // namespace mfn
