"""The hot-path pass of maskflownet_amd/hotpath.py on the CPU oracle (TEST INFRASTRUCTURE ONLY).

Used by tests/, __graft_entry__.smoke() (as the checker) and bench.py's cpu_baseline leg (as the
reported CPU baseline, kind "port").  Same operator sequence as MaskFlownet_S.hybrid_forward
(/root/reference/network/MaskFlownet.py:215-311), same seeded inputs.
"""
import numpy as np

from maskflownet_amd.hotpath import MD, MD_CASCADE, SCALE, STRIDES

from . import ref as oracle


def _leaky(x):
    return np.where(x > 0, x, np.float32(0.1) * x).astype(np.float32)


def oracle_pass(host, n_pairs, kind="S", mode="dropin"):
    """Run the pass on the first `n_pairs` samples of the synthetic batch `host` (numpy dict).
    kind "full" adds the cascade of MaskFlownet.hybrid_forward (:459-527; mode "fused" applies the LeakyReLUs
    the fused operators carry), kind "train" the backward of the S pass's correlations and deformable convs."""
    sl = slice(0, n_pairs)
    out = {}
    offs = {}
    out["corr6"] = oracle.correlation(host["c1_6"][sl], host["c2_6"][sl], max_displacement=MD, pad_size=MD)
    for l in (5, 4, 3, 2):
        off = offs[l] = oracle.offsets_from_flow(host["flow_%d" % l][sl], SCALE, float(STRIDES[l]))
        out["deform%d" % l] = oracle.deformable_convolution(host["c2_%d" % l][sl], off, host["w_%d" % l],
                                                            host["b_%d" % l], kernel=(3, 3), pad=(1, 1))
        out["corr%d" % l] = oracle.correlation(host["c1_%d" % l][sl], out["deform%d" % l], max_displacement=MD,
                                               pad_size=MD)
    out["warp"] = oracle.warp(host["img2"][sl], host["flow_full"][sl], clip_grid=False)
    if kind == "full":
        act = _leaky if mode == "fused" else (lambda x: x)
        for l in (6, 5, 4, 3, 2):
            off = oracle.offsets_from_flow(host["flow_%d" % l][sl], SCALE, float(STRIDES[l]))
            wu = act(oracle.deformable_convolution(host["c2_%d" % l][sl], off, host["wu_%d" % l], host["bu_%d" % l],
                                                   kernel=(3, 3), pad=(1, 1)))
            out["deform_u%d" % l] = wu
            out["corr_u%d" % l] = act(oracle.correlation(host["c1_%d" % l][sl], wu, max_displacement=MD_CASCADE,
                                                         pad_size=MD_CASCADE))
            out["corr_v%d" % l] = act(oracle.correlation(host["c3_%d" % l][sl], host["c4_%d" % l][sl],
                                                         max_displacement=MD_CASCADE, pad_size=MD_CASCADE))
    if kind == "train":
        for l in (2, 3, 4, 5, 6):
            d2 = out["deform%d" % l] if l != 6 else host["c2_6"][sl]
            g1, g2 = oracle.correlation_backward(host["gcorr_%d" % l][sl], host["c1_%d" % l][sl], d2,
                                                 max_displacement=MD, pad_size=MD)
            out["g_c1_%d" % l], out["g_warp_%d" % l] = g1, g2
            if l != 6:
                gx, goff, gw, gb = oracle.deformable_convolution_backward(g2, host["c2_%d" % l][sl], offs[l],
                                                                          host["w_%d" % l], with_bias=True,
                                                                          kernel=(3, 3), pad=(1, 1))
                out["g_c2_%d" % l], out["gw_%d" % l], out["gb_%d" % l] = gx, gw, gb
                if mode == "fused":   # d/dflow = scale / stride * sum over the nine taps (MaskFlownet.py:230)
                    n_, _, h_, w_ = goff.shape
                    out["g_flow_%d" % l] = (goff.reshape(n_, 9, 2, h_, w_).sum(axis=1)
                                            * (np.float32(SCALE) / np.float32(STRIDES[l]))).astype(np.float32)
                else:
                    out["g_offset_%d" % l] = goff
    return out
