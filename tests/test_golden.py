"""Committed golden vectors (tests/golden/kat_v1.npz, made by tests/golden/make_golden.py from the fp32 CPU oracle).

The reference ships no vectors for this path, so the fixture does two jobs: it pins the oracle against
silent change (CPU), and it is a checker for the kernels that needs nothing but numpy on the GPU box -- through the
kernel-logic emulation here, through libmfn_hip.so with `-m gpu`."""
import os

import numpy as np
import pytest

from tests import parity_cases as pc

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kat_v1.npz"))
ident = lambda a: a


def test_generator_is_deterministic_and_oracle_unchanged(oracle):
    from tests.golden import make_golden
    fresh = make_golden.build()
    assert sorted(fresh) == sorted(G.files)
    for k in G.files:
        np.testing.assert_array_equal(fresh[k], G[k], err_msg=k)  # bit-exact: same seeds, same fp32 oracle


def _run_forward(ops, dev, host):
    for tag, md in (("corr4", 4), ("corr2", 2)):
        got = host(ops.Correlation(dev(G[tag + "_f1"]), dev(G[tag + "_f2"]), kernel_size=1, max_displacement=md, stride1=1,
                                   stride2=1, pad_size=md, is_multiply=True))
        pc.check_close(got, G[tag + "_out"], what="golden " + tag)
    pc.check_close(host(ops.warp(dev(G["warp_x"]), dev(G["warp_flow"]), clip_grid=False)), G["warp_out"], what="golden warp")
    pc.check_close(host(ops.warp(dev(G["warp_x"]), dev(G["warp_flow"]), clip_grid=True)), G["warp_out_clip"], what="golden warp clip")
    off = ops.offsets_from_flow(dev(G["dc_flow"]), 20.0, 8.0)
    np.testing.assert_array_equal(host(off), G["dc_offset"])
    kw = dict(kernel=(3, 3), pad=(1, 1), num_filter=14)
    pc.check_close(host(ops.DeformableConvolution(dev(G["dc_x"]), off, dev(G["dc_w"]), dev(G["dc_b"]), **kw)), G["dc_out"],
                   what="golden deform shared")
    pc.check_close(host(ops.deformable_convolution_shared(dev(G["dc_x"]), dev(G["dc_flow"]), 20.0, 8.0, dev(G["dc_w"]),
                                                          dev(G["dc_b"]))), G["dc_out"], what="golden deform fused")
    pc.check_close(host(ops.DeformableConvolution(dev(G["dc_x"]), dev(G["dc_offset_pertap"]), dev(G["dc_w"]), dev(G["dc_b"]),
                                                  **kw)), G["dc_out_pertap"], what="golden deform per-tap")


def _run_backward(ops, dev, host):
    for tag, md in (("corr4", 4), ("corr2", 2)):
        g1, g2 = ops.Correlation_backward(dev(G[tag + "_gout"]), dev(G[tag + "_f1"]), dev(G[tag + "_f2"]), kernel_size=1,
                                          max_displacement=md, stride1=1, stride2=1, pad_size=md, is_multiply=True)
        pc.check_close(host(g1), G[tag + "_g1"], what="golden %s g1" % tag)
        pc.check_close(host(g2), G[tag + "_g2"], what="golden %s g2" % tag)
    gx, gf = ops.warp_backward(dev(G["warp_gout"]), dev(G["warp_x"]), dev(G["warp_flow"]), clip_grid=False)
    pc.check_close(host(gx), G["warp_gx"], what="golden warp gx")
    pc.check_close(host(gf), G["warp_gflow"], tol=5e-5, what="golden warp gflow")
    gx, goff, gw, gb = ops.DeformableConvolution_backward(dev(G["dc_gout"]), dev(G["dc_x"]), dev(G["dc_offset_pertap"]),
                                                          dev(G["dc_w"]), kernel=(3, 3), pad=(1, 1))
    pc.check_close(host(gx), G["dc_gx"], tol=5e-5, what="golden deform gx")
    pc.check_close(host(goff), G["dc_goffset"], tol=5e-5, what="golden deform goffset")
    pc.check_close(host(gw), G["dc_gw"], tol=5e-5, what="golden deform gw")
    pc.check_close(host(gb), G["dc_gb"], tol=5e-5, what="golden deform gb")


def test_kernel_sources_match_golden_through_the_emulation():
    from tests.emu import emu_ops
    ops = emu_ops.emu_ops()
    _run_forward(ops, ident, ident)
    _run_backward(ops, ident, ident)


@pytest.mark.gpu
def test_hip_kernels_match_golden():
    import torch
    from maskflownet_amd import ops as o
    assert torch.cuda.is_available()
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    host = lambda t: t.detach().cpu().numpy()
    _run_forward(o.default_ops(), dev, host)
    _run_backward(o.default_ops(), dev, host)
