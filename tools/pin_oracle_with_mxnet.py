#!/usr/bin/env python3
"""Pin the CPU oracle against Apache MXNet itself -- for a machine that HAS MXNet (this image has not: SURVEY.md 8c).

    pip install mxnet==1.5.1          # what the reference was tested with (/root/reference/README.md:27); Python <= 3.7
    python tools/pin_oracle_with_mxnet.py            # writes tests/golden/mxnet_kat_v1.npz and prints the comparison
    python -m pytest tests/test_oracle_pinned_by_mxnet.py     # the same comparison as a test (skipped while the file is absent)

What it does: feeds the INPUTS of tests/golden/kat_v1.npz (the seeded known-answer shapes the oracle and the HIP kernels
are already checked on) to the four MXNet operators the hot path reaches, forward and backward through mx.autograd, on
mx.cpu() (mx.gpu(0) for DeformableConvolution if the CPU build has no kernel for it -- question Q5 below), stores MXNet's
outputs, and runs a probe per open question of SURVEY.md Appendix A.5:

  Q1  Correlation: channel order (dx fastest), 1/(kernel^2 C) normaliser, zero padding         -> corr4_out / corr2_out
  Q2  DeformableConvolution: a tap at h_im in (-1, 0) contributes ZERO (the `h_im >= 0` test)    -> probe_q2
  Q3  DeformableConvolution: a tap in [H-1, H) is clamped to the last row (weight 1)           -> probe_q3
  Q4  DeformableConvolution: bilinear fractions from the (h_in, w_in)-relative map_h (mode 0) or from h_im (mode 1)
                                                                                               -> dc_out, both oracle modes
  Q5  DeformableConvolution has a CPU kernel in this MXNet build                                -> meta_dc_context
  Q6  Correlation / BilinearSampler loop order (fp32 summation order): bit-exact or 1 ulp class  -> reported as max ulps
  NET the reference's own MaskFlownet_S (network/MaskFlownet.py, from MFN_REFERENCE) under MXNet with the seeded weights of
      oracle/network_ref.Params(seed=3) against the restatement oracle/network_ref.Net                -> net_pred*, net_occlusion

Call sites restated here (test infrastructure; the product never imports this):
  /root/reference/network/MaskFlownet.py:193-195, :440-441   F.Correlation(..., pad_size=md, kernel_size=1, max_displacement=md,
                                                             stride1=1, stride2=1, is_multiply=1)
  /root/reference/network/layer.py:14-18, :26-30             GridGenerator(flow.flip(axis=1), 'warp') [.clip(-1, 1)] + BilinearSampler
  /root/reference/network/layer.py:117-121, kwargs :91-95    contrib.DeformableConvolution(x, offset, weight, bias, kernel, stride,
                                                             dilate, pad, num_filter, num_group, num_deformable_group, no_bias)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden", "kat_v1.npz")
OUT = os.path.join(ROOT, "tests", "golden", "mxnet_kat_v1.npz")


def probes():
    """Inputs of the single-purpose probes (Q2, Q3): one channel, one filter whose only non-zero weight is the centre tap,
    a linear ramp as data, all nine taps sharing one offset -- the output IS the sampled value at (y + dy, x + dx)."""
    H, W = 5, 6
    x = (np.arange(H, dtype=np.float32)[:, None] * 10 + np.arange(W, dtype=np.float32)[None, :] + 1).reshape(1, 1, H, W)
    w = np.zeros((1, 1, 3, 3), np.float32)
    w[0, 0, 1, 1] = 1.0
    off2 = np.zeros((1, 18, H, W), np.float32)
    off2[:, 0::2] = -0.5            # Q2: row 0 samples h_im = -0.5 -> 0 under `h_im >= 0`; 0.5 * x[0] under `h_im > -1`
    off3 = np.zeros((1, 18, H, W), np.float32)
    off3[:, 0::2] = 0.5             # Q3: row H-1 samples h_im = H - 0.5 -> x[H-1] (clamp) ; 0.5 * x[H-1] without it
    return x, w, off2, off3


def run_mxnet():
    import mxnet as mx
    from mxnet import autograd, nd
    G = np.load(GOLD)
    out = {"meta_mxnet_version": np.array(mx.__version__)}
    ctx = mx.cpu()

    def A(a, c=ctx):
        return nd.array(np.ascontiguousarray(a), ctx=c, dtype="float32")

    # ---- Correlation, md = 4 and md = 2 (MaskFlownet.py:193-195, :440-441) ----
    for tag, md in (("corr4", 4), ("corr2", 2)):
        f1, f2 = A(G[tag + "_f1"]), A(G[tag + "_f2"])
        f1.attach_grad()
        f2.attach_grad()
        with autograd.record():
            y = nd.Correlation(f1, f2, pad_size=md, kernel_size=1, max_displacement=md, stride1=1, stride2=1, is_multiply=1)
        y.backward(A(G[tag + "_gout"]))
        out[tag + "_out"], out[tag + "_g1"], out[tag + "_g2"] = y.asnumpy(), f1.grad.asnumpy(), f2.grad.asnumpy()

    # ---- Reconstruction2D / Reconstruction2DSmooth (layer.py:14-18, :26-30) ----
    x, fl = A(G["warp_x"]), A(G["warp_flow"])
    x.attach_grad()
    fl.attach_grad()
    with autograd.record():
        grid = nd.GridGenerator(data=fl.flip(axis=1), transform_type="warp")
        y = nd.BilinearSampler(x, grid)
    y.backward(A(G["warp_gout"]))
    out["warp_out"], out["warp_gx"], out["warp_gflow"] = y.asnumpy(), x.grad.asnumpy(), fl.grad.asnumpy()
    grid = nd.GridGenerator(data=A(G["warp_flow"]).flip(axis=1), transform_type="warp").clip(-1, 1)
    out["warp_out_clip"] = nd.BilinearSampler(A(G["warp_x"]), grid).asnumpy()

    # ---- contrib.DeformableConvolution (layer.py:117-121) ----
    kw = dict(kernel=(3, 3), stride=(1, 1), dilate=(1, 1), pad=(1, 1), num_filter=14, num_group=1, num_deformable_group=1,
              no_bias=False)

    def dc(c, data, offset, weight, bias, gout=None, **k):
        arrs = [A(a, c) for a in (data, offset, weight, bias)]
        for a in arrs:
            a.attach_grad()
        with autograd.record():
            y = nd.contrib.DeformableConvolution(*arrs, **k)
        if gout is not None:
            y.backward(A(gout, c))
        return [y.asnumpy()] + ([a.grad.asnumpy() for a in arrs] if gout is not None else [])

    dc_ctx = ctx
    try:
        dc(ctx, G["dc_x"], G["dc_offset"], G["dc_w"], G["dc_b"], **kw)
        nd.waitall()
    except Exception as e:   # Q5: MXNet <= 1.5 may ship deformable_im2col for the GPU only
        print("DeformableConvolution on mx.cpu() failed (%s): trying mx.gpu(0)" % (str(e).splitlines()[0],))
        dc_ctx = mx.gpu(0)
    out["meta_dc_context"] = np.array(str(dc_ctx))
    out["dc_out"] = dc(dc_ctx, G["dc_x"], G["dc_offset"], G["dc_w"], G["dc_b"], **kw)[0]
    r = dc(dc_ctx, G["dc_x"], G["dc_offset_pertap"], G["dc_w"], G["dc_b"], gout=G["dc_gout"], **kw)
    out["dc_out_pertap"], out["dc_gx"], out["dc_goffset"], out["dc_gw"], out["dc_gb"] = r
    px, pw, off2, off3 = probes()
    pk = dict(kw, num_filter=1)
    zb = np.zeros(1, np.float32)
    out["probe_q2"] = dc(dc_ctx, px, off2, pw, zb, **pk)[0]
    out["probe_q3"] = dc(dc_ctx, px, off3, pw, zb, **pk)[0]
    nd.waitall()
    return out


def run_network(out):
    """Network-level probe: the reference's OWN MaskFlownet_S (network/MaskFlownet.py:66-315, imported
    from MFN_REFERENCE, default /root/reference) on mx.cpu() with the seeded weights of oracle/network_ref.Params(seed=3) on the
    synthetic pair of the EPE tests -> its predictions, occlusion mask and warped image.  One MXNet run then pins the harness
    (oracle/network_ref.py, which bench.py's EPE delta and tests/test_network_epe.py are measured against) as well as the
    operators.  Parameter names: Gluon's are '<model prefix><block prefix>...weight'; the restatement's keys are
    '<block prefix>.weight' -- matched by the longest block prefix contained in the name, and the match must be a bijection."""
    import importlib
    import types
    import mxnet as mx
    from oracle import network_ref as nr
    ref_root = os.environ.get("MFN_REFERENCE", "/root/reference")
    pkg = types.ModuleType("mfn_refnet_pin")
    pkg.__path__ = [os.path.join(ref_root, "network")]   # a bare namespace: network/__init__.py (pipeline, trainer) is not executed
    sys.modules["mfn_refnet_pin"] = pkg
    net_mod = importlib.import_module("mfn_refnet_pin.MaskFlownet")

    class _Knob:
        def get(self, default=None):
            return default

    class _Section:
        def __getattr__(self, name):
            return _Knob()

    class _Config:   # network/config.py's Reader with an empty file: every knob at its default
        network = _Section()
        optimizer = _Section()

    im1, im2 = nr.synthetic_pair(1, 64, 64, seed=5)
    P = nr.Params(seed=3)
    want = nr.Net(P, nr.OracleMatching(), "cpu").forward(im1, im2)    # creates every key of P.store
    net = net_mod.MaskFlownet_S(_Config())
    net.initialize(mx.initializer.Zero(), ctx=mx.cpu())
    a, b = mx.nd.array(im1), mx.nd.array(im2)
    net(a, b)                                                          # shapes the deferred parameters
    layers = sorted({k.rsplit(".", 1)[0] for k in P.store}, key=len, reverse=True)
    used = set()
    for name, par in net.collect_params().items():
        kind = "weight" if name.endswith("weight") else ("bias" if name.endswith("bias") else None)
        hit = next((l for l in layers if l in name), None)
        if kind is None or hit is None:
            raise RuntimeError("cannot map Gluon parameter %r to a layer of the restatement (layers: %s ...)" % (name, layers[:6]))
        key = hit + "." + kind
        if key in used:
            raise RuntimeError("two Gluon parameters map to %r (second: %r)" % (key, name))
        used.add(key)
        par.set_data(mx.nd.array(P.get(key, tuple(par.shape))))
    if used != set(P.store):
        raise RuntimeError("parameters of the restatement without a Gluon counterpart: %s" % sorted(set(P.store) - used)[:8])
    preds, occ, srcs = net(a, b)
    for i, pr in enumerate(preds):
        out["net_pred%d" % i] = pr.asnumpy()
    out["net_occlusion"] = occ[0].asnumpy()
    out["net_warped"] = srcs[4].asnumpy()[:, :3]
    out["net_npreds"] = np.array(len(preds))
    return want


def compare_network(M, verbose=True):
    """oracle/network_ref.Net against the stored outputs of the reference's MaskFlownet_S under MXNet ({name: rel err})."""
    from oracle import network_ref as nr
    if "net_npreds" not in M:
        return {}
    im1, im2 = nr.synthetic_pair(1, 64, 64, seed=5)
    want = nr.Net(nr.Params(seed=3), nr.OracleMatching(), "cpu").forward(im1, im2)
    res = {}
    rel = lambda got, ref: float(np.abs(np.asarray(got, np.float64) - ref).max() / max(np.abs(ref).max(), 1e-30))
    for i in range(int(M["net_npreds"])):
        res["net_pred%d" % i] = rel(want["predictions"][i], M["net_pred%d" % i])
    res["net_occlusion"] = rel(want["occlusion"], M["net_occlusion"])
    res["net_warped"] = rel(want["warped"], M["net_warped"])
    if verbose:
        for k in sorted(res):
            print("  %-16s rel err %.3e   (restated MaskFlownet_S vs the reference's own under MXNet)" % (k, res[k]))
    return res


def ulps(a, b):
    """Largest difference in units of the last place of the larger magnitude (0 = bit-exact)."""
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    scale = np.maximum(np.abs(a), np.abs(b))
    with np.errstate(divide="ignore", invalid="ignore"):
        u = np.abs(a.astype(np.float64) - b) / np.spacing(np.maximum(scale, np.float32(1e-30)).astype(np.float32))
    return float(np.nanmax(u)) if u.size else 0.0


def compare(M, verbose=True):
    """Oracle vs the stored MXNet outputs: {name: (max |diff| / max |mxnet|, max ulps)} plus the answers to Q2 - Q5.
    Used by tests/test_oracle_pinned_by_mxnet.py."""
    from oracle import ref
    G = np.load(GOLD)
    res, answers = {}, {}

    def put(name, got):
        want = M[name]
        res[name] = (float(np.abs(got.astype(np.float64) - want).max() / max(np.abs(want).max(), 1e-30)), ulps(got, want))

    for tag, md in (("corr4", 4), ("corr2", 2)):
        put(tag + "_out", ref.correlation(G[tag + "_f1"], G[tag + "_f2"], max_displacement=md, pad_size=md))
        g1, g2 = ref.correlation_backward(G[tag + "_gout"], G[tag + "_f1"], G[tag + "_f2"], max_displacement=md, pad_size=md)
        put(tag + "_g1", g1)
        put(tag + "_g2", g2)
    put("warp_out", ref.warp(G["warp_x"], G["warp_flow"], clip_grid=False))
    put("warp_out_clip", ref.warp(G["warp_x"], G["warp_flow"], clip_grid=True))
    gx, gf = ref.warp_backward(G["warp_gout"], G["warp_x"], G["warp_flow"], clip_grid=False)
    put("warp_gx", gx)
    put("warp_gflow", gf)
    kw = dict(kernel=(3, 3), pad=(1, 1))
    per_mode = {}
    for mode in (0, 1):   # Q4
        ref.set_dc_fraction_mode(mode)
        got = ref.deformable_convolution(G["dc_x"], G["dc_offset"], G["dc_w"], G["dc_b"], **kw)
        per_mode[mode] = (float(np.abs(got.astype(np.float64) - M["dc_out"]).max()), ulps(got, M["dc_out"]))
    ref.set_dc_fraction_mode(0)
    answers["Q4_fraction_source"] = {"mode0_map_h_relative": per_mode[0], "mode1_absolute_h_im": per_mode[1],
                                     "closer": 0 if per_mode[0][0] <= per_mode[1][0] else 1}
    put("dc_out", ref.deformable_convolution(G["dc_x"], G["dc_offset"], G["dc_w"], G["dc_b"], **kw))
    put("dc_out_pertap", ref.deformable_convolution(G["dc_x"], G["dc_offset_pertap"], G["dc_w"], G["dc_b"], **kw))
    gx, goff, gw, gb = ref.deformable_convolution_backward(G["dc_gout"], G["dc_x"], G["dc_offset_pertap"], G["dc_w"], **kw)
    for n_, g_ in (("dc_gx", gx), ("dc_goffset", goff), ("dc_gw", gw), ("dc_gb", gb)):
        put(n_, g_)
    px, pw, off2, off3 = probes()
    zb = np.zeros(1, np.float32)
    q2 = ref.deformable_convolution(px, off2, pw, zb, **kw)
    q3 = ref.deformable_convolution(px, off3, pw, zb, **kw)
    put("probe_q2", q2)
    put("probe_q3", q3)
    # the rule MXNet itself follows, read off its outputs: row 0 of the Q2 probe, row H-1 of the Q3 probe
    answers["Q2_row0_is_zero"] = bool(np.all(M["probe_q2"][0, 0, 0] == 0))            # `h_im >= 0`: zero contribution
    answers["Q3_last_row_is_clamped"] = bool(np.allclose(M["probe_q3"][0, 0, -1], px[0, 0, -1]))   # weight 1 on row H-1
    answers["Q5_dc_context"] = str(M["meta_dc_context"])
    answers["mxnet_version"] = str(M["meta_mxnet_version"])
    if verbose:
        print("MXNet %s, DeformableConvolution ran on %s" % (answers["mxnet_version"], answers["Q5_dc_context"]))
        for k in sorted(res):
            print("  %-16s rel err %.3e   %.1f ulps" % (k, res[k][0], res[k][1]))
        for k, v in answers.items():
            print("  %-24s %s" % (k, v))
    return res, answers


if __name__ == "__main__":
    if "--compare-only" not in sys.argv:
        out = run_mxnet()
        if "--no-network" not in sys.argv:
            run_network(out)
        np.savez_compressed(OUT, **out)
        print("wrote", OUT)
    M = np.load(OUT)
    compare(M)
    compare_network(M)
