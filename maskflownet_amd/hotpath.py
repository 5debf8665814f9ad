"""One pass of the MaskFlownet-S matching hot path over one batch -- the unit bench.py times.

The operator sequence is the one MaskFlownet_S.hybrid_forward issues per forward
(/root/reference/network/MaskFlownet.py:215-311):

    L6            corr(c16, c26)                                     :216
    L5..L2        deform_l(c2l, repeat9(flow_l * scale / stride_l))  :230,248,266,284
                  corr(c1l, warp_l)                                  :234,252,270,288
    full res      warp(c20, Upsample(4)(flow2) * scale)              :311

with the pyramid shapes of :79-96 (C = 196,128,96,64,32 at strides 64..4).  Everything between
these calls (pyramid / decoder convolutions, gating, upsampling) is out of scope (SURVEY.md
section 2 rows 5-7), so their outputs are replaced by synthetic, seeded tensors of the right
shape that stay resident in HBM; the deformable-conv output does feed the following correlation,
as in the network.

Two more passes are built from the same operators (BASELINE.json configs[3], configs[4]):

    kind "full"   (cfg4)  the S pass above plus the cascade of MaskFlownet.hybrid_forward (:459-527): per level
                          6..2 deform_l(c2l, repeat9(flow_l*scale/stride_l)) [+LeakyReLU], corr_u = corr(c1l, warp_u)
                          and corr_v = corr(c3l, c4l), both md=2 (:322) with LeakyReLU -- 5 more deformable convs
                          (own parameters, level 6 included) and 10 25-channel cost volumes.
    kind "train"  (cfg5)  forward of the S pass followed by the backward of its correlations and deformable convs
                          (a7; pipeline.py:112-113): corr_bwd -> (g c1l, g warp_l), deform_bwd(g warp_l) -> (g c2l,
                          g offset, g weight, g bias).  The out-gradients of the cost volumes (what the decoder's
                          backward would hand over) are synthetic.  Parameter gradients of all four levels live in one
                          flat bucket: the step's only exchange is one all-reduce(sum) of it (dist.allreduce_bucket).
                          The warp is forward-only (c40 feeds the cascade only).

`mode="dropin"`  : offsets are materialised (a6) and go through DeformableConvolution's own
                   (N,18,H,W) signature -- exactly the reference's operator boundary.
`mode="fused"`   : deformable_convolution_shared, the offset tensor never exists (f-1).
Launches go to a private stream; capture()/replay() wrap the ~14 launches in one hipGraph.
"""
import ctypes
import os

import numpy as np

from . import _lib
from .ops import default_ops

CHANNELS = {6: 196, 5: 128, 4: 96, 3: 64, 2: 32}      # MaskFlownet.py:79-96
STRIDES = {6: 64, 5: 32, 4: 16, 3: 8, 2: 4}           # MaskFlownet.py:71
SCALE = 20.0                                          # MaskFlownet.py:69 (flow_multiplier = 1)
MD = 4                                                # MaskFlownet.py:70

MD_CASCADE = 2                                        # MaskFlownet.py:322

CONFIGS = {
    # name: (batch per GPU, H, W[, kind])  -- BASELINE.json configs[1], configs[2]; configs[3] = batch 32 over 4 GPUs,
    # configs[4] = batch 64 over 8 GPUs: 8 pairs per GPU either way
    "cfg2": (8, 384, 512),
    "cfg3": (4, 448, 1024),
    "cfg4": (8, 384, 512, "full"),
    "cfg5": (8, 384, 512, "train"),
    "tiny": (2, 64, 128),
    "tiny_full": (2, 64, 128, "full"),
    "tiny_train": (2, 64, 128, "train"),
}
KINDS = ("S", "full", "train")


def level_shapes(N, H, W):
    return {l: (N, CHANNELS[l], H // STRIDES[l], W // STRIDES[l]) for l in (6, 5, 4, 3, 2)}


def algorithmic_bytes(N, H, W, mode="dropin", kind="S"):
    """SURVEY.md 8(d) per-op byte counts summed over one pass (fp32)."""
    out = {}
    if kind == "full":
        D2 = (2 * MD_CASCADE + 1) ** 2
        for l, (n, c, h, w) in level_shapes(N, H, W).items():
            out["deform_u%d" % l] = 4 * (n * h * w * (c + 18 + c) + 9 * c * c + c)
            if mode == "dropin":
                out["offsets_u%d" % l] = 4 * n * h * w * (2 + 18)
            out["corr_u%d" % l] = out["corr_v%d" % l] = 4 * n * h * w * (2 * c + D2)
    if kind == "train":
        for l, (n, c, h, w) in level_shapes(N, H, W).items():
            out["corr_bwd%d" % l] = 4 * n * h * w * ((2 * MD + 1) ** 2 + 4 * c)       # gout, f1, f2 in; g1, g2 out
            if l != 6:                                                             # gout, x, offset in; gx, goffset, gw, gb out
                out["deform_bwd%d" % l] = 4 * (n * h * w * (3 * c + 36) + 2 * 9 * c * c + c)
    for l, (n, c, h, w) in level_shapes(N, H, W).items():
        out["corr%d" % l] = 4 * n * h * w * (2 * c + (2 * MD + 1) ** 2)
        if l != 6:
            out["deform%d" % l] = 4 * (n * h * w * (c + 18 + c) + 9 * c * c + c)
            if mode == "dropin":
                out["offsets%d" % l] = 4 * n * h * w * (2 + 18)
    out["warp"] = 4 * N * H * W * (2 * 3 + 2)
    return out


def algorithmic_flops(N, H, W, kind="S"):
    out = {}
    if kind == "full":
        for l, (n, c, h, w) in level_shapes(N, H, W).items():
            out["deform_u%d" % l] = 2 * n * h * w * c * c * 9
            out["corr_u%d" % l] = out["corr_v%d" % l] = 2 * n * h * w * c * (2 * MD_CASCADE + 1) ** 2
    if kind == "train":
        for l, (n, c, h, w) in level_shapes(N, H, W).items():
            out["corr_bwd%d" % l] = 4 * n * h * w * c * (2 * MD + 1) ** 2
            if l != 6:
                out["deform_bwd%d" % l] = 4 * n * h * w * c * c * 9
    for l, (n, c, h, w) in level_shapes(N, H, W).items():
        out["corr%d" % l] = 2 * n * h * w * c * (2 * MD + 1) ** 2
        if l != 6:
            out["deform%d" % l] = 2 * n * h * w * c * c * 9
    return out


def upsample2(a):
    """Upsample(2) of /root/reference/network/MaskFlownet.py:35-62: out[2i] = in[i],
    out[2i+1] = (in[i] + in[i+1]) / 2 with the last row/column replicated."""
    a = np.asarray(a, np.float64)
    p = np.pad(a, ((0, 0), (0, 0), (0, 1), (0, 1)), mode="edge")
    n, c, h, w = a.shape
    out = np.empty((n, c, 2 * h, 2 * w))
    out[:, :, 0::2, 0::2] = p[:, :, :h, :w]
    out[:, :, 0::2, 1::2] = 0.5 * (p[:, :, :h, :w] + p[:, :, :h, 1:w + 1])
    out[:, :, 1::2, 0::2] = 0.5 * (p[:, :, :h, :w] + p[:, :, 1:h + 1, :w])
    out[:, :, 1::2, 1::2] = 0.25 * (p[:, :, :h, :w] + p[:, :, :h, 1:w + 1] + p[:, :, 1:h + 1, :w] + p[:, :, 1:h + 1, 1:w + 1])
    return out


FLOW_MODELS = ("smooth", "rough")


def rough_flow(rng, n, h, w):
    """SURVEY.md 8(d)'s op-level flow: N(0, sigma = 2 px) per pixel (level pixels), 2 % of the pixels uniform in
    [-h, h] (out of range).  No spatial coherence at all: a wave's 4x4 neighbourhoods do not share a window."""
    fl = rng.standard_normal((n, 2, h, w)) * 2.0
    out = rng.random((n, 1, h, w)) < 0.02
    return np.where(out, rng.uniform(-h, h, (n, 2, h, w)), fl).astype(np.float32)


def synth_inputs(N, H, W, seed=20260925, flow_model="smooth"):
    """Seeded synthetic tensors (numpy) for every level -- SURVEY.md 8(d) 'op level'.  flow_model "smooth" (default):
    flow_l is what it is in the network, the reference's Upsample(2) applied recursively to a coarse field (piecewise
    linear); "rough": 8(d)'s i.i.d. N(0, 2 px) + 2 % outliers per pixel, the adversarial case for the gather."""
    if flow_model not in FLOW_MODELS:
        raise ValueError("flow_model must be one of %s" % (FLOW_MODELS,))
    data = {}
    for l, shp in level_shapes(N, H, W).items():
        rng = np.random.default_rng(seed + l)
        for name in ("c1_%d" % l, "c2_%d" % l):  # post-activation features: leaky_relu(N(0,1), 0.1)
            x = rng.standard_normal(shp).astype(np.float32)
            data[name] = np.where(x > 0, x, np.float32(0.1) * x).astype(np.float32)
        if l != 6:
            n, c, h, w = shp
            # flow_l = Upsample(2)(flow_{l+1}) in the network (MaskFlownet.py:228), recursively: a smooth,
            # piecewise-linear field.  Model: N(0, 2 level-px) noise three levels up (1/8 resolution),
            # upsampled 3x with the reference's Upsample(2), plus a global shift per sample.
            ch8, cw8 = max(1, (h + 7) // 8), max(1, (w + 7) // 8)
            # (MFN_DIAG_FLOW_NOISE: measurement sessions only -- the coarse field's sigma, e.g. 0.5 for flows whose windows all fit)
            fl = rng.standard_normal((n, 2, ch8, cw8)) * float(os.environ.get("MFN_DIAG_FLOW_NOISE", 2.0)) + rng.uniform(-3, 3, (n, 2, 1, 1))
            for _ in range(3):
                fl = upsample2(fl)
            fl = fl[:, :, :h, :w].astype(np.float32)
            if flow_model == "rough":
                fl = rough_flow(rng, n, h, w)
            # level-pixel offsets = flow * SCALE / stride  ->  store the network-unit flow
            data["flow_%d" % l] = (fl * np.float32(STRIDES[l] / SCALE)).astype(np.float32)
            fan = 9.0 * c
            data["w_%d" % l] = (rng.standard_normal((c, c, 3, 3)) * np.sqrt(2.0 / (1.01 * fan))).astype(np.float32)
            data["b_%d" % l] = (rng.standard_normal((c,)) * 0.1).astype(np.float32)
    rng = np.random.default_rng(seed)
    data["img2"] = rng.standard_normal((N, 3, H, W)).astype(np.float32)
    # Upsample(4)(flow2) * scale (MaskFlownet.py:311): the same smooth level-2 field, 8 full-res px sigma
    ff = rng.standard_normal((N, 2, max(1, H // 32), max(1, W // 32))) * 2.0
    for _ in range(5):
        ff = upsample2(ff)
    data["flow_full"] = (ff[:, :, :H, :W] * 4.0).astype(np.float32)
    if flow_model == "rough":
        data["flow_full"] = (rough_flow(rng, N, H, W) * 4.0).astype(np.float32)   # sigma 8 full-resolution px
    return data


def _smooth_flow(rng, n, h, w, stride):
    """The flow model of synth_inputs at one level, in network units."""
    ch8, cw8 = max(1, (h + 7) // 8), max(1, (w + 7) // 8)
    fl = rng.standard_normal((n, 2, ch8, cw8)) * 2.0 + rng.uniform(-3, 3, (n, 2, 1, 1))
    for _ in range(3):
        fl = upsample2(fl)
    return (fl[:, :, :h, :w].astype(np.float32) * np.float32(stride / SCALE)).astype(np.float32)


def synth_inputs_full(N, H, W, seed=20260925):
    """What the cascade of MaskFlownet.hybrid_forward (:436-527) adds to synth_inputs: the second pyramid's features
    c3_l / c4_l (:441-455), the level-6 flow (flows[0], :458) and the cascade's own deformable-conv parameters
    deform6..deform2 (:403-407; bias on, :324)."""
    data = {}
    for l, shp in level_shapes(N, H, W).items():
        rng = np.random.default_rng(seed + 100 + l)
        n, c, h, w = shp
        for name in ("c3_%d" % l, "c4_%d" % l):
            x = rng.standard_normal(shp).astype(np.float32)
            data[name] = np.where(x > 0, x, np.float32(0.1) * x).astype(np.float32)
        if l == 6:
            data["flow_6"] = _smooth_flow(rng, n, h, w, STRIDES[6])
        data["wu_%d" % l] = (rng.standard_normal((c, c, 3, 3)) * np.sqrt(2.0 / (1.01 * 9.0 * c))).astype(np.float32)
        data["bu_%d" % l] = (rng.standard_normal((c,)) * 0.1).astype(np.float32)
    return data


def synth_inputs_train(N, H, W, seed=20260925):
    """Out-gradients of the five cost volumes (the decoder's backward is out of scope): N(0,1)/81."""
    data = {}
    for l, (n, c, h, w) in level_shapes(N, H, W).items():
        rng = np.random.default_rng(seed + 200 + l)
        data["gcorr_%d" % l] = (rng.standard_normal((n, (2 * MD + 1) ** 2, h, w)) / 81.0).astype(np.float32)
    return data


def grad_bucket_layout(N, H, W):
    """(name, offset, shape) of every parameter gradient of the S pass inside the flat all-reduce bucket, and its
    length in floats: deform5..deform2 weight + bias (MaskFlownet.py:155-158), 276 800 floats = 1.1 MB."""
    lay, off = [], 0
    for l in (5, 4, 3, 2):
        c = CHANNELS[l]
        for name, shp in (("gw_%d" % l, (c, c, 3, 3)), ("gb_%d" % l, (c,))):
            n = int(np.prod(shp))
            lay.append((name, off, shp))
            off += n
    return lay, off


def output_names(kind="S", mode="dropin"):
    """Names of everything a pass of `kind` writes, in launch order of the forward / cascade / backward parts (the training
    pass: d/doffset (18 planes) in drop-in mode, d/dflow (2 planes) in fused mode)."""
    keys = ["corr6"] + [k for l in (5, 4, 3, 2) for k in ("deform%d" % l, "corr%d" % l)] + ["warp"]
    if kind == "full":
        keys += [k % l for l in (6, 5, 4, 3, 2) for k in ("deform_u%d", "corr_u%d", "corr_v%d")]
    if kind == "train":
        keys += [k % l for l in (6, 5, 4, 3, 2) for k in ("g_c1_%d", "g_warp_%d")]
        keys += [k % l for l in (5, 4, 3, 2) for k in ("g_c2_%d", "g_offset_%d" if mode == "dropin" else "g_flow_%d", "gw_%d", "gb_%d")]
    return keys


class _TorchBuffers:
    """Device buffers of a workload: torch-ROCm tensors on `device`, launches on a private stream."""

    def __init__(self, device):
        import torch
        self.torch = torch
        self.device = torch.device(device)
        self.ops = default_ops()
        self.stream = torch.cuda.Stream(device=self.device)

    def to_device(self, a):
        return self.torch.from_numpy(a).to(self.device)

    def empty(self, shape):
        return self.torch.empty(tuple(shape), device=self.device)

    def zeros(self, n):
        return self.torch.zeros(int(n), device=self.device)

    def view(self, flat, off, shape):
        return flat[off:off + int(np.prod(shape))].view(*shape)

    def launching(self):
        return self.torch.cuda.stream(self.stream)

    def as_collective_tensor(self, flat):
        return flat  # torch.distributed reduces the device tensor in place

    def synchronize(self):
        self.stream.synchronize()


class HotPathWorkload:
    def __init__(self, cfg="cfg2", device="cuda", mode="dropin", seed=20260925, prepack=True, buffers=None,
                 flow_model="smooth"):
        """prepack=True: the deformable-conv weights are laid out once here, as layer.DeformableConv2D does
        for a block's constant parameters at inference; False re-packs inside every operator call (what a
        stateless MXNet operator sees).  Outputs are bit-identical either way.
        buffers: where the tensors live and which operator set runs the calls -- the torch-ROCm device buffers by
        default; the CPU test-suite passes numpy buffers with the kernel-emulation operator set to check the call
        lists without a GPU (no streams, no graphs there)."""
        bufs = buffers if buffers is not None else _TorchBuffers(device)
        self.bufs = bufs
        torch = self.torch = getattr(bufs, "torch", None)
        if isinstance(cfg, str):
            cfg = CONFIGS[cfg]
        self.N, self.H, self.W = cfg[:3]
        self.kind = cfg[3] if len(cfg) > 3 else "S"
        if self.kind not in KINDS:
            raise ValueError("unknown pass kind %r" % (self.kind,))
        self.mode = mode
        self.device = getattr(bufs, "device", None)
        self.ops = bufs.ops
        self.lib = _lib.lib() if buffers is None else None
        self.stream = getattr(bufs, "stream", None)
        self.flow_model = flow_model
        self.host = synth_inputs(self.N, self.H, self.W, seed, flow_model)
        if self.kind == "full":
            self.host.update(synth_inputs_full(self.N, self.H, self.W, seed))
        if self.kind == "train":
            self.host.update(synth_inputs_train(self.N, self.H, self.W, seed))
        # uploads and the weight packing below run on the workload's own stream (torch streams are non-blocking: work
        # left on the default stream would not be ordered before the first launch / the captured graph)
        with bufs.launching():
            self.t = {k: bufs.to_device(v) for k, v in self.host.items()}
        shp = level_shapes(self.N, self.H, self.W)
        D2 = (2 * MD + 1) ** 2
        self.o = {}
        for l, (n, c, h, w) in shp.items():
            self.o["corr%d" % l] = bufs.empty((n, D2, h, w))
            if l != 6:
                self.o["deform%d" % l] = bufs.empty((n, c, h, w))
                if mode == "dropin":
                    self.o["offset%d" % l] = bufs.empty((n, 18, h, w))
        self.o["warp"] = bufs.empty((self.N, 3, self.H, self.W))
        if self.kind == "full":
            D2u = (2 * MD_CASCADE + 1) ** 2
            for l, (n, c, h, w) in shp.items():
                self.o["deform_u%d" % l] = bufs.empty((n, c, h, w))
                self.o["corr_u%d" % l] = bufs.empty((n, D2u, h, w))
                self.o["corr_v%d" % l] = bufs.empty((n, D2u, h, w))
                if mode == "dropin":
                    self.o["offset_u%d" % l] = bufs.empty((n, 18, h, w))
        if self.kind == "train":
            lay, nfl = grad_bucket_layout(self.N, self.H, self.W)
            self.grad_bucket = bufs.zeros(nfl)
            for name, off, s_ in lay:
                self.o[name] = bufs.view(self.grad_bucket, off, s_)
            for l, (n, c, h, w) in shp.items():
                self.o["g_c1_%d" % l] = bufs.empty((n, c, h, w))
                self.o["g_warp_%d" % l] = bufs.empty((n, c, h, w))   # d loss / d data2 of corr_l
                if l != 6:
                    self.o["g_c2_%d" % l] = bufs.empty((n, c, h, w))
                    if mode == "dropin":
                        self.o["g_offset_%d" % l] = bufs.empty((n, 18, h, w))
                    else:
                        self.o["g_flow_%d" % l] = bufs.empty((n, 2, h, w))
        self.graph = None
        self.prepack = bool(prepack)
        self.packed = {}
        if self.prepack:
            with bufs.launching():
                for l in (5, 4, 3, 2):
                    self.packed[l] = self.ops.pack_deform_weights(self.t["w_%d" % l], shp[l], kernel=(3, 3), pad=(1, 1))
                if self.kind == "full":
                    for l in (6, 5, 4, 3, 2):
                        self.packed["u%d" % l] = self.ops.pack_deform_weights(self.t["wu_%d" % l], shp[l],
                                                                              kernel=(3, 3), pad=(1, 1))
        bufs.synchronize()
        if torch is not None:
            torch.cuda.synchronize(self.device)   # allocations / fills made on other streams are complete too

    # the operator sequence of one forward, as (name, thunk) pairs
    def calls(self):
        ops, t, o = self.ops, self.t, self.o
        seq = [("corr6", lambda: ops.Correlation(t["c1_6"], t["c2_6"], 1, MD, 1, 1, MD, True, out=o["corr6"]))]
        for l in (5, 4, 3, 2):
            if self.mode == "dropin":
                seq.append(("offsets%d" % l, lambda l=l: ops.offsets_from_flow(t["flow_%d" % l], SCALE, STRIDES[l],
                                                                               out=o["offset%d" % l])))
                seq.append(("deform%d" % l, lambda l=l: ops.DeformableConvolution(
                    t["c2_%d" % l], o["offset%d" % l], t["w_%d" % l], t["b_%d" % l], kernel=(3, 3), stride=(1, 1),
                    dilate=(1, 1), pad=(1, 1), num_filter=CHANNELS[l], out=o["deform%d" % l], packed=self.packed.get(l))))
            else:
                seq.append(("deform%d" % l, lambda l=l: ops.deformable_convolution_shared(
                    t["c2_%d" % l], t["flow_%d" % l], SCALE, STRIDES[l], t["w_%d" % l], t["b_%d" % l],
                    out=o["deform%d" % l], packed=self.packed.get(l))))
            seq.append(("corr%d" % l, lambda l=l: ops.Correlation(t["c1_%d" % l], o["deform%d" % l], 1, MD, 1, 1, MD, True,
                                                                  out=o["corr%d" % l])))
        seq.append(("warp", lambda: ops.warp(t["img2"], t["flow_full"], clip_grid=False, out=o["warp"])))
        if self.kind == "full":
            seq += self._cascade_calls()
        if self.kind == "train":
            seq += self._backward_calls()
        return seq

    def _cascade_calls(self):
        """MaskFlownet.hybrid_forward :459-527, hot-path operators only.  dropin: the reference's operator boundary
        (offset tensor, DeformableConvolution, Correlation; the LeakyReLUs between them are MXNet's own elementwise
        ops and not launched, so corr_u reads the raw deformable-conv output).  fused: deformable_matching
        (offset builder + LeakyReLU in the epilogue, no gating: :460-461) and Correlation(activation='leaky')."""
        ops, t, o = self.ops, self.t, self.o
        seq = []
        md = MD_CASCADE
        act = "leaky" if self.mode == "fused" else None
        for l in (6, 5, 4, 3, 2):
            pk = self.packed.get("u%d" % l)
            if self.mode == "dropin":
                seq.append(("offsets_u%d" % l, lambda l=l: ops.offsets_from_flow(t["flow_%d" % l], SCALE, STRIDES[l],
                                                                                 out=o["offset_u%d" % l])))
                seq.append(("deform_u%d" % l, lambda l=l, pk=pk: ops.DeformableConvolution(
                    t["c2_%d" % l], o["offset_u%d" % l], t["wu_%d" % l], t["bu_%d" % l], kernel=(3, 3), stride=(1, 1),
                    dilate=(1, 1), pad=(1, 1), num_filter=CHANNELS[l], out=o["deform_u%d" % l], packed=pk)))
            else:
                seq.append(("deform_u%d" % l, lambda l=l, pk=pk: ops.deformable_matching(
                    t["c2_%d" % l], t["flow_%d" % l], SCALE, STRIDES[l], t["wu_%d" % l], t["bu_%d" % l], leaky=True,
                    out=o["deform_u%d" % l], packed=pk)))
            seq.append(("corr_u%d" % l, lambda l=l: ops.Correlation(t["c1_%d" % l], o["deform_u%d" % l], 1, md, 1, 1, md,
                                                                    True, out=o["corr_u%d" % l], activation=act)))
            seq.append(("corr_v%d" % l, lambda l=l: ops.Correlation(t["c3_%d" % l], t["c4_%d" % l], 1, md, 1, 1, md, True,
                                                                    out=o["corr_v%d" % l], activation=act)))
        return seq

    def _backward_calls(self):
        """Backward of the S pass's correlations and deformable convs, finest level first (reverse of the forward)."""
        ops, t, o = self.ops, self.t, self.o
        seq = []
        for l in (2, 3, 4, 5):
            seq.append(("corr_bwd%d" % l, lambda l=l: ops.Correlation_backward(
                t["gcorr_%d" % l], t["c1_%d" % l], o["deform%d" % l], 1, MD, 1, 1, MD, True,
                g1=o["g_c1_%d" % l], g2=o["g_warp_%d" % l])))
            if self.mode == "dropin":
                seq.append(("deform_bwd%d" % l, lambda l=l: ops.DeformableConvolution_backward(
                    o["g_warp_%d" % l], t["c2_%d" % l], o["offset%d" % l], t["w_%d" % l], kernel=(3, 3), stride=(1, 1),
                    dilate=(1, 1), pad=(1, 1),
                    out=(o["g_c2_%d" % l], o["g_offset_%d" % l], o["gw_%d" % l], o["gb_%d" % l]))))
            else:   # the fused call's own backward: d/dflow instead of the 18 offset planes
                seq.append(("deform_bwd%d" % l, lambda l=l: ops.deformable_convolution_shared_backward(
                    o["g_warp_%d" % l], t["c2_%d" % l], t["flow_%d" % l], SCALE, STRIDES[l], t["w_%d" % l],
                    out=(o["g_c2_%d" % l], o["g_flow_%d" % l], o["gw_%d" % l], o["gb_%d" % l]))))
        seq.append(("corr_bwd6", lambda: ops.Correlation_backward(
            t["gcorr_6"], t["c1_6"], t["c2_6"], 1, MD, 1, 1, MD, True, g1=o["g_c1_6"], g2=o["g_warp_6"])))
        return seq

    def _enqueue(self):
        for _, fn in self.calls():
            fn()

    def run_eager(self):
        with self.bufs.launching():
            self._enqueue()
        self.bufs.synchronize()
        return self.outputs()

    def capture(self):
        """Capture one pass into a hipGraph (after one eager pass so the workspace exists)."""
        self.run_eager()
        s = self.stream.cuda_stream
        with self.torch.cuda.stream(self.stream):
            _lib.check(self.lib.graph_begin_capture(s))
            try:
                self._enqueue()
            finally:
                g = ctypes.c_void_p()
                rc = self.lib.graph_end_capture(s, ctypes.byref(g))
            _lib.check(rc)
        self.graph = g
        return self

    def replay(self):
        """Enqueue one pass (graph replay if captured) on the workload's stream; no sync."""
        if self.graph is not None:
            _lib.check(self.lib.graph_launch(self.graph, self.stream.cuda_stream))
        else:
            with self.bufs.launching():
                self._enqueue()

    def step(self, dist=None, global_batch=None):
        """One step of the pass: replay(), then the training pass's exchange().  No host synchronisation."""
        self.replay()
        self.exchange(dist, global_batch)

    def exchange(self, dist=None, global_batch=None):
        """The training pass's only collective: all-reduce(sum) of the flat gradient bucket over the ranks (RCCL; enqueued
        behind the pass on the workload's stream, the next replay waits for it) and trainer.step's 1/batch
        (pipeline.py:114).  Nothing to do for the forward passes or without a process group."""
        if self.kind == "train" and dist is not None:
            from .dist import allreduce_bucket
            with self.bufs.launching():
                allreduce_bucket(self.bufs.as_collective_tensor(self.grad_bucket), dist, batch_size=global_batch)

    def synchronize(self):
        self.bufs.synchronize()

    def outputs(self):
        return [self.o[k] for k in self.output_names()]

    def output_names(self):
        return output_names(self.kind, self.mode)

    def checksum(self):
        """[sum |out|, element count] over all outputs -- the 2-float record ranks all-reduce."""
        if self.torch is None:   # numpy buffers (CPU dry run of the call lists)
            import torch
            outs = [np.asarray(x, np.float64) for x in self.outputs()]
            return torch.tensor([sum(float(np.abs(x).sum()) for x in outs), float(sum(x.size for x in outs))],
                                dtype=torch.float64)
        tot = self.torch.zeros(2, device=self.device, dtype=self.torch.float64)
        for x in self.outputs():
            tot[0] += x.abs().sum(dtype=self.torch.float64)
            tot[1] += x.numel()
        return tot

    def __del__(self):
        try:
            if self.graph is not None:
                self.lib.graph_destroy(self.graph)
        except Exception:
            pass
