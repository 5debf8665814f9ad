"""Network-level parity harness (TEST INFRASTRUCTURE ONLY): MaskFlownet-S end to end with the matching hot path
plugged in either from libmfn_hip.so or from the CPU oracle, so that the second half of BASELINE.json's metric --
"EPE delta vs CPU ref", north_star "<= 1e-4 EPE vs reference" -- can be measured.

What is restated here, in torch as plain glue (none of it is product code, none of it is on the hot path):
  * the dataflow of MaskFlownet_S.hybrid_forward (/root/reference/network/MaskFlownet.py:197-315): two 6-level
    pyramids (:79-96), per level 6..2 [Upsample(2) of flow / mask, deformable warp of c2 gated by sigmoid(mask) plus the
    trade-off convolution, cost volume + LeakyReLU, densely connected decoder, flow / mask / feature heads], the
    dilated context network (:138-144, :304-305), predictions = flow * scale (:307), and the full-resolution image
    warp c40 (:311);
  * PipelineFlownet's pre/post-processing around it (network/pipeline.py:26 MSRAPrelu(slope=0.1) init, :85-87
    centralize, :136 Upsample(4) of the last prediction);
  * EPE of pipeline.py / MaskFlownet.py:548-560.
The pretrained weights are missing blobs and there is no dataset here, so weights are seeded MSRAPrelu
draws and the image pair is the synthetic one SURVEY.md 8(d) prescribes.

Only tests/, bench.py's `epe` leg (a checker, like cpu_baseline) and nothing under maskflownet_amd/ import this.
"""
import math

import numpy as np

from . import ref as oracle

SCALE = 20.0                                   # MaskFlownet.py:69, flow_multiplier = 1
MD = 4                                         # :70
STRIDES = {6: 64, 5: 32, 4: 16, 3: 8, 2: 4}    # :71
PYRAMID = {1: 16, 2: 32, 3: 64, 4: 96, 5: 128, 6: 196}   # :79-96
DECODER = (128, 128, 96, 64, 32)               # conv{l}_0 .. conv{l}_4, :101-129
UPFEAT = 16                                    # upfeat_ch default, :73
CONTEXT = ((128, 1), (128, 2), (128, 4), (96, 8), (64, 16), (32, 1))   # dc_conv1..6 (channels, dilation), :131-136
SLOPE = 0.1


class Params:
    """Lazily created, seeded MSRAPrelu(factor_type='avg', slope=0.1) parameters keyed by the reference's layer names.
    The draw of a layer depends on (seed, name, shape) only, never on creation order."""

    def __init__(self, seed=0):
        self.seed = seed
        self.store = {}

    def get(self, name, shape):
        import zlib
        key = name
        if key not in self.store:
            if name.endswith("bias"):
                self.store[key] = np.zeros(shape, np.float32)     # Gluon: bias_initializer='zeros'
            else:
                rng = np.random.default_rng([self.seed, zlib.crc32(name.encode())])
                hw = int(np.prod(shape[2:]))
                fan_in, fan_out = shape[1] * hw, shape[0] * hw   # mx.initializer.Xavier on the stored layout
                std = math.sqrt(2.0 / (1.0 + SLOPE ** 2) / ((fan_in + fan_out) / 2.0))
                self.store[key] = (rng.standard_normal(shape) * std).astype(np.float32)
        a = self.store[key]
        assert tuple(a.shape) == tuple(shape), (name, a.shape, shape)
        return a

    def count(self):
        return sum(int(a.size) for a in self.store.values())


class OracleMatching:
    """Correlation / deformable conv / warp / Upsample from the CPU oracle (numpy in, numpy out)."""
    name = "oracle"

    def corr(self, a, b, md=MD):
        return oracle.correlation(a, b, kernel_size=1, max_displacement=md, stride1=1, stride2=1, pad_size=md)

    def deform(self, x, offset, w, b):
        return oracle.deformable_convolution(x, offset, w, b, kernel=(3, 3), stride=(1, 1), dilate=(1, 1), pad=(1, 1))

    def warp(self, x, flow):
        return oracle.warp(x, flow, clip_grid=False)

    def upsample(self, x, f):
        return oracle.upsample(x, f)


class HipMatching:
    """The same four operators from libmfn_hip.so through the package's operator front-end (torch-ROCm tensors)."""
    name = "hip"

    def __init__(self, device="cuda:0"):
        import torch
        from maskflownet_amd import ops
        self.torch, self.ops, self.device = torch, ops.default_ops(), torch.device(device)

    def _d(self, a):
        t = self.torch
        return a.to(self.device) if isinstance(a, t.Tensor) else t.from_numpy(np.ascontiguousarray(a)).to(self.device)

    def corr(self, a, b, md=MD):
        return self.ops.Correlation(self._d(a), self._d(b), kernel_size=1, max_displacement=md, stride1=1, stride2=1,
                                    pad_size=md, is_multiply=True)

    def deform(self, x, offset, w, b):
        return self.ops.DeformableConvolution(self._d(x), self._d(offset), self._d(w), self._d(b), kernel=(3, 3),
                                              stride=(1, 1), dilate=(1, 1), pad=(1, 1), num_filter=int(w.shape[0]))

    def warp(self, x, flow):
        return self.ops.warp(self._d(x), self._d(flow), clip_grid=False)

    def upsample(self, x, f):
        return self.ops.Upsample(self._d(x), f)


class Net:
    """forward(im1, im2) -> dict(flow_full, predictions[5], occlusion, warped).  `matching` supplies the hot-path
    operators; everything else (convolutions, activations, concat) is torch on `conv_device`."""

    def __init__(self, params, matching, conv_device="cpu", prefix=""):
        import torch
        self.P, self.M, self.torch = params, matching, torch
        self.dev = torch.device(conv_device)
        self.prefix = prefix   # "MaskFlownet_S." when this is the head of the full model (Gluon's structural key of the child block)
        self.calls = []   # (operator, shape) of every hot-path call, in order
        self.srcs = None  # what MaskFlownet_S hands to the cascade (MaskFlownet.py:305-314), kept by forward()

    # ---- plumbing between the glue (torch on conv_device) and the matching operators ---------------------------
    def _to_m(self, t):
        return t if self.M.name == "hip" else t.detach().cpu().numpy()

    def _from_m(self, a):
        t = self.torch
        return (a if isinstance(a, t.Tensor) else t.from_numpy(np.ascontiguousarray(a))).to(self.dev)

    def _w(self, name, shape):
        return self.torch.from_numpy(self.P.get(self.prefix + name, shape)).to(self.dev)

    # ---- layers (nn.Conv2D / Conv2DTranspose / LeakyReLU(0.1) of the reference's conv(), deconv(), predict_*()) --
    def conv(self, name, x, cout, stride=1, dilation=1, act=True):
        F = self.torch.nn.functional
        x = x.contiguous()
        w = self._w(name + ".weight", (cout, x.shape[1], 3, 3))
        b = self._w(name + ".bias", (cout,))
        y = F.conv2d(x, w, b, stride=stride, padding=dilation, dilation=dilation)
        return F.leaky_relu(y, SLOPE) if act else y

    def deconv(self, name, x, cout):
        F = self.torch.nn.functional
        w = self._w(name + ".weight", (x.shape[1], cout, 4, 4))   # Conv2DTranspose stores (in, out, kh, kw)
        b = self._w(name + ".bias", (cout,))
        return F.leaky_relu(F.conv_transpose2d(x, w, b, stride=2, padding=1), SLOPE)

    def pyramid(self, im):
        feats, x = {}, im
        for l in range(1, 7):
            for k, s in (("a", 2), ("b", 1), ("c", 1)):
                x = self.conv("conv%d%s" % (l, k), x, PYRAMID[l], stride=s)
            feats[l] = x
        return feats

    # ---- hot-path operators ------------------------------------------------------------------------------------
    def corr(self, a, b, md=MD):
        self.calls.append(("correlation", tuple(a.shape)) if md == MD else ("correlation_md%d" % md, tuple(a.shape)))
        return self._from_m(self.M.corr(self._to_m(a.contiguous()), self._to_m(b.contiguous()), md))

    def deform(self, l, x, flow_l):
        c = x.shape[1]
        # offset = repeat(expand_dims(flow * scale / stride, 1), 9, 1).reshape((0, -3, -2)): one (dy, dx) for all taps
        off = (flow_l * SCALE / STRIDES[l]).unsqueeze(1).repeat(1, 9, 1, 1, 1).reshape(x.shape[0], 18, *x.shape[2:])
        w = self.P.get(self.prefix + "deform%d.weight" % l, (c, c, 3, 3))
        b = self.P.get(self.prefix + "deform%d.bias" % l, (c,))
        self.calls.append(("deformable_conv", tuple(x.shape)))
        return self._from_m(self.M.deform(self._to_m(x), self._to_m(off.contiguous()), w, b))

    def upsample(self, x, f):
        return self._from_m(self.M.upsample(self._to_m(x.contiguous()), f))

    def forward(self, im1, im2):
        t = self.torch
        F = t.nn.functional
        with t.no_grad():
            im1, im2 = self._from_m(im1), self._from_m(im2)
            c1, c2 = self.pyramid(im1), self.pyramid(im2)
            flows, flow, mask, feat, x = {}, None, None, None, None
            for l in (6, 5, 4, 3, 2):
                if l == 6:
                    x = F.leaky_relu(self.corr(c1[6], c2[6]), SLOPE)
                    flow_up = None
                else:
                    flow_up = self.upsample(flow, 2)
                    mask_up = self.upsample(mask, 2)
                    warp = self.deform(l, c2[l], flow_up)
                    warp = warp * t.sigmoid(mask_up) + self.conv("conv%df" % l, feat, PYRAMID[l], act=False)
                    warp = F.leaky_relu(warp, SLOPE)
                    cost = F.leaky_relu(self.corr(c1[l], warp), SLOPE)
                    x = t.cat([cost, c1[l], feat, flow_up], dim=1)
                for k, ch in enumerate(DECODER):
                    x = t.cat([self.conv("conv%d_%d" % (l, k), x, ch), x], dim=1)
                delta = self.conv("pred_flow%d" % l, x, 2, act=False)
                flow = delta if flow_up is None else flow_up + delta
                if l > 2:
                    mask = self.conv("pred_mask%d" % l, x, 1, act=False)
                    feat = self.deconv("upfeat%d" % (l - 1), x, UPFEAT)
                else:
                    mask = mask_up   # mask2 is the upsampled level-3 mask (MaskFlownet.py:283, :309)
                    y = x
                    for i, (ch, dil) in enumerate(CONTEXT[:4]):
                        y = self.conv("dc_conv%d" % (i + 1), y, ch, dilation=dil)
                    z = y
                    for i, (ch, dil) in enumerate(CONTEXT[4:]):
                        z = self.conv("dc_conv%d" % (i + 5), z, ch, dilation=dil)
                    flow = flow + self.conv("dc_conv7", z, 2, act=False)
                flows[l] = flow
            preds = [flows[l] * SCALE for l in (6, 5, 4, 3, 2)]
            flow_full = self.upsample(preds[-1], 4)                       # pipeline.py:136
            self.calls.append(("warp", tuple(im2.shape)))
            warped = self._from_m(self.M.warp(self._to_m(im2), self._to_m((self.upsample(flows[2], 4) * SCALE).contiguous())))
            # srcs (MaskFlownet.py:305-314).  c2s = [c21, c12, c13, c24, c25, c26]: levels 2 and 3 of the "second image" list
            # are IMAGE-1 features -- the released full-model weights were trained with that aliasing
            mask0 = t.sigmoid(self.upsample(mask, 4)) - 0.5
            self.srcs = {"c1": c1, "c2": {l: (c1[l] if l in (2, 3) else c2[l]) for l in range(1, 7)}, "flows": dict(flows),
                         "c30": t.cat([im1, t.zeros_like(mask0)], dim=1), "c40": t.cat([warped, mask0], dim=1)}
            return {"flow_full": flow_full.cpu().numpy(), "predictions": [p.cpu().numpy() for p in preds],
                    "occlusion": t.sigmoid(mask).cpu().numpy(), "warped": warped.cpu().numpy()}


MD_CASCADE = 2                                 # MaskFlownet.py:322
PYRAMID_X = "xyz"                              # conv{l}x / y / z: the second (4-channel) pyramid, :332-349


class NetFull:
    """The full MaskFlownet (MaskFlownet.hybrid_forward, /root/reference/network/MaskFlownet.py:436-545): the S head, a second
    pyramid over (image 1 | zeros) and (warped image 2 | occlusion mask - 0.5), and per level 6..2 a deformable warp of the
    head's features by the cascade's own flow (LeakyReLU, no gating: :460-461), two md = 2 cost volumes (u: c1 vs the warp,
    v: the two new pyramids), a densely connected decoder and a flow head; the dilated context network at level 2.
    Parameters: the head's under 'MaskFlownet_S.<block>', the cascade's under the reference's block names."""

    def __init__(self, params, matching, conv_device="cpu"):
        self.head = Net(params, matching, conv_device, prefix="MaskFlownet_S.")
        self.body = Net(params, matching, conv_device, prefix="")     # layer helpers with the cascade's parameter names
        self.P, self.M, self.torch, self.dev = params, matching, self.head.torch, self.head.dev
        self.calls = self.body.calls

    def pyramid4(self, x):
        feats = {}
        for l in range(1, 7):
            for k, s_ in zip(PYRAMID_X, (2, 1, 1)):
                x = self.body.conv("conv%d%s" % (l, k), x, PYRAMID[l], stride=s_)
            feats[l] = x
        return feats

    def forward(self, im1, im2):
        t = self.torch
        F = t.nn.functional
        B = self.body
        with t.no_grad():
            head_out = self.head.forward(im1, im2)
            S = self.head.srcs
            self.calls[:0] = self.head.calls
            c1, c2 = S["c1"], S["c2"]
            c3, c4 = self.pyramid4(S["c30"]), self.pyramid4(S["c40"])
            flows, flow, x = {}, None, None
            for l in (6, 5, 4, 3, 2):
                flow = S["flows"][6] if l == 6 else B.upsample(flow, 2)
                self.calls.append(("deform_source", "c1" if c2[l] is c1[l] else "c2"))
                warp_u = F.leaky_relu(B.deform(l, c2[l], flow), SLOPE)
                corr_u = F.leaky_relu(B.corr(c1[l], warp_u, MD_CASCADE), SLOPE)
                corr_v = F.leaky_relu(B.corr(c3[l], c4[l], MD_CASCADE), SLOPE)
                if l == 6:
                    x = t.cat([corr_u, corr_v, flow], dim=1)
                else:
                    feat = B.deconv("upfeat%d" % l, x, UPFEAT)
                    x = t.cat([c1[l], feat, corr_u, corr_v, flow, S["flows"][l]], dim=1)
                for k, ch in enumerate(DECODER):
                    x = t.cat([B.conv("conv%d_%d" % (l, k), x, ch), x], dim=1)
                flow = flow + B.conv("pred_flow%d" % l, x, 2, act=False)
                flows[l] = flow
            y = x
            for i, (ch, dil) in enumerate(CONTEXT[:4]):
                y = B.conv("dc_conv%d" % (i + 1), y, ch, dilation=dil)
            for i, (ch, dil) in enumerate(CONTEXT[4:]):
                y = B.conv("dc_conv%d" % (i + 5), y, ch, dilation=dil)
            flows[2] = flows[2] + B.conv("dc_conv7", y, 2, act=False)
            preds = [flows[l] * SCALE for l in (6, 5, 4, 3, 2)]
            flow_full = B.upsample(preds[-1], 4)                          # pipeline.py:136
            return {"flow_full": flow_full.cpu().numpy(), "predictions": [p.cpu().numpy() for p in preds],
                    "visual": flows[2][:, :1].cpu().numpy(),               # `visuals` (:543), what do_batch up-samples as "occ_mask"
                    "head": head_out}


def synthetic_pair(N=1, H=384, W=512, seed=20260925, shift=(3, -5)):
    """SURVEY.md 8(d) network-level input: image1 = low-pass filtered uint8 noise, image2 = image1 translated by
    (+3, -5) px; /255 (pipeline.py:99) and centralize (pipeline.py:85-87)."""
    from scipy.ndimage import gaussian_filter
    rng = np.random.default_rng(seed)
    ims = []
    for n in range(N):
        raw = rng.integers(0, 256, (3, H + 32, W + 32)).astype(np.float32)
        smooth = np.stack([gaussian_filter(ch, 2.5) for ch in raw])
        smooth = (smooth - smooth.min()) / (smooth.max() - smooth.min()) * 255.0
        ims.append(np.round(smooth).astype(np.uint8))
    big = np.stack(ims)
    dy, dx = shift
    im1 = big[:, :, 16:16 + H, 16:16 + W].astype(np.float32) / 255.0
    im2 = big[:, :, 16 - dy:16 - dy + H, 16 - dx:16 - dx + W].astype(np.float32) / 255.0
    mean = np.concatenate([im1, im2], axis=2).mean(axis=(2, 3)).reshape(N, 3, 1, 1)
    return (im1 - mean).astype(np.float32), (im2 - mean).astype(np.float32)


def epe(a, b):
    """mean over pixels of ||a - b||_2 (EpeLoss, MaskFlownet.py:548-560, eps = 0)."""
    d = np.asarray(a, np.float64) - np.asarray(b, np.float64)
    return float(np.sqrt((d ** 2).sum(axis=1)).mean())


def epe_delta(out_a, out_b):
    """EPE between two runs' final flows, absolute (px) and relative to the mean flow magnitude of run b."""
    mag = epe(out_b["flow_full"], np.zeros_like(out_b["flow_full"]))
    d = epe(out_a["flow_full"], out_b["flow_full"])
    return {"epe_delta_px": d, "mean_flow_px": mag, "epe_delta_rel": d / max(mag, 1e-30)}
