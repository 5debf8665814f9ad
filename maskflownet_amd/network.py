"""MaskFlownet-S and the full MaskFlownet forward, end to end on libmfn_hip.so (inference).

The caller of the hot path: /root/reference/network/MaskFlownet.py:66-315 (`MaskFlownet_S`).  Every layer with
arithmetic in it is one of this package's HIP operators -- Convolution / Deconvolution, Correlation,
the fused deformable matching step, Upsample, warp -- and torch supplies device memory plus the element-wise adds /
scales between them.  Nothing here is a port of the reference's graph code: the pyramid runs both images as one batch,
every densely connected decoder stage writes its output channels straight into the level's concat buffer and the next
stage reads the buffer's channel suffix in place (x = concat(conv(x), x), MaskFlownet.py:219-223, without a concat copy),
the cost volume, the up-sampled features and the skip features land in that buffer directly, and one forward is a fixed,
allocation-free launch sequence that can be captured into a hipGraph (capture() / replay()).

`MaskFlownet` (the full model, MaskFlownet.py:318-545) runs the S head and then the cascade in the same launch sequence:
a second pyramid over (image 1 | zeros) and (warped image 2 | occlusion mask - 0.5), per level a deformable warp by the
cascade's flow, two md = 2 cost volumes and the decoder; its parameters are keyed 'MaskFlownet_S.<block>' for the head
and '<block>' for the cascade, as in the reference's checkpoints.

Parameters use the reference's block names ('conv1a', 'conv6_0', 'pred_flow5', 'upfeat4', 'deform3', 'conv2f',
'dc_conv7', ... with '.weight' / '.bias'); a Gluon checkpoint's structural keys carry an extra sequence index
('conv1a.0.weight'), see `from_reference_keys`.
"""
import ctypes

import numpy as np

from . import _lib
from .ops import default_ops

SCALE = 20.0                                             # MaskFlownet.py:69
MD = 4                                                   # :70
STRIDES = {6: 64, 5: 32, 4: 16, 3: 8, 2: 4}              # :71
PYRAMID = {1: 16, 2: 32, 3: 64, 4: 96, 5: 128, 6: 196}   # :79-96
DECODER = (128, 128, 96, 64, 32)                         # conv{l}_0..4, :101-129
UPFEAT = 16                                              # :73
CONTEXT = ((128, 1), (128, 2), (128, 4), (96, 8), (64, 16), (32, 1))   # dc_conv1..6 (channels, dilation), :131-136
DEC_SUM = sum(DECODER)                                   # 448 channels a decoder prepends to its input


MD_CASCADE = 2                                           # MaskFlownet.py:322
HEAD = "MaskFlownet_S."                                  # the head's structural key inside the full model, :328


def from_reference_keys(params):
    """{'conv1a.0.weight': ...} (Gluon save_parameters, network/pipeline.py:52-54) -> {'conv1a.weight': ...};
    'MaskFlownet_S.conv1a.0.weight' -> 'MaskFlownet_S.conv1a.weight'."""
    out = {}
    for k, v in params.items():
        parts = k.split(".")
        if len(parts) >= 3 and parts[-2].isdigit():
            k = ".".join(parts[:-2] + parts[-1:])
        out[k] = v
    return out


def layer_shapes():
    """[(block name, weight shape, bias shape)] of MaskFlownet_S's 71 parametrised layers (MaskFlownet.py:79-163):
    10 514 256 parameters.  Conv2D weights are (out, in, 3, 3); Conv2DTranspose (upfeat*) weights are (in, out, 4, 4)."""
    out = []
    cin = 3
    for l in range(1, 7):
        for k in "abc":
            out.append(("conv%d%s" % (l, k), (PYRAMID[l], cin, 3, 3), (PYRAMID[l],)))
            cin = PYRAMID[l]
    for l in (6, 5, 4, 3, 2):
        c = 81 if l == 6 else 81 + PYRAMID[l] + UPFEAT + 2
        if l < 6:
            out.append(("conv%df" % l, (PYRAMID[l], UPFEAT, 3, 3), (PYRAMID[l],)))
            out.append(("deform%d" % l, (PYRAMID[l], PYRAMID[l], 3, 3), (PYRAMID[l],)))
        for k, ch in enumerate(DECODER):
            out.append(("conv%d_%d" % (l, k), (ch, c, 3, 3), (ch,)))
            c += ch
        out.append(("pred_flow%d" % l, (2, c, 3, 3), (2,)))
        if l > 2:
            out.append(("pred_mask%d" % l, (1, c, 3, 3), (1,)))
            out.append(("upfeat%d" % (l - 1), (c, UPFEAT, 4, 4), (UPFEAT,)))
        else:
            for i, (ch, _) in enumerate(CONTEXT):
                out.append(("dc_conv%d" % (i + 1), (ch, c, 3, 3), (ch,)))
                c = ch
            out.append(("dc_conv7", (2, c, 3, 3), (2,)))
    return out


def cascade_input_channels(l):
    """Channels of the cascade decoder's input at level l: level 6 concat(corr_u, corr_v, flow) (MaskFlownet.py:466);
    below, concat(c1, upfeat, corr_u, corr_v, flow, head flow) (:480)."""
    nd = (2 * MD_CASCADE + 1) ** 2
    return 2 * nd + 2 if l == 6 else PYRAMID[l] + UPFEAT + 2 * nd + 4


def layer_shapes_full():
    """MaskFlownet's parametrised layers (MaskFlownet.py:328-407): the head's 71 under 'MaskFlownet_S.', then the
    cascade's 18 pyramid convolutions over 4-channel inputs, 5 deformable convolutions (level 6 included), 25 decoder
    convolutions, 5 flow heads, 4 feature deconvolutions and the 7-layer context network."""
    out = [(HEAD + n, w, b) for n, w, b in layer_shapes()]
    cin = 4
    for l in range(1, 7):
        for k in "xyz":
            out.append(("conv%d%s" % (l, k), (PYRAMID[l], cin, 3, 3), (PYRAMID[l],)))
            cin = PYRAMID[l]
    for l in (6, 5, 4, 3, 2):
        c = cascade_input_channels(l)
        out.append(("deform%d" % l, (PYRAMID[l], PYRAMID[l], 3, 3), (PYRAMID[l],)))
        for k, ch in enumerate(DECODER):
            out.append(("conv%d_%d" % (l, k), (ch, c, 3, 3), (ch,)))
            c += ch
        out.append(("pred_flow%d" % l, (2, c, 3, 3), (2,)))
        if l > 2:
            out.append(("upfeat%d" % (l - 1), (c, UPFEAT, 4, 4), (UPFEAT,)))
        else:
            for i, (ch, _) in enumerate(CONTEXT):
                out.append(("dc_conv%d" % (i + 1), (ch, c, 3, 3), (ch,)))
                c = ch
            out.append(("dc_conv7", (2, c, 3, 3), (2,)))
    return out


def random_params(seed=0, slope=0.1, full=False):
    """Seeded MSRAPrelu(factor_type='avg', slope=0.1) weights and zero biases (network/pipeline.py:26) for every layer."""
    rng = np.random.default_rng(seed)
    params = {}
    for name, ws, bs in (layer_shapes_full() if full else layer_shapes()):
        hw = ws[2] * ws[3]
        std = np.sqrt(2.0 / (1.0 + slope ** 2) / ((ws[0] * hw + ws[1] * hw) / 2.0))
        params[name + ".weight"] = (rng.standard_normal(ws) * std).astype(np.float32)
        params[name + ".bias"] = np.zeros(bs, np.float32)
    return params


class MaskFlownetS:
    def __init__(self, params, batch, H, W, device="cuda:0"):
        import torch
        if H % 64 or W % 64:
            raise ValueError("MaskFlownetS: H and W must be multiples of 64 (the pipeline resizes to that, pipeline.py:139-147)")
        self.torch, self.ops = torch, default_ops()
        self.dev = torch.device(device)
        self.N, self.H, self.W = int(batch), int(H), int(W)
        self.stream = torch.cuda.Stream(device=self.dev)
        self.graph = None
        self.lib = _lib.lib()
        self._packed = {}
        self._flops = {}
        self.b = {}     # named, preallocated device buffers
        N = self.N
        with torch.cuda.stream(self.stream):
            self.P = {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).to(self.dev) for k, v in params.items()}
            # pred_flow_l and pred_mask_l read the same decoder output: one convolution with three filters
            for l in (6, 5, 4, 3):
                self.P["heads%d.weight" % l] = torch.cat([self.P["pred_flow%d.weight" % l], self.P["pred_mask%d.weight" % l]]).contiguous()
                self.P["heads%d.bias" % l] = torch.cat([self.P["pred_flow%d.bias" % l], self.P["pred_mask%d.bias" % l]]).contiguous()
            e = lambda *shape: torch.empty(*shape, device=self.dev)
            self.b["im"] = e(2 * N, 3, H, W)
            h, w = H, W
            for l in range(1, 7):
                h, w = h // 2, w // 2
                for k in "abc":
                    self.b["c%d%s" % (l, k)] = e(2 * N, PYRAMID[l], h, w)
            for l in (6, 5, 4, 3, 2):
                h, w = H // STRIDES[l], W // STRIDES[l]
                cin0 = 81 if l == 6 else 81 + PYRAMID[l] + UPFEAT + 2
                self.b["x%d" % l] = e(N, DEC_SUM + cin0, h, w)
                self.b["flow%d" % l] = e(N, 2, h, w)
                self.b["dflow%d" % l] = e(N, 3 if l > 2 else 2, h, w)   # flow increment (2) [+ mask (1)]
                if l > 2:
                    self.b["mask%d" % l] = e(N, 1, h, w)
                if l < 6:
                    self.b["flow_up%d" % l] = e(N, 2, h, w)
                    self.b["mask_up%d" % l] = e(N, 1, h, w)
                    self.b["trade%d" % l] = e(N, PYRAMID[l], h, w)
                    self.b["warp%d" % l] = e(N, PYRAMID[l], h, w)
            h, w = H // 4, W // 4
            for i, (ch, _) in enumerate(CONTEXT):
                self.b["dc%d" % (i + 1)] = e(N, ch, h, w)
            self.b["dc7"] = e(N, 2, h, w)
            self.b["pred2"] = e(N, 2, h, w)
            self.b["flow_full"] = e(N, 2, H, W)
            self.b["flow_up0"] = e(N, 2, H, W)
            self.b["warped"] = e(N, 3, H, W)
            self.b["occlusion"] = e(N, 1, h, w)
        self.stream.synchronize()
        torch.cuda.synchronize(self.dev)

    # ---- layers ------------------------------------------------------------------------------------------------------
    def _conv(self, name, x, out, stride=1, dilation=1, act=True, transposed=False):
        w, b = self.P[name + ".weight"], self.P[name + ".bias"]
        key = (name, tuple(x.shape))
        pk = self._packed.get(key)
        if name not in self._flops:   # useful multiply-adds: a 4x4 / stride-2 transposed conv touches 4 taps per output
            self._flops[name] = 2.0 * out.shape[0] * out.shape[2] * out.shape[3] * x.shape[1] * out.shape[1] * (4 if transposed else 9)
        if transposed:
            if pk is None:
                pk = self._packed[key] = self.ops.pack_conv_weights(w, tuple(x.shape), kernel=(4, 4), stride=(2, 2), pad=(1, 1), transposed=True)
            return self.ops.Deconvolution(x, w, b, kernel=(4, 4), stride=(2, 2), pad=(1, 1), out=out,
                                          activation="leaky" if act else None, packed=pk)
        if pk is None:
            pk = self._packed[key] = self.ops.pack_conv_weights(w, tuple(x.shape), kernel=(3, 3), stride=(stride, stride),
                                                                dilate=(dilation, dilation), pad=(dilation, dilation))
        return self.ops.Convolution(x, w, b, kernel=(3, 3), stride=(stride, stride), dilate=(dilation, dilation),
                                    pad=(dilation, dilation), out=out, activation="leaky" if act else None, packed=pk)

    def _forward(self):
        t, b, ops, N = self.torch, self.b, self.ops, self.N
        x = b["im"]
        for l in range(1, 7):                       # both images as one batch: c1* = [:N], c2* = [N:]
            for k, s in (("a", 2), ("b", 1), ("c", 1)):
                x = self._conv("conv%d%s" % (l, k), x, b["c%d%s" % (l, k)], stride=s)
        for l in (6, 5, 4, 3, 2):
            C = PYRAMID[l]
            c1, c2 = b["c%dc" % l][:N], b["c%dc" % l][N:]
            xb = b["x%d" % l]
            if l == 6:
                ops.Correlation(c1, c2, 1, MD, 1, 1, MD, True, out=xb[:, DEC_SUM:DEC_SUM + 81], activation="leaky")
            else:
                ops.Upsample(b["flow%d" % (l + 1)], 2, out=b["flow_up%d" % l])
                ops.Upsample(b["mask%d" % (l + 1)], 2, out=b["mask_up%d" % l])
                feat = xb[:, DEC_SUM + 81 + C:DEC_SUM + 81 + C + UPFEAT]        # written by the level above (upfeat)
                self._conv("conv%df" % l, feat, b["trade%d" % l], act=False)
                # warp = LeakyReLU(deform(c2, repeat9(flow*scale/stride)) * sigmoid(mask) + conv_f(feat)): one launch
                ops.deformable_matching(c2, b["flow_up%d" % l], SCALE, float(STRIDES[l]), self.P["deform%d.weight" % l],
                                        self.P["deform%d.bias" % l], mask=b["mask_up%d" % l], tradeoff=b["trade%d" % l],
                                        leaky=True, out=b["warp%d" % l], packed=self._dpack(l, c2))
                ops.Correlation(c1, b["warp%d" % l], 1, MD, 1, 1, MD, True, out=xb[:, DEC_SUM:DEC_SUM + 81], activation="leaky")
                xb[:, DEC_SUM + 81:DEC_SUM + 81 + C].copy_(c1)
                xb[:, DEC_SUM + 81 + C + UPFEAT:].copy_(b["flow_up%d" % l])
            off = DEC_SUM
            for k, ch in enumerate(DECODER):        # x = concat(conv(x), x): outputs are prepended in place
                self._conv("conv%d_%d" % (l, k), xb[:, off:], xb[:, off - ch:off])
                off -= ch
            self._conv(("heads%d" if l > 2 else "pred_flow%d") % l, xb, b["dflow%d" % l], act=False)
            if l == 6:
                b["flow6"].copy_(b["dflow6"][:, :2])
            else:
                t.add(b["flow_up%d" % l], b["dflow%d" % l][:, :2], out=b["flow%d" % l])
            if l > 2:
                b["mask%d" % l].copy_(b["dflow%d" % l][:, 2:3])
                nxt, Cn = b["x%d" % (l - 1)], PYRAMID[l - 1]
                self._conv("upfeat%d" % (l - 1), xb, nxt[:, DEC_SUM + 81 + Cn:DEC_SUM + 81 + Cn + UPFEAT], transposed=True)
        y = b["x2"]
        for i, (ch, dil) in enumerate(CONTEXT):
            y = self._conv("dc_conv%d" % (i + 1), y, b["dc%d" % (i + 1)], dilation=dil)
        self._conv("dc_conv7", y, b["dc7"], act=False)
        b["flow2"].add_(b["dc7"])
        t.mul(b["flow2"], SCALE, out=b["pred2"])
        ops.Upsample(b["pred2"], 4, out=b["flow_full"])                    # pipeline.py:136
        ops.Upsample(b["flow2"], 4, out=b["flow_up0"])                     # MaskFlownet.py:311
        b["flow_up0"].mul_(SCALE)
        ops.warp(b["im"][N:], b["flow_up0"], clip_grid=False, out=b["warped"])
        t.sigmoid(b["mask_up2"], out=b["occlusion"])                       # :309

    def _dpack(self, l, x, scope="", corr_channels=81):
        C, hw = x.shape[1], x.shape[2] * x.shape[3]
        name = "%sdeform%d" % (scope, l)
        self._flops[name] = 2.0 * x.shape[0] * hw * C * C * 9
        self._flops["%scorr%d" % (scope, l)] = 2.0 * x.shape[0] * hw * C * corr_channels
        key = (name, tuple(x.shape))
        pk = self._packed.get(key)
        if pk is None:
            pk = self._packed[key] = self.ops.pack_deform_weights(self.P[name + ".weight"], tuple(x.shape), kernel=(3, 3), pad=(1, 1))
        return pk

    # ---- running it ------------------------------------------------------------------------------------------------------
    def set_input(self, im1, im2):
        t = self.torch
        with t.cuda.stream(self.stream):
            self.b["im"][:self.N].copy_(t.as_tensor(im1).to(self.dev, non_blocking=True))
            self.b["im"][self.N:].copy_(t.as_tensor(im2).to(self.dev, non_blocking=True))

    def run_eager(self):
        with self.torch.cuda.stream(self.stream):
            self._forward()
        return self

    def capture(self):
        """One eager forward (packs the weights, sizes the workspaces), then the same launch sequence captured into a
        hipGraph: replay() costs one graph launch."""
        self.run_eager()
        self.stream.synchronize()
        s = self.stream.cuda_stream
        with self.torch.cuda.stream(self.stream):
            _lib.check(self.lib.graph_begin_capture(s))
            try:
                self._forward()
            finally:
                g = ctypes.c_void_p()
                rc = self.lib.graph_end_capture(s, ctypes.byref(g))
            _lib.check(rc)
        self.graph = g
        return self

    def replay(self):
        if self.graph is not None:
            _lib.check(self.lib.graph_launch(self.graph, self.stream.cuda_stream))
        else:
            self.run_eager()

    def synchronize(self):
        self.stream.synchronize()

    def __call__(self, im1, im2):
        """-> dict(flow_full (N,2,H,W) = Upsample(4)(predictions[-1]), predictions [5] (flow * scale, levels 6..2),
        occlusion (N,1,H/4,W/4), warped (N,3,H,W)) as in network_ref / MaskFlownet_S.hybrid_forward + pipeline.do_batch."""
        self.set_input(im1, im2)
        self.replay()
        self.synchronize()
        b = self.b
        return {"flow_full": b["flow_full"], "predictions": [b["flow%d" % l] * SCALE for l in (6, 5, 4, 3, 2)],
                "occlusion": b["occlusion"], "warped": b["warped"]}

    def flops(self):
        """GEMM flops of one forward (every convolution / deconvolution / deformable conv / cost volume), known after
        the first run."""
        return float(sum(self._flops.values()))

    def __del__(self):
        try:
            if self.graph is not None:
                self.lib.graph_destroy(self.graph)
        except Exception:
            pass


CASCADE = "cascade/"      # scope of the cascade's parameters and buffers inside the object (the head keeps the bare names)


class MaskFlownet(MaskFlownetS):
    """The full model (MaskFlownet.hybrid_forward, /root/reference/network/MaskFlownet.py:436-545): `params` holds the head
    under 'MaskFlownet_S.<block>.weight|bias' and the cascade under '<block>.weight|bias'.  One forward = the head's launch
    sequence followed by the cascade's, on one stream, capturable as one hipGraph."""

    def __init__(self, params, batch, H, W, device="cuda:0"):
        head = {k[len(HEAD):]: v for k, v in params.items() if k.startswith(HEAD)}
        if not head:
            raise ValueError("MaskFlownet: no 'MaskFlownet_S.*' parameters (is this an S checkpoint? use MaskFlownetS)")
        super().__init__(head, batch, H, W, device)
        torch, N = self.torch, self.N
        with torch.cuda.stream(self.stream):
            for k, v in params.items():
                if not k.startswith(HEAD):
                    self.P[CASCADE + k] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).to(self.dev)
            e = lambda *shape: torch.empty(*shape, device=self.dev)
            b = self.b
            b["im4"] = torch.zeros(2 * N, 4, H, W, device=self.dev)     # [:N] = (image 1 | 0), [N:] = (warped image 2 | mask0)
            b["mask0"] = e(N, 1, H, W)
            h, w = H, W
            for l in range(1, 7):
                h, w = h // 2, w // 2
                for k in "xyz":
                    b["d%d%s" % (l, k)] = e(2 * N, PYRAMID[l], h, w)
            for l in (6, 5, 4, 3, 2):
                h, w = H // STRIDES[l], W // STRIDES[l]
                b["y%d" % l] = e(N, DEC_SUM + cascade_input_channels(l), h, w)
                b["gflow%d" % l] = e(N, 2, h, w)
                b["gdflow%d" % l] = e(N, 2, h, w)
                b["gwarp%d" % l] = e(N, PYRAMID[l], h, w)
                if l < 6:
                    b["gflow_up%d" % l] = e(N, 2, h, w)
            h, w = H // 4, W // 4
            for i, (ch, _) in enumerate(CONTEXT):
                b["gdc%d" % (i + 1)] = e(N, ch, h, w)
            b["gdc7"] = e(N, 2, h, w)
            b["gpred2"] = e(N, 2, h, w)
            b["gflow_full"] = e(N, 2, H, W)
        self.stream.synchronize()
        torch.cuda.synchronize(self.dev)

    def _forward(self):
        super()._forward()
        t, b, ops, N = self.torch, self.b, self.ops, self.N
        nd = (2 * MD_CASCADE + 1) ** 2
        # c30 = (image 1 | zeros), c40 = (warp(image 2, Upsample(4)(flow2) * scale) | sigmoid(Upsample(4)(mask2)) - 0.5), :309-314
        im4 = b["im4"]
        im4[:N, :3].copy_(b["im"][:N])
        im4[N:, :3].copy_(b["warped"])
        ops.Upsample(b["mask_up2"], 4, out=b["mask0"])
        b["mask0"].sigmoid_().sub_(0.5)
        im4[N:, 3:].copy_(b["mask0"])
        x = im4
        for l in range(1, 7):                       # c3* = [:N], c4* = [N:]
            for k, s in (("x", 2), ("y", 1), ("z", 1)):
                x = self._conv(CASCADE + "conv%d%s" % (l, k), x, b["d%d%s" % (l, k)], stride=s)
        for l in (6, 5, 4, 3, 2):
            C = PYRAMID[l]
            c1 = b["c%dc" % l][:N]
            c2 = c1 if l in (2, 3) else b["c%dc" % l][N:]     # c2s = [c21, c12, c13, c24, c25, c26] (:307)
            c3, c4 = b["d%dz" % l][:N], b["d%dz" % l][N:]
            yb = b["y%d" % l]
            if l == 6:
                flow_in, o = b["flow6"], DEC_SUM              # the head's coarsest flow starts the cascade (:457)
            else:
                flow_in, o = b["gflow_up%d" % l], DEC_SUM + C + UPFEAT
                ops.Upsample(b["gflow%d" % (l + 1)], 2, out=flow_in)
                yb[:, DEC_SUM:DEC_SUM + C].copy_(c1)
                yb[:, o + 2 * nd + 2:].copy_(b["flow%d" % l])  # the head's flow at this level
            ops.deformable_matching(c2, flow_in, SCALE, float(STRIDES[l]), self.P[CASCADE + "deform%d.weight" % l],
                                    self.P[CASCADE + "deform%d.bias" % l], leaky=True, out=b["gwarp%d" % l],
                                    packed=self._dpack(l, c2, CASCADE, 2 * nd))
            ops.Correlation(c1, b["gwarp%d" % l], 1, MD_CASCADE, 1, 1, MD_CASCADE, True, out=yb[:, o:o + nd], activation="leaky")
            ops.Correlation(c3, c4, 1, MD_CASCADE, 1, 1, MD_CASCADE, True, out=yb[:, o + nd:o + 2 * nd], activation="leaky")
            yb[:, o + 2 * nd:o + 2 * nd + 2].copy_(flow_in)
            off = DEC_SUM
            for k, ch in enumerate(DECODER):
                self._conv(CASCADE + "conv%d_%d" % (l, k), yb[:, off:], yb[:, off - ch:off])
                off -= ch
            self._conv(CASCADE + "pred_flow%d" % l, yb, b["gdflow%d" % l], act=False)
            t.add(flow_in, b["gdflow%d" % l], out=b["gflow%d" % l])
            if l > 2:
                nxt, Cn = b["y%d" % (l - 1)], PYRAMID[l - 1]
                self._conv(CASCADE + "upfeat%d" % (l - 1), yb, nxt[:, DEC_SUM + Cn:DEC_SUM + Cn + UPFEAT], transposed=True)
        y = b["y2"]
        for i, (ch, dil) in enumerate(CONTEXT):
            y = self._conv(CASCADE + "dc_conv%d" % (i + 1), y, b["gdc%d" % (i + 1)], dilation=dil)
        self._conv(CASCADE + "dc_conv7", y, b["gdc7"], act=False)
        b["gflow2"].add_(b["gdc7"])
        t.mul(b["gflow2"], SCALE, out=b["gpred2"])
        ops.Upsample(b["gpred2"], 4, out=b["gflow_full"])                  # pipeline.py:136

    def __call__(self, im1, im2):
        """-> dict(flow_full, predictions [5], visual = flow2[:, :1] (`visuals`, :543), head = the S head's outputs)."""
        head = super().__call__(im1, im2)
        b = self.b
        return {"flow_full": b["gflow_full"], "predictions": [b["gflow%d" % l] * SCALE for l in (6, 5, 4, 3, 2)],
                "visual": b["gflow2"][:, :1], "head": head}
