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

`mode="dropin"`  : offsets are materialised (a6) and go through DeformableConvolution's own
                   (N,18,H,W) signature -- exactly the reference's operator boundary.
`mode="fused"`   : deformable_convolution_shared, the offset tensor never exists (f-1).
Launches go to a private stream; capture()/replay() wrap the ~14 launches in one hipGraph.
"""
import ctypes

import numpy as np

from . import _lib
from .ops import default_ops

CHANNELS = {6: 196, 5: 128, 4: 96, 3: 64, 2: 32}      # MaskFlownet.py:79-96
STRIDES = {6: 64, 5: 32, 4: 16, 3: 8, 2: 4}           # MaskFlownet.py:71
SCALE = 20.0                                          # MaskFlownet.py:69 (flow_multiplier = 1)
MD = 4                                                # MaskFlownet.py:70

CONFIGS = {
    # name: (batch per GPU, H, W)  -- BASELINE.json configs[1], configs[2]
    "cfg2": (8, 384, 512),
    "cfg3": (4, 448, 1024),
    "tiny": (2, 64, 128),
}


def level_shapes(N, H, W):
    return {l: (N, CHANNELS[l], H // STRIDES[l], W // STRIDES[l]) for l in (6, 5, 4, 3, 2)}


def algorithmic_bytes(N, H, W, mode="dropin"):
    """SURVEY.md 8(d) per-op byte counts summed over one pass (fp32)."""
    out = {}
    for l, (n, c, h, w) in level_shapes(N, H, W).items():
        out["corr%d" % l] = 4 * n * h * w * (2 * c + (2 * MD + 1) ** 2)
        if l != 6:
            out["deform%d" % l] = 4 * (n * h * w * (c + 18 + c) + 9 * c * c + c)
            if mode == "dropin":
                out["offsets%d" % l] = 4 * n * h * w * (2 + 18)
    out["warp"] = 4 * N * H * W * (2 * 3 + 2)
    return out


def algorithmic_flops(N, H, W):
    out = {}
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


def synth_inputs(N, H, W, seed=20260925):
    """Seeded synthetic tensors (numpy) for every level -- SURVEY.md 8(d) 'op level'."""
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
            fl = rng.standard_normal((n, 2, ch8, cw8)) * 2.0 + rng.uniform(-3, 3, (n, 2, 1, 1))
            for _ in range(3):
                fl = upsample2(fl)
            fl = fl[:, :, :h, :w].astype(np.float32)
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
    return data


class HotPathWorkload:
    def __init__(self, cfg="cfg2", device="cuda", mode="dropin", seed=20260925, prepack=True):
        """prepack=True: the deformable-conv weights are laid out once here, as layer.DeformableConv2D does
        for a block's constant parameters at inference; False re-packs inside every operator call (what a
        stateless MXNet operator sees).  Outputs are bit-identical either way."""
        import torch
        self.torch = torch
        if isinstance(cfg, str):
            cfg = CONFIGS[cfg]
        self.N, self.H, self.W = cfg
        self.mode = mode
        self.device = torch.device(device)
        self.ops = default_ops()
        self.lib = _lib.lib()
        self.stream = torch.cuda.Stream(device=self.device)
        self.host = synth_inputs(self.N, self.H, self.W, seed)
        self.t = {k: torch.from_numpy(v).to(self.device) for k, v in self.host.items()}
        shp = level_shapes(self.N, self.H, self.W)
        D2 = (2 * MD + 1) ** 2
        self.o = {}
        for l, (n, c, h, w) in shp.items():
            self.o["corr%d" % l] = torch.empty((n, D2, h, w), device=self.device)
            if l != 6:
                self.o["deform%d" % l] = torch.empty((n, c, h, w), device=self.device)
                if mode == "dropin":
                    self.o["offset%d" % l] = torch.empty((n, 18, h, w), device=self.device)
        self.o["warp"] = torch.empty((self.N, 3, self.H, self.W), device=self.device)
        self.graph = None
        self.prepack = bool(prepack)
        self.packed = {}
        if self.prepack:
            for l in (5, 4, 3, 2):
                self.packed[l] = self.ops.pack_deform_weights(self.t["w_%d" % l], shp[l], kernel=(3, 3), pad=(1, 1))
        torch.cuda.synchronize(self.device)

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
        return seq

    def _enqueue(self):
        for _, fn in self.calls():
            fn()

    def run_eager(self):
        with self.torch.cuda.stream(self.stream):
            self._enqueue()
        self.stream.synchronize()
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
            with self.torch.cuda.stream(self.stream):
                self._enqueue()

    def synchronize(self):
        self.stream.synchronize()

    def outputs(self):
        keys = ["corr6"] + [k for l in (5, 4, 3, 2) for k in ("deform%d" % l, "corr%d" % l)] + ["warp"]
        return [self.o[k] for k in keys]

    def output_names(self):
        return ["corr6"] + [k for l in (5, 4, 3, 2) for k in ("deform%d" % l, "corr%d" % l)] + ["warp"]

    def checksum(self):
        """[sum |out|, element count] over all outputs -- the 2-float record ranks all-reduce."""
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
