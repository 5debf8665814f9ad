"""The training step of MaskFlownet-S with EVERY layer's forward and backward on libmfn_hip.so.

Caller of the hot path on the training side: /root/reference/network/pipeline.py:89-114 (`train_batch`: forward under
autograd, `MultiscaleEpe` loss, `loss.backward()`, `trainer.step`) over /root/reference/network/MaskFlownet.py:197-315
(`MaskFlownet_S.hybrid_forward`).  MXNet's autograd is torch's here (device memory, tape and the optimizer are plumbing);
what the tape calls is the library:

* convolutions / transposed convolutions with their fused LeakyReLU: `mfn_conv2d_fwd` / `mfn_conv2d_bwd` (`_ConvFn`),
* cost volumes, the deformable matching step, `Upsample`: the autograd functions of `layer.py`
  (`mfn_correlation_fwd/_bwd`, `mfn_deform_conv_shared_fwd/_bwd`, `mfn_upsample_fwd/_bwd`),
* concatenation, the gating `warp * sigmoid(mask) + tradeoff`, the loss arithmetic: torch element-wise glue, as the
  reference's are MXNet's.

`network.MaskFlownetS` is the inference form of the same graph (pre-allocated concat buffers, packed weights, one
hipGraph); the parameter names are shared (`network.random_params`, the reference checkpoint's keys).
"""
import numpy as np
import torch
from torch import nn

from . import layer, ops
from .network import CONTEXT, DECODER, MD, PYRAMID, SCALE, STRIDES, UPFEAT


class _ConvFn(torch.autograd.Function):
    """Convolution / Deconvolution + bias [+ LeakyReLU(0.1)] in one launch; backward through mfn_conv2d_bwd (which undoes the
    fused activation from the saved output)."""

    @staticmethod
    def forward(ctx, x, w, b, stride, dilation, act, transposed):
        o = ops.default_ops()
        x = x.contiguous()
        if transposed:
            y = o.Deconvolution(x, w, b, kernel=(4, 4), stride=(2, 2), pad=(1, 1), activation="leaky" if act else None)
        else:
            y = o.Convolution(x, w, b, kernel=(3, 3), stride=(stride, stride), dilate=(dilation, dilation),
                              pad=(dilation, dilation), activation="leaky" if act else None)
        ctx.save_for_backward(x, w, y)
        ctx.cfg = (stride, dilation, act, transposed)
        return y

    @staticmethod
    def backward(ctx, gout):
        x, w, y = ctx.saved_tensors
        stride, dilation, act, transposed = ctx.cfg
        o = ops.default_ops()
        need = ctx.needs_input_grad
        req = tuple("write" if need[i] else "null" for i in range(3))
        gout = gout.contiguous()
        if transposed:
            gx, gw, gb = o.Deconvolution_backward(gout, x, w, output=y, kernel=(4, 4), stride=(2, 2), pad=(1, 1),
                                                  activation="leaky" if act else None, req=req)
        else:
            gx, gw, gb = o.Convolution_backward(gout, x, w, output=y, kernel=(3, 3), stride=(stride, stride),
                                                dilate=(dilation, dilation), pad=(dilation, dilation),
                                                activation="leaky" if act else None, req=req)
        return gx, gw, gb, None, None, None, None


class LibraryBackend:
    """The differentiable layer set of the network, every member's forward AND backward a libmfn_hip.so call.  (The tests pass
    a pure-torch implementation of the same four members to check the gradients of the composition.)"""

    def __init__(self):
        self._up = {}

    def conv(self, x, w, b, stride=1, dilation=1, act=True, transposed=False):
        return _ConvFn.apply(x, w, b, stride, dilation, act, transposed)

    def correlation(self, a, b, md):     # MaskFlownet.py:193-195 + the LeakyReLU around every call (:217)
        return layer.correlation(a.contiguous(), b.contiguous(), md, leaky=True)

    def deform(self, x, flow, w, b, scale, stride):   # MaskFlownet.py:230: deform(x, repeat9(flow * scale / stride))
        return layer._SharedDeformConvFn.apply(x.contiguous(), flow.contiguous(), w, b, scale, stride,
                                               {"kernel": (3, 3), "dilate": (1, 1), "pad": (1, 1), "num_group": 1}, None)

    def upsample(self, x, factor):
        if factor not in self._up:
            self._up[factor] = layer.Upsample(factor)
        return self._up[factor](x.contiguous())

    def warp(self, x, flow):             # layer.py:14-18 (Reconstruction2D); only ever called on the frozen head's outputs
        return ops.warp(x.contiguous(), flow.contiguous(), clip_grid=False)


def _key(name):
    return name.replace(".", "__")


class MaskFlownetSTrainable(nn.Module):
    """MaskFlownet_S.hybrid_forward (MaskFlownet.py:197-315) as a differentiable module on the library.
    forward(im1, im2) -> ([flow6 .. flow2] x scale, [sigmoid(mask2)]) = the reference's (predictions, occlusion_masks)."""

    def __init__(self, params, backend=None, dtype=torch.float32):
        super().__init__()
        self.P = nn.ParameterDict({_key(k): nn.Parameter(torch.as_tensor(np.ascontiguousarray(v)).to(dtype))
                                   for k, v in params.items()})
        self.B = backend if backend is not None else LibraryBackend()

    def _p(self, name):
        return self.P[_key(name + ".weight")], self.P[_key(name + ".bias")]

    def conv(self, name, x, stride=1, dilation=1, act=True, transposed=False):
        w, b = self._p(name)
        return self.B.conv(x, w, b, stride, dilation, act, transposed)

    def forward(self, im1, im2):
        r = self.run(im1, im2)
        return r["predictions"], [r["occlusion"]]

    def run(self, im1, im2):
        """forward() with what MaskFlownet_S hands to the cascade (`srcs`, MaskFlownet.py:305-314) kept: the pyramid c[l] (both
        images as one batch), the flows per level (network units) and mask2."""
        N = im1.shape[0]
        x = torch.cat([im1, im2], 0)
        c = {}
        for l in range(1, 7):                        # both images as one batch through the shared pyramid
            for k, s in (("a", 2), ("b", 1), ("c", 1)):
                x = self.conv("conv%d%s" % (l, k), x, stride=s)
            c[l] = x
        preds, flows = [], {}
        flow = mask = feat = None
        for l in (6, 5, 4, 3, 2):
            c1, c2 = c[l][:N], c[l][N:]
            if l == 6:
                x = self.B.correlation(c1, c2, MD)
            else:
                flow_up, mask_up = self.B.upsample(flow, 2), self.B.upsample(mask, 2)
                trade = self.conv("conv%df" % l, feat, act=False)
                w, b = self._p("deform%d" % l)
                # MaskFlownet.py:230-233: deform(c2, repeat9(flow * scale / stride)) * sigmoid(mask) + tradeoff, LeakyReLU
                warp = self.B.deform(c2, flow_up, w, b, SCALE, float(STRIDES[l]))
                warp = torch.nn.functional.leaky_relu(warp * torch.sigmoid(mask_up) + trade, 0.1)
                corr = self.B.correlation(c1, warp, MD)
                x = torch.cat([corr, c1, feat, flow_up], 1)
            for k in range(len(DECODER)):            # x = concat(conv(x), x)
                x = torch.cat([self.conv("conv%d_%d" % (l, k), x), x], 1)
            if l > 2:
                d = torch.cat([self.conv("pred_flow%d" % l, x, act=False), self.conv("pred_mask%d" % l, x, act=False)], 1)
            else:
                d = self.conv("pred_flow2", x, act=False)
            flow = d[:, :2].contiguous() if l == 6 else flow_up + d[:, :2]
            if l > 2:
                mask = d[:, 2:3].contiguous()
                feat = self.conv("upfeat%d" % (l - 1), x, transposed=True)
                preds.append(flow)
                flows[l] = flow
        y = x
        for i, (_, dil) in enumerate(CONTEXT):
            y = self.conv("dc_conv%d" % (i + 1), y, dilation=dil)
        flow = flow + self.conv("dc_conv7", y, act=False)
        preds.append(flow)
        flows[2] = flow
        return {"predictions": [f * SCALE for f in preds], "occlusion": torch.sigmoid(mask_up),      # MaskFlownet.py:303-305
                "c": c, "flows": flows, "mask2": mask_up}


class MaskFlownetTrainable(nn.Module):
    """The full MaskFlownet (MaskFlownet.hybrid_forward, MaskFlownet.py:436-545) for its training stage: the S head frozen
    (`fix_head`, :413-415, main.py:139) and run without a tape, the cascade -- second pyramid over (image 1 | 0) and (warped image 2 |
    occlusion mask - 0.5), per level a deformable warp by the cascade's own flow, two md = 2 cost volumes, decoder, flow head, the
    context network -- differentiable on the library.  `params`: the head under 'MaskFlownet_S.<block>', the cascade under '<block>'.
    forward -> ([gflow6 .. gflow2] x scale, [the head's occlusion mask])."""

    def __init__(self, params, backend=None, dtype=torch.float32):
        super().__init__()
        from .network import HEAD
        self.B = backend if backend is not None else LibraryBackend()
        self.head = MaskFlownetSTrainable({k[len(HEAD):]: v for k, v in params.items() if k.startswith(HEAD)}, self.B, dtype)
        for q in self.head.parameters():
            q.requires_grad_(False)
        self.P = nn.ParameterDict({_key(k): nn.Parameter(torch.as_tensor(np.ascontiguousarray(v)).to(dtype))
                                   for k, v in params.items() if not k.startswith(HEAD)})

    def conv(self, name, x, stride=1, dilation=1, act=True, transposed=False):
        return self.B.conv(x, self.P[_key(name + ".weight")], self.P[_key(name + ".bias")], stride, dilation, act, transposed)

    def forward(self, im1, im2):
        from .network import MD_CASCADE
        N = im1.shape[0]
        with torch.no_grad():
            h = self.head.run(im1, im2)
            warped = self.B.warp(im2, self.B.upsample(h["flows"][2], 4) * SCALE)                       # :311
            mask0 = torch.sigmoid(self.B.upsample(h["mask2"], 4)) - 0.5                                 # :309-310
            x = torch.cat([torch.cat([im1, torch.zeros_like(mask0)], 1), torch.cat([warped, mask0], 1)], 0)   # c30, c40 (:312-313)
        d = {}
        for l in range(1, 7):                        # the second pyramid, both 4-channel inputs as one batch
            for k, s in (("x", 2), ("y", 1), ("z", 1)):
                x = self.conv("conv%d%s" % (l, k), x, stride=s)
            d[l] = x
        preds = []
        flow = feat = None
        for l in (6, 5, 4, 3, 2):
            c1 = h["c"][l][:N]
            c2 = c1 if l in (2, 3) else h["c"][l][N:]          # c2s = [c21, c12, c13, c24, c25, c26] (:307)
            c3, c4 = d[l][:N], d[l][N:]
            flow_in = h["flows"][6] if l == 6 else self.B.upsample(flow, 2)                              # :457
            w, b = self.P[_key("deform%d.weight" % l)], self.P[_key("deform%d.bias" % l)]
            warp = torch.nn.functional.leaky_relu(self.B.deform(c2, flow_in, w, b, SCALE, float(STRIDES[l])), 0.1)   # :460-461
            cu = self.B.correlation(c1, warp, MD_CASCADE)
            cv = self.B.correlation(c3, c4, MD_CASCADE)
            x = torch.cat([cu, cv, flow_in], 1) if l == 6 else torch.cat([c1, feat, cu, cv, flow_in, h["flows"][l]], 1)   # :466, :480
            for k in range(len(DECODER)):
                x = torch.cat([self.conv("conv%d_%d" % (l, k), x), x], 1)
            flow = flow_in + self.conv("pred_flow%d" % l, x, act=False)
            if l > 2:
                feat = self.conv("upfeat%d" % (l - 1), x, transposed=True)
                preds.append(flow)
        y = x
        for i, (_, dil) in enumerate(CONTEXT):
            y = self.conv("dc_conv%d" % (i + 1), y, dilation=dil)
        flow = flow + self.conv("dc_conv7", y, act=False)
        preds.append(flow)
        return [f * SCALE for f in preds], [h["occlusion"]]


class MultiscaleEpe(nn.Module):
    """MaskFlownet.py:585-611 with match='upsampling' (pipeline.py:42-44): sum_s w_s * EpeLossWithMask(Upsample(s)(pred_s) ,
    label, mask); EpeLossWithMask (:563-583) = sum(sqrt(sum_c (p - l)^2 + eps) * mask) / sum(mask) per sample.
    `label` in pixels, as the predictions (flow x scale)."""

    def __init__(self, scales=(64, 32, 16, 8, 4), weights=(.005, .01, .02, .08, .32), eps=1e-8, backend=None):
        super().__init__()
        self.scales, self.weights, self.eps = tuple(scales), tuple(weights), float(eps)
        self.B = backend if backend is not None else LibraryBackend()

    def forward(self, label, mask, *preds):
        total = 0.
        for p, w, s in zip(preds, self.weights, self.scales):
            e = torch.sqrt(((self.B.upsample(p, s) - label) ** 2).sum(1) + self.eps) * mask[:, 0]
            total = total + w * e.flatten(1).sum(1) / mask.flatten(1).sum(1)
        return total


class GradientBuckets:
    """The training step's gradient exchange (/root/reference/network/pipeline.py:27 kvstore='device', :95 batch
    shards, :114 `trainer.step(batch_size)`): every parameter gradient of the network (142 tensors, 42.06 MB for MaskFlownet-S)
    lives in one of `n_buckets` flat fp32 buffers -- `p.grad` is a VIEW into its bucket, autograd accumulates in place -- laid
    out in REVERSE registration order, which is roughly the order backward completes them (context network and decoders first,
    the image pyramid last).  A post-accumulate hook per parameter counts its bucket down; the bucket whose last gradient
    arrives is all-reduced (sum) at once with async_op=True, so the collective of the decoders' gradients travels over
    RCCL / xGMI while backward is still in the pyramid.  finish() waits for the handles and applies 1 / global_batch.
    Four buckets of ~10.5 MB: large enough for xGMI's per-link bandwidth, few enough launches, early enough first launch.
    Without a process group (or world 1) only the 1 / global_batch remains."""

    def __init__(self, params, n_buckets=4, dist=None):
        params = [p for p in params if p.requires_grad]
        self.dist = dist if (dist is not None and dist.is_initialized() and dist.get_world_size() > 1) else None
        order = list(reversed(params))
        total = sum(p.numel() for p in order)
        target = (total + n_buckets - 1) // max(1, n_buckets)
        groups, cur, size = [], [], 0
        for p in order:
            cur.append(p)
            size += p.numel()
            if size >= target and len(groups) < n_buckets - 1:
                groups.append(cur)
                cur, size = [], 0
        if cur:
            groups.append(cur)
        self.buckets, self.members, self._of, self._hooks = [], groups, {}, []
        for bi, grp in enumerate(groups):
            flat = torch.zeros(sum(p.numel() for p in grp), dtype=grp[0].dtype, device=grp[0].device)
            off = 0
            for p in grp:
                p.grad = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
                self._of[p] = bi
                self._hooks.append(p.register_post_accumulate_grad_hook(self._arrived))
            self.buckets.append(flat)
        self._left = [len(g) for g in groups]
        self._handles = []
        self.launch_order = []   # bucket indices in the order their all-reduce was launched by the last backward (tests read it)

    def zero(self):
        """Before a step's backward (instead of opt.zero_grad(set_to_none=True), which would drop the views)."""
        for b in self.buckets:
            b.zero_()
        self._left = [len(g) for g in self.members]
        self._handles, self.launch_order = [], []

    def _arrived(self, p):
        bi = self._of[p]
        b = self.buckets[bi]
        if p.grad is None or not (b.data_ptr() <= p.grad.data_ptr() < b.data_ptr() + b.numel() * b.element_size()):
            raise RuntimeError("a parameter's .grad no longer views its gradient bucket (zero the buckets with GradientBuckets.zero())")
        self._left[bi] -= 1
        if self._left[bi] == 0:
            self.launch_order.append(bi)
            if self.dist is not None:
                self._handles.append(self.dist.all_reduce(self.buckets[bi], op=self.dist.ReduceOp.SUM, async_op=True))

    def finish(self, global_batch):
        """Behind backward: every bucket reduced (whatever arrived late is launched now), then trainer.step's 1 / batch_size."""
        for bi, left in enumerate(self._left):
            if left > 0:   # parameters that took no gradient this step (frozen branches): their bucket still has to travel
                self._left[bi] = 0
                self.launch_order.append(bi)
                if self.dist is not None:
                    self._handles.append(self.dist.all_reduce(self.buckets[bi], op=self.dist.ReduceOp.SUM, async_op=True))
        for h in self._handles:
            h.wait()
        self._handles = []
        for b in self.buckets:
            b.mul_(1.0 / float(global_batch))

    def nbytes(self):
        return sum(b.numel() * b.element_size() for b in self.buckets)


def train_step(net, loss_fn, opt, im1, im2, label, mask, buckets=None, global_batch=None):
    """pipeline.py:95-114: forward, loss, `loss.backward()` (the per-sample losses summed), the gradient exchange, and
    `trainer.step(batch_size)` -- the optimizer sees (sum over the GLOBAL batch of the per-sample gradients) / global_batch.
    buckets: a GradientBuckets over net.parameters() (its all-reduces start inside backward); None = one device, the gradients
    scaled in place.  global_batch defaults to this call's batch (one device).  Returns the per-sample loss of the local shard."""
    gb = int(global_batch if global_batch is not None else im1.shape[0])
    if buckets is not None:
        buckets.zero()
    else:
        opt.zero_grad(set_to_none=True)
    preds, _ = net(im1, im2)
    loss = loss_fn(label, mask, *preds)
    loss.sum().backward()
    if buckets is not None:
        buckets.finish(gb)
    else:
        for p in net.parameters():
            if p.grad is not None:
                p.grad.mul_(1.0 / gb)
    opt.step()
    return loss.detach()
