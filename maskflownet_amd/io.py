"""Flow and checkpoint IO next to the hot path -- host side only, numpy.

* Middlebury `.flo`: b'PIEH' (the float 202021.25), int32 width, int32 height, then H*W (u, v) float32 pairs, little
  endian (/root/reference/reader/chairs/flo.py:4-25 hard-codes the 512x384 header; reader/sintel.py:46-73 reads the
  general one).  `.flo` stores (u, v) = (dx, dy); the network uses channel 0 = dy (pipeline.py:105,176,219), so
  `flo_to_network` / `network_to_flo` do the flip and the HWC <-> CHW move.
* MXNet `.params` (`save_parameters`, pipeline.py:52-63): an NDArray list -- uint64 magic 0x112, uint64 reserved,
  uint64 count, count x NDArray, uint64 name count, names as uint64 length + bytes.  An NDArray (V2, magic 0xF993FAC9)
  is int32 storage type (0 = dense), shape as uint32 ndim + int64 dims, context (int32 dev_type, int32 dev_id), int32
  dtype flag, raw little-endian data; V1 (0xF993FAC8) has no storage type; older files start with the uint32 ndim and
  carry uint32 dims.  Restated from MXNet 1.x `src/ndarray/ndarray.cc`; MXNet cannot run here and the reference's
  weights are absent (.MISSING_LARGE_BLOBS), so the reader is pinned by tests/golden/mxnet_v2.params -- assembled byte
  by byte from that layout by tests/golden/make_params_fixture.py, independently of the writer below -- and by round
  trips.
"""
import struct

import numpy as np

FLO_TAG = 202021.25
_LIST_MAGIC = 0x112
_V1, _V2, _V3 = 0xF993FAC8, 0xF993FAC9, 0xF993FACA
_DTYPES = {0: np.float32, 1: np.float64, 2: np.float16, 3: np.uint8, 4: np.int32, 5: np.int8, 6: np.int64}
_FLAGS = {np.dtype(v): k for k, v in _DTYPES.items()}


def read_flo(path):
    """-> (H, W, 2) float32, last axis (u, v) = (dx, dy)."""
    with open(path, "rb") as f:
        head = f.read(12)
        if len(head) != 12:
            raise ValueError("%s: truncated .flo header" % path)
        tag, w, h = struct.unpack("<fii", head)
        if tag != FLO_TAG:
            raise ValueError("%s: bad .flo tag %r (expected 'PIEH' = %r)" % (path, tag, FLO_TAG))
        if w <= 0 or h <= 0 or w > 99999 or h > 99999:
            raise ValueError("%s: implausible .flo size %dx%d" % (path, w, h))
        data = np.fromfile(f, dtype="<f4", count=2 * w * h)
    if data.size != 2 * w * h:
        raise ValueError("%s: truncated .flo payload" % path)
    return data.reshape(h, w, 2).astype(np.float32, copy=False)


def write_flo(path, flow_hw2):
    a = np.ascontiguousarray(flow_hw2, dtype="<f4")
    if a.ndim != 3 or a.shape[2] != 2:
        raise ValueError("write_flo: expected (H, W, 2), got %s" % (a.shape,))
    with open(path, "wb") as f:
        f.write(struct.pack("<fii", FLO_TAG, a.shape[1], a.shape[0]))
        a.tofile(f)


def flo_to_network(flow_hw2):
    """(H, W, 2) (u, v) -> (2, H, W) with channel 0 = dy, as the network and mfn_warp_fwd expect."""
    return np.ascontiguousarray(np.asarray(flow_hw2)[..., ::-1].transpose(2, 0, 1))


def network_to_flo(flow_2hw):
    return np.ascontiguousarray(np.asarray(flow_2hw).transpose(1, 2, 0)[..., ::-1])


def _read_ndarray(buf, pos):
    (magic,) = struct.unpack_from("<I", buf, pos)
    if magic in (_V2, _V3):
        pos += 4
        (stype,) = struct.unpack_from("<i", buf, pos)
        pos += 4
        if stype != 0:
            raise ValueError("sparse NDArray (storage type %d) is not supported" % stype)
        (ndim,) = struct.unpack_from("<I" if magic == _V2 else "<i", buf, pos)
        pos += 4
        dims = struct.unpack_from("<%dq" % ndim, buf, pos)
        pos += 8 * ndim
    elif magic == _V1:
        pos += 4
        (ndim,) = struct.unpack_from("<I", buf, pos)
        pos += 4
        dims = struct.unpack_from("<%dq" % ndim, buf, pos)
        pos += 8 * ndim
    else:  # legacy: the first word is ndim, dims are uint32
        ndim = magic
        if ndim > 16:
            raise ValueError("not an MXNet NDArray (first word 0x%x)" % magic)
        pos += 4
        dims = struct.unpack_from("<%dI" % ndim, buf, pos)
        pos += 4 * ndim
    if ndim == 0:
        return np.zeros((0,), np.float32), pos
    pos += 8  # context: dev_type, dev_id
    (flag,) = struct.unpack_from("<i", buf, pos)
    pos += 4
    if flag not in _DTYPES:
        raise ValueError("unknown MXNet dtype flag %d" % flag)
    dt = np.dtype(_DTYPES[flag]).newbyteorder("<")
    n = int(np.prod(dims, dtype=np.int64))
    arr = np.frombuffer(buf, dtype=dt, count=n, offset=pos).reshape(dims).copy()
    return arr, pos + n * dt.itemsize


def load_params(path):
    """MXNet NDArray-list file -> {name: ndarray}.  Gluon prefixes ('arg:' / 'aux:') are stripped, so the structural
    keys of save_parameters ('deform5.weight', 'conv1a.0.weight', ...) come back as they are."""
    buf = open(path, "rb").read()
    magic, _reserved, count = struct.unpack_from("<QQQ", buf, 0)
    if magic != _LIST_MAGIC:
        raise ValueError("%s: not an MXNet NDArray list (magic 0x%x)" % (path, magic))
    pos, arrays = 24, []
    for _ in range(count):
        a, pos = _read_ndarray(buf, pos)
        arrays.append(a)
    (nnames,) = struct.unpack_from("<Q", buf, pos)
    pos += 8
    names = []
    for _ in range(nnames):
        (ln,) = struct.unpack_from("<Q", buf, pos)
        pos += 8
        names.append(buf[pos:pos + ln].decode())
        pos += ln
    if nnames not in (0, count):
        raise ValueError("%s: %d names for %d arrays" % (path, nnames, count))
    if not names:
        names = [str(i) for i in range(count)]
    strip = lambda k: k.split(":", 1)[1] if k[:4] in ("arg:", "aux:") else k
    return {strip(k): a for k, a in zip(names, arrays)}


def save_params(path, params):
    """Writer of the same format (V2 dense NDArrays on cpu(0)); used to pin load_params by round trips."""
    with open(path, "wb") as f:
        f.write(struct.pack("<QQQ", _LIST_MAGIC, 0, len(params)))
        for a in params.values():
            a = np.ascontiguousarray(a)
            if a.dtype not in _FLAGS:
                raise ValueError("dtype %s has no MXNet flag" % a.dtype)
            f.write(struct.pack("<IiI", _V2, 0, a.ndim))
            f.write(struct.pack("<%dq" % a.ndim, *a.shape))
            f.write(struct.pack("<iii", 1, 0, _FLAGS[a.dtype]))
            f.write(a.astype(a.dtype.newbyteorder("<"), copy=False).tobytes())
        f.write(struct.pack("<Q", len(params)))
        for k in params:
            kb = k.encode()
            f.write(struct.pack("<Q", len(kb)))
            f.write(kb)


def load_into(module, params, strict=True):
    """Copy checkpoint arrays into a torch module that uses the reference's parameter names (layer.DeformableConv2D:
    '<prefix>weight' / '<prefix>bias').  Blocks built the way the reference builds them -- DeformableConv2D with
    in_channels=0, MaskFlownet.py:155-158: weight creation deferred to the first forward -- are materialised here from
    the checkpoint's weight shape first, so that their parameters exist to be loaded (and are seen by an optimizer
    built afterwards).  strict: every parameter must be in the checkpoint AND every checkpoint key must match a
    parameter.  Returns (missing, unexpected)."""
    import torch
    from .layer import DeformableConv2D
    for name, mod in module.named_modules():
        if isinstance(mod, DeformableConv2D) and mod.weight is None:
            key = (name + "." if name else "") + "weight"
            if key in params:
                w = params[key]
                if w.ndim != 4 or w.shape[0] != mod._channels or tuple(w.shape[2:]) != tuple(mod._kwargs["kernel"]):
                    raise ValueError("%s: checkpoint shape %s does not fit DeformableConv2D(%d, kernel %s)"
                                     % (key, w.shape, mod._channels, mod._kwargs["kernel"]))
                mod._materialize(int(w.shape[1]) * mod._kwargs["num_group"], torch.device("cpu"))
    own = dict(module.named_parameters())
    missing = [k for k in own if k not in params]
    unexpected = [k for k in params if k not in own]
    if strict and missing:
        raise KeyError("checkpoint lacks %s" % missing)
    if strict and unexpected:
        raise KeyError("checkpoint keys without a parameter: %s" % unexpected)
    with torch.no_grad():
        for k, p in own.items():
            if k in params:
                if tuple(p.shape) != tuple(params[k].shape):
                    raise ValueError("%s: checkpoint shape %s, module shape %s" % (k, params[k].shape, tuple(p.shape)))
                p.copy_(torch.from_numpy(np.ascontiguousarray(params[k])))
    return missing, unexpected
