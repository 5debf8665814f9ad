"""Assembles tests/golden/mxnet_v2.params BY HAND from the documented layout of an MXNet 1.5 NDArray-list file
(src/ndarray/ndarray.cc: NDArray::Save(dmlc::Stream*, vector<NDArray>, vector<string>) and NDArray::Save(strm)),
WITHOUT going through maskflownet_amd.io -- the reader is then checked against these bytes
(tests/test_io.py::test_params_reader_against_hand_assembled_bytes).  What Gluon's save_parameters writes for the
reference (network/pipeline.py:52-54) is this container with structural keys such as 'deform5.weight'.

    uint64 0x112 (kMXAPINDArrayListMagic) | uint64 0 (reserved) | uint64 count
    count x NDArray:
        V2: uint32 0xF993FAC9 | int32 storage type (0 = default/dense) | uint32 ndim | ndim x int64 dims
            | int32 dev_type (1 = cpu) | int32 dev_id | int32 type flag (0 = float32, 4 = int32, ...) | raw LE data
        V1: uint32 0xF993FAC8 | uint32 ndim | ndim x int64 dims | context | type flag | data
        legacy (MXNet < 0.9): uint32 ndim | ndim x uint32 dims | context | type flag | data
    uint64 name count | per name: uint64 length | bytes

Run from the repo root:  python tests/golden/make_params_fixture.py
"""
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def le(value, nbytes):
    return int(value).to_bytes(nbytes, "little", signed=value < 0)


def f32(x):
    import struct
    return struct.pack("<f", x)


def main():
    out = bytearray()
    out += le(0x112, 8) + le(0, 8) + le(4, 8)
    # 1. 'deform5.weight' (2, 1, 3, 3) float32, V2, values k/4 - 2
    out += le(0xF993FAC9, 4) + le(0, 4) + le(4, 4) + le(2, 8) + le(1, 8) + le(3, 8) + le(3, 8)
    out += le(1, 4) + le(0, 4) + le(0, 4)
    for k in range(18):
        out += f32(k / 4.0 - 2.0)
    # 2. 'deform5.bias' (2,) float32, V2, context gpu(3) (a checkpoint saved from a GPU run)
    out += le(0xF993FAC9, 4) + le(0, 4) + le(1, 4) + le(2, 8) + le(2, 4) + le(3, 4) + le(0, 4) + f32(0.5) + f32(-1.25)
    # 3. 'arg:steps' (3,) int32, V1
    out += le(0xF993FAC8, 4) + le(1, 4) + le(3, 8) + le(1, 4) + le(0, 4) + le(4, 4) + le(7, 4) + le(-2, 4) + le(100000, 4)
    # 4. 'aux:legacy' (2, 2) float32, pre-0.9 layout
    out += le(2, 4) + le(2, 4) + le(2, 4) + le(1, 4) + le(0, 4) + le(0, 4) + f32(1.0) + f32(2.0) + f32(3.0) + f32(4.0)
    names = [b"deform5.weight", b"deform5.bias", b"arg:steps", b"aux:legacy"]
    out += le(len(names), 8)
    for n in names:
        out += le(len(n), 8) + n
    path = os.path.join(HERE, "mxnet_v2.params")
    with open(path, "wb") as f:
        f.write(bytes(out))
    print(path, len(out), "bytes")


if __name__ == "__main__":
    main()
