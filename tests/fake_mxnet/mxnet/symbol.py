"""mxnet.symbol of the stub: only what `from mxnet import symbol` (network/layer.py:6) needs to import."""


class Symbol:
    pass
