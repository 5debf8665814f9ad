"""mxnet.initializer of the stub: MSRAPrelu (/root/reference/network/pipeline.py:26) and zeros."""
import math

import numpy as np


class Initializer:
    def init(self, name, shape, rng):
        raise NotImplementedError


class Zero(Initializer):
    def init(self, name, shape, rng):
        return np.zeros(shape, np.float32)


class MSRAPrelu(Initializer):
    """Xavier('gaussian', factor_type, magnitude = 2 / (1 + slope^2)): N(0, sqrt(magnitude / factor))."""

    def __init__(self, factor_type="avg", slope=0.25):
        self.factor_type, self.slope = factor_type, slope

    def init(self, name, shape, rng):
        if name.endswith("bias"):
            return np.zeros(shape, np.float32)
        hw = int(np.prod(shape[2:])) if len(shape) > 2 else 1
        fan_in, fan_out = shape[1] * hw, shape[0] * hw
        factor = {"avg": (fan_in + fan_out) / 2.0, "in": fan_in, "out": fan_out}[self.factor_type]
        return (rng.standard_normal(shape) * math.sqrt(2.0 / (1 + self.slope ** 2) / factor)).astype(np.float32)


def create(spec):
    if spec is None:
        return None
    if isinstance(spec, Initializer):
        return spec
    if spec == "zeros":
        return Zero()
    raise ValueError("stub initializer %r" % (spec,))
