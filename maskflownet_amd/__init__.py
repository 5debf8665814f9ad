"""maskflownet_amd -- MI355X (gfx950) implementation of MaskFlownet's matching hot path.

Correlation cost volume, flow warp and DeformableConvolution as hand-written HIP kernels behind
the C ABI of include/mfn_hip.h, with a Python front-end that mirrors the operator and layer
signatures of /root/reference/network/layer.py and network/MaskFlownet.py:193-195.
"""
__version__ = "0.1.0"
