"""The hot-path pass of maskflownet_amd/hotpath.py on the CPU oracle (TEST INFRASTRUCTURE ONLY).

Used by tests/, __graft_entry__.smoke() (as the checker) and bench.py's cpu_baseline leg (as the
reported CPU baseline, kind "port").  Same operator sequence as MaskFlownet_S.hybrid_forward
(/root/reference/network/MaskFlownet.py:215-311), same seeded inputs.
"""
from maskflownet_amd.hotpath import MD, SCALE, STRIDES

from . import ref as oracle


def oracle_pass(host, n_pairs):
    """Run the pass on the first `n_pairs` samples of the synthetic batch `host` (numpy dict)."""
    sl = slice(0, n_pairs)
    out = {}
    out["corr6"] = oracle.correlation(host["c1_6"][sl], host["c2_6"][sl], max_displacement=MD, pad_size=MD)
    for l in (5, 4, 3, 2):
        off = oracle.offsets_from_flow(host["flow_%d" % l][sl], SCALE, float(STRIDES[l]))
        out["deform%d" % l] = oracle.deformable_convolution(host["c2_%d" % l][sl], off, host["w_%d" % l],
                                                            host["b_%d" % l], kernel=(3, 3), pad=(1, 1))
        out["corr%d" % l] = oracle.correlation(host["c1_%d" % l][sl], out["deform%d" % l], max_displacement=MD,
                                               pad_size=MD)
    out["warp"] = oracle.warp(host["img2"][sl], host["flow_full"][sl], clip_grid=False)
    return out
