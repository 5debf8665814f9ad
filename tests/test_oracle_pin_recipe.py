"""The MXNet pin recipe proves itself.  MXNet cannot run in this image, so tests/golden/mxnet_kat_v1.npz does not
exist and tests/test_oracle_pinned_by_mxnet.py skips -- which would also hide a recipe that can only ever pass.  Here the recipe's
`--compare-only` path (tools/pin_oracle_with_mxnet.compare) and the assertions of that test file run against STAND-IN fixtures:

  * one built from the oracle's own outputs: every assertion passes (the plumbing works end to end);
  * the same fixture with ONE rule of SURVEY.md A.5 deliberately answered the other way, as an MXNet that differs from the
    restatement would answer it: the assertion that guards that rule must FAIL -- a maintainer's first real run cannot pass silently.

Test infrastructure only; nothing here is evidence about MXNet itself ("parity unpinned")."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def recipe():
    import pin_oracle_with_mxnet as pin
    from oracle import ref
    from tests import test_oracle_pinned_by_mxnet as guards
    G = np.load(pin.GOLD)
    M = {"meta_mxnet_version": np.array("stand-in (the oracle itself)"), "meta_dc_context": np.array("cpu(0)")}
    for tag, md in (("corr4", 4), ("corr2", 2)):
        M[tag + "_out"] = ref.correlation(G[tag + "_f1"], G[tag + "_f2"], max_displacement=md, pad_size=md)
        M[tag + "_g1"], M[tag + "_g2"] = ref.correlation_backward(G[tag + "_gout"], G[tag + "_f1"], G[tag + "_f2"], max_displacement=md, pad_size=md)
    M["warp_out"] = ref.warp(G["warp_x"], G["warp_flow"], clip_grid=False)
    M["warp_out_clip"] = ref.warp(G["warp_x"], G["warp_flow"], clip_grid=True)
    M["warp_gx"], M["warp_gflow"] = ref.warp_backward(G["warp_gout"], G["warp_x"], G["warp_flow"], clip_grid=False)
    kw = dict(kernel=(3, 3), pad=(1, 1))
    M["dc_out"] = ref.deformable_convolution(G["dc_x"], G["dc_offset"], G["dc_w"], G["dc_b"], **kw)
    M["dc_out_pertap"] = ref.deformable_convolution(G["dc_x"], G["dc_offset_pertap"], G["dc_w"], G["dc_b"], **kw)
    M["dc_gx"], M["dc_goffset"], M["dc_gw"], M["dc_gb"] = ref.deformable_convolution_backward(G["dc_gout"], G["dc_x"], G["dc_offset_pertap"], G["dc_w"], **kw)
    px, pw, off2, off3 = pin.probes()
    zb = np.zeros(1, np.float32)
    M["probe_q2"] = ref.deformable_convolution(px, off2, pw, zb, **kw)
    M["probe_q3"] = ref.deformable_convolution(px, off3, pw, zb, **kw)
    return pin, guards, M, px


ALL_GUARDS = ("test_q1_q6_correlation_matches_mxnet", "test_warp_pair_matches_mxnet", "test_q2_negative_fraction_row_contributes_zero",
              "test_q3_last_row_is_clamped", "test_q4_fraction_source", "test_deformable_convolution_forward_and_backward_match_mxnet")


def _run_guards(pin, guards, M):
    pinned = pin.compare(M, verbose=False)
    failed = []
    for name in ALL_GUARDS:
        try:
            getattr(guards, name)(pinned)
        except AssertionError:
            failed.append(name)
    return failed


def test_recipe_passes_on_a_fixture_that_agrees_with_the_oracle(recipe):
    pin, guards, M, _ = recipe
    assert _run_guards(pin, guards, M) == []
    assert pin.compare_network(M, verbose=False) == {}      # no network probe in this fixture: the network guard skips


def test_recipe_fails_when_mxnet_would_sample_rows_in_minus_one_to_zero(recipe):
    """Q2 answered the other way (`h_im > -1` instead of `h_im >= 0`): row 0 of the probe is 0.5 * x[0] instead of 0."""
    pin, guards, M, px = recipe
    m = dict(M)
    q2 = M["probe_q2"].copy()
    q2[0, 0, 0] = 0.5 * px[0, 0, 0]
    m["probe_q2"] = q2
    assert "test_q2_negative_fraction_row_contributes_zero" in _run_guards(pin, guards, m)


def test_recipe_fails_when_mxnet_would_not_clamp_the_last_row(recipe):
    """Q3 answered the other way: a tap in [H-1, H) interpolates towards a zero row H (0.5 * x[H-1]) instead of reading row H-1."""
    pin, guards, M, px = recipe
    m = dict(M)
    q3 = M["probe_q3"].copy()
    q3[0, 0, -1] = 0.5 * px[0, 0, -1]
    m["probe_q3"] = q3
    assert "test_q3_last_row_is_clamped" in _run_guards(pin, guards, m)


@pytest.mark.parametrize("mutation", ["dy_fastest", "no_normaliser"])
def test_recipe_fails_when_the_cost_volume_convention_differs(recipe, mutation):
    """Q1 answered the other way: displacement channels ordered dy-fastest, or the sum not divided by kernel^2 C."""
    pin, guards, M, _ = recipe
    m = dict(M)
    out = M["corr4_out"]
    n, d2, h, w = out.shape
    d = int(round(d2 ** 0.5))
    m["corr4_out"] = out.reshape(n, d, d, h, w).transpose(0, 2, 1, 3, 4).reshape(n, d2, h, w).copy() if mutation == "dy_fastest" \
        else out * np.float32(np.load(pin.GOLD)["corr4_f1"].shape[1])
    assert "test_q1_q6_correlation_matches_mxnet" in _run_guards(pin, guards, m)


def test_recipe_fails_when_the_sampler_semantics_differ(recipe):
    """The warp pair with align_corners = False semantics (a half-pixel shift) is caught by the warp guard."""
    pin, guards, M, _ = recipe
    m = dict(M)
    m["warp_out"] = np.roll(M["warp_out"], 1, axis=3)
    assert "test_warp_pair_matches_mxnet" in _run_guards(pin, guards, m)
