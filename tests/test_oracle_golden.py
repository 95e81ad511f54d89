"""Pins the CPU oracle against (a) golden vectors captured from the reference
(tests/golden/make_golden.py) and (b) the reference's own known-answer tests.
CPU only."""
import math

import numpy as np
import pytest
import torch

from oracle import timesblock_oracle as orc


def _params(ftn, man, hyp, case):
    h = hyp[case["hyper"]]
    C = case["C"]
    d_ff = C if h["d_ff_mult"] is None else C * h["d_ff_mult"]
    ks = [tuple(k) for k in h["kernel_set"]]
    p = ftn.synth.make_inception_params(C, d_ff, ks, h["ratio"], case["seed"])
    wsum = float(sum(np.abs(v).sum(dtype=np.float64) for v in p.values()))
    assert math.isclose(wsum, case["wsum"], rel_tol=1e-12), "weight generator drifted from the fixtures"
    return {k: torch.from_numpy(v) for k, v in p.items()}, ks, h["act"]


@pytest.fixture(scope="module")
def hypers():
    import json
    from conftest import GOLDEN
    return json.loads((GOLDEN / "manifest.json").read_text())["hypers"]


BLOCKS = ["b_tiny_min", "b_tiny_pipe", "b_c0_min", "b_c0_pipe", "b_c0_rect", "b_c0_wide1", "b_odd_min",
          "b_odd_pipe", "b_c1_min", "b_c1_pipe", "b_c2_pipe_k5", "b_noise_pipe"]


@pytest.mark.parametrize("name", BLOCKS)
def test_block_matches_reference(name, manifest, golden, hypers, ftn):
    case, g = manifest[name], golden(name)
    P, ks, act = _params(ftn, manifest, hypers, case)
    x = torch.from_numpy(g["x"])
    # regenerated input must equal the stored one (guards the input generator)
    x2 = ftn.synth.make_input(case["B"], case["L"], case["C"], seed=case["seed"],
                              planted=None if case["planted"] is None else tuple(case["planted"]))
    assert np.array_equal(x2, g["x"])
    y, aux = orc.timesblock_forward(x, P, ks, act, case["K"], case["L"], 1)
    assert aux.sel.periods == g["periods"].tolist()          # bit-exact integers
    assert aux.sel.freq_idx == g["freq_idx"].tolist()
    assert aux.groups.periods == g["g_periods"].tolist()
    assert aux.groups.pad == g["g_pad"].tolist()
    assert aux.groups.cycles == g["g_cycles"].tolist()
    assert aux.groups.mapping == g["mapping"].tolist()
    np.testing.assert_allclose(aux.sel.amps.numpy(), g["amps"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(y.numpy(), g["y"], rtol=1e-4, atol=2e-5)


STUBS = ["s_dup", "s_mixed_pad", "s_448", "s_wide_rows", "s_p1"]


@pytest.mark.parametrize("name", STUBS)
def test_stub_selector_matches_reference(name, manifest, golden, hypers, ftn):
    case, g = manifest[name], golden(name)
    P, ks, act = _params(ftn, manifest, hypers, case)
    x = torch.from_numpy(g["x"])
    y, aux = orc.timesblock_forward(x, P, ks, act, 0, case["L"], 1,
                                    periods=g["periods"].tolist(), amps=torch.from_numpy(g["amps"]))
    assert len(aux.groups.periods) == case["groups"]
    np.testing.assert_allclose(y.numpy(), g["y"], rtol=1e-4, atol=2e-5)


SELECTS = ["sel_kat_256", "sel_bounds", "sel_c1", "sel_odd", "sel_L720", "sel_thr7", "sel_even_c"]


@pytest.mark.parametrize("name", SELECTS)
def test_selector_matches_reference(name, manifest, golden):
    case, g = manifest[name], golden(name)
    r = orc.period_select(torch.from_numpy(g["x"]), case["K"], case["pmax"], case["min_thr"])
    assert r.periods == g["periods"].tolist()
    assert r.freq_idx == g["freq_idx"].tolist()
    np.testing.assert_allclose(r.median.numpy(), g["median"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(r.amp_mean.numpy(), g["amp_mean"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(r.amps.numpy(), g["amps"], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("name", ["lrtc_a", "lrtc_b", "lrtc_c"])
def test_lrtc_matches_reference(name, manifest, golden):
    case, g = manifest[name], golden(name)
    np.testing.assert_allclose(orc.lrtc_basis(case["L"], case["R"]).numpy(), g["basis"], rtol=1e-6, atol=1e-7)
    ctx = orc.lrtc_forward(torch.from_numpy(g["coeff"]), case["L"], case["scale"])
    np.testing.assert_allclose(ctx.numpy(), g["ctx"], rtol=1e-5, atol=1e-6)


# ---- the reference's own known-answer tests, restated against the oracle ----
def test_kat_shared_periods_and_ordering():
    """reference tests/test_fft_period_selector.py:14-40"""
    torch.manual_seed(0)
    B, L, C = 2, 256, 3
    t = torch.arange(L, dtype=torch.float32)
    sig = 3.0 * torch.sin(2 * math.pi * 4 * t / L) + 1.5 * torch.sin(2 * math.pi * 8 * t / L)
    x = torch.stack([torch.stack([sig + 0.01 * torch.randn(L) for _ in range(C)], 1) for _ in range(B)], 0)
    r = orc.period_select(x, 2, L)
    assert r.periods == [64, 32]
    assert r.amps.shape == (B, 2) and bool(torch.all(r.amps[:, 0] >= r.amps[:, 1]))


def test_kat_bounds_zero_k_and_min_cycles():
    """reference tests/test_fft_period_selector.py:43-70,105-118"""
    L = 64
    t = torch.arange(L, dtype=torch.float32)
    x = (2.0 * torch.sin(2 * math.pi * 2 * t / L) + torch.sin(2 * math.pi * 20 * t / L)).view(1, L, 1)
    r = orc.period_select(x, 2, 16, 5)
    assert r.periods == [16, 5] and bool(torch.all(r.amps > 0))
    assert orc.period_select(x, 0, L).periods == []
    L = 28
    x = torch.sin(2 * math.pi * torch.arange(L, dtype=torch.float32) / 7).view(1, L, 1)
    r = orc.period_select(x, 3, L)
    assert r.periods and all(p < L and (L + p - 1) // p >= 2 for p in r.periods)


def test_kat_softmax_weighting_and_identity():
    """reference tests/test_times_block.py:101-136 (analytic residual; invalid periods)."""
    amps = torch.tensor([[2.0, 0.0]])
    grp = orc.period_group([2, 4], 8, 1, 8)
    w = orc.group_weights(amps, grp.mapping, len(grp.periods))
    assert torch.allclose(w, torch.softmax(amps, dim=1))
    assert orc.period_group([0, -1], 5, 1, 5).periods == []


def test_kat_duplicates_group_once_and_mass_is_conserved():
    """reference tests/test_times_block.py:157-180, tests/test_timesblock_vectorized.py:132-168"""
    grp = orc.period_group([4, 4, 4, 8], 32, 1, 32)
    assert grp.periods == [4, 8] and grp.mapping == [0, 0, 0, 1]
    amps = torch.randn(5, 4)
    w = orc.group_weights(amps, grp.mapping, 2)
    assert torch.allclose(w.sum(1), torch.ones(5), atol=1e-6)
    lse = torch.stack([torch.logsumexp(amps[:, :3], 1), amps[:, 3]], 1)
    assert torch.allclose(w, torch.softmax(lse, 1), atol=1e-6)
