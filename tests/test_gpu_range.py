"""Domain of the default engine (f16x2: activations travel as fp16 pieces, |v| < 65504) against the reference's fp32
(models/timesnet.py:1047-1056 up-casts to fp32 for the convs): input scales from 1e-6 to beyond the fp16 maximum, the
automatic repeat on bf16x3 when a value does not fit, and the degenerate batches the reference's sales data contain -
all zero, constant, a row of NaN.

Error is measured against the size of what the block adds, max|y_ref - x| (rtol 1e-4 of BASELINE.json's north star),
plus one fp32 ulp of x for the final ``x + delta``."""
import warnings

import numpy as np
import pytest
import torch

from oracle import timesblock_oracle as orc

pytestmark = pytest.mark.gpu
KS = [(3, 3), (5, 5), (7, 7)]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need a ROCm device"
    return torch.device("cuda:0")


def _block(ftn, dev, C=64, K=3, L=96, engine=None, seed=3):
    T = ftn.models.timesnet
    blk = T.TimesBlock(C, KS, 0.0, "gelu", d_ff=4 * C, bottleneck_ratio=4.0)
    blk.engine = engine
    sd = ftn.synth.make_inception_params(C, 4 * C, KS, 4.0, seed)
    P = {k: torch.from_numpy(v) for k, v in sd.items()}
    blk.inception.load_state_dict(P, strict=True)
    blk.period_selector = T.FFTPeriodSelector(K, L)
    return blk.eval().to(dev), P


def _unit_input(ftn, B, L, C, seed):
    base = ftn.synth.make_input(B, L, C, seed=seed, planted=(24, 12, 8))
    return (base / np.abs(base).max()).astype(np.float32)         # max |x| = 1


def _close(y, y_ref, x):
    err = float(np.abs(y - y_ref).max())
    delta = float(np.abs(y_ref - x).max())
    return err, 1e-4 * delta + 2.0 * np.finfo(np.float32).eps * float(np.abs(x).max())


@pytest.mark.parametrize("scale", [1e-6, 1e-3, 1.0, 1e3, 3e4])
def test_default_engine_across_input_scales(scale, ftn, dev):
    """max|x| = scale.  Whatever the stage outputs grow to, the result must be the reference's: either the fp16 pieces
    held every value, or the kernels flagged the call and the block repeated it on bf16x3."""
    B, L, C, K = 4, 96, 64, 3
    blk, P = _block(ftn, dev, C, K, L)
    x = torch.from_numpy(_unit_input(ftn, B, L, C, 5) * np.float32(scale))
    y_ref, aux = orc.timesblock_forward(x, P, KS, "gelu", K, L)
    with torch.inference_mode(), warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        y = blk(x.to(dev))
        used = blk.check_range()                                  # (reading blk._last_engine does the same)
    assert blk._last_backend == "hip" and used in ("f16x2", "bf16x3")
    assert blk.period_selector.last_selected_periods.tolist() == aux.sel.periods
    err, tol = _close(y.cpu().numpy(), y_ref.numpy(), x.numpy())
    assert np.isfinite(y.cpu().numpy()).all() and err <= tol, (scale, used, err, tol)


def test_out_of_range_call_is_repeated_on_bf16x3(ftn, dev):
    """max|x| = 1e5 > 65504: the f16x2 kernels must notice, and the tensor the block returned must end up holding
    exactly what engine bf16x3 computes (finite, the reference's result) - without the caller doing anything but
    letting the block look at its flag (check_range / _last_engine / its next call)."""
    B, L, C, K = 3, 96, 64, 3
    blk, P = _block(ftn, dev, C, K, L)
    safe, _ = _block(ftn, dev, C, K, L, engine="bf16x3")
    x = torch.from_numpy(_unit_input(ftn, B, L, C, 6) * np.float32(1e5))
    y_ref, _ = orc.timesblock_forward(x, P, KS, "gelu", K, L)
    with torch.inference_mode():
        want = safe(x.to(dev))
        assert safe.check_range() == "bf16x3" and safe._range_fallbacks == 0
        with pytest.warns(RuntimeWarning, match="fp16 range"):
            y = blk(x.to(dev))
            assert blk._last_engine == "bf16x3"
        assert blk._range_fallbacks == 1
        assert torch.isfinite(y).all() and torch.equal(y, want)
        err, tol = _close(y.cpu().numpy(), y_ref.numpy(), x.numpy())
        assert err <= tol, (err, tol)
        # the next in-range call runs on f16x2 again, and a flagged call is also repaired by the NEXT call of the block
        x1 = torch.from_numpy(_unit_input(ftn, B, L, C, 7)).to(dev)
        y1 = blk(x1)
        assert blk.check_range() == "f16x2" and blk._range_fallbacks == 1
        with pytest.warns(RuntimeWarning, match="fp16 range"):
            y2 = blk(x.to(dev))
            torch.cuda.synchronize()
            blk(x1)                                               # looks at the finished call's flag first
        assert blk._range_fallbacks == 2 and torch.equal(y2, want)
        assert torch.equal(blk(x1), y1)


def test_large_hidden_values_without_large_inputs(ftn, dev):
    """x stays small but a weight matrix is scaled so that the stage outputs leave the fp16 range: the guard is on
    the values that are split, not on x."""
    B, L, C, K = 2, 96, 64, 3
    T = ftn.models.timesnet
    blk = T.TimesBlock(C, KS, 0.0, "gelu", d_ff=4 * C, bottleneck_ratio=4.0)
    sd = ftn.synth.make_inception_params(C, 4 * C, KS, 4.0, 9)
    for k in list(sd):
        if k.startswith("0.paths.") and k.endswith("branch.0.weight"):       # first 1x1 of every branch: a = W_in1 x + b
            sd[k] = sd[k] * np.float32(3e5)
    P = {k: torch.from_numpy(v) for k, v in sd.items()}
    blk.inception.load_state_dict(P, strict=True)
    blk.period_selector = T.FFTPeriodSelector(K, L)
    blk = blk.eval().to(dev)
    x = torch.from_numpy(_unit_input(ftn, B, L, C, 10))
    y_ref, _ = orc.timesblock_forward(x, P, KS, "gelu", K, L)
    with torch.inference_mode(), pytest.warns(RuntimeWarning, match="fp16 range"):
        y = blk(x.to(dev))
        assert blk._last_engine == "bf16x3"
    err, tol = _close(y.cpu().numpy(), y_ref.numpy(), x.numpy())
    assert err <= tol, (err, tol)


def test_all_zero_batch(ftn, dev):
    """Zero series are the common case in the reference's sales data: every amplitude is exactly 0, the scores are the
    tie-break penalty alone, and the selector must return bins 1..K (lowest index first) - bit-exact."""
    B, L, C, K = 8, 336, 64, 5
    blk, P = _block(ftn, dev, C, K, L)
    x = torch.zeros(B, L, C)
    sel = orc.period_select(x, K, L)
    assert sel.freq_idx == [1, 2, 3, 4, 5]
    y_ref, aux = orc.timesblock_forward(x, P, KS, "gelu", K, L)
    with torch.inference_mode():
        y = blk(x.to(dev))
        assert blk.check_range() == "f16x2"
    assert blk.period_selector.last_frequency_indices.tolist() == sel.freq_idx
    assert blk.period_selector.last_selected_periods.tolist() == sel.periods
    assert blk._last_group_count == len(aux.groups.periods)
    np.testing.assert_allclose(y.cpu().numpy(), y_ref.numpy(), rtol=1e-4, atol=2e-5)


def test_constant_batch(ftn, dev):
    """x = const: only the DC bin (which the selector kills, :119-120) carries signal.  pocketfft returns exact zeros
    elsewhere; a DFT-as-GEMM returns rounding noise, so WHICH bins win is not pinned - the amplitudes must be noise
    (below 1e-5 of the DC term), the periods valid, and the block's output the reference's for those periods."""
    B, L, C, K = 6, 336, 64, 5
    blk, P = _block(ftn, dev, C, K, L)
    x = torch.full((B, L, C), 3.25)
    with torch.inference_mode():
        y = blk(x.to(dev))
        assert blk.check_range() == "f16x2"
    sel = blk.period_selector
    periods = sel.last_selected_periods.tolist()
    assert len(periods) == K and all(2 <= p < L for p in periods)
    med, _ = ftn.runtime.spectrum(x.to(dev))
    assert float(med[:, 1:].abs().max()) <= 1e-5 * 3.25 * L
    amps = med[:, sel.last_frequency_indices].cpu()
    y_ref, _ = orc.timesblock_forward(x, P, KS, "gelu", 0, L, 1, periods=periods, amps=amps)
    np.testing.assert_allclose(y.cpu().numpy(), y_ref.numpy(), rtol=1e-4, atol=2e-5)


def test_nan_row_stays_nan(ftn, dev):
    """A batch row of NaN: the reference's rfft / median / mean propagate it (:109-112), its output row is NaN.  Here
    the row must come back NaN as well (never finite garbage) and the other rows finite; which periods a NaN spectrum
    selects is torch.topk's undefined order in the reference, so nothing else is pinned."""
    B, L, C, K = 4, 96, 64, 3
    blk, _ = _block(ftn, dev, C, K, L)
    xs = _unit_input(ftn, B, L, C, 11)
    xs[2] = np.nan
    with torch.inference_mode(), warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        y = blk(torch.from_numpy(xs).to(dev))
        blk.check_range()
    y = y.cpu().numpy()
    assert np.isnan(y[2]).all()
    assert np.isfinite(np.delete(y, 2, axis=0)).all()
