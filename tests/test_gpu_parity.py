"""GPU parity tests (run with ``-m gpu`` on an MI355X): the HIP path, called
through the C ABI by the drop-in modules, against (a) golden vectors captured
from the reference and (b) the CPU oracle on the same seeded inputs.

Tolerances: period / frequency indices and all grouping integers bit-exact;
floating point within rtol 1e-4 (BASELINE.json north_star) plus a small atol for
values that cancel to ~0 (outputs are O(1))."""
import json

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import timesblock_oracle as orc

pytestmark = pytest.mark.gpu

HYP = json.loads((GOLDEN / "manifest.json").read_text())["hypers"]
RTOL, ATOL = 1e-4, 2e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need a ROCm device"
    return torch.device("cuda:0")


def _hyper(case):
    h = HYP[case["hyper"]]
    C = case["C"]
    d_ff = C if h["d_ff_mult"] is None else C * h["d_ff_mult"]
    return C, d_ff, [tuple(k) for k in h["kernel_set"]], h["ratio"], h["act"], h["d_ff_mult"] is None


ENGINES = ["f32", "bf16x3", "f16x2"]


def _block(ftn, case, dev, engine=None):
    T = ftn.models.timesnet
    C, d_ff, ks, ratio, act, dff_none = _hyper(case)
    blk = T.TimesBlock(C, ks, 0.0, act, d_ff=None if dff_none else d_ff, bottleneck_ratio=ratio)
    blk.engine = engine
    sd = ftn.synth.make_inception_params(C, d_ff, ks, ratio, case["seed"])
    blk.inception.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return blk.eval().to(dev), {k: torch.from_numpy(v) for k, v in sd.items()}, ks, act


def test_mfma_lane_maps(ftn, dev):
    out = ftn.runtime.selftest_mfma(dev).cpu().numpy()
    i, k, j = np.arange(16)[:, None, None], np.arange(8)[None, :, None], np.arange(16)[None, None, :]
    want = ((i * 8 + k + 1) * ((k + 1) * 100 + j)).sum(1).astype(np.float32)
    np.testing.assert_array_equal(out, want)


SELECTS = ["sel_kat_256", "sel_bounds", "sel_c1", "sel_odd", "sel_L720", "sel_thr7", "sel_even_c"]


@pytest.mark.parametrize("name", SELECTS)
def test_selector_matches_reference(name, manifest, golden, ftn, dev):
    case, g = manifest[name], golden(name)
    T = ftn.models.timesnet
    sel = T.FFTPeriodSelector(case["K"], case["pmax"], case["min_thr"])
    x = torch.from_numpy(g["x"]).to(dev)
    with torch.inference_mode():
        periods, amps = sel(x)
    assert periods.dtype == torch.long and periods.tolist() == g["periods"].tolist()     # bit-exact
    assert sel.last_frequency_indices.tolist() == g["freq_idx"].tolist()
    np.testing.assert_allclose(amps.cpu().numpy(), g["amps"], rtol=RTOL, atol=1e-4 * float(np.abs(g["median"]).max()))
    med, psum = ftn.runtime.spectrum(x.contiguous())
    # the device-built descriptor (wave-parallel grouping in k_finalize) == the host's ftn_desc_from_periods
    dd = sel._pending.host() if sel._pending is not None else None
    if dd is not None:
        n = int(dd.n_sel)
        hd = ftn.lib.desc_from_periods(list(dd.sel_period[:n]), case["L"], sel.min_period_threshold, sel.pmax)
        for fld in ("g_period", "g_pad", "g_cycles", "g_px_off", "g_tw", "g_th", "g_ntx", "g_nty", "g_tile_off",
                    "sel_group"):
            assert list(getattr(dd, fld)) == list(getattr(hd, fld)), fld
        assert (dd.n_groups, dd.total_px, dd.tiles_per_row) == (hd.n_groups, hd.total_px, hd.tiles_per_row)
    scale = float(np.abs(g["median"]).max())
    np.testing.assert_allclose(med.cpu().numpy(), g["median"], rtol=RTOL, atol=2e-6 * scale)
    np.testing.assert_allclose(psum.cpu().numpy() / case["B"], g["amp_mean"], rtol=RTOL, atol=2e-6 * scale)


@pytest.mark.parametrize("L", [500, 1024, 2100])
def test_selector_top_k_both_forms(L, ftn, dev):
    """The finalize workgroup picks its top-k by rank counting over 64-bit keys up to F = 256 bins and by k rounds of
    a single-wave shuffle arg-max beyond (ftn_finalize.h): L = 500 (F = 251) against L = 1024 (F = 513) and L = 2100
    (F = 1051), each against the oracle, bit-exact indices, with planted periods so that the winners are not near-ties."""
    from oracle import timesblock_oracle as orc

    B, C, K = 4, 8, 5
    x = torch.from_numpy(ftn.synth.make_input(B, L, C, seed=21, planted=(50, 21, 12, 7)))
    want = orc.period_select(x, K, L)
    sel = ftn.models.timesnet.FFTPeriodSelector(K, L)
    with torch.inference_mode():
        periods, _ = sel(x.to(dev))
    assert sel.last_frequency_indices.tolist() == want.freq_idx
    assert periods.tolist() == want.periods


BLOCKS = ["b_tiny_min", "b_tiny_pipe", "b_c0_min", "b_c0_pipe", "b_c0_rect", "b_c0_wide1", "b_odd_min",
          "b_odd_pipe", "b_c1_min", "b_c1_pipe", "b_c2_pipe_k5", "b_noise_pipe"]


@pytest.mark.parametrize("engine", ENGINES)
@pytest.mark.parametrize("name", BLOCKS)
def test_block_matches_reference(name, engine, manifest, golden, ftn, dev):
    case, g = manifest[name], golden(name)
    blk, _, _, _ = _block(ftn, case, dev, engine)
    blk.period_selector = ftn.models.timesnet.FFTPeriodSelector(case["K"], case["L"])
    with torch.inference_mode():
        y = blk(torch.from_numpy(g["x"]).to(dev))
    assert blk._last_backend == "hip" and blk._hip_calls == 1
    assert blk.period_selector.last_selected_periods.tolist() == g["periods"].tolist()
    assert blk.period_selector.last_frequency_indices.tolist() == g["freq_idx"].tolist()
    assert blk._last_group_count == case["groups"]
    assert blk._last_raw_period_count == len(g["periods"])
    np.testing.assert_allclose(y.cpu().numpy(), g["y"], rtol=RTOL, atol=ATOL)


class _Stub(torch.nn.Module):
    def __init__(self, periods, amps):
        super().__init__()
        self.periods = torch.as_tensor(periods, dtype=torch.long)
        self.amps = torch.as_tensor(amps, dtype=torch.float32)

    def forward(self, x):
        a = self.amps
        if a.dim() == 1:
            a = a.unsqueeze(0)
        if a.size(0) == 1 and x.size(0) > 1:
            a = a.expand(x.size(0), -1)
        return self.periods.to(x.device), a.to(device=x.device, dtype=x.dtype)


STUBS = ["s_dup", "s_mixed_pad", "s_448", "s_wide_rows", "s_p1"]


@pytest.mark.parametrize("engine", ENGINES)
@pytest.mark.parametrize("name", STUBS)
def test_stub_selector_matches_reference(name, engine, manifest, golden, ftn, dev):
    """host grouping -> uploaded descriptor -> HIP conv path (reference tests inject selectors
    the same way, tests/test_times_block.py:93)"""
    case, g = manifest[name], golden(name)
    blk, _, _, _ = _block(ftn, case, dev, engine)
    object.__setattr__(blk, "period_selector", _Stub(g["periods"], g["amps"]))
    with torch.inference_mode():
        y = blk(torch.from_numpy(g["x"]).to(dev))
    assert blk._last_backend == "hip"
    assert blk._last_group_count == case["groups"]
    np.testing.assert_allclose(y.cpu().numpy(), g["y"], rtol=RTOL, atol=ATOL)


def test_invalid_periods_are_identity(ftn, dev):
    """reference tests/test_times_block.py:123-136"""
    T = ftn.models.timesnet
    blk = T.TimesBlock(2, [(3, 3)], 0.0, "gelu").eval().to(dev)
    object.__setattr__(blk, "period_selector", _Stub([0, -1], [1.0, 1.0]))
    x = torch.randn(2, 5, 2, device=dev)
    with torch.inference_mode():
        assert torch.equal(blk(x), x)
    blk.period_selector = T.FFTPeriodSelector(0, 5)          # k = 0 -> no periods
    with torch.inference_mode():
        assert torch.equal(blk(x), x)


@pytest.mark.parametrize("name", ["lrtc_a", "lrtc_b", "lrtc_c"])
def test_lrtc_matches_reference(name, manifest, golden, ftn, dev):
    case, g = manifest[name], golden(name)
    mod = ftn.models.timesnet.LowRankTemporalContext(case["R"], case["scale"]).to(dev)
    coeff = torch.from_numpy(g["coeff"]).to(dev)
    with torch.inference_mode():
        ctx = mod(coeff, case["L"])
        fused = mod(coeff, case["L"], add_to=torch.ones(case["B"], case["L"], case["N"], device=dev))
    assert mod._last_backend == "hip"
    scale = float(np.abs(g["ctx"]).max())
    np.testing.assert_allclose(ctx.cpu().numpy(), g["ctx"], rtol=RTOL, atol=1e-5 * scale)
    np.testing.assert_allclose(fused.cpu().numpy() - 1.0, g["ctx"], rtol=1e-3, atol=1e-6 + 1e-5 * scale)
    basis = ftn.runtime.state(dev).lrtc_basis(case["L"], case["R"]).cpu().numpy()
    np.testing.assert_allclose(basis[: case["L"] * case["R"]].reshape(case["L"], case["R"]), g["basis"],
                               rtol=1e-5, atol=1e-6)


# ---- same seeded inputs vs the oracle, sizes the oracle finishes in seconds -------
@pytest.mark.parametrize("B,L,C,K,hyper,seed", [
    (8, 336, 64, 3, "pipeline", 11), (8, 336, 64, 5, "pipeline", 12), (6, 336, 64, 3, "minimal", 13),
    (4, 720, 128, 3, "pipeline", 14), (5, 200, 24, 4, "rect", 15), (3, 97, 5, 3, "wide1", 16),
    (32, 96, 16, 2, "minimal", 17),
])
@pytest.mark.parametrize("engine", ENGINES)
def test_block_matches_oracle_seeded(B, L, C, K, hyper, seed, engine, ftn, dev):
    case = dict(hyper=hyper, C=C, seed=seed)
    blk, P, ks, act = _block(ftn, case, dev, engine)
    blk.period_selector = ftn.models.timesnet.FFTPeriodSelector(K, L)
    x = torch.from_numpy(ftn.synth.make_input(B, L, C, seed=seed))
    y_ref, aux = orc.timesblock_forward(x, P, ks, act, K, L)
    with torch.inference_mode():
        y = blk(x.to(dev))
    assert blk._last_backend == "hip"
    assert blk.period_selector.last_selected_periods.tolist() == aux.sel.periods
    assert blk._last_group_count == len(aux.groups.periods)
    np.testing.assert_allclose(y.cpu().numpy(), y_ref.numpy(), rtol=RTOL, atol=ATOL)


# ---- awkward geometries through the stub path: multi-tile grids, ragged batch chunks,
#      period L-1 (two very wide rows), period 1/2/3 (very tall grids), single batch row
@pytest.mark.parametrize("engine", ENGINES)
@pytest.mark.parametrize("B,L,C,hyper,periods", [
    (5, 720, 64, "pipeline", [24, 719, 7]),          # 30x24 (2 row tiles), 2x719 (5 column tiles), 103x7
    (7, 336, 64, "pipeline", [335, 2, 100]),         # 2x335, 168x2, 4x100 (pad 64): B not a multiple of 4
    (1, 97, 64, "pipeline", [1, 3, 96]),             # period 1 (97x1), 33x3, 2x96 (pad 95)
    (3, 400, 16, "pipeline", [64, 65, 399]),         # C=16: fp32 stage C beside the bf16 conv engine
    (2, 336, 64, "minimal", [24, 168]),              # single-conv mode (always the fp32 engine)
    (6, 1024, 64, "pipeline", [512, 16, 37]),        # long window
    (3, 250, 128, "pipeline", [7, 249, 60]),         # d_model 128 (k_mlp_bf_c128): ragged last workgroup, pads
])
def test_awkward_geometries_match_oracle(B, L, C, hyper, periods, engine, ftn, dev):
    case = dict(hyper=hyper, C=C, seed=21)
    blk, P, ks, act = _block(ftn, case, dev, engine)
    rs = np.random.RandomState(5)
    amps = rs.standard_normal(size=(B, len(periods))).astype(np.float32)
    object.__setattr__(blk, "period_selector", _Stub(periods, amps))
    x = torch.from_numpy(ftn.synth.make_input(B, L, C, seed=9, planted=()))
    y_ref, aux = orc.timesblock_forward(x, P, ks, act, 0, L, 1, periods=periods, amps=torch.from_numpy(amps))
    with torch.inference_mode():
        y = blk(x.to(dev))
    assert blk._last_backend == "hip" and blk._last_group_count == len(aux.groups.periods)
    np.testing.assert_allclose(y.cpu().numpy(), y_ref.numpy(), rtol=RTOL, atol=ATOL)


# ---- position-major stage C (k_mlp_pos): more groups than one pass holds (5 with two activation pieces, 4 with
#      three), every group padded (tail pixels t >= L, several 16-pixel tail units per row), ragged last unit of a row
@pytest.mark.parametrize("engine", ["bf16x3", "f16x2", "bf16"])
@pytest.mark.parametrize("B,L,periods", [
    (3, 336, [5, 9, 16, 24, 33, 50, 100]),           # 7 groups: two passes; pads 4, 6, 0, 0, 27, 14, 64
    (2, 150, [70, 7, 11, 13, 17, 19, 23, 29, 31]),   # 9 groups, L % 16 != 0; period 70 pads 60 = four tail units per row
    (9, 100, [33]),                                  # one group, pad 32: tail units of exactly 16 pixels
])
def test_position_major_passes_and_tails(B, L, periods, engine, ftn, dev):
    case = dict(hyper="pipeline", C=64, seed=23)
    blk, P, ks, act = _block(ftn, case, dev, engine)
    rs = np.random.RandomState(8)
    amps = rs.standard_normal(size=(B, len(periods))).astype(np.float32)
    object.__setattr__(blk, "period_selector", _Stub(periods, amps))
    x = torch.from_numpy(ftn.synth.make_input(B, L, 64, seed=10, planted=()))
    y_ref, aux = orc.timesblock_forward(x, P, ks, act, 0, L, 1, periods=periods, amps=torch.from_numpy(amps))
    with torch.inference_mode():
        y = blk(x.to(dev))
    assert blk._last_backend == "hip" and blk._last_group_count == len(aux.groups.periods)
    if engine == "bf16":                                        # plain bf16 products: BASELINE configs[2], not an fp32 claim
        assert float((y.cpu() - y_ref).abs().max()) < 0.05 * float(y_ref.abs().max())
    else:
        np.testing.assert_allclose(y.cpu().numpy(), y_ref.numpy(), rtol=RTOL, atol=ATOL)


def test_engines_agree_and_plain_bf16_is_close(ftn, dev):
    """f32 and bf16x3 differ at rounding level; plain bf16 (BASELINE configs[2]) is a
    reduced-precision path with its own tolerance."""
    case = dict(hyper="pipeline", C=64, seed=4)
    x = torch.from_numpy(ftn.synth.make_input(8, 336, 64, seed=4)).to(dev)
    ys = {}
    for eng in ("f32", "bf16x3", "f16x2", "bf16"):
        blk, _, _, _ = _block(ftn, case, dev, eng)
        blk.period_selector = ftn.models.timesnet.FFTPeriodSelector(5, 336)
        with torch.inference_mode():
            ys[eng] = blk(x).cpu().numpy()
    np.testing.assert_allclose(ys["bf16x3"], ys["f32"], rtol=2e-5, atol=5e-6)
    np.testing.assert_allclose(ys["f16x2"], ys["f32"], rtol=2e-5, atol=5e-6)
    np.testing.assert_allclose(ys["bf16"], ys["f32"], rtol=5e-2, atol=5e-2)
    assert np.abs(ys["bf16"] - ys["f32"]).max() > 1e-5          # it really is a different arithmetic


# ---- BASELINE full size (B=256 L=336 C=64): size-independent properties ---------
def test_full_size_properties(ftn, dev):
    B, L, C, K = 256, 336, 64, 5
    case = dict(hyper="pipeline", C=C, seed=0)
    blk, P, ks, act = _block(ftn, case, dev)
    blk.period_selector = ftn.models.timesnet.FFTPeriodSelector(K, L)
    xh = ftn.synth.make_input(B, L, C, seed=0)
    x = torch.from_numpy(xh).to(dev)
    with torch.inference_mode():
        y1 = blk(x)
        y2 = blk(x)
    assert torch.equal(y1, y2)                                    # deterministic (no atomics)
    periods = blk.period_selector.last_selected_periods.tolist()
    assert sorted(periods) == sorted([24, 168, 7, 12, 84])        # the planted periods
    assert torch.isfinite(y1).all()
    # with the periods and per-row amplitudes fixed, rows are independent: any sub-batch,
    # in any order, must give the same rows -> check against the CPU oracle on 4 rows
    amps = blk.period_selector(x)[1].cpu()
    rows = [0, 17, 128, 255]
    y_ref, _ = orc.timesblock_forward(torch.from_numpy(xh[rows]), P, ks, act, K, L, periods=periods, amps=amps[rows])
    np.testing.assert_allclose(y1[rows].cpu().numpy(), y_ref.numpy(), rtol=RTOL, atol=ATOL)
    # batch permutation equivariance through the stub path (same descriptor, permuted rows)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(0))
    object.__setattr__(blk, "period_selector", _Stub(periods, amps[perm].numpy()))
    with torch.inference_mode():
        yp = blk(x[perm.to(dev)])
    np.testing.assert_allclose(yp.cpu().numpy(), y1[perm.to(dev)].cpu().numpy(), rtol=1e-5, atol=1e-6)


def test_half_precision_input_roundtrip(ftn, dev):
    """bf16 activations keep fp32 parameters and fp32 compute; output dtype == input dtype
    (reference :30-34, :941, :1068-1069)."""
    case = dict(hyper="pipeline", C=16, seed=3)
    blk, P, ks, act = _block(ftn, case, dev)
    blk.period_selector = ftn.models.timesnet.FFTPeriodSelector(3, 96)
    x = torch.from_numpy(ftn.synth.make_input(4, 96, 16, seed=3, planted=(24, 12, 8))).to(dev)
    with torch.inference_mode():
        y32 = blk(x)
        y16 = blk(x.bfloat16())
    assert y16.dtype == torch.bfloat16
    assert next(blk.inception.parameters()).dtype == torch.float32
    np.testing.assert_allclose(y16.float().cpu().numpy(), y32.cpu().numpy(), rtol=0.05, atol=0.1)


# ---- full model shell: mirror TimesNet with HIP blocks + HIP LRTC vs the reference ----
def _model_from_fixture(ftn, case, g, dev):
    cfg = dict(case["cfg"])
    cfg["kernel_set"] = [tuple(k) for k in cfg["kernel_set"]]
    model = ftn.models.TimesNet(**cfg).eval()
    kw = {}
    for key in ("series_static", "series_ids", "x_mark"):
        if key in g:
            kw[key] = torch.from_numpy(g[key])
    with torch.no_grad():
        model(torch.from_numpy(g["x"]), **kw)                     # materialise lazy layers on CPU
    sd = {k[4:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd::")}
    d_ff = cfg["d_ff"] if cfg.get("d_ff") else cfg["d_model"]
    for li in range(cfg["n_layers"]):
        prm = ftn.synth.make_inception_params(cfg["d_model"], d_ff, cfg["kernel_set"],
                                              cfg.get("bottleneck_ratio", 1.0), seed=100 + li)
        for k, v in prm.items():
            sd[f"blocks.{li}.inception.{k}"] = torch.from_numpy(v)
    model.load_state_dict(sd, strict=True)
    return model.to(dev), {k: v.to(dev) for k, v in kw.items()}


@pytest.mark.parametrize("name", ["m_context", "m_pipeline"])
def test_full_model_matches_reference(name, manifest, golden, ftn, dev):
    case, g = manifest[name], golden(name)
    model, kw = _model_from_fixture(ftn, case, g, dev)
    with torch.inference_mode():
        rate, disp = model(torch.from_numpy(g["x"]).to(dev), **kw)
    assert all(b._last_backend == "hip" for b in model.blocks)
    assert model._last_head_backend == "hip" and model._last_embed_backend == "hip"
    if model.temporal_context is not None:
        assert model.temporal_context._last_backend == "fused"
    assert model.period_selector.last_selected_periods.tolist() == g["periods"].tolist()
    np.testing.assert_allclose(rate.cpu().numpy(), g["rate"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(disp.cpu().numpy(), g["dispersion"], rtol=RTOL, atol=ATOL)


def test_lazy_build_under_inference_mode(ftn, dev):
    """Parameters created by the lazy build inside torch.inference_mode are inference tensors (no
    version counter); the packed-weight cache must cope and the HIP path must still be the one used."""
    T = ftn.models.timesnet
    blk = T.TimesBlock(None, [3, 5], 0.0, "gelu", d_ff=None, bottleneck_ratio=2.0).eval()
    blk.period_selector = T.FFTPeriodSelector(3, 48)
    x = torch.from_numpy(ftn.synth.make_input(4, 48, 32, seed=3, planted=(12, 8, 6))).to(dev)
    with torch.inference_mode():
        y1 = blk(x)
        y2 = blk(x)
    assert blk._last_backend == "hip"
    assert torch.equal(y1, y2)
    P = {k: v.detach().cpu() for k, v in blk.inception.state_dict().items()}
    ref, _ = orc.timesblock_forward(x.cpu(), P, [(3, 3), (5, 5)], "gelu", 3, 48, 1)
    np.testing.assert_allclose(y1.cpu().numpy(), ref.numpy(), rtol=RTOL, atol=ATOL)


# ---- per-block residual + shared LayerNorm epilogue (reference TimesNet.forward :2050-2058) ----
@pytest.mark.parametrize("hyper,C,L", [("pipeline", 64, 96), ("pipeline", 24, 50), ("pipeline", 128, 48),
                                       ("minimal", 16, 96), ("rect", 32, 60)])
@pytest.mark.parametrize("engine", ENGINES)
def test_post_norm_epilogue(hyper, C, L, engine, ftn, dev):
    """block(x, post_norm=ln) == LayerNorm(x + (block(x) - x)): fused into k_out for d_model <= 64 in
    bottleneck mode, the in-place row kernel otherwise; checked against torch ops on the HIP block's own
    output and against the CPU oracle end to end."""
    case = dict(hyper=hyper, C=C, seed=11)
    blk, P, ks, act = _block(ftn, case, dev, engine)
    blk.period_selector = ftn.models.timesnet.FFTPeriodSelector(3, L)
    ln = torch.nn.LayerNorm(C).to(dev)
    with torch.no_grad():
        ln.weight.copy_(torch.linspace(0.5, 1.5, C))
        ln.bias.copy_(torch.linspace(-0.2, 0.3, C))
    x = torch.from_numpy(ftn.synth.make_input(5, L, C, seed=4, planted=(12, 8, 6))).to(dev)
    with torch.inference_mode():
        y = blk(x)
        z = blk(x, post_norm=ln)
        want = torch.nn.functional.layer_norm(x + (y - x), (C,), ln.weight, ln.bias, ln.eps)
    assert blk._last_backend == "hip"
    np.testing.assert_allclose(z.cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=2e-6)
    ref, _ = orc.timesblock_forward(x.cpu(), P, ks, act, 3, L, 1)
    xc = x.cpu()
    with torch.no_grad():
        want_o = torch.nn.functional.layer_norm(xc + (ref - xc), (C,), ln.weight.cpu(), ln.bias.cpu(), ln.eps)
    np.testing.assert_allclose(z.cpu().numpy(), want_o.numpy(), rtol=RTOL, atol=5e-5)


def test_post_norm_when_block_is_identity(ftn, dev):
    """No valid period -> the block returns x (:796-797) and the shell normalises x + (x - x)."""
    case = dict(hyper="pipeline", C=16, seed=2)
    blk, P, ks, act = _block(ftn, case, dev)

    class Stub(torch.nn.Module):
        def forward(self, x):
            return torch.tensor([0, -3], device=x.device), torch.ones(x.size(0), 2, device=x.device)

    blk.period_selector = Stub()
    ln = torch.nn.LayerNorm(16).to(dev)
    x = torch.randn(3, 40, 16, device=dev)
    with torch.inference_mode():
        z = blk(x, post_norm=ln)
        want = ln(x)
    np.testing.assert_allclose(z.cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=2e-6)


# ---- fused rate / dispersion heads (reference TimesNet.forward :2066-2102) ----
@pytest.mark.parametrize("B,S,D,N,hist,late,floor", [
    (3, 6, 8, 5, 6, "shared", "scalar"),        # ragged N (scalar loads/stores)
    (2, 12, 64, 512, 12, "batch", "vector"),    # vector path, 8 series slices
    (4, 5, 16, 72, 3, None, "scalar"),          # history shorter than the horizon: tail edge-padded
    (2, 1, 128, 200, 1, "shared", "vector"),    # recursive mode (one step), d_model 128
    (1, 24, 24, 8, 24, "batch", "scalar"),      # d_model not a multiple of 16
])
def test_heads_match_oracle(B, S, D, N, hist, late, floor, ftn, dev):
    g = torch.Generator().manual_seed(B * 100 + S)
    T = hist + 5
    x = 3.0 * torch.randn(B, T, N, generator=g)
    hidden = torch.randn(B, S, D, generator=g)
    w_mu, w_sg = 0.3 * torch.randn(N, D, generator=g), 0.3 * torch.randn(N, D, generator=g)
    b_mu, b_sg = torch.randn(N, generator=g), torch.randn(N, generator=g)
    hidden[0, 0] *= 30.0                          # drive some pre-activations past the softplus threshold
    lt = None if late is None else 0.5 * torch.randn(B if late == "batch" else 1, S, N, generator=g)
    fv = torch.rand(N, generator=g) if floor == "vector" else None
    tail = x[:, -hist:, :]
    tail_full = tail if hist == S else torch.cat([tail, tail[:, -1:, :].expand(-1, S - hist, -1)], dim=1)
    # the oracle's dot products in fp64: with hidden[0, 0] x 30 the sum of |terms| reaches ~1e3, where an fp32 F.linear's
    # own rounding (1e-7 of that) is as large as the tolerance below; the kernel's bf16x3 products carry ~1e-8 of it
    dd = lambda t: None if t is None else t.double()
    want_r, want_d = orc.model_heads(dd(hidden), dd(w_mu), dd(b_mu), dd(w_sg), dd(b_sg), dd(tail_full), dd(lt),
                                     fv.view(1, 1, N) if fv is not None else 1e-3)
    to = lambda t: None if t is None else t.to(dev)
    xd = x.to(dev)
    rate, disp, bad = ftn.runtime.head_forward(to(hidden), to(w_mu), to(b_mu), to(w_sg), to(b_sg), xd[:, -hist:, :],
                                               hist, to(lt), to(fv), 1e-3)
    assert int(bad.item()) == 0
    np.testing.assert_allclose(rate.cpu().numpy(), want_r.numpy(), rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(disp.cpu().numpy(), want_d.numpy(), rtol=2e-5, atol=1e-6)


def test_heads_flag_bad_outputs(ftn, dev):
    """A non-finite pre-activation must surface (the reference raises RuntimeError, :2095-2098)."""
    B, S, D, N = 2, 4, 8, 8
    hidden = torch.randn(B, S, D, device=dev)
    w = torch.randn(N, D, device=dev)
    b = torch.zeros(N, device=dev)
    x = torch.randn(B, S, N, device=dev)
    x[1, 2, 3] = float("inf")
    _, _, bad = ftn.runtime.head_forward(hidden, w, b, w, b, x, S, None, None, 1e-3)
    assert int(bad.item()) == 1
    hidden[0, 1, 0] = float("nan")
    _, _, bad = ftn.runtime.head_forward(hidden, w, b, w, b, x, S, None, None, 1e-3)
    assert int(bad.item()) == 3
    model = ftn.models.TimesNet(input_len=16, pred_len=4, d_model=8, n_layers=1, k_periods=2, kernel_set=[(3, 3)],
                                dropout=0.0, activation="gelu", mode="direct").eval()
    xin = torch.randn(2, 16, 8, device=dev)
    xin[0, -1, 0] = float("inf")
    with torch.inference_mode(), pytest.raises(RuntimeError, match="rate must be finite"):
        model(xin)


# ---- value embedding with the context front-end folded in (reference :1958-1996, :1283-1325) ----
@pytest.mark.parametrize("B,L,N,D,add_b,ln,T", [
    (3, 24, 5, 8, "shared", False, 24),       # ragged N: scalar loads
    (2, 336, 512, 64, "shared", False, 340),  # headline shape of one window pair, window = view into a longer x
    (4, 40, 72, 128, "batch", True, 40),      # d_model 128, per-sample add, LayerNorm epilogue
    (5, 17, 200, 24, "batch", True, 17),      # rows not a multiple of the tile, d_model not a multiple of 16
    (1, 50, 64, 16, None, False, 50),
])
def test_embed_matches_oracle(B, L, N, D, add_b, ln, T, ftn, dev):
    g = torch.Generator().manual_seed(L * 7 + N)
    x = torch.randn(B, T, N, generator=g)
    w = torch.randn(D, N, generator=g) / N ** 0.5
    bias = torch.randn(D, generator=g)
    aux = None if add_b is None else torch.randn(B if add_b == "batch" else 1, L, D, generator=g)
    lnp = (torch.rand(D, generator=g) + 0.5, torch.randn(D, generator=g), 1e-5) if ln else None
    win = x[:, -L:, :]
    want = orc.embed_front(win, w, bias, aux if aux is not None else torch.zeros(1, L, D), ln=lnp)
    add = bias.view(1, 1, D).expand(1, L, D) if aux is None else aux + bias
    to = lambda t: t.to(dev)
    out = ftn.runtime.embed_forward(to(x)[:, -L:, :], to(w), to(add.contiguous()),
                                    None if lnp is None else (to(lnp[0]), to(lnp[1]), lnp[2]))
    np.testing.assert_allclose(out.cpu().numpy(), want.numpy(), rtol=2e-5, atol=2e-5)


def test_embed_context_fold_matches_oracle(ftn, dev):
    """The temporal context and the constant context bias enter through ``add`` (pushed through W)
    instead of being added to x first: same result as the reference order of operations."""
    B, L, N, D, R = 3, 48, 20, 16, 4
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, L, N, generator=g)
    w, bias = torch.randn(D, N, generator=g) / N ** 0.5, torch.randn(D, generator=g)
    coeff = torch.randn(B, N, R, generator=g)
    cb = 0.3 * torch.randn(B, N, generator=g)
    aux = torch.randn(1, L, D, generator=g)
    lrtc = ftn.models.LowRankTemporalContext(R, 0.05)
    signal = orc.lrtc_forward(coeff, L, torch.tensor(0.05))
    want = orc.embed_front(x, w, bias, aux, signal=signal, cbias=cb)
    lrtc = lrtc.to(dev)
    with torch.inference_mode():
        add = aux.to(dev) + bias.to(dev) + lrtc.project(coeff.to(dev), L, w.to(dev)) \
            + torch.matmul(cb.to(dev), w.to(dev).t()).unsqueeze(1)
        out = ftn.runtime.embed_forward(x.to(dev), w.to(dev), add.contiguous(), None)
    np.testing.assert_allclose(out.cpu().numpy(), want.numpy(), rtol=2e-5, atol=2e-5)


# ---- HIP-graph replay of whole forwards --------------------------------------------------------
def test_graphed_model_forward_matches_eager_and_reference(manifest, golden, ftn, dev):
    case, g = manifest["m_context"], golden("m_context")
    model, kw = _model_from_fixture(ftn, case, g, dev)
    x = torch.from_numpy(g["x"]).to(dev)
    with torch.inference_mode():
        rate_e, disp_e = model(x, **kw)
    gf = ftn.graph.GraphedForward(model, x, **kw)
    rate, disp = gf(x, **kw)
    assert torch.equal(rate, rate_e) and torch.equal(disp, disp_e)
    np.testing.assert_allclose(rate.cpu().numpy(), g["rate"], rtol=RTOL, atol=ATOL)
    assert model.period_selector.last_selected_periods.tolist() == g["periods"].tolist()
    # new data through the same graph: period selection happens on the device inside the replay
    x2 = torch.roll(x, 3, dims=1) * 1.5
    with torch.inference_mode():
        want = model(x2, **kw)
    got = gf(x2, **kw)
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    # the deferred finite-positive check still raises
    x3 = x.clone()
    x3[0, -1, 0] = float("inf")
    with pytest.raises(RuntimeError, match="rate must be finite"):
        gf(x3, **kw)


def test_graphed_block_forward(ftn, dev):
    case = dict(hyper="pipeline", C=32, seed=5)
    blk, P, ks, act = _block(ftn, case, dev)
    blk.period_selector = ftn.models.timesnet.FFTPeriodSelector(3, 96)
    x = torch.from_numpy(ftn.synth.make_input(4, 96, 32, seed=1, planted=(24, 12, 8))).to(dev)
    gf = ftn.graph.GraphedForward(blk, x)
    y = gf(x).clone()
    with torch.inference_mode():
        want = blk(x)
    assert torch.equal(y, want)
    x2 = torch.from_numpy(ftn.synth.make_input(4, 96, 32, seed=2, planted=(16, 6, 32))).to(dev)
    with torch.inference_mode():
        want2 = blk(x2)
    assert torch.equal(gf(x2), want2)
    assert not torch.equal(want2, want)


def test_pipeline_shaped_call_hip_vs_cpu_mirror(ftn, dev):
    """How the reference's predict_once calls the model (predict.py:797-966, one series per batch row):
    x[B=193, L=28, N=1] with per-sample static features [B,1,F] and ids [B,1].  The CPU mirror is pinned to
    the reference by tests/test_predict_pipeline_dropin.py; here the HIP path must reproduce it."""
    B, L, H = 193, 28, 7
    cfg = dict(input_len=L, pred_len=H, d_model=16, d_ff=32, n_layers=2, k_periods=2, kernel_set=[3, 5], dropout=0.0,
               activation="gelu", mode="direct", bottleneck_ratio=2.0, id_embed_dim=6, static_proj_dim=5,
               use_zero_mean_context=True, context_rank=3, context_scale=0.05, use_constant_context_bias=True)
    g = torch.Generator().manual_seed(11)
    t = torch.arange(L, dtype=torch.float32).view(1, L, 1)
    x = torch.rand(B, L, 1, generator=g) * 5.0 + 2.0 * torch.sin(2 * torch.pi * t / 7.0)
    static = torch.randn(B, 1, 4, generator=g)
    ids = torch.arange(B).view(B, 1)
    torch.manual_seed(0)
    cpu = ftn.models.TimesNet(**cfg).eval()
    warm = dict(series_static=static[:1, 0], series_ids=torch.tensor([B - 1]))   # max id sizes the table (:1437)
    with torch.no_grad():
        cpu(torch.zeros(1, L, 1), **warm)
        for p in cpu.parameters():
            if float(p.abs().sum()) == 0.0:
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
        want_r, want_d = cpu(x, series_static=static, series_ids=ids)
    gpu = ftn.models.TimesNet(**cfg).eval()
    with torch.no_grad():
        gpu(torch.zeros(1, L, 1, device=dev), **{k: v.to(dev) for k, v in warm.items()})
    gpu.load_state_dict(cpu.state_dict(), strict=True)
    with torch.inference_mode():
        rate, disp = gpu(x.to(dev), series_static=static.to(dev), series_ids=ids.to(dev))
    assert all(b._last_backend == "hip" for b in gpu.blocks)
    assert gpu._last_head_backend == "hip" and gpu._last_embed_backend == "hip"
    assert gpu.period_selector.last_selected_periods.tolist() == cpu.period_selector.last_selected_periods.tolist()
    np.testing.assert_allclose(rate.cpu().numpy(), want_r.numpy(), rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(disp.cpu().numpy(), want_d.numpy(), rtol=RTOL, atol=ATOL)


def test_gelu_accuracy(ftn, dev):
    """The kernels' GELU (one v_exp_f32 + a degree-6 polynomial, ftn_common.h) against the erf form in fp64."""
    lib = ftn.lib.load()
    v = torch.cat([torch.linspace(-12, 12, 1_000_001), torch.tensor([0.0, -0.0, 1e-30, -1e-30, 50.0, -50.0, 3e4, -3e4])])
    v = torch.cat([v, torch.zeros((-v.numel()) % 4)]).to(dev)
    out = torch.empty_like(v)
    ftn.lib.check(lib.ftn_selftest_gelu(v.data_ptr(), out.data_ptr(), v.numel(),
                                        torch.cuda.current_stream().cuda_stream), "ftn_selftest_gelu")
    want = torch.nn.functional.gelu(v.double().cpu())
    err = (out.double().cpu() - want).abs()
    assert float(err.max()) < 4e-7, float(err.max())
    big = v.cpu().abs() > 12
    assert float((out.cpu()[big] - torch.clamp(v.cpu()[big], min=0.0)).abs().max()) < 1e-13   # |v| clamps at 8


@pytest.mark.parametrize("K,L,C,hyper", [(12, 720, 64, "pipeline"), (16, 336, 32, "pipeline"), (16, 200, 16, "minimal")])
def test_many_periods(K, L, C, hyper, ftn, dev):
    """k_periods up to FTN_KMAX = 16: descriptor capacity, worst-case grids / workspace for many groups, and
    duplicate periods collapsing into fewer groups (white-noise input: the candidates are whatever top-k finds)."""
    case = dict(hyper=hyper, C=C, seed=31)
    blk, P, ks, act = _block(ftn, case, dev)
    blk.period_selector = ftn.models.timesnet.FFTPeriodSelector(K, L)
    x = torch.from_numpy(ftn.synth.make_input(3, L, C, seed=31, planted=(24, 7, 12, 50, 9, 33, 5, 16)))
    y_ref, aux = orc.timesblock_forward(x, P, ks, act, K, L)
    with torch.inference_mode():
        y = blk(x.to(dev))
    assert blk._last_backend == "hip"
    assert blk.period_selector.last_selected_periods.tolist() == aux.sel.periods
    assert blk._last_group_count == len(aux.groups.periods) and blk._last_raw_period_count == len(aux.sel.periods)
    np.testing.assert_allclose(y.cpu().numpy(), y_ref.numpy(), rtol=RTOL, atol=ATOL)


# ---- wide blocks: more than 16 stage-C output tiles, or a hidden chunk beyond the LDS budget, run
#      the chain as generic pointwise launches; the reference's search space tunes d_model up to 512
@pytest.mark.parametrize("engine", ["f16x2", "f32"])
@pytest.mark.parametrize("C,d_ff,ratio,ks,L,B", [
    (192, 768, 4.0, [(3, 3), (5, 5), (7, 7)], 96, 2),      # 9 + 12 = 21 output tiles
    (256, 1024, 4.0, [(3, 3), (5, 5), (7, 7)], 96, 2),     # 12 + 16 = 28
    (512, 2048, 4.0, [(3, 3), (5, 5)], 48, 2),             # mid 128: 16 + 32 = 48, hidden chunks of 2 x 64 fragments
    (128, 512, 2.0, [(3, 3), (5, 5), (7, 7)], 96, 3),      # ratio 2: mid 64 -> 12 + 8 = 20
    (384, None, 4.0, [(3, 3), (5, 5), (7, 7)], 60, 2),     # d_ff == d_model: identity res_proj twice, 18 tiles
    (72, 288, 1.5, [(3, 3), (5, 5)], 80, 3),               # odd ratio: mid 48, d_model not a multiple of 16
])
def test_wide_blocks_match_oracle(C, d_ff, ratio, ks, L, B, engine, ftn, dev):
    T = ftn.models.timesnet
    blk = T.TimesBlock(C, ks, 0.0, "gelu", d_ff=d_ff, bottleneck_ratio=ratio)
    blk.engine = engine
    sd = ftn.synth.make_inception_params(C, d_ff if d_ff is not None else C, ks, ratio, 41)
    blk.inception.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    blk = blk.eval().to(dev)
    blk.period_selector = T.FFTPeriodSelector(3, L)
    x = torch.from_numpy(ftn.synth.make_input(B, L, C, seed=8, planted=(24, 12, 8)))
    P = {k: torch.from_numpy(v) for k, v in sd.items()}
    y_ref, aux = orc.timesblock_forward(x, P, ks, "gelu", 3, L)
    ln = torch.nn.LayerNorm(C).to(dev)
    with torch.inference_mode():
        y = blk(x.to(dev))
        z = blk(x.to(dev), post_norm=ln)
    assert blk._last_backend == "hip"
    assert blk.period_selector.last_selected_periods.tolist() == aux.sel.periods
    np.testing.assert_allclose(y.cpu().numpy(), y_ref.numpy(), rtol=RTOL, atol=ATOL)
    with torch.inference_mode():
        want = torch.nn.functional.layer_norm(y, (C,), ln.weight, ln.bias, ln.eps)
    np.testing.assert_allclose(z.cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=3e-6)
