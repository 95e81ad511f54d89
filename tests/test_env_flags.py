"""The reference's environment-flag paths pinned to fixtures the reference itself
produced (tests/golden/make_golden_env.py): TIMES_PERIOD_MAX_UNIQ, TIMES_PERIOD_BINNING, per-depth schedules with
``block_index``, and the TIMESBLOCK_VEC_DISABLE loop path (reference models/timesnet.py:162-272, 350-437, 806,
820-953).  CPU: the host PeriodGrouper mirror and the mirror TimesBlock (torch backend); ``-m gpu``: the HIP block
fed the flagged grouping through ftn_desc_from_periods."""
import json

import numpy as np
import pytest
import torch

from conftest import GOLDEN

_ALL = json.loads((GOLDEN / "manifest_env.json").read_text())["cases"]
ENV_CASES = {k: v for k, v in _ALL.items() if v["kind"] != "half"}
HALF_CASES = {k: v for k, v in _ALL.items() if v["kind"] == "half"}
DT = {"bfloat16": torch.bfloat16, "float16": torch.float16}
HYP = json.loads((GOLDEN / "manifest.json").read_text())["hypers"]
FLAGS = ("TIMES_PERIOD_MAX_UNIQ", "TIMES_PERIOD_BINNING", "TIMESBLOCK_VEC_DISABLE")


def _load(name):
    with np.load(GOLDEN / f"{name}.npz") as z:
        return {k: z[k] for k in z.files}


def _setenv(monkeypatch, env):
    for f in FLAGS:
        monkeypatch.delenv(f, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)


class _Stub(torch.nn.Module):
    def __init__(self, periods, amps):
        super().__init__()
        self.periods = torch.as_tensor(periods, dtype=torch.long)
        self.amps = torch.as_tensor(amps, dtype=torch.float32)

    def forward(self, x):
        a = self.amps
        if a.size(0) == 1 and x.size(0) > 1:
            a = a.expand(x.size(0), -1)
        return self.periods.to(x.device), a.to(device=x.device, dtype=x.dtype)


def _block(ftn, case, device="cpu"):
    h = HYP[case["hyper"]]
    C = case["C"]
    d_ff = C if h["d_ff_mult"] is None else C * h["d_ff_mult"]
    ks = [tuple(k) for k in h["kernel_set"]]
    blk = ftn.models.timesnet.TimesBlock(C, ks, 0.0, h["act"], d_ff=None if h["d_ff_mult"] is None else d_ff,
                                         bottleneck_ratio=h["ratio"])
    sd = ftn.synth.make_inception_params(C, d_ff, ks, h["ratio"], case["seed"])
    blk.inception.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return blk.eval().to(device)


@pytest.mark.parametrize("name", sorted(ENV_CASES))
def test_host_grouper_matches_reference_under_flags(name, ftn, monkeypatch):
    case, g = ENV_CASES[name], _load(name)
    _setenv(monkeypatch, case["env"])
    amps = torch.from_numpy(g["amps"])
    if amps.shape[0] == 1 and case["B"] > 1:
        amps = amps.expand(case["B"], -1)
    kw = dict(min_period=1, max_period=case["L"]) if case["kind"] == "env_native" else {}
    grp = ftn.models.timesnet.PeriodGrouper(torch.from_numpy(g["periods"]), amps, seq_len=case["L"],
                                            block_index=case["block_index"], **kw).group()
    assert grp.periods.tolist() == g["g_periods"].tolist()                      # integers: bit-exact
    assert grp.pad_lengths.tolist() == g["g_pad"].tolist() and grp.cycles.tolist() == g["g_cycles"].tolist()
    assert grp.mapping.tolist() == g["mapping"].tolist() and grp.valid_mask.tolist() == g["valid_mask"].tolist()
    np.testing.assert_allclose(grp.logits.numpy(), g["logits"], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("name", sorted(ENV_CASES))
def test_mirror_block_matches_reference_under_flags(name, ftn, monkeypatch):
    case, g = ENV_CASES[name], _load(name)
    blk = _block(ftn, case)
    if case["kind"] == "env_native":
        blk.period_selector = ftn.models.timesnet.FFTPeriodSelector(case["K"], case["L"])
    else:
        object.__setattr__(blk, "period_selector", _Stub(g["periods"], g["amps"]))
    blk.block_index = case["block_index"]
    _setenv(monkeypatch, case["env"])
    with torch.no_grad():
        y = blk(torch.from_numpy(g["x"]))
    assert blk._last_backend == "torch"
    assert blk._last_group_count == case["groups"] and blk._last_loop_iterations == case["loop_iterations"]
    assert blk._vec_calls == case["vec_calls"] and blk._last_valid_period_count == case["valid"]
    np.testing.assert_allclose(y.numpy(), g["y"], rtol=1e-5, atol=1e-6)


def test_max_unique_reduces_loop_iterations(ftn, monkeypatch):
    """Restates reference tests/test_times_block.py:183-212 on the mirror."""
    torch.manual_seed(5)
    blk = ftn.models.timesnet.TimesBlock(d_model=2, kernel_set=[(3, 3)], dropout=0.0, activation="gelu")
    object.__setattr__(blk, "period_selector", _Stub([3, 4, 6, 12], [[0.5, -0.2, 1.0, -1.5]]))
    x = torch.randn(1, 48, 2)
    monkeypatch.setenv("TIMESBLOCK_VEC_DISABLE", "1")
    blk(x)
    baseline = blk._last_loop_iterations
    monkeypatch.setenv("TIMES_PERIOD_MAX_UNIQ", "2")
    blk(x)
    reduced = blk._last_loop_iterations
    assert baseline > 0 and reduced <= 2 and reduced < baseline


def test_vectorized_matches_loop(ftn, monkeypatch):
    """Restates reference tests/test_timesblock_vectorized.py:43-62 and test_timesblock_maskless.py:31-47."""
    torch.manual_seed(0)
    blk = ftn.models.timesnet.TimesBlock(d_model=4, kernel_set=[(3, 3)], dropout=0.0, activation="gelu").eval()
    object.__setattr__(blk, "period_selector", _Stub([3, 5], [[0.1, -0.4], [1.2, 0.7]]))
    x = torch.randn(2, 28, 4)
    monkeypatch.setenv("TIMESBLOCK_VEC_DISABLE", "1")
    with torch.no_grad():
        loop_out = blk(x)
    assert blk._last_loop_iterations == 2
    monkeypatch.delenv("TIMESBLOCK_VEC_DISABLE", raising=False)
    blk._vec_calls = 0
    with torch.no_grad():
        vec_out = blk(x)
    assert blk._vec_calls >= 1 and blk._last_loop_iterations == 0
    assert float((loop_out - vec_out).abs().max()) < 1e-5


@pytest.mark.parametrize("raw,depth,want", [
    ("0:4,2:2,default:3", 0, 4), ("0:4,2:2,default:3", 1, 4), ("0:4,2:2,default:3", 2, 2), ("0:4,2:2,default:3", 7, 2),
    ("0:4,2:2,default:3", None, 3), ("1:2,default:3", 0, 3), ("5", 3, 5), ("3,7", None, 7), ("2=6;junk", 2, None),
    ("", 1, None), ("0:-1,default:0", 0, None), ("1:2", 0, 2), ("x:3,*:4", 9, 4), ("2.9", None, 2),
])
def test_schedule_parser_cases(raw, depth, want, ftn):
    """The per-depth schedule syntax (reference :162-232): explicit depth keys (largest key <= depth wins),
    ``default`` / ``*``, bare values, non-positive -> None."""
    from flow_timesnet_amd import grouping
    assert grouping._resolve_scheduled_int(raw, depth) == want


@pytest.mark.parametrize("raw,want", [("log:2", 2.0), ("3", 3.0), ("log", 2.0), ("off", None), ("logscale:1.5", 1.5),
                                      ("1.0", None), ("0", None), ("nonsense", 2.0), ("2.5:junk", 2.5)])
def test_log_binning_parser_cases(raw, want, ftn):
    from flow_timesnet_amd import grouping
    assert grouping._resolve_log_binning_base(raw, None) == want


# ---- half-precision inputs: the reference rounds amplitudes, softmax weights, every per-group delta, the weighted
#      terms' sum and x + sum to the input dtype (:124-159, :1000-1009, :1068-1069, :1092, :818)
def _half_block(ftn, case, g, device):
    blk = _block(ftn, case, device)
    if case["K"]:
        blk.period_selector = ftn.models.timesnet.FFTPeriodSelector(case["K"], case["L"])
    else:
        object.__setattr__(blk, "period_selector", _Stub(g["periods"], g["amps"]))
    return blk


@pytest.mark.parametrize("name", sorted(HALF_CASES))
def test_mirror_block_half_input_matches_reference_bitwise(name, ftn):
    case, g = HALF_CASES[name], _load(name)
    blk = _half_block(ftn, case, g, "cpu")
    with torch.no_grad():
        y = blk(torch.from_numpy(g["x"]).to(DT[case["dtype"]]))
    assert y.dtype == DT[case["dtype"]] and blk._last_group_count == case["groups"]
    assert next(blk.inception.parameters()).dtype == torch.float32
    np.testing.assert_array_equal(y.float().numpy(), g["y"])


# ------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("engine", ["f16x2", "f32"])
@pytest.mark.parametrize("name", sorted(HALF_CASES))
def test_hip_block_half_input_matches_reference(name, engine, ftn):
    """The HIP path with bf16 / fp16 activations: same rounding points as the reference, so the output agrees
    with the reference's own half-precision output to one unit in the last place of that dtype (the fp32 conv
    results under the roundings differ at the 1e-6 level, which can flip a rounding)."""
    dev = torch.device("cuda:0")
    case, g = HALF_CASES[name], _load(name)
    dt = DT[case["dtype"]]
    blk = _half_block(ftn, case, g, dev)
    blk.engine = engine
    with torch.inference_mode():
        y = blk(torch.from_numpy(g["x"]).to(dt).to(dev))
    assert blk._last_backend == "hip" and y.dtype == dt and blk._last_group_count == case["groups"]
    if case["K"]:
        assert blk.period_selector.last_selected_periods.tolist() == g["periods"].tolist()
    got, want = y.float().cpu().numpy(), g["y"]
    ulp = np.abs(want) * (2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10) + 1e-30
    assert np.all(np.abs(got - want) <= 1.001 * ulp), float(np.max(np.abs(got - want) / ulp))
    assert np.mean(got == want) > 0.97
@pytest.mark.gpu
@pytest.mark.parametrize("engine", ["f16x2", "f32"])
@pytest.mark.parametrize("name", sorted(ENV_CASES))
def test_hip_block_matches_reference_under_flags(name, engine, ftn, monkeypatch):
    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    case, g = ENV_CASES[name], _load(name)
    blk = _block(ftn, case, dev)
    blk.engine = engine
    if case["kind"] == "env_native":
        blk.period_selector = ftn.models.timesnet.FFTPeriodSelector(case["K"], case["L"])
    else:
        object.__setattr__(blk, "period_selector", _Stub(g["periods"], g["amps"]))
    blk.block_index = case["block_index"]
    _setenv(monkeypatch, case["env"])
    with torch.inference_mode():
        y = blk(torch.from_numpy(g["x"]).to(dev))
    assert blk._last_backend == "hip"
    assert blk._last_group_count == case["groups"] and blk._last_loop_iterations == case["loop_iterations"]
    if case["kind"] == "env_native":
        # native selector: the flagged grouping ran in the device finalize kernel (no host grouping round trip) and
        # the descriptor it wrote equals what the reference's PeriodGrouper returned under the same flags
        d = blk.period_selector._pending.host()
        G, n = int(d.n_groups), int(d.n_sel)
        assert list(d.sel_period[:n]) == g["periods"].tolist()
        assert list(d.g_period[:G]) == g["g_periods"].tolist() and list(d.g_pad[:G]) == g["g_pad"].tolist()
        assert list(d.g_cycles[:G]) == g["g_cycles"].tolist() and list(d.sel_group[:n]) == g["mapping"].tolist()
    np.testing.assert_allclose(y.cpu().numpy(), g["y"], rtol=1e-4, atol=2e-5)
