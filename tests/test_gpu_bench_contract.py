"""The driver parses ONE JSON line from `python bench.py --gpus N --steps K --warmup W`: this test runs the bench on
the GPU with a tiny K and checks the line's shape (fields the task contract names, their types, the roofline object)."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def test_bench_prints_one_contract_line():
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "8", "--warmup", "2",
                        "--no-cpu-baseline", "--no-extras"], capture_output=True, text=True, timeout=600, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, f"expected one line on stdout, got {len(lines)}"
    d = json.loads(lines[0])
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str),
                     ("data", str), ("config", dict), ("roofline", dict)):
        assert key in d and isinstance(d[key], typ), key
    assert d["n_gpus"] == 1 and d["steps"] == 8 and d["warmup"] == 2
    assert d["vs_baseline"] is None                       # BASELINE.md publishes no number for this metric
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 256 * 512 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6   # series/s = B N / t
    rf = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in rf, key
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s")
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and 0.0 < rf["frac"] < 1.0
    assert rf["traffic"] is None or rf["traffic"] > 0
    assert (rf["traffic"] is None) == (rf["traffic_source"] is None)
    # what the timed region is preceded by is part of the line
    assert d["run_in_steps"] >= 0 and d["stage_events_every"] >= 1
    assert d["config"]["B_per_rank"] == 256 and d["config"]["B_global"] == 256
    rc = d["roofline_conv"]
    for key in ("bound", "us", "achieved", "peak", "unit", "frac", "mfma_util_pmc"):
        assert key in rc, key
    assert rc["bound"] == "mfma" and 0.0 < rc["frac"] < 1.0


def test_bench_strong_scaling_and_extras():
    """--scaling strong at one rank is the same batch; the N = 1 extras carry the batch sweep a strong-scaling rank
    sees, the stock PyTorch-ROCm composition on the same tensor, and the CPU baseline at its faster thread count."""
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "8", "--warmup", "2",
                        "--scaling", "strong"], capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["scaling"] == "strong" and d["config"]["B_per_rank"] == 256
    sweep = d["batch_sweep"]
    assert set(sweep) == {"32", "64", "128", "256"}
    assert all(v["ms_per_step"] > 0 and v["ms_per_step_hip_graph"] > 0 for v in sweep.values())
    assert sweep["32"]["ms_per_step_hip_graph"] < sweep["256"]["ms_per_step_hip_graph"]
    tr = d["torch_rocm_baseline"]
    assert "error" not in tr, tr
    assert tr["value"] > 0 and tr["max_abs_diff_vs_hip"] < 1e-2
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] >= cb["threads8"]["value"] * 0.999
    assert abs(d["vs_cpu_baseline"] - d["value"] / cb["value"]) / d["vs_cpu_baseline"] < 1e-9
