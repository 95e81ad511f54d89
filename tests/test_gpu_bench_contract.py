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
