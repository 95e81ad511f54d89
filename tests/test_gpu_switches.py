"""Every environment switch README.md lists selects code that must stay correct: each one is set in a fresh process
(the library reads them once) and a slice of the parity suite - the d_model-64 pipeline block on the default engine,
golden fixtures and seeded oracle cases at L = 336 - is run under it."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]
SLICE = ("test_block_matches_reference and (b_c1_pipe-f16x2 or b_c2_pipe_k5-f16x2) or "
         "test_block_matches_oracle_seeded and f16x2 and pipeline")


@pytest.mark.parametrize("switch", ["FTN_MLP_SPLIT=0", "FTN_R_KEEPS_X=0", "FTN_FUSE_STAGE_A=0", "FTN_CONV_QUANT=0",
                                    "FTN_MLP_U1=0", "FTN_MLP_W16=1", "FTN_MLP_W4=1", "FTN_MLP_PFD=2",
                                    "FTN_CONV_GENERIC=1", "FTN_SEL_ROW=0", "FTN_SEL_ROW=1", "FTN_SEL_FLAT=1",
                                    "FTN_MLP_POS=0", "FTN_MLP_POS_NWV=8", "FTN_MLP_POS_GB=4", "FTN_OUT_H=0", "FTN_MLP_POS_PF=0"])
def test_parity_slice_under_switch(switch):
    name, value = switch.split("=")
    env = dict(os.environ, **{name: value})
    r = subprocess.run([sys.executable, "-m", "pytest", str(ROOT / "tests" / "test_gpu_parity.py"), "-m", "gpu", "-x", "-q",
                        "-k", SLICE], env=env, capture_output=True, text=True, timeout=600, cwd=str(ROOT))
    tail = (r.stdout + r.stderr)[-1500:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "failed" not in r.stdout, tail


SLICE_128 = ("test_block_matches_oracle_seeded and 720-128 and f16x2 or "
             "test_awkward_geometries_match_oracle and 250-128 and f16x2")


@pytest.mark.parametrize("switch", ["FTN_MLP_POS=0", "FTN_OUT_H=0", "FTN_CONV_GENERIC=1", "FTN_MLP_POS_GB=2", "FTN_SEL_ROW=0"])
def test_d_model_128_slice_under_switch(switch):
    """The d_model-128 forms added late in round 3 - channel-tiled selector, fast conv path for mid 32, position-major
    stage C (three or two groups per pass), k_out_h with eight output tiles - each against the form it replaced."""
    name, value = switch.split("=")
    env = dict(os.environ, **{name: value})
    r = subprocess.run([sys.executable, "-m", "pytest", str(ROOT / "tests" / "test_gpu_parity.py"), "-m", "gpu", "-x", "-q",
                        "-k", SLICE_128], env=env, capture_output=True, text=True, timeout=600, cwd=str(ROOT))
    tail = (r.stdout + r.stderr)[-1500:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "failed" not in r.stdout, tail
