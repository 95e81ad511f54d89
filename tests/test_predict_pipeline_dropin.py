"""the mirror ``TimesNet`` inside the reference's own inference pipeline, on the
reference's own data and artifact formats (build container only; skipped without /root/reference).

The reference's ``predict_once`` (predict.py:345-1000) reads ``config_used.yaml``, ``metadata.json``,
``schema.json``, ``scaler.pkl`` and a ``state_dict`` checkpoint, rebuilds ``TimesNet`` with a warm-up
call, resizes the id embedding, ``load_state_dict(strict=True)``, windows ``data/test/TEST_00.csv``
(193 series x 28 days, one series per batch row: ``x[B=193, L=28, N=1]`` with per-sample static
features and ids) and writes a submission CSV.  The test runs that pipeline twice on a checkpoint
written by the reference model - once untouched, once with ``predict.TimesNet`` replaced by
``flow_timesnet_amd.models.TimesNet`` - and the two CSVs must agree.  All artifacts are written by
this test with the reference's own writers; only the CSV under data/test is the reference's file."""
import importlib
import pickle
import shutil
import sys
from pathlib import Path

import numpy as np
import pandas as pd
import pytest
import torch
import yaml

REF_ROOT = Path("/root/reference")
REF = REF_ROOT / "src"
pytestmark = pytest.mark.skipif(not REF.exists(), reason="reference checkout not present")

DATE, SID, TARGET = "영업일자", "영업장명_메뉴명", "매출수량"
INPUT_LEN, PRED_LEN = 28, 7
MODEL = dict(mode="direct", input_len=INPUT_LEN, pred_len=PRED_LEN, d_model=16, d_ff=32, n_layers=2, k_periods=2,
             kernel_set=[3, 5], dropout=0.0, activation="gelu", bottleneck_ratio=2.0, use_embedding_norm=True,
             id_embed_dim=6, static_proj_dim=5, static_layernorm=True, min_period_threshold=1,
             use_zero_mean_context=True, context_rank=3, context_scale=0.05, use_constant_context_bias=True)


def _ref(name):
    if str(REF) not in sys.path:
        sys.path.insert(0, str(REF))
    return importlib.import_module(name)


def _write_artifacts(tmp_path, ids, static_np):
    cfgmod, io_utils, meta_utils = _ref("timesnet_forecast.config"), _ref("timesnet_forecast.utils.io"), \
        _ref("timesnet_forecast.utils.metadata")
    tsn = _ref("timesnet_forecast.models.timesnet")
    art = tmp_path / "artifacts"
    art.mkdir()
    n = len(ids)

    # a reference model in its trained shape: warm-up as train.py does, widen the id table, wake the heads
    torch.manual_seed(0)
    model = tsn.TimesNet(channels_last=False, use_checkpoint=False,
                         **{k: v for k, v in MODEL.items() if k not in ("min_period_threshold",)},
                         min_period_threshold=MODEL["min_period_threshold"])
    with torch.no_grad():
        model(torch.zeros(1, INPUT_LEN, 1), series_static=torch.from_numpy(static_np[:1]),
              series_ids=torch.tensor([n - 1]))
        g = torch.Generator().manual_seed(7)
        for p in model.parameters():
            if float(p.abs().sum()) == 0.0:
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
    assert model.series_embedding.num_embeddings == n
    torch.save(model.state_dict(), art / "timesnet.pth")

    tf = cfgmod.TimeFeatureConfig(enabled=False).to_dict()
    names = [f"f{i}" for i in range(static_np.shape[1])]
    with open(art / "scaler.pkl", "wb") as f:
        pickle.dump({"ids": ids, "method": "none", "scaler": None, "static_features": static_np,
                     "feature_names": names, "time_features": {"enabled": False, "feature_dim": 0, "config": tf}}, f)
    schema = io_utils.DataSchema.from_fields({"date": DATE, "id": SID, "target": TARGET},
                                             sources={"date": "override", "id": "override", "target": "override"})
    io_utils.save_schema_artifact(str(art / "schema.json"), schema,
                                  normalization={"method": "none", "per_series": True, "eps": 1e-8})
    window = cfgmod.WindowConfig(input_len=INPUT_LEN, pred_len=PRED_LEN, stride=1, short_series_strategy="error",
                                 pad_value=0.0)
    meta = meta_utils.MetadataArtifact.from_training(
        window=window, schema=schema,
        time_features={"config": tf, "enabled": False, "feature_dim": 0, "freq": None},
        static_features={"feature_names": names, "feature_dim": static_np.shape[1]})
    meta_utils.save_metadata_artifact(meta, str(art / "metadata.json"))

    test_dir = tmp_path / "test"
    test_dir.mkdir()
    shutil.copy(REF_ROOT / "data" / "test" / "TEST_00.csv", test_dir / "TEST_00.csv")
    cfg = {
        "artifacts": {"dir": str(art), "model_file": "timesnet.pth", "scaler_file": "scaler.pkl",
                      "schema_file": "schema.json", "config_file": "config_used.yaml",
                      "signature_file": "model_signature.json", "metadata_file": "metadata.json"},
        "data": {"test_dir": str(test_dir), "sample_submission": str(tmp_path / "absent.csv"),
                 "fill_missing_dates": False, "date_col": DATE, "id_col": SID, "target_col": TARGET,
                 "time_features": tf, "encoding": "utf-8-sig"},
        "preprocess": {"clip_negative": False, "normalize": "none", "normalize_per_series": True, "eps": 1e-8},
        "train": {"device": "cpu", "matmul_precision": "highest", "channels_last": False, "amp": False,
                  "use_checkpoint": False, "cuda_graphs": False,
                  "val": {"strategy": "holdout", "holdout_days": INPUT_LEN + PRED_LEN}},
        "model": dict(MODEL),
        "window": {"input_len": INPUT_LEN, "pred_len": PRED_LEN, "stride": 1, "short_series_strategy": "error",
                   "pad_value": 0.0},
        "submission": {"out_path": str(tmp_path / "out" / "submission.csv"), "format": "row_key", "date_col": DATE},
    }
    with open(art / "config_used.yaml", "w", encoding="utf-8") as f:
        yaml.safe_dump({k: cfg[k] for k in ("model", "train", "data", "artifacts", "submission", "preprocess")}, f,
                       allow_unicode=True)
    return cfg, model


def test_predict_once_with_mirror_model(tmp_path, ftn, monkeypatch):
    predict = _ref("timesnet_forecast.predict")
    raw = pd.read_csv(REF_ROOT / "data" / "test" / "TEST_00.csv", encoding="utf-8-sig")
    io_utils = _ref("timesnet_forecast.utils.io")
    ids = sorted({io_utils.normalize_id(v) for v in raw[SID].unique()})      # the ids training would have stored
    assert len(ids) > 100 and raw[DATE].nunique() == INPUT_LEN
    rng = np.random.RandomState(3)
    static_np = rng.randn(len(ids), 4).astype(np.float32)
    cfg, ref_model = _write_artifacts(tmp_path, ids, static_np)

    want_path = predict.predict_once(cfg)
    want = pd.read_csv(want_path)
    shutil.move(want_path, str(tmp_path / "want.csv"))

    built = []

    class Mirror(ftn.models.TimesNet):
        def __init__(self, *a, **kw):
            super().__init__(*a, **kw)
            built.append(self)

    monkeypatch.setattr(predict, "TimesNet", Mirror)
    got_path = predict.predict_once(cfg)
    got = pd.read_csv(got_path)

    assert len(built) == 1 and all(b._period_calls > 0 for b in built[0].blocks)
    assert list(got.columns) == list(want.columns) and len(got) == len(want) == PRED_LEN
    assert got.iloc[:, 0].tolist() == want.iloc[:, 0].tolist()
    a, b = got.iloc[:, 1:].to_numpy(np.float64), want.iloc[:, 1:].to_numpy(np.float64)
    assert np.isfinite(a).all() and (a >= 0).all() and a.max() > 0
    np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-6)
    # the checkpoint really was the reference's: same key set, strict load succeeded inside predict_once
    assert set(built[0].state_dict().keys()) == set(ref_model.state_dict().keys())
