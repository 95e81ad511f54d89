"""world_size-2 gloo test (CPU) of the batch-sharded path: the [F] partial-sum
exchange must make both ranks pick the periods of the full batch, and the
re-assembled output must equal the single-process result."""
import json
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN, ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, name, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, str(ROOT))
    import __graft_entry__ as ge
    pkg = ge.load_package()
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        man = json.loads((GOLDEN / "manifest.json").read_text())
        case, hyp = man["cases"][name], man["hypers"]
        h = hyp[case["hyper"]]
        C = case["C"]
        d_ff = C if h["d_ff_mult"] is None else C * h["d_ff_mult"]
        ks = [tuple(k) for k in h["kernel_set"]]
        T = pkg.models.timesnet
        blk = T.TimesBlock(C, ks, 0.0, h["act"], d_ff=d_ff, bottleneck_ratio=h["ratio"])
        sd = pkg.synth.make_inception_params(C, d_ff, ks, h["ratio"], case["seed"])
        blk.inception.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        blk.period_selector = T.FFTPeriodSelector(case["K"], case["L"])
        blk.eval()
        with np.load(GOLDEN / f"{name}.npz") as z:
            x = torch.from_numpy(z["x"])
        shard = x.chunk(world, dim=0)[rank]
        runner = pkg.dist.ShardedTimesBlock(blk)
        with torch.no_grad():
            y = runner(shard, gather=True)
            y_local = runner(shard, gather=False)
            y_async, work = runner(shard, gather="async")
            work.wait()
            assert torch.equal(y_async, y)
        assert blk.period_selector.shard_group is None          # restored after the call
        # the flagged grouping variants would rank groups on each rank's own rows: refused, not diverging
        os.environ["TIMES_PERIOD_MAX_UNIQ"] = "2"
        try:
            with pytest.raises(NotImplementedError):
                runner(shard, gather=False)
        finally:
            del os.environ["TIMES_PERIOD_MAX_UNIQ"]
        np.save(os.path.join(out_dir, f"y{rank}.npy"), y.numpy())
        np.save(os.path.join(out_dir, f"p{rank}.npy"), blk.period_selector.last_selected_periods.numpy())
        assert torch.equal(y.chunk(world, dim=0)[rank], y_local)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name", ["b_c0_pipe", "b_c1_min"])
def test_batch_sharded_equals_single_process(name, tmp_path, golden):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), name, str(tmp_path)), nprocs=world, join=True)
    g = golden(name)
    y0, y1 = np.load(tmp_path / "y0.npy"), np.load(tmp_path / "y1.npy")
    assert np.array_equal(y0, y1)                                # every rank holds the same gathered batch
    assert np.load(tmp_path / "p0.npy").tolist() == g["periods"].tolist()
    assert np.load(tmp_path / "p1.npy").tolist() == g["periods"].tolist()
    np.testing.assert_allclose(y0, g["y"], rtol=1e-4, atol=2e-5)


def _fixture_model(pkg, name="m_context"):
    man = json.loads((GOLDEN / "manifest.json").read_text())
    cfg = dict(man["cases"][name]["cfg"])
    cfg["kernel_set"] = [tuple(k) for k in cfg["kernel_set"]]
    with np.load(GOLDEN / f"{name}.npz") as z:
        g = {k: z[k] for k in z.files}
    kw = {k: torch.from_numpy(g[k]) for k in ("series_static", "series_ids") if k in g}
    x3 = torch.from_numpy(g["x"])
    x = torch.cat([x3, torch.roll(x3[:1], 5, dims=1) * 0.7 + 0.3], dim=0)        # 4 rows: splits over 2 ranks
    model = pkg.models.TimesNet(**cfg).eval()
    with torch.no_grad():
        model(x, **kw)
    sd = {k[4:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd::")}
    d_ff = cfg["d_ff"] if cfg.get("d_ff") else cfg["d_model"]
    for li in range(cfg["n_layers"]):
        prm = pkg.synth.make_inception_params(cfg["d_model"], d_ff, cfg["kernel_set"],
                                              cfg.get("bottleneck_ratio", 1.0), seed=100 + li)
        for k, v in prm.items():
            sd[f"blocks.{li}.inception.{k}"] = torch.from_numpy(v)
    model.load_state_dict(sd, strict=True)
    return model, x, kw


def _model_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, str(ROOT))
    import __graft_entry__ as ge
    pkg = ge.load_package()
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model, x, kw = _fixture_model(pkg)
        runner = pkg.dist.ShardedTimesNet(model)
        with torch.no_grad():
            rate, disp = runner(x.chunk(world, dim=0)[rank], gather=True, **kw)
            rate_l, _ = runner(x.chunk(world, dim=0)[rank], **kw)
        assert model.period_selector.shard_group is None
        assert torch.equal(rate.chunk(world, dim=0)[rank], rate_l)
        np.save(os.path.join(out_dir, f"rate{rank}.npy"), rate.numpy())
        np.save(os.path.join(out_dir, f"disp{rank}.npy"), disp.numpy())
        np.save(os.path.join(out_dir, f"per{rank}.npy"), model.period_selector.last_selected_periods.numpy())
    finally:
        dist.destroy_process_group()


def test_batch_sharded_model_equals_single_process(tmp_path, ftn):
    """Whole model, batch split over two ranks: every block's selector sees the full-batch spectrum through the
    partial-sum exchange, so the gathered outputs equal the single-process forward of the same mirror (which
    tests/test_host_logic.py pins to the reference on this fixture's weights)."""
    world = 2
    model, x, kw = _fixture_model(ftn)
    with torch.no_grad():
        want_r, want_d = model(x, **kw)
    want_p = model.period_selector.last_selected_periods.tolist()
    mp.spawn(_model_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "rate0.npy"), np.load(tmp_path / "rate1.npy")
    assert np.array_equal(r0, r1)
    assert np.load(tmp_path / "per0.npy").tolist() == want_p == np.load(tmp_path / "per1.npy").tolist()
    np.testing.assert_allclose(r0, want_r.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(np.load(tmp_path / "disp0.npy"), want_d.numpy(), rtol=1e-5, atol=1e-6)
