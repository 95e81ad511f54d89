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
