"""The multi-GPU exchange without a collective (dist.IpcExchange, FtnExchange in include/flowtimes.h): two ranks - two
processes that share this box's one GPU, which exercises the same IPC mapping, peer stores, sequence words and bounded
wait as two GPUs do - must select the periods of the full batch and produce its rows, call after call (the two halves
of the exchange buffer alternate), bit for bit what the all-gather exchange gives."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu
KS = [(3, 3), (5, 5), (7, 7)]
B, L, C, K = 8, 96, 64, 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.path.insert(0, str(ROOT))
    import torch.distributed as dist

    import __graft_entry__ as ge
    pkg = ge.load_package()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")                              # both ranks on the one GPU of the box
        T = pkg.models.timesnet
        blk = T.TimesBlock(C, KS, 0.0, "gelu", d_ff=4 * C, bottleneck_ratio=4.0)
        sd = pkg.synth.make_inception_params(C, 4 * C, KS, 4.0, 3)
        blk.inception.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        blk.period_selector = T.FFTPeriodSelector(K, L)
        blk = blk.eval().to(dev)
        xs = [torch.from_numpy(pkg.synth.make_input(B, L, C, seed=s, planted=p)).chunk(world, dim=0)[rank].to(dev)
              for s, p in ((5, (24, 12, 8)), (6, (16, 6, 32)), (7, (48, 4, 12)))]
        xch = pkg.dist.IpcExchange(None, dev, f_cap=128)
        via_ipc = pkg.dist.ShardedTimesBlock(blk, exchange=xch)
        via_gather = pkg.dist.ShardedTimesBlock(blk)
        with torch.inference_mode():
            for i, x in enumerate(xs * 2):                        # six exchanges: both halves of the buffer, three times
                y = via_ipc(x, gather=False)
                periods = blk.period_selector.last_selected_periods.tolist()
                y2 = via_gather(x, gather=False)
                assert blk.period_selector.last_selected_periods.tolist() == periods
                assert torch.equal(y, y2), f"call {i}: ipc and all-gather exchanges differ"
                if i < len(xs):
                    np.save(os.path.join(out_dir, f"y{i}_{rank}.npy"), y.cpu().numpy())
                    np.save(os.path.join(out_dir, f"p{i}_{rank}.npy"), np.asarray(periods))
        xch.check()
        assert int(xch.x.seq) == 6
        xch.close()
    finally:
        dist.destroy_process_group()


def test_ipc_exchange_two_ranks_on_one_gpu(ftn, tmp_path):
    from oracle import timesblock_oracle as orc

    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    sd = ftn.synth.make_inception_params(C, 4 * C, KS, 4.0, 3)
    P = {k: torch.from_numpy(v) for k, v in sd.items()}
    for i, (s, p) in enumerate(((5, (24, 12, 8)), (6, (16, 6, 32)), (7, (48, 4, 12)))):
        x = torch.from_numpy(ftn.synth.make_input(B, L, C, seed=s, planted=p))
        y_ref, aux = orc.timesblock_forward(x, P, KS, "gelu", K, L)
        y = np.concatenate([np.load(tmp_path / f"y{i}_{r}.npy") for r in range(world)], axis=0)
        for r in range(world):                                    # every rank selected the FULL batch's periods
            assert np.load(tmp_path / f"p{i}_{r}.npy").tolist() == aux.sel.periods
        np.testing.assert_allclose(y, y_ref.numpy(), rtol=1e-4, atol=2e-5)


def test_ipc_exchange_world_of_one(ftn):
    """A single rank publishes to its own buffer and waits for itself: the exchange path end to end in this process."""
    import torch.distributed as dist

    dev = torch.device("cuda:0")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        T = ftn.models.timesnet
        blk = T.TimesBlock(C, KS, 0.0, "gelu", d_ff=4 * C, bottleneck_ratio=4.0)
        blk.period_selector = T.FFTPeriodSelector(K, L)
        blk = blk.eval().to(dev)
        x = torch.from_numpy(ftn.synth.make_input(B, L, C, seed=5, planted=(24, 12, 8))).to(dev)
        xch = ftn.dist.IpcExchange(None, dev, f_cap=64)
        runner = ftn.dist.ShardedTimesBlock(blk, exchange=xch)
        os.environ["FTN_BENCH_FORCE_DIST"] = "1"                  # a world of one still takes the sharded path
        try:
            with torch.inference_mode():
                want = blk(x)
                for _ in range(3):
                    got = runner(x, gather=False)
                    assert torch.equal(got, want)
        finally:
            del os.environ["FTN_BENCH_FORCE_DIST"]
        xch.check()
        xch.close()
    finally:
        dist.destroy_process_group()
