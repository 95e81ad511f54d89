"""GPU tests at the real sizes of BASELINE.json's configs (run with ``-m gpu`` on an MI355X).

The oracle cannot run these sizes in seconds, so each test pins the full-size HIP result through
size-independent properties: determinism (no atomics), the planted periods recovered bit-exactly,
batch rows checked against the CPU oracle / CPU mirror with the device-selected periods injected
(rows are independent once the shared periods are fixed), and batch-permutation equivariance.

  c1  B=256 L=336 d_model=64  k=3, exact fp32 MFMA engine          (configs[1])
  c3  B=256 L=720 d_model=128 d_ff=512 k=5, both engines           (configs[3], one GPU's block)
  c4  whole model, per-GPU shard B=64 of 512, L=720->96, N=4096,   (configs[4])
      d_model=128, d_ff=512, 3 blocks, context rank 16, random heads
"""
import ctypes

import numpy as np
import pytest
import torch

from oracle import timesblock_oracle as orc
from test_gpu_parity import ATOL, RTOL, _Stub, _block

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need a ROCm device"
    return torch.device("cuda:0")


def _full_size_block_properties(ftn, dev, B, L, C, K, engine, planted, rows):
    case = dict(hyper="pipeline", C=C, seed=0)
    blk, P, ks, act = _block(ftn, case, dev, engine)
    blk.period_selector = ftn.models.timesnet.FFTPeriodSelector(K, L)
    xh = ftn.synth.make_input(B, L, C, seed=0)
    x = torch.from_numpy(xh).to(dev)
    with torch.inference_mode():
        y1 = blk(x)
        y2 = blk(x)
    assert blk._last_backend == "hip"
    assert torch.equal(y1, y2)                                    # deterministic (no atomics)
    periods = blk.period_selector.last_selected_periods.tolist()
    want = orc.period_select(torch.from_numpy(xh), K, L)          # the selector alone is cheap on the CPU
    assert periods == want.periods                                # integer period indices, bit-exact, score order
    assert blk.period_selector.last_frequency_indices.tolist() == want.freq_idx
    assert sorted(periods) == sorted(planted) and want.topk_gap > 0.1
    assert blk._last_group_count == len(set(planted))
    assert torch.isfinite(y1).all()
    # rows are independent once periods and per-row amplitudes are fixed -> CPU oracle on a few rows
    amps = blk.period_selector(x)[1].cpu()
    y_ref, _ = orc.timesblock_forward(torch.from_numpy(xh[rows]), P, ks, act, K, L, periods=periods, amps=amps[rows])
    np.testing.assert_allclose(y1[rows].cpu().numpy(), y_ref.numpy(), rtol=RTOL, atol=ATOL)
    # batch-permutation equivariance through the host-descriptor path (same periods, permuted rows)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(0))
    object.__setattr__(blk, "period_selector", _Stub(periods, amps[perm].numpy()))
    with torch.inference_mode():
        yp = blk(x[perm.to(dev)])
    np.testing.assert_allclose(yp.cpu().numpy(), y1[perm.to(dev)].cpu().numpy(), rtol=1e-5, atol=1e-6)


def test_c1_full_size_fp32_engine(ftn, dev):
    """BASELINE configs[1]: B=256 L=336 d_model=64 k=3 in exact fp32 MFMA arithmetic."""
    _full_size_block_properties(ftn, dev, 256, 336, 64, 3, "f32", [24, 168, 7], [0, 17, 128, 255])


@pytest.mark.parametrize("engine", ["f16x2", "bf16x3"])
def test_c2_full_size_split_engines(engine, ftn, dev):
    """BASELINE configs[2] = the bench shape: B=256 L=336 d_model=64 k=5 on the split engines.  At this size every
    conv workgroup walks several batch rows of a tile (the counted-vmcnt row barrier and the double-buffered region
    DMA of k_conv_bf_fast), the selector takes its row-resident quarter-fold form and stage A rides in the finalize
    launch - none of which the small fixtures reach.  Also checked against the exact fp32-MFMA engine on EVERY row."""
    _full_size_block_properties(ftn, dev, 256, 336, 64, 5, engine, [24, 168, 7, 12, 84], [0, 17, 128, 255])
    case = dict(hyper="pipeline", C=64, seed=0)
    x = torch.from_numpy(ftn.synth.make_input(256, 336, 64, seed=0)).to(dev)
    ys = []
    for eng in (engine, "f32"):
        blk, _, _, _ = _block(ftn, case, dev, eng)
        blk.period_selector = ftn.models.timesnet.FFTPeriodSelector(5, 336)
        with torch.inference_mode():
            ys.append(blk(x))
        assert blk._last_backend == "hip"
    np.testing.assert_allclose(ys[0].cpu().numpy(), ys[1].cpu().numpy(), rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("engine", ["f16x2", "bf16x3", "f32"])
def test_c3_full_size_block(engine, ftn, dev):
    """BASELINE configs[3] (one GPU's TimesBlock): B=256 L=720 d_model=128 d_ff=512 k=5 - 0.95 M grid pixels,
    the flat pixel index and the 1.8 GB workspace at their real sizes."""
    # planted 168 is not a divisor of 720: its energy lands in rFFT bin 4 -> period ceil(720/4) = 180
    _full_size_block_properties(ftn, dev, 256, 720, 128, 5, engine, [24, 180, 144, 12, 7], [0, 100, 255])


class _ReplaySelector(torch.nn.Module):
    """Returns, call by call, the periods / per-row amplitudes recorded from the device selector."""

    def __init__(self, recs, rows):
        super().__init__()
        self.recs, self.rows, self.i = recs, rows, 0
        self.min_period_threshold, self.pmax = recs[0][2], recs[0][3]

    def forward(self, x):
        periods, amps, _, _ = self.recs[self.i % len(self.recs)]
        self.i += 1
        return periods.clone(), amps[self.rows].to(x.dtype)


def test_c4_model_shard(ftn, dev):
    """BASELINE configs[4], one GPU's shard: the mirror TimesNet, B=64 (of 512) L=720->H=96 N=4096 d_model=128
    d_ff=512, 3 TimesBlocks, LRTC rank 16, randomised heads: eager == HIP-graph replay bit for bit, outputs
    finite and positive, and three batch rows against the CPU mirror (pinned to the reference by
    tests/test_dropin_reference.py) fed the periods / amplitudes each block selected on the device."""
    B, L, H, N, D = 64, 720, 96, 4096, 128
    ks = [(3, 3), (5, 5), (7, 7)]
    cfg = dict(input_len=L, pred_len=H, d_model=D, d_ff=4 * D, n_layers=3, k_periods=5, kernel_set=ks, dropout=0.0,
               activation="gelu", mode="direct", bottleneck_ratio=4.0, id_embed_dim=32,
               use_zero_mean_context=True, context_rank=16)
    torch.manual_seed(0)
    cpu = ftn.models.TimesNet(**cfg).eval()
    xh = torch.from_numpy(ftn.synth.make_input(B, L, N, seed=3))
    xh = xh.abs() + 0.5                                            # count-like, keeps softplus in its usual range
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        cpu(xh[:1])                                                # lazy build
        for p in cpu.parameters():
            if float(p.abs().sum()) == 0.0:                        # reference zero-inits heads / context (:1660-1662)
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
    gpu = ftn.models.TimesNet(**cfg).eval()
    x = xh.to(dev)
    with torch.no_grad():
        gpu(x[:1])
    gpu.load_state_dict(cpu.state_dict(), strict=True)

    recs = []
    sel = gpu.period_selector
    orig = sel.select_device

    def recording(t, **kw):
        s = orig(t, **kw)
        recs.append(s)
        return s

    sel.select_device = recording
    with torch.inference_mode():
        rate, disp = gpu(x)
    sel.select_device = orig
    assert len(recs) == 3
    assert all(b._last_backend == "hip" for b in gpu.blocks)
    assert gpu._last_head_backend == "hip" and gpu._last_embed_backend == "hip"
    assert rate.shape == (B, H, N) and disp.shape == (B, H, N)
    assert torch.isfinite(rate).all() and torch.isfinite(disp).all()
    assert float(rate.min()) > 0.0 and float(disp.min()) > 0.0

    gf = ftn.graph.GraphedForward(gpu, x)
    rate_g, disp_g = gf(x)
    assert torch.equal(rate_g, rate) and torch.equal(disp_g, disp)

    rows = [0, 31, 63]
    host = []
    for s in recs:
        d = s.host()
        n = int(d.n_sel)
        host.append((torch.tensor(list(d.sel_period[:n]), dtype=torch.long), s.amps[:, :n].cpu(),
                     sel.min_period_threshold, sel.pmax))
    cpu.period_selector = _ReplaySelector(host, rows)
    with torch.no_grad():
        want_r, want_d = cpu(xh[rows])
    np.testing.assert_allclose(rate[rows].cpu().numpy(), want_r.numpy(), rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(disp[rows].cpu().numpy(), want_d.numpy(), rtol=RTOL, atol=ATOL)


# ---- multi-GPU exchange on the HIP path, on one GPU ----------------------------------
@pytest.mark.parametrize("B,L,C,K", [(16, 336, 64, 5), (6, 97, 24, 3)])
def test_finalize_with_two_partial_sums_matches_one_shot(B, L, C, K, ftn, dev):
    """What two batch-sharded ranks do: each runs ftn_period_spectrum on its half, the [F] fp64 partial
    sums are stacked (the all-gather) and every rank calls ftn_period_finalize(nparts=2, Btotal=B).
    Descriptor and integer outputs must equal the unsharded call bit for bit, floats to 1e-6."""
    rt = ftn.runtime
    x = torch.from_numpy(ftn.synth.make_input(B, L, C, seed=5)).to(dev)
    med, psum = rt.spectrum(x)
    one = rt.finalize(psum, B, med, L, K, L, 1)
    h = B // 2
    med_a, ps_a = rt.spectrum(x[:h].contiguous())
    med_b, ps_b = rt.spectrum(x[h:].contiguous())
    parts = torch.stack([ps_a, ps_b])
    two_a = rt.finalize(parts, B, med_a, L, K, L, 1)
    two_b = rt.finalize(parts, B, med_b, L, K, L, 1)
    want = bytes(one.host())
    assert bytes(two_a.host()) == want and bytes(two_b.host()) == want
    n, G = int(one.host().n_sel), int(one.host().n_groups)
    assert n == K and G >= 1
    np.testing.assert_allclose(torch.cat([two_a.amps, two_b.amps]).cpu().numpy(), one.amps.cpu().numpy(),
                               rtol=1e-6, atol=0)
    np.testing.assert_allclose(torch.cat([two_a.weights, two_b.weights]).cpu().numpy(), one.weights.cpu().numpy(),
                               rtol=1e-6, atol=1e-7)
    assert two_a.max_groups == one.max_groups and two_a.px_bound == one.px_bound >= int(one.host().total_px)


# ---- autograd gating: trainable block weights behind a no-grad input ---------------------------
def test_trainable_block_behind_frozen_input_gets_gradients(ftn, dev):
    case = dict(hyper="pipeline", C=16, seed=2)
    blk, _, _, _ = _block(ftn, case, dev)
    blk.period_selector = ftn.models.timesnet.FFTPeriodSelector(2, 48)
    blk.train()                                                    # dropout 0: train mode alone must not matter
    x = torch.from_numpy(ftn.synth.make_input(3, 48, 16, seed=1, planted=(12, 8))).to(dev)
    assert not x.requires_grad
    y = blk(x)                                                     # grad mode on, parameters require grad
    assert blk._last_backend == "torch" and y.requires_grad
    y.square().mean().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in blk.inception.parameters())
    with torch.no_grad():
        y_hip = blk(x)
    assert blk._last_backend == "hip"
    np.testing.assert_allclose(y_hip.cpu().numpy(), y.detach().cpu().numpy(), rtol=RTOL, atol=ATOL)
    for p in blk.inception.parameters():
        p.requires_grad_(False)
    assert blk(x) is not None and blk._last_backend == "hip"        # frozen weights: nothing to record
    ln = torch.nn.LayerNorm(16).to(dev)                            # trainable post-norm stays outside the kernel
    z = blk(x, post_norm=ln)
    assert blk._last_backend == "hip" and z.requires_grad


# ---- workspace hygiene: a captured graph must survive later, larger calls ----------------------
def test_graph_replay_survives_larger_eager_call(ftn, dev):
    case = dict(hyper="pipeline", C=32, seed=5)
    blk, _, _, _ = _block(ftn, case, dev)
    blk.period_selector = ftn.models.timesnet.FFTPeriodSelector(3, 96)
    x = torch.from_numpy(ftn.synth.make_input(4, 96, 32, seed=1, planted=(24, 12, 8))).to(dev)
    gf = ftn.graph.GraphedForward(blk, x)
    want = gf(x).clone()
    big = torch.from_numpy(ftn.synth.make_input(64, 96, 32, seed=2, planted=(24, 12, 8))).to(dev)
    junk = []
    with torch.inference_mode():
        for _ in range(3):
            yb = blk(big)                                          # a bigger workspace than the captured call's
            junk.append(torch.full((1 << 22,), 7.0, device=dev))   # churn the allocator between replays
            assert torch.equal(gf(x), want)
    with torch.inference_mode():
        assert torch.equal(blk(big), yb)
    side = torch.cuda.Stream(dev)                                  # two streams, each with its own workspace
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.inference_mode():
        with torch.cuda.stream(side):
            ys = blk(big)
        ym = blk(x)
    torch.cuda.synchronize()
    assert torch.equal(ys, yb) and torch.equal(ym, want)


def test_descriptor_beyond_declared_bounds_is_identity(ftn, dev):
    """A descriptor with more pixels / groups than the caller declared (px_bound, max_groups) must never
    write past the workspace: the call degrades to y = x."""
    lib, rt = ftn.lib.load(), ftn.runtime
    case = dict(hyper="pipeline", C=16, seed=2)
    blk, _, _, _ = _block(ftn, case, dev)
    B, L = 3, 48
    x = torch.from_numpy(ftn.synth.make_input(B, L, 16, seed=1, planted=(12, 8))).to(dev)
    wblob, plan = blk._packed(dev)
    dh = ftn.lib.desc_from_periods([12, 8, 47], L, 1, L)
    w = torch.full((B, 3), 1.0 / 3.0)
    sel = rt.selection_from_host(dh, w, dev)
    y_ok = rt.timesblock_forward(x, plan, wblob, sel)
    assert not torch.equal(y_ok, x)
    for mg, pxb in ((2, int(dh.total_px)), (3, int(dh.total_px) - 1)):
        need = lib.ftn_timesblock_workspace_bytes(ctypes.byref(plan), B, L, mg, pxb)
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        y = torch.full_like(x, float("nan"))
        ftn.lib.check(lib.ftn_timesblock_forward(x.data_ptr(), y.data_ptr(), B, L, ctypes.byref(plan),
                                                 wblob.data_ptr(), sel.desc.data_ptr(), sel.weights.data_ptr(), mg, pxb, 0, 0,
                                                 ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream, None),
                      "ftn_timesblock_forward")
        torch.cuda.synchronize()
        assert torch.equal(y, x)


# ---- stage A in the selector's finalize launch (ftn_period_finalize_stage_a) --------------------------------
@pytest.mark.parametrize("engine", ["f32", "bf16x3", "f16x2"])
@pytest.mark.parametrize("B,L,C", [(16, 336, 64), (5, 97, 24)])
def test_fused_finalize_stage_a_equals_two_launches(B, L, C, engine, ftn, dev, monkeypatch):
    """The block's stage A rides in the finalize launch (workgroup 0 = S3-S5 + descriptor copy, the rest =
    stage A).  Descriptor, amplitudes, weights and the block output must equal the separate launches bit for bit."""
    rt = ftn.runtime
    torch.manual_seed(3)
    blk = ftn.models.TimesBlock(d_model=C, d_ff=2 * C, kernel_set=[(3, 3), (5, 5)], dropout=0.0, activation="gelu",
                                bottleneck_ratio=4.0).to(dev).eval()
    blk.engine = engine
    blk.period_selector = ftn.models.FFTPeriodSelector(4, L, 1)
    x = torch.from_numpy(ftn.synth.make_input(B, L, C, seed=9, planted=(24, 7))).to(dev)
    outs = []
    for fuse in ("1", "0"):
        monkeypatch.setenv("FTN_FUSE_STAGE_A", fuse)
        wblob, plan = blk._packed(dev)
        sel = blk.period_selector.select_device(x, stage_a=(plan, wblob))
        assert (getattr(sel, "stage_a", None) is not None) == (fuse == "1")
        y = rt.timesblock_forward(x, plan, wblob, sel)
        assert getattr(sel, "stage_a", None) is None
        with torch.inference_mode():
            y2 = blk(x)
        assert blk._last_backend == "hip"
        outs.append((sel.desc.cpu(), sel.amps.cpu(), sel.weights.cpu(), y.cpu(), y2.cpu()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert torch.equal(outs[0][3], outs[0][4])
    assert not torch.equal(outs[0][3], x.cpu())
    # the two halves launched apart (what a batch-sharded run does around the exchange of the partial sums)
    monkeypatch.setenv("FTN_FUSE_STAGE_A", "1")
    wblob, plan = blk._packed(dev)
    sm = blk.period_selector
    med, psum = rt.spectrum(x)
    pre = rt.stage_a_only(x, plan, wblob, sm.k, sm.pmax, sm.min_period_threshold)
    sel = rt.finalize(psum, B, med, L, sm.k, sm.pmax, sm.min_period_threshold, stage_a=(x, plan, wblob), pre=pre)
    assert sel.stage_a is not None
    y = rt.timesblock_forward(x, plan, wblob, sel)
    for a, b in zip((sel.desc.cpu(), sel.amps.cpu(), sel.weights.cpu(), y.cpu()), outs[0][:4]):
        assert torch.equal(a, b)


def test_fused_stage_a_rejects_a_foreign_input(ftn, dev):
    rt = ftn.runtime
    blk = ftn.models.TimesBlock(d_model=16, d_ff=32, kernel_set=[(3, 3)], dropout=0.0, activation="gelu",
                                bottleneck_ratio=2.0).to(dev).eval()
    blk.period_selector = ftn.models.FFTPeriodSelector(2, 48, 1)
    x = torch.from_numpy(ftn.synth.make_input(2, 48, 16, seed=1)).to(dev)
    wblob, plan = blk._packed(dev)
    sel = blk.period_selector.select_device(x, stage_a=(plan, wblob))
    with pytest.raises(RuntimeError, match="different input"):
        rt.timesblock_forward(x.clone(), plan, wblob, sel)


# ---- row-resident selector kernels (k_spectrum_row, k_spectrum_rowq) -----------------------------------------
@pytest.mark.parametrize("B,L,C", [(256, 336, 64), (70, 96, 64), (65, 97, 24), (64, 48, 7), (128, 3, 32), (64, 255, 33),
                                   (64, 128, 40), (64, 8, 5),
                                   # d_model > 64: the channel-tiled k_spectrum_rowq + k_median_rows (mode 2 / unset)
                                   (64, 720, 128), (8, 96, 100), (3, 16, 130), (5, 336, 256), (4, 8, 65)])
def test_row_resident_spectrum_forms_agree(B, L, C, dev, tmp_path):
    """ftn_period_spectrum has three kernels: k_spectrum (one workgroup per (row, 32-bin block)), k_spectrum_row
    (one workgroup per batch row, x[b] folded once into LDS; same MFMA sequence on the same operands: bit-identical)
    and, for L % 4 == 0, k_spectrum_rowq (folded a second time around L/4: half the MFMAs, rounds differently at the
    1e-7 level).  Each form is forced in a fresh process (FTN_SEL_ROW is read once per process)."""
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    out = {}
    for mode in ("0", "1", "2"):
        f = tmp_path / f"m{mode}.npz"
        code = (
            "import sys, torch, numpy as np; sys.path.insert(0, %r); import __graft_entry__ as ge; ftn = ge.load_package();"
            "x = torch.from_numpy(ftn.synth.make_input(%d, %d, %d, seed=11, planted=(24, 7))).cuda();"
            "med, psum = ftn.runtime.spectrum(x); torch.cuda.synchronize();"
            "np.savez(%r, med=med.cpu().numpy(), psum=psum.cpu().numpy())"
        ) % (str(root), B, L, C, str(f))
        env = dict(os.environ, FTN_SEL_ROW=mode)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        out[mode] = np.load(f)
    assert out["0"]["med"].tobytes() == out["1"]["med"].tobytes()
    assert out["0"]["psum"].tobytes() == out["1"]["psum"].tobytes()
    scale = float(np.abs(out["0"]["med"]).max())
    np.testing.assert_allclose(out["2"]["med"], out["0"]["med"], rtol=2e-6, atol=2e-6 * scale)
    np.testing.assert_allclose(out["2"]["psum"], out["0"]["psum"], rtol=2e-6, atol=2e-6 * scale * B)
