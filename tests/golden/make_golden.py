#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE implementation on CPU.

Usage (build container only; the reference never travels to the GPU box):

    python tests/golden/make_golden.py /root/reference

The reference is imported from ``<ref>/src`` (never copied).  Inputs and weights
come from ``flow-timesnet_amd/synth.py`` (seeded ``numpy.random.RandomState``),
so the fixtures only need to hold the seeds, the inputs (small cases) and the
reference's outputs.  Output: ``tests/golden/*.npz`` + ``manifest.json``.
"""
from __future__ import annotations

import importlib.util
import json
import os
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parents[1]


def _load_synth():
    spec = importlib.util.spec_from_file_location("ftn_synth", ROOT / "flow-timesnet_amd" / "synth.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


HYPERS = {
    # tests/test_times_block.py:86-91 style
    "minimal": dict(kernel_set=[(3, 3)], ratio=1.0, d_ff_mult=None, act="gelu"),
    # configs/default.yaml:75-80 + config.py:183
    "pipeline": dict(kernel_set=[(3, 3), (5, 5), (7, 7)], ratio=4.0, d_ff_mult=4, act="gelu"),
    # extra coverage: rectangular kernels, relu, ratio 2, d_ff = 2C
    "rect": dict(kernel_set=[(3, 5), (5, 1)], ratio=2.0, d_ff_mult=2, act="relu"),
    # ratio 1 with several kernels and d_ff != d_model (res_proj + merged conv)
    "wide1": dict(kernel_set=[(3, 3), (5, 5)], ratio=1.0, d_ff_mult=2, act="gelu"),
}

BLOCK_CASES = [
    # name, B, L, C, K, hyper, seed, planted periods (None = white noise + default)
    ("b_tiny_min", 2, 24, 3, 2, "minimal", 0, (6, 8)),
    ("b_tiny_pipe", 2, 24, 3, 2, "pipeline", 1, (6, 8)),
    ("b_c0_min", 4, 96, 16, 2, "minimal", 0, (24, 12, 8)),
    ("b_c0_pipe", 4, 96, 16, 3, "pipeline", 1, (24, 12, 8)),
    ("b_c0_rect", 4, 96, 16, 3, "rect", 2, (24, 12, 8)),
    ("b_c0_wide1", 3, 96, 16, 3, "wide1", 2, (24, 16, 8)),
    ("b_odd_min", 3, 150, 8, 3, "minimal", 2, (25, 10, 6)),
    ("b_odd_pipe", 3, 150, 8, 5, "pipeline", 0, (25, 10, 6, 15, 50)),
    ("b_c1_min", 2, 336, 64, 3, "minimal", 0, None),
    ("b_c1_pipe", 2, 336, 64, 3, "pipeline", 1, None),
    ("b_c2_pipe_k5", 2, 336, 64, 5, "pipeline", 2, None),
    ("b_noise_pipe", 2, 96, 16, 5, "pipeline", 3, ()),  # white noise: duplicate tiny periods
]

# Stub-selector cases (reference tests/test_times_block.py:14-30,157-180;
# tests/test_timesblock_vectorized.py:111-129): periods/amps given explicitly.
STUB_CASES = [
    ("s_dup", 2, 25, 3, "minimal", 3, [4, 4, 8, 4], [[0.5, -1.2, 0.3, 0.7]]),
    ("s_mixed_pad", 2, 30, 4, "minimal", 4, [7, 4, 15, 0, -1, 40], [[0.1, 0.9, -0.3, 2.0, 1.0, 0.5]]),
    ("s_448", 3, 32, 4, "pipeline", 5, [4, 4, 4, 8], [[1.0, 0.2, -0.5, 0.3], [0.0, 0.1, 0.2, 0.3], [2.0, -2.0, 0.5, 0.5]]),
    ("s_wide_rows", 2, 48, 8, "pipeline", 6, [47, 24, 2], [[0.3, 0.2, 0.1]]),
    ("s_p1", 1, 12, 2, "minimal", 7, [1, 3], [[0.0, 1.0]]),
]

SELECT_CASES = [
    # name, B, L, C, K, pmax, min_thr, seed, planted, noise
    ("sel_kat_256", 2, 256, 3, 2, 256, 1, 0, (64, 32), 0.01),
    ("sel_bounds", 1, 64, 1, 2, 16, 5, 0, (32, 3.2), 0.0),
    ("sel_c1", 4, 336, 64, 5, 336, 1, 1, None, 1.0),
    ("sel_odd", 3, 150, 8, 3, 150, 1, 2, (25, 10, 6), 1.0),
    ("sel_L720", 2, 720, 16, 5, 720, 1, 3, None, 1.0),
    ("sel_thr7", 3, 96, 5, 4, 96, 7, 4, (24, 12, 4, 3), 0.5),
    ("sel_even_c", 2, 60, 4, 3, 60, 1, 5, (20, 5), 1.0),
]

# Full model: reference TimesNet with non-zero heads; the whole state_dict is part
# of the fixture (small models), so the GPU test needs nothing but the mirror.
MODEL_CASES = {
    "m_context": dict(cfg=dict(input_len=48, pred_len=12, d_model=16, d_ff=32, n_layers=2, k_periods=3,
                               kernel_set=[(3, 3), (5, 5)], dropout=0.0, activation="gelu", mode="direct",
                               bottleneck_ratio=2.0, use_checkpoint=False, id_embed_dim=8, static_proj_dim=8,
                               use_zero_mean_context=True, context_rank=4, context_scale=0.05),
                      B=3, N=5, static=6, ids=[4, 0, 2, 7, 1], marks=0, extra_t=0),
    "m_pipeline": dict(cfg=dict(input_len=96, pred_len=24, d_model=64, d_ff=256, n_layers=2, k_periods=3,
                                kernel_set=[(3, 3), (5, 5), (7, 7)], dropout=0.0, activation="gelu", mode="direct",
                                bottleneck_ratio=4.0, use_checkpoint=True, id_embed_dim=4, min_period_threshold=2,
                                use_zero_mean_context=True, context_rank=16, context_scale=0.05),
                       B=4, N=8, static=0, ids=None, marks=2, extra_t=5),
}

LRTC_CASES = [("lrtc_a", 2, 24, 5, 4, 0.01, 0), ("lrtc_b", 3, 336, 7, 16, 0.5, 1), ("lrtc_c", 1, 150, 3, 1, -1.25, 2)]


def main() -> None:
    ref_root = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
    sys.path.insert(0, str(ref_root / "src"))
    for var in ("TIMES_PERIOD_MAX_UNIQ", "TIMES_PERIOD_BINNING", "TIMESBLOCK_VEC_DISABLE",
                "TIMESBLOCK_MEMORY_FORMAT", "TIMESBLOCK_BUCKET_MAX", "TIMESBLOCK_K_CHUNK", "TIMES_MP_CONV"):
        os.environ.pop(var, None)
    from timesnet_forecast.models.timesnet import (  # type: ignore
        FFTPeriodSelector, LowRankTemporalContext, PeriodGrouper, TimesBlock,
    )
    synth = _load_synth()
    torch.set_num_threads(4)
    manifest = {"torch": torch.__version__, "numpy": np.__version__, "cases": {},
                "hypers": {k: dict(v, kernel_set=[list(t) for t in v["kernel_set"]]) for k, v in HYPERS.items()}}

    def build_block(C, hyper, seed):
        h = HYPERS[hyper]
        d_ff = None if h["d_ff_mult"] is None else C * h["d_ff_mult"]
        blk = TimesBlock(d_model=C, kernel_set=h["kernel_set"], dropout=0.0, activation=h["act"],
                         d_ff=d_ff, bottleneck_ratio=h["ratio"])
        params = synth.make_inception_params(C, d_ff if d_ff is not None else C, h["kernel_set"], h["ratio"], seed)
        blk.inception.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
        blk.eval()
        wsum = float(sum(np.abs(v).sum(dtype=np.float64) for v in params.values()))
        return blk, wsum

    class Stub(torch.nn.Module):
        def __init__(self, periods, amps):
            super().__init__()
            self.periods = torch.as_tensor(periods, dtype=torch.long)
            self.amps = torch.as_tensor(amps, dtype=torch.float32)

        def forward(self, x):
            a = self.amps
            if a.dim() == 1:
                a = a.unsqueeze(0)
            if a.size(0) == 1 and x.size(0) > 1:
                a = a.expand(x.size(0), -1)
            return self.periods, a.to(x.dtype)

    with torch.no_grad():
        for name, B, L, C, K, hyper, seed, planted in BLOCK_CASES:
            x = synth.make_input(B, L, C, seed=seed, planted=planted)
            blk, wsum = build_block(C, hyper, seed)
            sel = FFTPeriodSelector(k_periods=K, pmax=L, min_period_threshold=1)
            object.__setattr__(blk, "period_selector", sel)
            xt = torch.from_numpy(x)
            y = blk(xt)
            periods, amps = sel(xt)
            grp = PeriodGrouper(periods, amps, seq_len=L, min_period=1, max_period=L).group()
            out = dict(x=x, y=y.numpy(), periods=periods.numpy(), amps=amps.numpy(),
                       freq_idx=sel.last_frequency_indices.numpy(),
                       g_periods=grp.periods.numpy(), g_pad=grp.pad_lengths.numpy(),
                       g_cycles=grp.cycles.numpy(), mapping=grp.mapping.numpy())
            np.savez_compressed(HERE / f"{name}.npz", **out)
            manifest["cases"][name] = dict(kind="block", B=B, L=L, C=C, K=K, hyper=hyper, seed=seed,
                                           planted=None if planted is None else list(planted),
                                           wsum=wsum, groups=int(blk._last_group_count))
            print(name, "periods", periods.tolist(), "groups", grp.periods.tolist())

        for name, B, L, C, hyper, seed, periods, amps in STUB_CASES:
            x = synth.make_input(B, L, C, seed=seed, planted=())
            blk, wsum = build_block(C, hyper, seed)
            object.__setattr__(blk, "period_selector", Stub(periods, amps))
            y = blk(torch.from_numpy(x))
            np.savez_compressed(HERE / f"{name}.npz", x=x, y=y.numpy(),
                                periods=np.asarray(periods, dtype=np.int64),
                                amps=np.asarray(amps, dtype=np.float32))
            manifest["cases"][name] = dict(kind="stub", B=B, L=L, C=C, hyper=hyper, seed=seed, wsum=wsum,
                                           groups=int(blk._last_group_count))
            print(name, "groups", blk._last_group_count)

        for name, B, L, C, K, pmax, thr, seed, planted, noise in SELECT_CASES:
            x = synth.make_input(B, L, C, seed=seed, planted=planted, noise=noise)
            sel = FFTPeriodSelector(k_periods=K, pmax=pmax, min_period_threshold=thr)
            xt = torch.from_numpy(x)
            periods, amps = sel(xt)
            med = torch.abs(torch.fft.rfft(xt, dim=1)).median(dim=2).values
            amp_mean = med.mean(dim=0)
            sc = amp_mean.clone()
            sc[0] = float("-inf")
            srt = torch.sort(sc, descending=True).values
            kk = min(K, sc.numel() - 1)
            gap = float((srt[kk - 1] - srt[kk]) / srt[kk - 1].abs().clamp_min(1e-30)) if sc.numel() - 1 > kk else float("inf")
            np.savez_compressed(HERE / f"{name}.npz", x=x, periods=periods.numpy(), amps=amps.numpy(),
                                freq_idx=sel.last_frequency_indices.numpy(), median=med.numpy(),
                                amp_mean=amp_mean.numpy())
            manifest["cases"][name] = dict(kind="select", B=B, L=L, C=C, K=K, pmax=pmax, min_thr=thr, seed=seed,
                                           topk_gap=gap)
            print(name, periods.tolist(), "gap", gap)

        for name, B, L, N, R, scale, seed in LRTC_CASES:
            rs = np.random.RandomState(seed)
            coeff = rs.standard_normal(size=(B, N, R)).astype(np.float32)
            mod = LowRankTemporalContext(rank=R, init_scale=scale)
            ctx = mod(torch.from_numpy(coeff), L)
            np.savez_compressed(HERE / f"{name}.npz", coeff=coeff, ctx=ctx.numpy(),
                                basis=mod._cached_basis.numpy())
            manifest["cases"][name] = dict(kind="lrtc", B=B, L=L, N=N, R=R, scale=scale, seed=seed)
            print(name, ctx.shape)

        from timesnet_forecast.models.timesnet import TimesNet  # type: ignore
        for name, mc in MODEL_CASES.items():
            cfg = mc["cfg"]
            B, N = mc["B"], mc["N"]
            L = cfg["input_len"] + mc["extra_t"]
            rs = np.random.RandomState(77)
            t = np.arange(L, dtype=np.float64).reshape(1, L, 1)
            x = (rs.standard_normal((B, L, N)) + 2.0 * np.sin(2 * np.pi * t / 12.0)
                 + np.sin(2 * np.pi * t / 8.0 + rs.uniform(0, 6.28, (B, 1, N)))).astype(np.float32)
            kw, arrs = {}, {"x": x}
            if mc["static"]:
                arrs["series_static"] = rs.standard_normal((N, mc["static"])).astype(np.float32)
                kw["series_static"] = torch.from_numpy(arrs["series_static"])
            if mc["ids"] is not None:
                arrs["series_ids"] = np.asarray(mc["ids"], dtype=np.int64)
                kw["series_ids"] = torch.from_numpy(arrs["series_ids"])
            if mc["marks"]:
                arrs["x_mark"] = rs.standard_normal((B, L, mc["marks"])).astype(np.float32)
                kw["x_mark"] = torch.from_numpy(arrs["x_mark"])
            torch.manual_seed(0)
            model = TimesNet(**cfg).eval()
            model(torch.from_numpy(x), **kw)
            gen = torch.Generator().manual_seed(5)
            for prm in model.parameters():
                if float(prm.detach().abs().sum()) == 0.0:
                    prm.copy_(0.1 * torch.randn(prm.shape, generator=gen))
            # block weights come from the seeded generator (seed 100 + layer), not from the fixture
            d_ff = cfg["d_ff"] if cfg.get("d_ff") else cfg["d_model"]
            for li, blk in enumerate(model.blocks):
                prm = synth.make_inception_params(cfg["d_model"], d_ff, cfg["kernel_set"],
                                                  cfg.get("bottleneck_ratio", 1.0), seed=100 + li)
                blk.inception.load_state_dict({k: torch.from_numpy(v) for k, v in prm.items()}, strict=True)
            rate, disp = model(torch.from_numpy(x), **kw)
            for k, v in model.state_dict().items():
                if not k.startswith("blocks."):
                    arrs["sd::" + k] = v.numpy()
            arrs["rate"], arrs["dispersion"] = rate.numpy(), disp.numpy()
            arrs["periods"] = model.period_selector.last_selected_periods.numpy()
            np.savez_compressed(HERE / f"{name}.npz", **arrs)
            manifest["cases"][name] = dict(kind="model", cfg={k: (list(map(list, v)) if k == "kernel_set" else v)
                                                               for k, v in cfg.items()})
            print(name, rate.shape, "last-layer periods", arrs["periods"].tolist())

    (HERE / "manifest.json").write_text(json.dumps(manifest, indent=1, sort_keys=True))
    total = sum(p.stat().st_size for p in HERE.glob("*.npz"))
    print(f"wrote {len(manifest['cases'])} cases, {total/1e6:.2f} MB")


if __name__ == "__main__":
    main()
