"""flow-timesnet_amd — MI355X-native (gfx950) TimesBlock forward path for
Flow-TimesNet, behind the reference's own PyTorch-module API.

Sub-modules
-----------
``synth``            seeded synthetic weights / inputs (numpy only)
``lib``              ctypes binding of ``csrc/libflowtimes_hip.so`` (the C-ABI in ``include/flowtimes.h``)
``pack``             host-side weight folding/packing for the HIP kernels
``models.timesnet``  drop-in mirrors of the reference modules
``models.shell``     mirror of the TimesNet model shell; HIP embedding / head kernels around the blocks
``dist``             batch-sharded multi-GPU forward (RCCL via torch.distributed)
``graph``            HIP-graph capture / replay of an inference forward
"""
from . import synth  # noqa: F401


def __getattr__(name):  # lazy: keeps `import flow_timesnet_amd.synth` torch-free
    import importlib

    if name in ("lib", "pack", "models", "dist", "grouping", "runtime", "graph"):
        return importlib.import_module(f"{__name__}.{name}")
    raise AttributeError(name)
