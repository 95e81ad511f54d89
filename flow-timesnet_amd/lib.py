"""ctypes binding of ``csrc/libflowtimes_hip.so`` (C ABI: ``include/flowtimes.h``).

The library is loaded lazily and loudly: there is no CPU or PyTorch substitute
behind these calls.  ``load()`` raises ``FlowTimesLibraryError`` if the shared
object is missing or does not export every symbol the header declares.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path
from typing import Optional

FTN_KMAX = 16
FTN_MAXBR = 8
ABI_VERSION = 10

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "csrc" / "libflowtimes_hip.so"


class FlowTimesLibraryError(RuntimeError):
    pass


FTN_XCHG_MAXWORLD = 16


class FtnExchange(C.Structure):
    """Mirror of ``struct FtnExchange`` (include/flowtimes.h): the peer-mapped exchange buffers of a batch-sharded run."""

    _fields_ = [("slots", C.c_void_p * FTN_XCHG_MAXWORLD), ("world", C.c_int32), ("rank", C.c_int32),
                ("F_cap", C.c_int32), ("seq", C.c_uint64)]


class FtnDesc(C.Structure):
    """Mirror of ``struct FtnDesc`` (include/flowtimes.h)."""

    _fields_ = [
        ("n_sel", C.c_int32), ("n_groups", C.c_int32), ("total_px", C.c_int32), ("tiles_per_row", C.c_int32),
        ("sel_freq", C.c_int32 * FTN_KMAX), ("sel_period", C.c_int32 * FTN_KMAX), ("sel_group", C.c_int32 * FTN_KMAX),
        ("g_period", C.c_int32 * FTN_KMAX), ("g_pad", C.c_int32 * FTN_KMAX), ("g_cycles", C.c_int32 * FTN_KMAX),
        ("g_px_off", C.c_int32 * (FTN_KMAX + 1)),
        ("g_tw", C.c_int32 * FTN_KMAX), ("g_th", C.c_int32 * FTN_KMAX),
        ("g_ntx", C.c_int32 * FTN_KMAX), ("g_nty", C.c_int32 * FTN_KMAX),
        ("g_tile_off", C.c_int32 * (FTN_KMAX + 1)),
    ]


DESC_INTS = C.sizeof(FtnDesc) // 4


class FtnPlan(C.Structure):
    """Mirror of ``struct FtnPlan`` (include/flowtimes.h)."""

    _fields_ = [
        ("C", C.c_int32), ("CP", C.c_int32), ("F", C.c_int32), ("FP", C.c_int32),
        ("mode", C.c_int32), ("act", C.c_int32), ("nbr", C.c_int32), ("MP", C.c_int32),
        ("kh", C.c_int32 * FTN_MAXBR), ("kw", C.c_int32 * FTN_MAXBR),
        ("res1", C.c_int32), ("res2", C.c_int32),
        ("w_in1", C.c_int64), ("b_in1", C.c_int64),
        ("w_conv1", C.c_int64 * FTN_MAXBR), ("b_conv1", C.c_int64),
        ("w_out1", C.c_int64), ("b_out1", C.c_int64),
        ("w_res1", C.c_int64), ("b_res1", C.c_int64),
        ("w_in2", C.c_int64), ("b_in2", C.c_int64),
        ("w_conv2", C.c_int64 * FTN_MAXBR), ("b_conv2", C.c_int64),
        ("w_out2", C.c_int64), ("b_out2", C.c_int64),
        ("w_res2", C.c_int64), ("b_res2", C.c_int64),
        ("w_c2", C.c_int64), ("b_c2", C.c_int64),
        ("w_cfrag", C.c_int64), ("cfrag_per_chunk", C.c_int32), ("n_hchunks", C.c_int32),
        ("w_convbf1", C.c_int64 * FTN_MAXBR), ("w_convbf2", C.c_int64 * FTN_MAXBR),
        ("engine", C.c_int32), ("cfragbf_per_chunk", C.c_int32), ("w_cfragbf", C.c_int64),
        ("b_conv1s", C.c_int64), ("b_conv2s", C.c_int64), ("b_out1s", C.c_int64), ("b_res1s", C.c_int64),
        ("b_c2s", C.c_int64),
        ("sc_conv1", C.c_float * FTN_MAXBR), ("sc_conv2", C.c_float * FTN_MAXBR),
        ("sc_out1", C.c_float), ("sc_res1", C.c_float), ("sc_a2", C.c_float), ("sc_r2", C.c_float),
        ("w_out2fb", C.c_int64), ("b_out2s", C.c_int64), ("sc_out2", C.c_float), ("reserved0", C.c_int32),
        ("total_floats", C.c_int64),
    ]


class FtnInceptionBlockWeights(C.Structure):
    """Mirror of ``struct FtnInceptionBlockWeights`` (include/flowtimes.h): raw host pointers to the reference
    ``state_dict`` tensors of one InceptionBlock."""

    _fields_ = [
        ("branch_w", (C.c_void_p * 3) * FTN_MAXBR), ("branch_b", (C.c_void_p * 3) * FTN_MAXBR),
        ("proj_w", C.c_void_p), ("proj_b", C.c_void_p), ("res_w", C.c_void_p), ("res_b", C.c_void_p),
    ]


_P = C.c_void_p
_SIGNATURES = {
    # name: (restype, argtypes)            -- one entry per declaration in flowtimes.h
    "ftn_abi_version": (C.c_int, []),
    "ftn_last_error": (C.c_char_p, []),
    "ftn_inception_pack_floats": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                              C.c_double, C.c_int]),
    "ftn_inception_pack_weights": (C.c_int, [C.POINTER(FtnInceptionBlockWeights), C.POINTER(FtnInceptionBlockWeights),
                                             C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                             C.c_double, C.c_int, C.c_int, _P, C.c_size_t, C.POINTER(FtnPlan)]),
    "ftn_dft_table_bytes": (C.c_size_t, [C.c_int]),
    "ftn_dft_table_init": (C.c_int, [_P, C.c_int, _P]),
    "ftn_period_spectrum_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "ftn_period_spectrum": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, C.POINTER(FtnExchange), _P]),
    "ftn_exchange_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "ftn_exchange_error": (C.c_int, [C.POINTER(FtnExchange), _P]),
    "ftn_exchange_alloc": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_void_p), _P]),
    "ftn_exchange_open": (C.c_int, [_P, C.POINTER(C.c_void_p)]),
    "ftn_exchange_close": (C.c_int, [_P]),
    "ftn_exchange_free": (C.c_int, [_P]),
    "ftn_period_finalize": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_double, _P, _P, _P, _P, C.POINTER(FtnExchange)]),
    "ftn_desc_from_periods": (C.c_int, [C.POINTER(C.c_int64), C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.POINTER(FtnDesc)]),
    "ftn_selector_px_bound": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "ftn_timesblock_workspace_bytes": (C.c_size_t, [C.POINTER(FtnPlan), C.c_int, C.c_int, C.c_int, C.c_int]),
    "ftn_timesblock_forward": (C.c_int, [_P, _P, C.c_int, C.c_int, C.POINTER(FtnPlan), _P, _P, _P, C.c_int, C.c_int,
                                         C.c_int, C.c_int, _P, C.c_size_t, _P, _P]),
    "ftn_timesblock_forward_norm": (C.c_int, [_P, _P, C.c_int, C.c_int, C.POINTER(FtnPlan), _P, _P, _P, C.c_int,
                                              C.c_int, C.c_int, _P, _P, C.c_float, _P, C.c_size_t, _P, _P]),
    "ftn_period_finalize_stage_a": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                              C.c_int, C.c_int, C.c_double, _P, _P, _P, _P, C.POINTER(FtnPlan), _P,
                                              C.c_int, C.c_int, _P, C.c_size_t, _P, _P, C.POINTER(FtnExchange)]),
    "ftn_residual_layernorm": (C.c_int, [_P, _P, _P, C.c_longlong, C.c_int, _P, _P, C.c_float, _P]),
    "ftn_head_forward": (C.c_int, [_P, C.c_longlong, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, C.c_longlong,
                                   C.c_int, _P, C.c_longlong, _P, C.c_float, _P, _P, _P, _P]),
    "ftn_embed_forward": (C.c_int, [_P, C.c_longlong, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P, C.c_longlong,
                                    _P, _P, C.c_float, _P, _P]),
    "ftn_lrtc_basis_floats": (C.c_size_t, [C.c_int, C.c_int]),
    "ftn_lrtc_basis": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "ftn_lrtc_forward": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "ftn_stage_timing": (C.c_int, [C.c_int]),
    "ftn_stage_times": (C.c_int, [C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_int)]),
    "ftn_debug_stamps": (C.c_int, [_P, C.c_size_t, C.c_int]),
    "ftn_selftest_mfma": (C.c_int, [_P, _P]),
    "ftn_selftest_gelu": (C.c_int, [_P, _P, C.c_longlong, _P]),
}
EXPORTS = tuple(_SIGNATURES)

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load the shared library (once) and bind every exported symbol."""
    global _lib
    if _lib is not None:
        return _lib
    path = Path(os.environ.get("FLOWTIMES_LIB", LIB_PATH))
    build_log = ""
    if "FLOWTIMES_LIB" not in os.environ:
        # In-tree build on every first use (hipcc cross-compiles gfx950 without a GPU): `make` is a no-op when
        # the library is newer than its sources and rebuilds it when it is not, so a stale .so (it is git-ignored
        # but travels with the tree) can never be what the tests validate.  A failed build is an error with the
        # compiler's output attached, never a fallback.
        import shutil
        import subprocess

        hipcc = shutil.which("hipcc") or ("/opt/rocm/bin/hipcc" if Path("/opt/rocm/bin/hipcc").exists() else None)
        csrc = _HERE / "csrc"

        def current() -> bool:
            """The library exists and is newer than every source it is built from."""
            if not path.exists():
                return False
            srcs = [*csrc.glob("*.hip"), *csrc.glob("*.h"), csrc / "Makefile", _HERE.parent / "include" / "flowtimes.h"]
            return all(not f.exists() or f.stat().st_mtime <= path.stat().st_mtime for f in srcs)

        if shutil.which("make") and hipcc and not (current() and not os.access(csrc, os.W_OK)):
            import fcntl

            # one builder at a time: the ranks of a multi-GPU launch all arrive here together
            try:
                lock = open(csrc / ".build.lock", "w")
            except OSError as exc:                              # read-only tree (an installed copy)
                if not current():
                    raise FlowTimesLibraryError(f"{csrc} is not writable and {path} is missing or older than its "
                                                f"sources: {exc}") from exc
                lock = None
            if lock is not None:
                with lock:
                    fcntl.flock(lock, fcntl.LOCK_EX)
                    try:
                        proc = subprocess.run(["make", "-C", str(csrc), "-j4"], check=False,
                                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
                        if proc.returncode != 0:
                            build_log = proc.stdout[-4000:]
                    finally:
                        fcntl.flock(lock, fcntl.LOCK_UN)
            if build_log:
                # a broken toolchain beside a library that is provably current (newer than every source) is not a
                # stale-library risk: use it and say so; anything else stays an error
                if not current():
                    raise FlowTimesLibraryError(f"building {path} failed; there is no fallback path:\n{build_log}")
                import warnings

                warnings.warn(f"`make` failed but {path.name} is newer than all of its sources - using it.\n{build_log[-500:]}",
                              RuntimeWarning, stacklevel=2)
    if not path.exists():
        raise FlowTimesLibraryError(
            f"{path} not found: build it with `make -C {_HERE / 'csrc'}` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`); there is no fallback path"
        )
    try:
        lib = C.CDLL(str(path))
    except OSError as exc:  # missing ROCm runtime etc.
        raise FlowTimesLibraryError(f"cannot load {path}: {exc}") from exc
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise FlowTimesLibraryError(f"{path} does not export {name}") from exc
        fn.restype = res
        fn.argtypes = args
    if lib.ftn_abi_version() != ABI_VERSION:
        raise FlowTimesLibraryError(f"ABI version {lib.ftn_abi_version()} != {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().ftn_last_error().decode("utf-8", "replace")
        kind = ValueError if rc < 0 else RuntimeError
        raise kind(f"{what} failed (rc={rc}): {msg}")


def desc_from_periods(periods, L: int, min_period: int, max_period: int) -> FtnDesc:
    """Host-side PeriodGrouper (default flags) + conv tiling, via the C helper."""
    lib = load()
    K = len(periods)
    arr = (C.c_int64 * max(K, 1))(*[int(p) for p in periods])
    d = FtnDesc()
    check(lib.ftn_desc_from_periods(arr, K, int(L), int(min_period), int(max_period), C.byref(d)),
          "ftn_desc_from_periods")
    return d
