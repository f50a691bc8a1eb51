"""Loader for the HIP shared library (csrc/librelmc.so, C ABI of include/relmc.h).

Fails loudly when the library is missing or cannot be loaded: this package has no CPU
evaluation path (the CPU restatement under ``oracle/`` is test infrastructure and is never
imported from here).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

from . import _abi

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.environ.get("RELMC_LIB_PATH") or os.path.join(_CSRC, "librelmc.so")   # override: profiling builds only

# every symbol include/relmc.h declares (checked at load time and by tests/test_abi.py)
EXPORTS = [
    "relmc_ctx_create", "relmc_ctx_destroy", "relmc_last_error", "relmc_version",
    "relmc_solver_opts_default", "relmc_nsq_opts_default", "relmc_case_load",
    "relmc_case_thresholds", "relmc_mc_sampling", "relmc_mc_sampling_dev",
    "relmc_mc_simulation", "relmc_mc_simulation_dev", "relmc_nsq_accumulate", "relmc_nsq_accumulate_distinct",
    "relmc_last_kernel_ms", "relmc_acc_zero", "relmc_acc_merge", "relmc_nsq_indices",
    "relmc_nsq_run", "relmc_hl1_load", "relmc_hl1_nsq",
    "relmc_comm_unique_id", "relmc_comm_init", "relmc_comm_allreduce_acc", "relmc_comm_destroy", "relmc_comm_set_host_allreduce", "relmc_comm_info",
    "relmc_db_reset", "relmc_nsq_db_batch", "relmc_db_accumulate", "relmc_db_size", "relmc_db_export", "relmc_db_import",
    "relmc_seq_load", "relmc_seq_mcsampling", "relmc_seq_mcsimulation", "relmc_seq_years", "relmc_retry_stats", "relmc_retry_overflow", "relmc_retry_dense_stats", "relmc_case_order",
    "relmc_case_order_hint", "relmc_tune_order",
    "relmc_comm_set_timeout", "relmc_comm_allreduce_f64", "relmc_comm_set_host_allreduce_f64", "relmc_device_pci_bus_id", "relmc_seq_opts_default", "relmc_seq_run",
]


class RelmcLibraryError(RuntimeError):
    pass


def build(verbose: bool = False) -> str:
    """Compile the HIP extension for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", _CSRC, "librelmc.so"]
    subprocess.check_call(cmd, stdout=None if verbose else subprocess.DEVNULL)
    return LIB_PATH


def code_object_sha256(path: str | None = None) -> str:
    """sha256 of the device code a library carries: the bytes of its `.hip_fatbin` ELF section (the clang offload bundles of every unit,
    i.e. the gfx950 code objects; host code is not part of it).  bench.py prints it and the committed profile summaries record it, so a line
    can tell whether the counters it quotes were measured on the binary that ran (scripts/summarize_profile.py)."""
    import hashlib
    import struct
    path = path or LIB_PATH
    with open(path, "rb") as fh:
        b = fh.read()
    if b[:4] != b"\x7fELF" or b[4] != 2 or b[5] != 1:
        raise RelmcLibraryError(f"{path}: not a little-endian ELF64 file")
    shoff, = struct.unpack_from("<Q", b, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", b, 0x3A)
    sec = lambda i: struct.unpack_from("<IIQQQQIIQQ", b, shoff + i * shentsize)       # name, type, flags, addr, offset, size, ...
    str_off = sec(shstrndx)[4]
    h = hashlib.sha256()
    found = False
    for i in range(shnum):
        nm, _, _, _, off, size = sec(i)[:6]
        name = b[str_off + nm:b.index(b"\0", str_off + nm)]
        if name == b".hip_fatbin":
            h.update(b[off:off + size]); found = True
    if not found:
        raise RelmcLibraryError(f"{path}: no .hip_fatbin section (not a HIP library)")
    return h.hexdigest()


_lib = None


def load():
    """ctypes handle with argtypes set.  Raises RelmcLibraryError if the extension is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RelmcLibraryError(
            f"HIP extension not built: {LIB_PATH} is missing. Run "
            f"`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C {_CSRC}`); "
            "there is no CPU fallback.")
    try:
        L = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover - depends on the machine
        raise RelmcLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    missing = [s for s in EXPORTS if not hasattr(L, s)]
    if missing:
        raise RelmcLibraryError(f"{LIB_PATH} lacks symbols {missing}")
    vp = C.c_void_p
    u8p, dp, i32p = _abi.c_uint8_p, _abi.c_double_p, _abi.c_int32_p
    L.relmc_ctx_create.argtypes = [C.c_int32, C.POINTER(vp)]
    L.relmc_ctx_create.restype = C.c_int32
    L.relmc_ctx_destroy.argtypes = [vp]
    L.relmc_ctx_destroy.restype = None
    L.relmc_last_error.argtypes = [vp]
    L.relmc_last_error.restype = C.c_char_p
    L.relmc_version.restype = C.c_char_p
    L.relmc_solver_opts_default.argtypes = [C.POINTER(_abi.SolverOpts)]
    L.relmc_solver_opts_default.restype = None
    L.relmc_nsq_opts_default.argtypes = [C.POINTER(_abi.NsqOpts)]
    L.relmc_nsq_opts_default.restype = None
    L.relmc_case_load.argtypes = [vp, C.POINTER(_abi.CaseDesc)]
    L.relmc_case_load.restype = C.c_int32
    L.relmc_case_thresholds.argtypes = [vp, _abi.c_uint32_p]
    L.relmc_case_thresholds.restype = C.c_int32
    L.relmc_mc_sampling.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_int64, u8p]
    L.relmc_mc_sampling.restype = C.c_int32
    L.relmc_mc_sampling_dev.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_int64, vp]
    L.relmc_mc_sampling_dev.restype = C.c_int32
    L.relmc_mc_simulation.argtypes = [vp, u8p, C.c_int64, C.POINTER(_abi.SolverOpts), dp, dp, i32p, i32p]
    L.relmc_mc_simulation.restype = C.c_int32
    L.relmc_mc_simulation_dev.argtypes = [vp, vp, C.c_int64, C.POINTER(_abi.SolverOpts), vp, vp, vp, vp]
    L.relmc_mc_simulation_dev.restype = C.c_int32
    L.relmc_nsq_accumulate.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_int64, C.POINTER(_abi.SolverOpts),
                                       C.POINTER(_abi.Acc)]
    L.relmc_nsq_accumulate.restype = C.c_int32
    L.relmc_nsq_accumulate_distinct.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_int64, C.POINTER(_abi.SolverOpts),
                                                C.POINTER(_abi.Acc), C.POINTER(C.c_int64)]
    L.relmc_nsq_accumulate_distinct.restype = C.c_int32
    L.relmc_last_kernel_ms.argtypes = [vp, _abi.c_double_p]
    L.relmc_last_kernel_ms.restype = C.c_int32
    L.relmc_acc_zero.argtypes = [C.POINTER(_abi.Acc)]
    L.relmc_acc_zero.restype = None
    L.relmc_acc_merge.argtypes = [C.POINTER(_abi.Acc), C.POINTER(_abi.Acc)]
    L.relmc_acc_merge.restype = None
    L.relmc_nsq_indices.argtypes = [C.POINTER(_abi.Acc), C.c_int32, C.c_int32, C.c_double,
                                    C.POINTER(_abi.Indices)]
    L.relmc_nsq_indices.restype = None
    L.relmc_nsq_run.argtypes = [vp, C.POINTER(_abi.NsqOpts), C.POINTER(_abi.NsqResult)]
    L.relmc_nsq_run.restype = C.c_int32
    L.relmc_comm_unique_id.argtypes = [u8p]
    L.relmc_comm_unique_id.restype = C.c_int32
    L.relmc_comm_init.argtypes = [vp, C.c_int32, C.c_int32, u8p]
    L.relmc_comm_init.restype = C.c_int32
    L.relmc_comm_allreduce_acc.argtypes = [vp, C.POINTER(_abi.Acc)]
    L.relmc_comm_allreduce_acc.restype = C.c_int32
    L.relmc_comm_destroy.argtypes = [vp]
    L.relmc_comm_destroy.restype = C.c_int32
    L.relmc_comm_set_host_allreduce.argtypes = [vp, C.c_int32, C.c_int32, _abi.ALLREDUCE_FN, C.c_void_p]
    L.relmc_comm_set_host_allreduce.restype = C.c_int32
    L.relmc_comm_info.argtypes = [vp, _abi.c_int32_p, _abi.c_int32_p, _abi.c_int32_p, _abi.c_int64_p, _abi.c_double_p]
    L.relmc_comm_info.restype = C.c_int32
    L.relmc_comm_set_host_allreduce_f64.argtypes = [vp, _abi.ALLREDUCE_F64_FN, vp]
    L.relmc_comm_set_host_allreduce_f64.restype = C.c_int32
    L.relmc_comm_set_timeout.argtypes = [vp, C.c_double]
    L.relmc_comm_set_timeout.restype = C.c_int32
    L.relmc_comm_allreduce_f64.argtypes = [vp, dp, C.c_int64]
    L.relmc_comm_allreduce_f64.restype = C.c_int32
    L.relmc_device_pci_bus_id.argtypes = [vp, C.c_char_p, C.c_int32]
    L.relmc_device_pci_bus_id.restype = C.c_int32
    L.relmc_db_reset.argtypes = [vp]
    L.relmc_db_reset.restype = C.c_int32
    L.relmc_nsq_db_batch.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_int64, C.POINTER(_abi.SolverOpts), C.POINTER(_abi.Acc),
                                     C.POINTER(_abi.DbStats)]
    L.relmc_nsq_db_batch.restype = C.c_int32
    L.relmc_db_accumulate.argtypes = [vp, C.POINTER(_abi.Acc)]
    L.relmc_db_accumulate.restype = C.c_int32
    L.relmc_db_size.argtypes = [vp, _abi.c_int64_p, _abi.c_int64_p]
    L.relmc_db_size.restype = C.c_int32
    L.relmc_retry_stats.argtypes = [vp, _abi.c_int64_p, _abi.c_int64_p]
    L.relmc_retry_stats.restype = C.c_int32
    L.relmc_retry_overflow.argtypes = [vp, _abi.c_int64_p]
    L.relmc_retry_overflow.restype = C.c_int32
    L.relmc_retry_dense_stats.argtypes = [vp, _abi.c_int64_p, _abi.c_int64_p]
    L.relmc_retry_dense_stats.restype = C.c_int32
    if hasattr(L, "relmc_debug_mc_simulation_dense"):
        L.relmc_debug_mc_simulation_dense.argtypes = [vp, u8p, C.c_int64, C.c_void_p, dp, dp, i32p, i32p]
        L.relmc_debug_mc_simulation_dense.restype = C.c_int32
    L.relmc_case_order.argtypes = [vp, i32p, i32p]
    L.relmc_case_order.restype = C.c_int32
    L.relmc_case_order_hint.argtypes = [vp, i32p, C.c_int32]
    L.relmc_case_order_hint.restype = C.c_int32
    L.relmc_tune_order.argtypes = [C.c_void_p, C.c_int32, C.c_uint64, i32p, i32p, i32p]
    L.relmc_tune_order.restype = C.c_int32
    L.relmc_db_export.argtypes = [vp, C.c_int64, C.c_int64, u8p, _abi.c_int64_p, dp, i32p, dp, i32p, i32p, u8p]
    L.relmc_db_export.restype = C.c_int32
    L.relmc_db_import.argtypes = [vp, C.c_void_p, C.c_int64, u8p, _abi.c_int64_p, dp, dp, i32p, i32p, u8p]
    L.relmc_db_import.restype = C.c_int32
    if hasattr(L, "relmc_debug_screen_states"):
        L.relmc_debug_screen_states.argtypes = [vp, u8p, dp, C.c_int64, u8p]
        L.relmc_debug_screen_states.restype = C.c_int32
    if hasattr(L, "relmc_debug_set"):
        L.relmc_debug_set.argtypes = [vp, C.c_char_p, C.c_int32]
        L.relmc_debug_set.restype = C.c_int32
    if hasattr(L, "relmc_dpp_probe"):
        L.relmc_dpp_probe.argtypes = [vp, dp, dp]
        L.relmc_dpp_probe.restype = C.c_int32
    _lib = L
    return L
