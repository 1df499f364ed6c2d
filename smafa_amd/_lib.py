"""ctypes loader for libsmafa_amd.so (the C ABI of include/smafa_amd.h).

The library is built in-tree by ``smafa_amd/csrc/Makefile`` (hipcc, gfx950).  There is no
Python or CPU fallback: if the shared object is missing this module raises, and every scan
entry point fails with SMAFA_ERR_DEVICE when no HIP device is visible.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# SMAFA_AMD_LIB: an alternative build of the same library (A/B runs of kernel variants: tools/variants.sh)
LIB_PATH = os.environ.get("SMAFA_AMD_LIB") or os.path.join(_HERE, "lib", "libsmafa_amd.so")
CLI_PATH = os.path.join(_HERE, "bin", "smafa")

OK = 0
ERR_INVALID, ERR_DEVICE, ERR_CAPACITY, ERR_IO, ERR_FORMAT, ERR_PANIC, ERR_NOMEM = -1, -2, -3, -4, -5, -6, -7
NONE = 0xFFFFFFFF
ALPHABET_NT, ALPHABET_AA = 0, 1

# every symbol include/smafa_amd.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "smafa_last_error", "smafa_device_count", "smafa_set_verbosity", "smafa_encode", "smafa_decode",
    "smafa_db_create", "smafa_db_append", "smafa_db_save", "smafa_db_load", "smafa_db_info", "smafa_db_set_stream", "smafa_db_destroy",
    "smafa_scan_hits", "smafa_distances", "smafa_qset_create", "smafa_qset_destroy", "smafa_scan_launch",
    "smafa_scan_each", "smafa_last_call_stats", "smafa_launch_device",
    "smafa_sync", "smafa_last_scan_ms", "smafa_last_scan_plan", "smafa_last_scan_kernel", "smafa_build_id", "smafa_hbm_read_probe", "smafa_set_query_block", "smafa_set_prefilter", "smafa_set_zone_level", "smafa_db_build_index", "smafa_db_drop_index", "smafa_index_info", "smafa_set_index", "smafa_select_rows", "smafa_write_rows",
    "smafa_dbfile_write", "smafa_dbfile_read", "smafa_fastx_load", "smafa_fastx_load_partial", "smafa_fastx_load_part", "smafa_free",
    "smafa_group_create", "smafa_group_load", "smafa_group_append", "smafa_group_scan_hits", "smafa_group_build_index", "smafa_group_size",
    "smafa_group_member", "smafa_group_destroy",
    "smafa_qsession_open", "smafa_qsession_info", "smafa_qsession_scan_part", "smafa_qsession_write", "smafa_qsession_close",
    "smafa_makedb", "smafa_makedb_packed", "smafa_query", "smafa_query_multi", "smafa_cluster", "smafa_cluster_multi", "smafa_cluster_sharded", "smafa_count",
]


# smafa_allgather_fn: int (*)(void *ctx, const void *send, uint64_t n, const void **recv, uint64_t *recv_n)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64))


class Hit(C.Structure):
    _fields_ = [("query", C.c_uint32), ("subject", C.c_uint32), ("dist", C.c_uint32)]


class DbInfo(C.Structure):
    _fields_ = [
        ("n_subjects", C.c_uint64), ("seq_len", C.c_uint32), ("alphabet", C.c_int32), ("device", C.c_int32),
        ("planes", C.c_uint32), ("words_per_plane", C.c_uint32), ("hbm_bytes", C.c_uint64),
        ("bytes_per_subject", C.c_uint64),
    ]


class IndexInfo(C.Structure):
    _fields_ = [
        ("mode", C.c_int32), ("current", C.c_int32), ("blocks", C.c_uint32), ("usable_blocks", C.c_uint32),
        ("max_div_served", C.c_uint32), ("probe_launches", C.c_uint32), ("bytes", C.c_uint64), ("longest_run", C.c_uint64),
        ("candidates_per_query", C.c_double), ("build_ms", C.c_double),
    ]


def build(force: bool = False) -> None:
    """Compile the HIP extension and the CLI for gfx950 (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC, "-j8"]
    if force:
        args.append("-B")
    r = subprocess.run(args, capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError("building libsmafa_amd.so failed:\n" + r.stdout + r.stderr)


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C smafa_amd/csrc`). "
            "smafa_amd has no CPU fallback.")
    l = C.CDLL(LIB_PATH)
    u8p, u32p, u64p, vp = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.c_void_p
    l.smafa_last_error.restype = C.c_char_p
    l.smafa_device_count.restype = C.c_int
    l.smafa_set_verbosity.argtypes = [C.c_int]
    l.smafa_set_verbosity.restype = None
    l.smafa_encode.argtypes = [C.c_int, vp, C.c_uint64, vp, u64p]
    l.smafa_decode.argtypes = [C.c_int, vp, C.c_uint64, vp]
    l.smafa_db_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_uint32]
    l.smafa_db_append.argtypes = [vp, vp, C.c_uint64]
    l.smafa_db_save.argtypes = [vp, C.c_char_p]
    l.smafa_db_load.argtypes = [C.POINTER(vp), C.c_int, C.c_char_p]
    l.smafa_makedb_packed.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int]
    l.smafa_db_info.argtypes = [vp, C.POINTER(DbInfo)]
    l.smafa_db_set_stream.argtypes = [vp, vp]
    l.smafa_db_destroy.argtypes = [vp]
    l.smafa_db_destroy.restype = None
    l.smafa_scan_hits.argtypes = [vp, vp, C.c_uint64, C.c_uint32, C.c_uint32, vp, C.c_uint64, u64p]
    l.smafa_distances.argtypes = [vp, vp, vp]
    l.smafa_qset_create.argtypes = [C.POINTER(vp), vp, vp, C.c_uint64]
    l.smafa_qset_destroy.argtypes = [vp]
    l.smafa_qset_destroy.restype = None
    l.smafa_scan_launch.argtypes = [vp, vp, C.c_uint32, C.c_uint32, vp, C.c_uint64, vp]
    l.smafa_scan_each.argtypes = [vp, vp, C.c_uint32, vp, C.c_uint64, vp, C.c_int]
    l.smafa_last_call_stats.argtypes = [vp, C.POINTER(C.c_float), u32p, u32p]
    l.smafa_launch_device.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_uint64)]
    l.smafa_sync.argtypes = [vp]
    l.smafa_last_scan_ms.argtypes = [vp, C.POINTER(C.c_float), u32p]
    l.smafa_last_scan_plan.argtypes = [vp, u32p, u32p, u32p]
    l.smafa_last_scan_kernel.argtypes = [vp, C.c_char_p, C.c_uint64]
    l.smafa_build_id.restype = C.c_char_p
    l.smafa_hbm_read_probe.argtypes = [C.c_int, C.c_uint64, C.POINTER(C.c_double)]
    l.smafa_set_query_block.argtypes = [vp, C.c_uint32]
    l.smafa_set_prefilter.argtypes = [vp, C.c_int]
    l.smafa_set_zone_level.argtypes = [vp, C.c_int]
    l.smafa_db_build_index.argtypes = [vp, C.c_uint32]
    l.smafa_db_drop_index.argtypes = [vp]
    l.smafa_index_info.argtypes = [vp, C.POINTER(IndexInfo)]
    l.smafa_set_index.argtypes = [vp, C.c_int]
    l.smafa_group_create.argtypes = [C.POINTER(vp), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_uint32]
    l.smafa_group_load.argtypes = [C.POINTER(vp), C.POINTER(C.c_int), C.c_int, C.c_char_p]
    l.smafa_group_append.argtypes = [vp, vp, C.c_uint64]
    l.smafa_group_scan_hits.argtypes = [vp, vp, C.c_uint64, C.c_uint32, C.c_uint32, vp, C.c_uint64, u64p]
    l.smafa_group_build_index.argtypes = [vp, C.c_uint32]
    l.smafa_group_size.argtypes = [vp]
    l.smafa_group_member.argtypes = [vp, C.c_int]
    l.smafa_group_member.restype = vp
    l.smafa_group_destroy.argtypes = [vp]
    l.smafa_group_destroy.restype = None
    l.smafa_qsession_open.argtypes = [C.POINTER(vp), C.c_char_p, C.c_int]
    l.smafa_qsession_info.argtypes = [vp, u64p, u32p, C.POINTER(C.c_int)]
    l.smafa_qsession_scan_part.argtypes = [vp, C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                                           C.POINTER(vp), u64p, u64p, u64p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    l.smafa_qsession_write.argtypes = [vp, vp, C.c_uint64, C.c_int]
    l.smafa_qsession_close.argtypes = [vp]
    l.smafa_qsession_close.restype = None
    l.smafa_select_rows.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64, vp, C.c_uint32, C.c_uint32, C.c_uint32,
                                    C.c_uint32, vp, C.c_uint64, u64p]
    l.smafa_write_rows.argtypes = [vp, C.c_uint64, vp, C.c_uint64, C.c_uint32, C.c_int, C.c_uint32, C.c_int]
    l.smafa_dbfile_write.argtypes = [C.c_char_p, C.c_int, vp, C.c_uint64, C.c_uint32]
    l.smafa_dbfile_read.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(vp), u64p, u32p]
    l.smafa_fastx_load.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp), u64p, u32p]
    l.smafa_fastx_load_partial.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp), u64p, u32p, C.POINTER(C.c_int)]
    l.smafa_fastx_load_part.argtypes = [C.c_char_p, C.c_int, C.c_uint32, C.c_uint32, C.POINTER(vp), u64p, u32p, C.POINTER(C.c_int),
                                        C.POINTER(C.c_int)]
    l.smafa_free.argtypes = [vp]
    l.smafa_free.restype = None
    l.smafa_makedb.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
    l.smafa_query.argtypes = [C.c_char_p, C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int]
    l.smafa_query_multi.argtypes = [C.c_char_p, C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_int), C.c_int]
    l.smafa_cluster.argtypes = [C.c_char_p, C.c_uint32, C.c_int, C.c_int, C.c_int]
    l.smafa_cluster_multi.argtypes = [C.c_char_p, C.c_uint32, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int]
    l.smafa_cluster_sharded.argtypes = [C.c_char_p, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint32,
                                        ALLGATHER_FN, vp]
    l.smafa_count.argtypes = [C.POINTER(C.c_char_p), C.c_uint64, C.c_int]
    _lib = l
    return l


class SmafaError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(message)
        self.code = code


class SmafaPanic(SmafaError):
    """An input on which the reference panics (message preserved)."""


def check(rc: int) -> None:
    if rc != OK:
        msg = lib().smafa_last_error().decode(errors="replace")
        raise (SmafaPanic if rc == ERR_PANIC else SmafaError)(rc, msg)
