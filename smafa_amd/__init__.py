"""smafa_amd — MI355X-native engine for smafa's fixed-length Hamming scan (query / cluster).

HIP kernels for gfx950 behind the C ABI of ``include/smafa_amd.h``; this package is the ctypes
mirror of the reference crate's interface for that path.  No CPU fallback.
"""
from ._lib import ALPHABET_AA, ALPHABET_NT, NONE, SmafaError, SmafaPanic, build  # noqa: F401
from .api import (HIT_DTYPE, QuerySet, SubjectGroup, SubjectStore, build_id, cluster, hbm_read_probe, count, decode, device_count, encode,  # noqa: F401
                  encode_rows, load_fastx, load_fastx_part, makedb, makedb_packed, query, read_db, select_rows, write_db, write_rows)
