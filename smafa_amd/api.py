"""Host-side mirror of the reference crate's interface for the hot path.

``makedb`` / ``query`` / ``cluster`` / ``count`` take the arguments of the reference's public
functions (/root/reference/src/lib.rs:137,198,378; src/cluster.rs:13) — paths, and ``None`` for an
absent option — and produce the same bytes.  ``SubjectStore`` is the ``WindowSet`` of
src/lib.rs:54-135 living in HBM: ``push``/``get_distances``/``scan``.  Everything goes through the C
ABI of libsmafa_amd.so; nothing here computes distances on the CPU.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from . import _lib
from ._lib import ALPHABET_AA, ALPHABET_NT, NONE, Hit, SmafaError, SmafaPanic, check, lib

HIT_DTYPE = np.dtype([("query", "<u4"), ("subject", "<u4"), ("dist", "<u4")])


def _opt(v: Optional[int]) -> int:
    return NONE if v is None else int(v)


def device_count() -> int:
    return lib().smafa_device_count()


def build_id() -> str:
    return lib().smafa_build_id().decode()


def hbm_read_probe(device: int = 0, nbytes: int = 8 << 30) -> float:
    """Empirical HBM read-stream rate of the device in GB/s (a trivial sum kernel over `nbytes`)."""
    out = C.c_double(0.0)
    check(lib().smafa_hbm_read_probe(device, nbytes, C.byref(out)))
    return out.value


def encode(seq: bytes, alphabet: int = ALPHABET_NT) -> np.ndarray:
    """ASCII -> code bytes (create_lut / from_bytes, src/lib.rs:29-52,167-196)."""
    out = np.empty(len(seq), dtype=np.uint8)
    bad = C.c_uint64(0)
    check(lib().smafa_encode(alphabet, seq, len(seq), out.ctypes.data, C.byref(bad)))
    return out


def encode_rows(ascii_rows: np.ndarray, alphabet: int = ALPHABET_NT) -> np.ndarray:
    a = np.ascontiguousarray(ascii_rows, dtype=np.uint8)
    out = np.empty_like(a)
    bad = C.c_uint64(0)
    check(lib().smafa_encode(alphabet, a.ctypes.data, a.size, out.ctypes.data, C.byref(bad)))
    return out


def decode(codes: np.ndarray, alphabet: int = ALPHABET_NT) -> bytes:
    """code bytes -> subject string (get_as_string, src/lib.rs:113-135)."""
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    out = C.create_string_buffer(codes.size)
    check(lib().smafa_decode(alphabet, codes.ctypes.data, codes.size, out))
    return out.raw


class QuerySet:
    """A packed query batch resident in HBM."""

    def __init__(self, store: "SubjectStore", query_codes: np.ndarray):
        q = np.ascontiguousarray(query_codes, dtype=np.uint8)
        assert q.ndim == 2 and q.shape[1] == store.seq_len
        self.n = q.shape[0]
        self._h = C.c_void_p()
        self._store = store
        check(lib().smafa_qset_create(C.byref(self._h), store._h, q.ctypes.data, self.n))

    def close(self):
        if self._h:
            lib().smafa_qset_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SubjectStore:
    """WindowSet (src/lib.rs:54-135) resident in HBM as bit-planes."""

    def __init__(self, seq_len: int, alphabet: int = ALPHABET_NT, device: int = 0):
        self._h = C.c_void_p()
        self.seq_len = int(seq_len)
        self.alphabet = alphabet
        check(lib().smafa_db_create(C.byref(self._h), device, alphabet, self.seq_len))

    @classmethod
    def load(cls, path: str, device: int = 0) -> "SubjectStore":
        """a store from a packed store file (smafa_db_load): mmap + three host-to-device copies"""
        self = cls.__new__(cls)
        self._h = C.c_void_p()
        check(lib().smafa_db_load(C.byref(self._h), device, os.fsencode(path)))
        info = self.info()
        self.seq_len, self.alphabet = int(info.seq_len), int(info.alphabet)
        return self

    def save(self, path: str) -> None:
        check(lib().smafa_db_save(self._h, os.fsencode(path)))

    # push_encoding x n (src/lib.rs:91-111)
    def push(self, codes: np.ndarray) -> None:
        c = np.ascontiguousarray(codes, dtype=np.uint8)
        if c.ndim != 2 or c.shape[1] != self.seq_len:
            raise SmafaPanic(_lib.ERR_PANIC, "WindowSet seq length is %d, got a new sequence of length %d"
                             % (self.seq_len, c.shape[-1]))
        check(lib().smafa_db_append(self._h, c.ctypes.data, c.shape[0]))

    def __len__(self) -> int:
        return int(self.info().n_subjects)

    def info(self) -> _lib.DbInfo:
        info = _lib.DbInfo()
        check(lib().smafa_db_info(self._h, C.byref(info)))
        return info

    # get_distances (src/lib.rs:71-89)
    def get_distances(self, query_codes: np.ndarray) -> np.ndarray:
        q = np.ascontiguousarray(query_codes, dtype=np.uint8).reshape(-1)
        if q.size != self.seq_len:
            raise SmafaPanic(_lib.ERR_PANIC, "Cannot compute distances between seq of length %d and windows of lengths %d"
                             % (q.size, self.seq_len))
        out = np.zeros(len(self), dtype=np.uint32)
        check(lib().smafa_distances(self._h, q.ctypes.data, out.ctypes.data))
        return out

    def scan(self, query_codes: np.ndarray, max_divergence: Optional[int] = None,
             max_num_hits: Optional[int] = None) -> np.ndarray:
        """All (query, subject, dist) rows within the bounds, ordered (query, dist, subject)."""
        q = np.ascontiguousarray(query_codes, dtype=np.uint8)
        assert q.ndim == 2 and q.shape[1] == self.seq_len
        cap = 1 << 16
        while True:
            out = np.zeros(cap, dtype=HIT_DTYPE)
            n_out = C.c_uint64(0)
            rc = lib().smafa_scan_hits(self._h, q.ctypes.data, q.shape[0], _opt(max_divergence), _opt(max_num_hits),
                                       out.ctypes.data, cap, C.byref(n_out))
            if rc == _lib.ERR_CAPACITY:
                cap = int(n_out.value)
                continue
            check(rc)
            return out[: n_out.value]

    # ---- device-resident form -------------------------------------------------------------
    def set_stream(self, hip_stream: int) -> None:
        check(lib().smafa_db_set_stream(self._h, C.c_void_p(hip_stream)))

    def set_query_block(self, n: int) -> None:
        check(lib().smafa_set_query_block(self._h, n))

    def set_prefilter(self, enabled: bool) -> None:
        check(lib().smafa_set_prefilter(self._h, 1 if enabled else 0))

    def set_zone_level(self, mode: int) -> None:
        check(lib().smafa_set_zone_level(self._h, int(mode)))

    def build_index(self, max_divergence: int) -> dict:
        """the block index of the resident store (smafa_db_build_index): fixed bounds up to max_divergence are then answered
        by max_divergence + 1 probes per query wherever the store's blocks are selective enough"""
        check(lib().smafa_db_build_index(self._h, int(max_divergence)))
        return self.index_info()

    def drop_index(self) -> None:
        check(lib().smafa_db_drop_index(self._h))

    def set_index(self, mode: int) -> None:
        check(lib().smafa_set_index(self._h, int(mode)))

    def index_info(self) -> dict:
        info = _lib.IndexInfo()
        check(lib().smafa_index_info(self._h, C.byref(info)))
        d = {k: getattr(info, k) for k, _ in _lib.IndexInfo._fields_}
        d["max_div_served"] = None if d["max_div_served"] == _lib.NONE else d["max_div_served"]
        return d

    def scan_launch(self, qset: QuerySet, max_divergence: Optional[int], max_num_hits: Optional[int],
                    d_hits: int, cap: int, d_count: int) -> None:
        check(lib().smafa_scan_launch(self._h, qset._h, _opt(max_divergence), _opt(max_num_hits),
                                      C.c_void_p(d_hits), cap, C.c_void_p(d_count)))

    def scan_each(self, qset: QuerySet, max_divergence: Optional[int], d_hits: int, cap_per_query: int, d_counts: int,
                  use_graph: bool = False) -> None:
        """one store pass per query, enqueued back to back (smafa_scan_each)"""
        check(lib().smafa_scan_each(self._h, qset._h, _opt(max_divergence), C.c_void_p(d_hits), cap_per_query,
                                    C.c_void_p(d_counts), 1 if use_graph else 0))

    def last_call_stats(self) -> dict:
        ms, n, k = C.c_float(0), C.c_uint32(0), C.c_uint32(0)
        check(lib().smafa_last_call_stats(self._h, C.byref(ms), C.byref(n), C.byref(k)))
        return {"kernel_ms": ms.value, "launches": n.value, "scans": k.value}

    def launch_device(self) -> tuple[int, int]:
        """(HIP device current on the launching thread at the last scan launch, launches issued off the handle's device)"""
        dev, off = C.c_int(-1), C.c_uint64(0)
        check(lib().smafa_launch_device(self._h, C.byref(dev), C.byref(off)))
        return dev.value, off.value

    def sync(self) -> None:
        check(lib().smafa_sync(self._h))

    def last_scan_ms(self) -> tuple[float, int]:
        ms, n = C.c_float(0), C.c_uint32(0)
        check(lib().smafa_last_scan_ms(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def last_scan_plan(self) -> dict:
        a, b, c = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        check(lib().smafa_last_scan_plan(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return {"filter_plane_resident": bool(a.value), "tiles_per_wave": b.value, "query_blocks": c.value}

    def last_scan_kernel(self) -> str:
        buf = C.create_string_buffer(128)
        check(lib().smafa_last_scan_kernel(self._h, buf, 128))
        return buf.value.decode()

    def close(self):
        if self._h:
            lib().smafa_db_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SubjectGroup:
    """One WindowSet replicated over several GPUs behind one handle (smafa_group_*): `scan` has the contract of
    SubjectStore.scan, the batch cut into contiguous blocks, one per device (the loop of src/lib.rs:232-318 sharded)."""

    def __init__(self, seq_len: int, alphabet: int = ALPHABET_NT, devices=(0,)):
        self._h = C.c_void_p()
        self.seq_len, self.alphabet = int(seq_len), alphabet
        arr = (C.c_int * len(devices))(*[int(d) for d in devices])
        check(lib().smafa_group_create(C.byref(self._h), arr, len(devices), alphabet, self.seq_len))

    @classmethod
    def load(cls, path: str, devices=(0,)) -> "SubjectGroup":
        self = cls.__new__(cls)
        self._h = C.c_void_p()
        arr = (C.c_int * len(devices))(*[int(d) for d in devices])
        check(lib().smafa_group_load(C.byref(self._h), arr, len(devices), os.fsencode(path)))
        info = _lib.DbInfo()
        check(lib().smafa_db_info(lib().smafa_group_member(self._h, 0), C.byref(info)))
        self.seq_len, self.alphabet = int(info.seq_len), int(info.alphabet)
        return self

    def __len__(self) -> int:
        return lib().smafa_group_size(self._h)

    def build_index(self, max_divergence: int) -> list:
        """smafa_group_build_index: every replica builds its block index; -> per member smafa_index_info()"""
        check(lib().smafa_group_build_index(self._h, int(max_divergence)))
        out = []
        for g in range(len(self)):
            info = _lib.IndexInfo()
            check(lib().smafa_index_info(lib().smafa_group_member(self._h, g), C.byref(info)))
            out.append({k: getattr(info, k) for k, _ in _lib.IndexInfo._fields_})
        return out

    def members(self):
        """per member: (smafa_db_info().device, device current at its last scan launch, launches off its device)"""
        out = []
        for g in range(len(self)):
            h = lib().smafa_group_member(self._h, g)
            info = _lib.DbInfo()
            check(lib().smafa_db_info(h, C.byref(info)))
            dev, off = C.c_int(-1), C.c_uint64(0)
            check(lib().smafa_launch_device(h, C.byref(dev), C.byref(off)))
            out.append((int(info.device), dev.value, off.value))
        return out

    def push(self, codes: np.ndarray) -> None:
        c = np.ascontiguousarray(codes, dtype=np.uint8)
        if c.ndim != 2 or c.shape[1] != self.seq_len:
            raise SmafaPanic(_lib.ERR_PANIC, "WindowSet seq length is %d, got a new sequence of length %d"
                             % (self.seq_len, c.shape[-1]))
        check(lib().smafa_group_append(self._h, c.ctypes.data, c.shape[0]))

    def scan(self, query_codes: np.ndarray, max_divergence: Optional[int] = None,
             max_num_hits: Optional[int] = None, cap: int = 1 << 16) -> np.ndarray:
        q = np.ascontiguousarray(query_codes, dtype=np.uint8)
        assert q.ndim == 2 and q.shape[1] == self.seq_len
        while True:
            out = np.zeros(max(cap, 1), dtype=HIT_DTYPE)
            n_out = C.c_uint64(0)
            rc = lib().smafa_group_scan_hits(self._h, q.ctypes.data, q.shape[0], _opt(max_divergence), _opt(max_num_hits),
                                             out.ctypes.data, cap, C.byref(n_out))
            if rc == _lib.ERR_CAPACITY:
                cap = int(n_out.value)
                continue
            check(rc)
            return out[: n_out.value]

    def close(self):
        if self._h:
            lib().smafa_group_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def select_rows(hits: np.ndarray, n_queries: int, n_subjects: int, subject_codes: Optional[np.ndarray], seq_len: int,
                max_divergence: Optional[int], max_num_hits: Optional[int],
                limit_per_sequence: Optional[int]) -> np.ndarray:
    """Selection rules of src/lib.rs:241-315 over an ordered hit list."""
    hits = np.ascontiguousarray(hits, dtype=HIT_DTYPE)
    cap = max(len(hits), 1)
    rows = np.zeros(cap, dtype=HIT_DTYPE)
    n_rows = C.c_uint64(0)
    codes_ptr = None
    if subject_codes is not None:
        subject_codes = np.ascontiguousarray(subject_codes, dtype=np.uint8)
        codes_ptr = subject_codes.ctypes.data
    check(lib().smafa_select_rows(hits.ctypes.data, len(hits), n_queries, n_subjects, codes_ptr, seq_len,
                                  _opt(max_divergence), _opt(max_num_hits), _opt(limit_per_sequence),
                                  rows.ctypes.data, cap, C.byref(n_rows)))
    return rows[: n_rows.value]


def read_db(path: str) -> tuple[int, np.ndarray]:
    """DB file -> (alphabet, code rows)."""
    alphabet, ptr, n, L = C.c_int(0), C.c_void_p(), C.c_uint64(0), C.c_uint32(0)
    check(lib().smafa_dbfile_read(os.fsencode(path), C.byref(alphabet), C.byref(ptr), C.byref(n), C.byref(L)))
    try:
        size = n.value * L.value
        arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(max(size, 1),))[:size].copy()
    finally:
        lib().smafa_free(ptr)
    return alphabet.value, arr.reshape(n.value, L.value)


def load_fastx(path: str, alphabet: int = ALPHABET_NT) -> np.ndarray:
    """FASTA/FASTQ(+gzip) of equal-length records -> code rows (needletail + from_bytes, src/lib.rs:221,235)."""
    ptr, n, L = C.c_void_p(), C.c_uint64(0), C.c_uint32(0)
    check(lib().smafa_fastx_load(os.fsencode(path), alphabet, C.byref(ptr), C.byref(n), C.byref(L)))
    try:
        size = n.value * L.value
        arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(max(size, 1),))[:size].copy()
    finally:
        lib().smafa_free(ptr)
    return arr.reshape(n.value, L.value)


def load_fastx_partial(path: str, alphabet: int = ALPHABET_NT):
    """-> (code rows in front of the first offending record, None or the SmafaError that record raises)"""
    ptr, n, L, pending = C.c_void_p(), C.c_uint64(0), C.c_uint32(0), C.c_int(0)
    check(lib().smafa_fastx_load_partial(os.fsencode(path), alphabet, C.byref(ptr), C.byref(n), C.byref(L), C.byref(pending)))
    err = None
    if pending.value != _lib.OK:
        msg = lib().smafa_last_error().decode(errors="replace")
        err = (SmafaPanic if pending.value == _lib.ERR_PANIC else SmafaError)(pending.value, msg)
    try:
        size = n.value * L.value
        arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(max(size, 1),))[:size].copy()
    finally:
        lib().smafa_free(ptr)
    return arr.reshape(n.value, L.value), err


def load_fastx_part(path: str, part: int, parts: int, alphabet: int = ALPHABET_NT):
    """-> (code rows of the records that start in this part's byte range, pending error or None, usable)"""
    ptr, n, L, pending, usable = C.c_void_p(), C.c_uint64(0), C.c_uint32(0), C.c_int(0), C.c_int(0)
    check(lib().smafa_fastx_load_part(os.fsencode(path), alphabet, part, parts, C.byref(ptr), C.byref(n), C.byref(L),
                                      C.byref(pending), C.byref(usable)))
    err = None
    if pending.value != _lib.OK:
        msg = lib().smafa_last_error().decode(errors="replace")
        err = (SmafaPanic if pending.value == _lib.ERR_PANIC else SmafaError)(pending.value, msg)
    try:
        size = n.value * L.value
        arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(max(size, 1),))[:size].copy() if ptr else np.zeros(0, np.uint8)
    finally:
        lib().smafa_free(ptr)
    return arr.reshape(n.value, L.value) if L.value else arr.reshape(0, 0), err, bool(usable.value)


def write_db(path: str, codes: np.ndarray, alphabet: int = ALPHABET_NT) -> None:
    c = np.ascontiguousarray(codes, dtype=np.uint8)
    check(lib().smafa_dbfile_write(os.fsencode(path), alphabet, c.ctypes.data, c.shape[0], c.shape[1]))


def write_rows(rows: np.ndarray, subject_codes: np.ndarray, alphabet: int, out_fd: int = 1, query_offset: int = 0) -> None:
    """TSV rows of `query` (src/lib.rs:292,310) for an ordered row list, written to out_fd."""
    rows = np.ascontiguousarray(rows, dtype=HIT_DTYPE)
    subject_codes = np.ascontiguousarray(subject_codes, dtype=np.uint8)
    n, L = subject_codes.shape
    check(lib().smafa_write_rows(rows.ctypes.data, len(rows), subject_codes.ctypes.data, n, L, alphabet, query_offset, out_fd))


# ------------------------------------------------------------------ the crate's pub fns
def makedb(subject_fasta: str, db_path: str, alphabet: int = ALPHABET_NT) -> None:
    """makedb(subject_fasta, db_path) — src/lib.rs:137-165."""
    check(lib().smafa_makedb(os.fsencode(subject_fasta), os.fsencode(db_path), alphabet))


def makedb_packed(subject_fasta: str, db_path: str, alphabet: int = ALPHABET_NT, device: int = 0) -> None:
    """makedb writing the packed store file; packed on GPU `device`, or by host threads when device < 0 or there is no GPU."""
    check(lib().smafa_makedb_packed(os.fsencode(subject_fasta), os.fsencode(db_path), alphabet, device))


def query(db_path: str, query_fasta: str, max_divergence: Optional[int] = None, max_num_hits: Optional[int] = None,
          limit_per_sequence: Optional[int] = None, out_fd: int = 1, device: int = 0, devices=None) -> None:
    """query(db_path, query_fasta, max_divergence, max_num_hits, limit_per_sequence) — src/lib.rs:198-325.
    `devices`: a list of GPU ordinals (entries may repeat) = one process, one handle and host thread per entry."""
    if devices is not None:
        arr = (C.c_int * len(devices))(*[int(d) for d in devices])
        check(lib().smafa_query_multi(os.fsencode(db_path), os.fsencode(query_fasta), _opt(max_divergence),
                                      _opt(max_num_hits), _opt(limit_per_sequence), out_fd, arr, len(devices)))
        return
    check(lib().smafa_query(os.fsencode(db_path), os.fsencode(query_fasta), _opt(max_divergence), _opt(max_num_hits),
                            _opt(limit_per_sequence), out_fd, device))


def cluster(input_fasta: str, max_divergence: int, out_fd: int = 1, device: int = 0,
            alphabet: int = ALPHABET_NT, devices=None) -> None:
    """cluster(input_fasta, max_divergence, print_stream) — src/cluster.rs:13-94.
    `devices`: a list of GPU ordinals (entries may repeat) = one process, one host thread and centroid replica per entry."""
    if devices is not None:
        arr = (C.c_int * len(devices))(*[int(d) for d in devices])
        check(lib().smafa_cluster_multi(os.fsencode(input_fasta), int(max_divergence), out_fd, arr, len(devices), alphabet))
        return
    check(lib().smafa_cluster(os.fsencode(input_fasta), int(max_divergence), out_fd, device, alphabet))


def count(paths, out_fd: int = 1) -> None:
    """count(paths) — src/lib.rs:378-398."""
    arr = (C.c_char_p * len(paths))(*[os.fsencode(p) for p in paths])
    check(lib().smafa_count(arr, len(paths), out_fd))
