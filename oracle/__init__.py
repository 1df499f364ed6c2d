"""CPU oracle for the smafa hot path — TEST INFRASTRUCTURE ONLY.

ctypes front end of ``oracle/_build/liboracle.so`` (built from ``smafa_oracle.c`` by
``oracle/Makefile``).  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module, and only as the checker.
The product package ``smafa_amd`` never imports it.

Parity status: nucleotide path PINNED by the reference's golden vectors
(tests/test_oracle_golden.py); amino-acid/code-byte functions are a build-defined
extension, UNPINNED upstream (the reference rejects amino-acid letters,
/root/reference/src/lib.rs:171-178).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
BUILD_DIR = os.path.join(_HERE, "_build")
CLI = os.path.join(BUILD_DIR, "smafa_oracle")
NO_LIMIT = -1
ALPHABET_NT = 0
ALPHABET_AA = 1

HIT_DTYPE = np.dtype([("query", "<u4"), ("subject", "<u4"), ("dist", "<u4")])


_made = False


def build(force: bool = False) -> None:
    """Compile the C restatement (gcc; seconds)."""
    global _made
    if force or not _made:  # make decides by file times: a stale build never outlives an edit of the sources
        subprocess.run(["make", "-C", _HERE] + (["-B"] if force else []), check=True, capture_output=True)
        _made = True


_libs: dict[str, C.CDLL] = {}


class _WindowSet(C.Structure):
    _fields_ = [
        ("version", C.c_uint32),
        ("n", C.c_size_t),
        ("len", C.c_size_t),
        ("nw", C.c_size_t),
        ("data", C.POINTER(C.c_uint64)),
        ("cap", C.c_size_t),
    ]


def lib(native: bool = False) -> C.CDLL:
    name = "liboracle_native.so" if native else "liboracle.so"
    if name in _libs:
        return _libs[name]
    build()
    l = C.CDLL(os.path.join(BUILD_DIR, name))
    l.orc_last_error.restype = C.c_char_p
    l.orc_lut_nt.restype = C.c_uint8
    l.orc_lut_nt.argtypes = [C.c_uint8]
    l.orc_code.restype = C.c_uint8
    l.orc_code.argtypes = [C.c_int, C.c_uint8]
    l.orc_words_for.restype = C.c_size_t
    l.orc_words_for.argtypes = [C.c_size_t]
    l.orc_encode_onehot.restype = C.c_int
    l.orc_encode_onehot.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_size_t)]
    for fn in (l.orc_scan_onehot, l.orc_scan_codes):
        fn.restype = C.c_int64
        fn.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint32, C.c_void_p, C.c_size_t]
    l.orc_distances_codes.restype = None
    l.orc_distances_codes.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    l.orc_bench_besthit_codes.restype = C.c_int64
    l.orc_bench_besthit_codes.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int64]
    l.orc_bench_besthit_onehot.restype = C.c_int64
    l.orc_bench_besthit_onehot.argtypes = [C.POINTER(_WindowSet), C.c_void_p, C.c_size_t, C.c_size_t, C.c_int64]
    l.orc_bench_kmode_onehot.restype = C.c_int64
    l.orc_bench_kmode_onehot.argtypes = [C.POINTER(_WindowSet), C.c_void_p, C.c_size_t, C.c_size_t, C.c_int64, C.c_int64]
    l.orc_ws_init.argtypes = [C.POINTER(_WindowSet), C.c_uint32]
    l.orc_ws_push.argtypes = [C.POINTER(_WindowSet), C.c_void_p, C.c_size_t]
    l.orc_ws_free.argtypes = [C.POINTER(_WindowSet)]
    l.orc_cluster_codes.restype = C.c_int
    l.orc_cluster_codes.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint32, C.c_void_p, C.c_void_p]
    _libs[name] = l
    return l


def last_error() -> str:
    return lib().orc_last_error().decode()


# --------------------------------------------------------------------------- encoding
def lut_nt(byte: int) -> int:
    return lib().orc_lut_nt(byte)


def encode_onehot(seq: bytes) -> list[int]:
    """src/lib.rs:29-52 — raises ValueError(position) for a non-nucleotide byte."""
    l = lib()
    nw = l.orc_words_for(len(seq))
    out = (C.c_uint64 * max(nw, 1))()
    bad = C.c_size_t(0)
    if l.orc_encode_onehot(seq, len(seq), out, C.byref(bad)) != 0:
        raise ValueError(bad.value)
    return [int(out[i]) for i in range(nw)]


def codes_from_ascii(ascii_rows: np.ndarray, alphabet: int) -> np.ndarray:
    """ASCII byte matrix -> code bytes (the oracle's own LUT, one call per distinct byte)."""
    l = lib()
    table = np.array([l.orc_code(alphabet, b) for b in range(256)], dtype=np.uint8)
    codes = table[ascii_rows]
    if (codes == 255).any():
        raise ValueError("byte outside alphabet")
    return codes


# ------------------------------------------------------------------------------- scans
def _scan(fn, subjects: np.ndarray, queries: np.ndarray, max_div: int) -> np.ndarray:
    subjects = np.ascontiguousarray(subjects, dtype=np.uint8)
    queries = np.ascontiguousarray(queries, dtype=np.uint8)
    n, L = subjects.shape if subjects.ndim == 2 else (0, queries.shape[1])
    q = queries.shape[0]
    assert queries.shape[1] == L
    cap = 1 << 16
    while True:
        out = np.zeros(cap, dtype=HIT_DTYPE)
        total = fn(subjects.ctypes.data, n, queries.ctypes.data, q, L, max_div, out.ctypes.data, cap)
        if total < 0:
            raise ValueError("encoding error in oracle scan")
        if total <= cap:
            return out[:total]
        cap = int(total)


def scan_onehot(subjects_ascii: np.ndarray, queries_ascii: np.ndarray, max_div: int) -> np.ndarray:
    """Reference-faithful scan (5-bit one-hot, xor+popcount/2): all hits d<=max_div, ordered (q,d,j)."""
    return _scan(lib().orc_scan_onehot, subjects_ascii, queries_ascii, max_div)


def scan_codes(subject_codes: np.ndarray, query_codes: np.ndarray, max_div: int) -> np.ndarray:
    """Code-byte scan (any alphabet): all hits d<=max_div, ordered (q,d,j)."""
    return _scan(lib().orc_scan_codes, subject_codes, query_codes, max_div)


def distances_codes(subject_codes: np.ndarray, query_codes: np.ndarray) -> np.ndarray:
    subject_codes = np.ascontiguousarray(subject_codes, dtype=np.uint8)
    query_codes = np.ascontiguousarray(query_codes, dtype=np.uint8)
    n, L = subject_codes.shape
    out = np.zeros(n, dtype=np.uint32)
    lib().orc_distances_codes(subject_codes.ctypes.data, n, query_codes.ctypes.data, L, out.ctypes.data)
    return out


def cluster_codes(codes: np.ndarray, max_div: int, alphabet: int = ALPHABET_NT) -> np.ndarray:
    """Greedy clustering of src/cluster.rs:13-94 on code bytes -> centroid ordinal per record
    (0xFFFFFFFF = exact duplicate, skipped)."""
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    n, L = codes.shape
    out = np.zeros(n, dtype=np.uint32)
    rc = lib().orc_cluster_codes(alphabet, codes.ctypes.data, codes.ctypes.data, n, L, max_div, None, out.ctypes.data)
    assert rc == 0
    return out


# ------------------------------------------------------------------ black-box CLI runs
def run_cli(*args: str) -> subprocess.CompletedProcess:
    build()
    return subprocess.run([CLI, *args], capture_output=True, text=True)


def write_fasta(path: str, rows, names=None) -> None:
    with open(path, "wb") as f:
        for i, r in enumerate(rows):
            name = names[i] if names else f"s{i}"
            f.write(b">" + name.encode() + b"\n" + (r if isinstance(r, bytes) else bytes(r)) + b"\n")


def query_text(subject_rows, query_rows, *flags: str) -> str:
    """makedb + query through the oracle CLI; returns stdout (raises on failure)."""
    with tempfile.TemporaryDirectory() as td:
        s, q, d = os.path.join(td, "s.fna"), os.path.join(td, "q.fna"), os.path.join(td, "db")
        write_fasta(s, subject_rows)
        write_fasta(q, query_rows)
        r = run_cli("makedb", "-i", s, "-d", d)
        if r.returncode:
            raise RuntimeError(r.stderr)
        r = run_cli("query", "-d", d, "-q", q, *flags)
        if r.returncode:
            raise RuntimeError(r.stderr)
        return r.stdout


def cluster_text(rows, max_div: int) -> str:
    with tempfile.TemporaryDirectory() as td:
        s = os.path.join(td, "s.fna")
        write_fasta(s, rows)
        r = run_cli("cluster", "-i", s, "-d", str(max_div))
        if r.returncode:
            raise RuntimeError(r.stderr)
        return r.stdout


# --------------------------------------------------------------------- cpu_baseline
def bench_besthit_codes(subject_codes: np.ndarray, query_codes: np.ndarray, max_div: int, native: bool = False) -> int:
    subject_codes = np.ascontiguousarray(subject_codes, dtype=np.uint8)
    query_codes = np.ascontiguousarray(query_codes, dtype=np.uint8)
    n, L = subject_codes.shape
    return lib(native).orc_bench_besthit_codes(subject_codes.ctypes.data, n, query_codes.ctypes.data,
                                               query_codes.shape[0], L, max_div)


class OnehotDB:
    """An orc_windowset built from ASCII nucleotide rows (for the reference-mirroring CPU baseline)."""

    def __init__(self, subjects_ascii: np.ndarray, native: bool = False):
        self._l = lib(native)
        self._ws = _WindowSet()
        self._l.orc_ws_init(C.byref(self._ws), 2)
        subjects_ascii = np.ascontiguousarray(subjects_ascii, dtype=np.uint8)
        self.L = subjects_ascii.shape[1]
        self._l.orc_ws_from_ascii.argtypes = [C.POINTER(_WindowSet), C.c_void_p, C.c_size_t, C.c_size_t]
        if self._l.orc_ws_from_ascii(C.byref(self._ws), subjects_ascii.ctypes.data, subjects_ascii.shape[0], self.L):
            raise ValueError(self._l.orc_last_error().decode())

    def encode_queries(self, queries_ascii: np.ndarray) -> np.ndarray:
        nw = self._l.orc_words_for(self.L)
        queries_ascii = np.ascontiguousarray(queries_ascii, dtype=np.uint8)
        out = np.zeros((queries_ascii.shape[0], nw), dtype=np.uint64)
        self._l.orc_encode_rows.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
        if self._l.orc_encode_rows(queries_ascii.ctypes.data, queries_ascii.shape[0], self.L, out.ctypes.data):
            raise ValueError(self._l.orc_last_error().decode())
        return out

    def bench_besthit(self, query_enc: np.ndarray, max_div: int) -> int:
        return self._l.orc_bench_besthit_onehot(C.byref(self._ws), query_enc.ctypes.data, query_enc.shape[0], self.L, max_div)

    def bench_kmode(self, query_enc: np.ndarray, max_div: int, max_num_hits: int) -> int:
        """rows the K branch (src/lib.rs:242-295) would print; max_div = NO_LIMIT for none"""
        return self._l.orc_bench_kmode_onehot(C.byref(self._ws), query_enc.ctypes.data, query_enc.shape[0], self.L,
                                              max_div, max_num_hits)

    def close(self):
        self._l.orc_ws_free(C.byref(self._ws))
