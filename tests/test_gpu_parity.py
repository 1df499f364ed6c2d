"""GPU parity: the HIP path, called through the C ABI (ctypes) and the `smafa` CLI, against the CPU
oracle on the same seeded inputs — bit-exact (integer work), same rows, same order.

Sizes here are ones the oracle finishes in seconds; the full BASELINE sizes are covered by
size-independent properties in test_gpu_fullsize.py.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle
import smafa_amd
from smafa_amd import _lib, synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NT_ASCII = np.frombuffer(b"ACGTN", dtype=np.uint8)


@pytest.fixture(scope="module", autouse=True)
def _built():
    smafa_amd.build()
    assert smafa_amd.device_count() >= 1, "GPU tests need a HIP device (no CPU fallback exists)"


def planted(rng, n, L, n_letters, q, max_subs, dup=True):
    s = rng.integers(0, n_letters, size=(n, L), dtype=np.uint8)
    if dup and n > 4:
        s[n // 2] = s[1]
        s[n - 1] = s[1]
    qs = s[rng.integers(0, n, size=q)].copy()
    for r in qs:
        for _ in range(rng.integers(0, max_subs + 1)):
            r[rng.integers(0, L)] = rng.integers(0, n_letters)
    return s, qs


def expected_with_k(all_hits, k):
    """apply "dist <= k-th smallest distance of the query" to an ordered complete hit list"""
    out = []
    i = 0
    while i < len(all_hits):
        j = i
        while j < len(all_hits) and all_hits[j]["query"] == all_hits[i]["query"]:
            j += 1
        grp = all_hits[i:j]
        kth = grp[k - 1]["dist"] if len(grp) >= k else 0xFFFFFFFF
        out.append(grp[grp["dist"] <= kth])
        i = j
    return np.concatenate(out) if out else all_hits[:0]


# ------------------------------------------------------------------------------ scan vs oracle
@pytest.mark.parametrize("alphabet,n_letters", [(0, 5), (1, 28)])
@pytest.mark.parametrize("L", [1, 11, 31, 32, 33, 60, 64, 65, 96, 128, 129, 200])
def test_scan_hits_equals_oracle(alphabet, n_letters, L):
    rng = np.random.default_rng(1000 * alphabet + L)
    for n in (1, 255, 256, 257, 1025, 3000):
        s, q = planted(rng, n, L, n_letters, 70, min(L, 8))
        store = smafa_amd.SubjectStore(L, alphabet)
        store.push(s)
        assert len(store) == n
        for D in sorted({0, min(3, L), min(6, L), L}):
            got = store.scan(q, max_divergence=D)
            want = oracle.scan_codes(s, q, D)
            assert got.tobytes() == want.tobytes(), (alphabet, L, n, D)
        store.close()


@pytest.mark.parametrize("alphabet,n_letters", [(0, 4), (0, 5), (1, 28)])
@pytest.mark.parametrize("L", [65, 96, 128, 129, 160, 161, 257, 700])
def test_wide_lengths_all_modes_equal_oracle(alphabet, n_letters, L):
    """more than 64 columns: scan_wide_kernel while the bound prunes (small D), scan_kernel (up to 128 columns) or
    scan_generic_kernel beyond;
    fixed bounds, best hit and k-th bounds, clustered neighbourhoods (levels 2 and 3 are exercised)"""
    rng = np.random.default_rng(77 * L + n_letters)
    n = 5000
    s, q = planted(rng, n, L, n_letters, 90, 9)
    s[1000:1400] = s[3]                       # a dense neighbourhood: 400 copies ...
    for r in s[1000:1400]:
        for _ in range(rng.integers(0, 5)):   # ... each 0-4 substitutions away
            r[rng.integers(0, L)] = rng.integers(0, min(n_letters, 4))
    q[:10] = s[3]
    store = smafa_amd.SubjectStore(L, alphabet)
    store.push(s)
    for D in (0, 2, 7, 8, 40, L):
        got = store.scan(q, max_divergence=D)
        assert got.tobytes() == oracle.scan_codes(s, q, D).tobytes(), (L, D)
        plan = store.last_scan_plan()
        # W = 3, 4 keep the per-length kernels (8 subjects per lane); SMAFA_WIDE_FROM=3 moves them
        if (os.environ.get("SMAFA_FILTER", "1") == "0" or os.environ.get("SMAFA_LAZY", "1") == "0"
                or os.environ.get("SMAFA_ZONE", "1") == "2"):
            continue  # forced kernel forms: rows only
        moved = int(os.environ.get("SMAFA_WIDE_FROM", "5")) <= 3
        assert plan["filter_plane_resident"] == (D < 8)
        zone = store.last_scan_kernel().startswith("smafa::scan_zone")  # sorted store, bound its zone level prunes at
        assert plan["tiles_per_wave"] == (4 if L > 128 or zone else (4 if moved else 2) if D < 8 else 1)
    for D, k in ((6, 1), (None, 1), (5, 3), (None, 4), (7, 450)):
        got = store.scan(q, max_divergence=D, max_num_hits=k)
        want = expected_with_k(oracle.scan_codes(s, q, L if D is None else D), k)
        assert got.tobytes() == want.tobytes(), (L, D, k)
    store.set_prefilter(False)
    assert store.scan(q, max_divergence=3).tobytes() == oracle.scan_codes(s, q, 3).tobytes()
    store.close()


def test_nt_scan_equals_reference_arithmetic():
    """NT path vs the reference's own arithmetic (5-bit one-hot, xor + popcount / 2) on ASCII input"""
    rng = np.random.default_rng(2)
    letters = np.frombuffer(b"ACGTUNRYKMacgtn-", dtype=np.uint8)
    for L in (5, 12, 13, 60, 61):
        subj = letters[rng.integers(0, len(letters), size=(2000, L))]
        qry = subj[rng.integers(0, 2000, size=50)].copy()
        for r in qry:
            for _ in range(rng.integers(0, 5)):
                r[rng.integers(0, L)] = letters[rng.integers(0, len(letters))]
        store = smafa_amd.SubjectStore(L, smafa_amd.ALPHABET_NT)
        store.push(smafa_amd.encode_rows(subj))
        for D in (0, 4, L):
            got = store.scan(smafa_amd.encode_rows(qry), max_divergence=D)
            want = oracle.scan_onehot(subj, qry, D)
            assert got.tobytes() == want.tobytes(), (L, D)
        store.close()


def test_aa_kernel_on_nucleotide_data_equals_reference_arithmetic():
    """the 5-plane (amino-acid) kernel is the same code path; on nucleotide-alphabet data it must
    reproduce the reference-pinned one-hot result (cross-pin of the unpinned extension)"""
    rng = np.random.default_rng(3)
    codes = rng.integers(0, 5, size=(3000, 60), dtype=np.uint8)
    q = codes[rng.integers(0, 3000, size=64)].copy()
    q[:, 7] = (q[:, 7] + 1) % 5
    store = smafa_amd.SubjectStore(60, smafa_amd.ALPHABET_AA)  # AA store, codes 0..4 only
    store.push(codes)
    got = store.scan(q, max_divergence=5)
    want = oracle.scan_onehot(NT_ASCII[codes], NT_ASCII[q], 5)
    assert got.tobytes() == want.tobytes()
    store.close()


@pytest.mark.parametrize("alphabet,n_letters,L", [(0, 5, 60), (1, 28, 60), (0, 5, 7), (1, 28, 150)])
def test_get_distances_equals_oracle(alphabet, n_letters, L):
    rng = np.random.default_rng(4)
    s, q = planted(rng, 2500, L, n_letters, 5, 6)
    store = smafa_amd.SubjectStore(L, alphabet)
    store.push(s)
    for i in range(5):
        assert (store.get_distances(q[i]) == oracle.distances_codes(s, q[i])).all()
    store.close()


def test_append_in_pieces_equals_single_push():
    rng = np.random.default_rng(5)
    s, q = planted(rng, 2100, 60, 28, 40, 6)
    one = smafa_amd.SubjectStore(60, 1)
    one.push(s)
    pieces = smafa_amd.SubjectStore(60, 1)
    at = 0
    for step in (1, 62, 1, 64, 128, 300, 700, 3, 841):
        pieces.push(s[at:at + step])
        at += step
    assert at == 2100 and len(pieces) == 2100
    a = one.scan(q, max_divergence=8)
    b = pieces.scan(q, max_divergence=8)
    assert a.tobytes() == b.tobytes() == oracle.scan_codes(s, q, 8).tobytes()
    one.close()
    pieces.close()


@pytest.mark.parametrize("k", [1, 2, 5, 40])
@pytest.mark.parametrize("max_div", [None, 4, 30])
def test_kth_bound_modes(k, max_div):
    """max_num_hits = k: rows within the k-th smallest distance (ties included), src/lib.rs:250-256;
    exercises the seed + growing-segment launches that tighten the per-query bounds"""
    rng = np.random.default_rng(6)
    L = 60
    s, q = planted(rng, 9000, L, 4, 80, 7)  # 4 letters: distances spread out, plenty of ties
    s[100:140] = s[7]
    store = smafa_amd.SubjectStore(L, 0)
    store.push(s)
    got = store.scan(q, max_divergence=max_div, max_num_hits=k)
    full = oracle.scan_codes(s, q, L if max_div is None else max_div)
    want = expected_with_k(full, k)
    assert got.tobytes() == want.tobytes()
    store.close()


@pytest.mark.parametrize("alphabet,n_letters", [(0, 4), (1, 20)])
@pytest.mark.parametrize("div,hist_seed", [(0, 1), (2, 1), (16, 1), (16, 0)])
def test_kth_modes_counting_a_sample_first(alphabet, n_letters, div, hist_seed, monkeypatch):
    """k >= 3 without a usable bound: the counting pass covers the first 1/div of the tiles only (an upper bound of every
    query's k-th distance), the rest of the store is counted, tightened and appended in one pass, the exact bounds come from
    the complete counts and the sample's tiles are scanned again with them (engine.hip scan_range).  Forced here on a small
    store (SMAFA_KTH_SAMPLE_MIN_TILES; div 0 = everything counted first): rows == the oracle's, ties, dense spots, bounds."""
    monkeypatch.setenv("SMAFA_KTH_SAMPLE", str(div))
    monkeypatch.setenv("SMAFA_KTH_HIST_SEED", str(hist_seed))  # the seed bound: LDS histogram over the first tiles / a counting launch
    monkeypatch.setenv("SMAFA_KTH_SAMPLE_MIN_TILES", "8")
    monkeypatch.setenv("SMAFA_TWO_PHASE", "0")  # no near-hit ladder in front: every query takes the path under test
    rng = np.random.default_rng(77 + alphabet)
    L, n = 60, 30000
    s = rng.integers(0, n_letters, size=(n, L), dtype=np.uint8)
    s[200:260] = s[11]          # 61 copies: more ties than any k below
    s[29000:29030] = s[12]      # a dense spot in the last tiles (not in any sample)
    qs = [s[11], s[12], s[29999]]
    for subs in (1, 3, 6, 10, 20, 35):
        for base in (11, 12, 5000, 29500):
            r = s[base].copy()
            cols = rng.choice(L, size=subs, replace=False)
            r[cols] = (r[cols] + 1 + rng.integers(0, n_letters - 1, size=subs)) % n_letters
            qs.append(r)
    qs += [rng.integers(0, n_letters, size=L, dtype=np.uint8) for _ in range(37)]
    q = np.array(qs, dtype=np.uint8)
    store = smafa_amd.SubjectStore(L, alphabet)
    store.push(s)
    for D in (None, 40, 25):
        full = oracle.scan_codes(s, q, L if D is None else D)
        for k in (3, 5, 40, 100):
            got = store.scan(q, max_divergence=D, max_num_hits=k)
            assert got.tobytes() == expected_with_k(full, k).tobytes(), (D, k)
    store.close()


@pytest.mark.parametrize("alphabet,n_letters", [(0, 4), (1, 20)])
def test_near_hit_probe_boundaries(alphabet, n_letters):
    """k-th-distance modes with a loose or absent bound: a ladder of bounded scans (bounds 5, 12, then 30 — 16 for a 2-plane store — at L = 60) finishes
    the queries with at least k rows within a step's bound, the rest take the tightening path as a compacted batch;
    queries sit exactly on both sides of every boundary, with ties"""
    rng = np.random.default_rng(31 + alphabet)
    L, n = 60, 20000
    s = rng.integers(0, n_letters, size=(n, L), dtype=np.uint8)
    s[5000:5003] = s[17]              # 4 copies of row 17 in all: ties at distance 0 ...
    qs = []
    for subs in (0, 1, 4, 5, 5, 6, 6, 7, 9, 11, 12, 12, 13, 13, 15, 16, 16, 17, 17, 18, 24, 29, 30, 30, 31, 32, 33, 36):  # 0..36 substitutions
        for base in (17, 400, 4242, 19999):
            r = s[base].copy()
            cols = rng.choice(L, size=subs, replace=False)
            r[cols] = (r[cols] + 1 + rng.integers(0, n_letters - 1, size=subs)) % n_letters
            qs.append(r)
    qs += [rng.integers(0, n_letters, size=L, dtype=np.uint8) for _ in range(8)]  # no near subject at all
    q = np.array(qs, dtype=np.uint8)
    store = smafa_amd.SubjectStore(L, alphabet)
    store.push(s)
    for D in (None, 6, 12, 13, 16, 17, 18, 20, 24, 30, 32, 33, 35):  # 13..17 / 18..32: the all-planes kernel's FOLD 1 / FOLD 2 forms
        full = oracle.scan_codes(s, q, L if D is None else D)
        if D is not None:
            assert store.scan(q, max_divergence=D).tobytes() == full.tobytes(), D
        for k in (1, 2, 4, 5, 60):
            got = store.scan(q, max_divergence=D, max_num_hits=k)
            assert got.tobytes() == expected_with_k(full, k).tobytes(), (D, k)
    store.close()


def test_device_resident_launch_all_modes():
    """smafa_scan_launch (rows and count stay in HBM) in every mode — tests/device_launch_worker.py, a process of
    its own because the device buffers come from torch, which has to initialise HIP before the library does"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "device_launch_worker.py")], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "device launch modes ok" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


def test_capacity_error_reports_rows_needed():
    rng = np.random.default_rng(7)
    s, q = planted(rng, 500, 20, 4, 10, 2)
    store = smafa_amd.SubjectStore(20, 0)
    store.push(s)
    want = oracle.scan_codes(s, q, 20)
    assert len(want) == 5000
    out = np.zeros(10, dtype=smafa_amd.HIT_DTYPE)
    n_out = C.c_uint64(0)
    rc = _lib.lib().smafa_scan_hits(store._h, q.ctypes.data, 10, 20, _lib.NONE, out.ctypes.data, 10, C.byref(n_out))
    assert rc == _lib.ERR_CAPACITY and n_out.value == 5000
    store.close()


def test_grow_and_retry_is_never_stale():
    """after SMAFA_ERR_CAPACITY the handle keeps the rows for the identical retry only: other queries, other
    bounds or a changed store are scanned afresh"""
    rng = np.random.default_rng(17)
    s, q = planted(rng, 3000, 40, 4, 30, 5)
    store = smafa_amd.SubjectStore(40, 0)
    store.push(s)
    l = _lib.lib()

    def call(queries, div, cap):
        out = np.zeros(max(cap, 1), dtype=smafa_amd.HIT_DTYPE)
        n_out = C.c_uint64(0)
        rc = l.smafa_scan_hits(store._h, queries.ctypes.data, len(queries), div, _lib.NONE, out.ctypes.data, cap, C.byref(n_out))
        return rc, n_out.value, out[: min(n_out.value, cap)]

    want = oracle.scan_codes(s, q, 12)
    assert len(want) >= 30
    rc, need, _ = call(q, 12, 3)
    assert rc == _lib.ERR_CAPACITY and need == len(want)
    rc, n, rows = call(q, 12, need)                      # the retry: same request, served from the kept rows
    assert rc == 0 and rows.tobytes() == want.tobytes()
    rc, n, rows = call(q, 12, need)                      # and once more: kept rows are gone, scanned afresh
    assert rc == 0 and rows.tobytes() == want.tobytes()
    # a different batch of the same shape after an overflow must not see the kept rows
    q2 = q[::-1].copy()
    assert call(q, 12, 3)[0] == _lib.ERR_CAPACITY
    rc, n, rows = call(q2, 12, 1 << 16)
    assert rc == 0 and rows.tobytes() == oracle.scan_codes(s, q2, 12).tobytes()
    # other bound
    assert call(q, 12, 3)[0] == _lib.ERR_CAPACITY
    rc, n, rows = call(q, 11, 1 << 16)
    assert rc == 0 and rows.tobytes() == oracle.scan_codes(s, q, 11).tobytes()
    # store changed between the overflow and the retry
    assert call(q, 12, 3)[0] == _lib.ERR_CAPACITY
    store.push(q[:4])
    s2 = np.concatenate([s, q[:4]])
    rc, n, rows = call(q, 12, 1 << 16)
    assert rc == 0 and rows.tobytes() == oracle.scan_codes(s2, q, 12).tobytes()
    store.close()


def test_dense_hits_overflow_path():
    """every pair qualifies: more rows than the device scratch holds -> query range is split"""
    rng = np.random.default_rng(8)
    s = rng.integers(0, 4, size=(70000, 16), dtype=np.uint8)
    q = rng.integers(0, 4, size=(100, 16), dtype=np.uint8)
    store = smafa_amd.SubjectStore(16, 0)
    store.push(s)
    got = store.scan(q, max_divergence=16)  # 7M rows > 4M scratch rows
    assert len(got) == 70000 * 100
    d = (s[got["subject"][::997]] != q[got["query"][::997]]).sum(axis=1)
    assert (d == got["dist"][::997]).all()
    key = got["query"].astype(np.int64) << 40 | got["dist"].astype(np.int64) << 32 | got["subject"]
    assert (np.diff(key) > 0).all()
    store.close()


# ------------------------------------------------------------------------- CLI: golden vectors
def cli(*args):
    return subprocess.run([_lib.CLI_PATH, *args], capture_output=True, text=True)


def rows(*r):
    return "".join("\t".join(map(str, x)) + "\n" for x in r)


FOUR = rows((0, 0, 0, "CTT"), (0, 1, 3, "AGG"), (1, 1, 0, "AGG"), (1, 0, 3, "CTT"))
TWO = rows((0, 0, 0, "CTT"), (1, 1, 0, "AGG"))


def test_golden_dna_makedb_and_query(golden, tmp_path):  # tests/test_cmdline.rs:10-25
    db = str(tmp_path / "t.db")
    f = os.path.join(golden, "random_3_2.fna")
    assert cli("makedb", "-i", f, "-d", db).returncode == 0
    r = cli("query", "-d", db, "-q", f)
    assert r.returncode == 0 and r.stdout == TWO


def test_golden_degenerate(golden, tmp_path):  # tests/test_cmdline.rs:44-74
    db = str(tmp_path / "t.db")
    f = os.path.join(golden, "degenerate.fna")
    assert cli("makedb", "-i", f, "-d", db).returncode == 0
    r = cli("query", "-d", db, "-q", f, "--max-num-hits", "99")
    assert r.returncode == 0, r.stderr
    assert r.stdout == rows(
        (0, 0, 0, "CTTNGG"), (0, 1, 5, "AGGTGA"), (0, 2, 6, "NACTTT"),
        (1, 1, 0, "AGGTGA"), (1, 0, 5, "CTTNGG"), (1, 2, 5, "NACTTT"),
        (2, 2, 0, "NACTTT"), (2, 1, 5, "AGGTGA"), (2, 0, 6, "CTTNGG"))


@pytest.mark.parametrize("flags,expected", [
    (["--max-divergence", "99", "--max-num-hits", "99"], FOUR),  # test_cmdline.rs:77-97
    (["--max-divergence", "2", "--max-num-hits", "99"], TWO),    # :100-118
    (["--max-divergence", "3", "--max-num-hits", "99"], FOUR),   # :121-141
    (["--max-num-hits", "1"], TWO),                              # :144-160
    (["--max-num-hits", "99"], FOUR),                            # :163-181
])
def test_golden_prebuilt_db(golden, flags, expected):
    r = cli("query", "-d", os.path.join(golden, "random_3_2.fna.smafadb"), "-q", os.path.join(golden, "random_3_2.fna"), *flags)
    assert r.returncode == 0, r.stderr
    assert r.stdout == expected


def test_golden_limit_per_sequence(golden):  # tests/test_cmdline.rs:204-247
    db = os.path.join(golden, "random_3_2_one_repeated.fna.smafadb")
    q = os.path.join(golden, "random_3_2.fna")
    r = cli("query", "-d", db, "-q", q, "--max-num-hits", "99")
    assert r.stdout == rows((0, 0, 0, "CTT"), (0, 1, 3, "AGG"), (0, 2, 3, "AGG"), (1, 1, 0, "AGG"), (1, 2, 0, "AGG"), (1, 0, 3, "CTT"))
    r = cli("query", "-d", db, "-q", q, "--max-num-hits", "99", "--limit-per-sequence", "1")
    assert r.stdout == FOUR
    r = cli("query", "-d", db, "-q", q, "--limit-per-sequence", "1")  # src/lib.rs:301-303
    assert r.returncode == 101 and "limit_per_sequence is implemented unless" in r.stderr


def test_golden_cluster(golden):  # src/cluster.rs:102-143
    r = cli("cluster", "-i", os.path.join(golden, "cluster_dummy1.fna"), "-d", "1")
    assert r.returncode == 0, r.stderr
    assert r.stdout == "ATGC\tATGC\nATGG\tATGC\nAAAA\tAAAA\n"
    for name in ("cluster_bug1.fna", "cluster_best_hit_changes.fna"):
        r = cli("cluster", "-i", os.path.join(golden, name), "-d", "2")
        assert r.stdout == "ATGCAAAAA\tATGCAAAAA\nATAAAAAAA\tATGCAAAAA\nTTAAAAAAA\tTTAAAAAAA\n"


# ------------------------------------------------------------- CLI: differential vs oracle CLI
def write_fasta(path, ascii_rows, fastq=False):
    with open(path, "wb") as f:
        for i, r in enumerate(ascii_rows):
            b = bytes(r)
            if fastq:
                f.write(b"@r%d\n" % i + b + b"\n+\n" + b"I" * len(b) + b"\n")
            else:
                f.write(b">r%d desc\n" % i + b[:25] + b"\n" + b[25:] + b"\n")


@pytest.mark.parametrize("flags", [
    [], ["--max-divergence", "5"], ["--max-divergence", "0"], ["--max-num-hits", "4"],
    ["--max-num-hits", "4", "--max-divergence", "6"], ["--max-num-hits", "100000"],
    ["--max-num-hits", "6", "--limit-per-sequence", "2"], ["--max-num-hits", "3", "--max-divergence", "60"],
])
def test_query_cli_equals_oracle_cli(tmp_path, flags):
    rng = np.random.default_rng(9)
    letters = np.frombuffer(b"ACGTN", dtype=np.uint8)
    s = letters[rng.integers(0, 4, size=(4000, 60))]
    s[rng.random(size=s.shape) < 0.01] = ord("N")
    s[300:330] = s[5]
    s[400:420, :55] = s[5, :55]
    q = s[rng.integers(0, 4000, size=150)].copy()
    for r in q:
        for _ in range(rng.integers(0, 8)):
            r[rng.integers(0, 60)] = letters[rng.integers(0, 5)]
    sf, qf, db = str(tmp_path / "s.fna"), str(tmp_path / "q.fq"), str(tmp_path / "db")
    write_fasta(sf, s)
    write_fasta(qf, q, fastq=True)
    assert cli("makedb", "-i", sf, "-d", db).returncode == 0
    got = cli("query", "-d", db, "-q", qf, *flags)
    want = oracle.run_cli("query", "-d", db, "-q", qf, *flags)
    assert got.returncode == 0 and want.returncode == 0, (got.stderr, want.stderr)
    assert got.stdout == want.stdout
    assert len(got.stdout) > 0


def test_query_cli_error_after_partial_output(tmp_path):
    """a bad query record: rows of the earlier queries are printed, then the reference's panic text"""
    sf, qf, db = str(tmp_path / "s.fna"), str(tmp_path / "q.fna"), str(tmp_path / "db")
    open(sf, "wb").write(b">a\nACGTACGT\n>b\nTTTTACGT\n")
    open(qf, "wb").write(b">q0\nACGTACGT\n>q1\nACGTAEGT\n>q2\nTTTTACGT\n")
    assert cli("makedb", "-i", sf, "-d", db).returncode == 0
    got = cli("query", "-d", db, "-q", qf)
    want = oracle.run_cli("query", "-d", db, "-q", qf)
    assert got.returncode == 101 and want.returncode == 101
    assert got.stdout == want.stdout == "0\t0\t0\tACGTACGT\n"
    assert 'Byte 69 cannot be interpreted as nucleotide, in sequence "q1" at position 5' in got.stderr
    open(qf, "wb").write(b">q0\nACGTACGT\n>q1\nACGTACG\n")
    got = cli("query", "-d", db, "-q", qf)
    assert got.returncode == 101 and got.stdout == "0\t0\t0\tACGTACGT\n"
    assert "Cannot compute distances between seq of length 7 and windows of lengths 8" in got.stderr


def clustered(rng, n_roots, members, L, n_letters, max_subs):
    roots = rng.integers(0, n_letters, size=(n_roots, L), dtype=np.uint8)
    recs = np.repeat(roots, members, axis=0)
    for r in recs:
        for _ in range(rng.integers(0, max_subs + 1)):
            r[rng.integers(0, L)] = rng.integers(0, n_letters)
    rng.shuffle(recs, axis=0)
    return recs


@pytest.mark.parametrize("D", [0, 2, 5, 9])
def test_cluster_cli_equals_oracle_cli(tmp_path, D):
    rng = np.random.default_rng(10 + D)
    recs = clustered(rng, 300, 20, 60, 4, 4)  # 6000 records, several batches, duplicates included
    ascii_rows = NT_ASCII[recs]
    lower = ascii_rows.copy()
    lower[::3] = np.char.lower(lower[::3].view("S1")).view(np.uint8)  # column 1 must echo the raw bytes
    f = str(tmp_path / "c.fna")
    with open(f, "wb") as fh:
        for i, r in enumerate(lower):
            fh.write(b">c%d\n" % i + bytes(r) + b"\n")
    got = cli("cluster", "-i", f, "-d", str(D))
    want = oracle.run_cli("cluster", "-i", f, "-d", str(D))
    assert got.returncode == 0 and want.returncode == 0, got.stderr
    assert got.stdout == want.stdout
    assert got.stdout.count("\n") > 1000


def test_cluster_aa_equals_oracle_codes(tmp_path):
    rng = np.random.default_rng(20)
    lc = synth.letter_codes(1)
    recs = lc[clustered(rng, 200, 25, 60, 20, 5)]
    assigned = oracle.cluster_codes(recs, 5, oracle.ALPHABET_AA)
    letters = np.array([ord(smafa_amd.decode(np.array([c], dtype=np.uint8), 1)) for c in range(28)], dtype=np.uint8)
    ascii_rows = letters[recs]
    f = str(tmp_path / "aa.faa")
    with open(f, "wb") as fh:
        for i, r in enumerate(ascii_rows):
            fh.write(b">p%d\n" % i + bytes(r) + b"\n")
    got = cli("cluster", "-i", f, "-d", "5", "--alphabet", "aa")
    assert got.returncode == 0, got.stderr
    cents, lines = [], []
    for i, a in enumerate(assigned):
        if a == 0xFFFFFFFF:
            continue
        if a == len(cents):
            cents.append(bytes(ascii_rows[i]))
        lines.append(bytes(ascii_rows[i]) + b"\t" + cents[a] + b"\n")
    assert got.stdout.encode() == b"".join(lines)


def test_python_api_query_and_cluster(golden, tmp_path):
    """the ctypes mirror of the crate's pub fns writes the same bytes as the CLI"""
    out = tmp_path / "out.tsv"
    fd = os.open(str(out), os.O_WRONLY | os.O_CREAT | os.O_TRUNC)
    try:
        smafa_amd.query(os.path.join(golden, "random_3_2.fna.smafadb"), os.path.join(golden, "random_3_2.fna"),
                        max_divergence=99, max_num_hits=99, out_fd=fd)
        smafa_amd.cluster(os.path.join(golden, "cluster_dummy1.fna"), 1, out_fd=fd)
    finally:
        os.close(fd)
    assert out.read_text() == FOUR + "ATGC\tATGC\nATGG\tATGC\nAAAA\tAAAA\n"


# ------------------------------------------------- 2-bit (N-free) nucleotide store and its upgrade
def test_nt_store_uses_two_planes_until_an_N_arrives():
    """BASELINE configs[2] "2-bit pack path": a nucleotide store with no N keeps 2 planes per subject (16 B at
    L=60); queries may still contain N (an N mismatches every subject).  The first appended N re-lays the store
    out with 3 planes; results equal the 5-symbol oracle at every stage."""
    rng = np.random.default_rng(31)
    L = 60
    s1 = rng.integers(0, 4, size=(3000, L), dtype=np.uint8)            # A C G T only
    q = s1[rng.integers(0, 3000, size=80)].copy()
    q[rng.random(size=q.shape) < 0.02] = 4                              # queries with N
    for r in q[:40]:
        for _ in range(rng.integers(0, 5)):
            r[rng.integers(0, L)] = rng.integers(0, 4)
    store = smafa_amd.SubjectStore(L, smafa_amd.ALPHABET_NT)
    store.push(s1)
    info = store.info()
    assert info.planes == 2 and info.bytes_per_subject == 16
    for D in (0, 3, 8, L):
        assert store.scan(q, max_divergence=D).tobytes() == oracle.scan_codes(s1, q, D).tobytes()
    assert (store.get_distances(q[5]) == oracle.distances_codes(s1, q[5])).all()
    assert store.scan(q, None, 1).tobytes() == expected_with_k(oracle.scan_codes(s1, q, L), 1).tobytes()
    # the reference's own arithmetic on the ASCII form
    assert store.scan(q, max_divergence=4).tobytes() == oracle.scan_onehot(NT_ASCII[s1], NT_ASCII[q], 4).tobytes()
    # now subjects with N arrive: 3 planes, old rows intact
    s2 = rng.integers(0, 5, size=(700, L), dtype=np.uint8)
    s2[:10] = q[:10]                                                    # exact copies, N included
    store.push(s2)
    info = store.info()
    assert info.planes == 3 and info.bytes_per_subject == 24 and info.n_subjects == 3700
    both = np.concatenate([s1, s2])
    for D in (0, 3, L):
        assert store.scan(q, max_divergence=D).tobytes() == oracle.scan_codes(both, q, D).tobytes()
    assert (store.get_distances(q[0]) == oracle.distances_codes(both, q[0])).all()
    store.close()


@pytest.mark.parametrize("L", [7, 33, 100, 140])
def test_nt_two_plane_store_other_lengths(L):
    rng = np.random.default_rng(32 + L)
    s = rng.integers(0, 4, size=(1500, L), dtype=np.uint8)
    q = s[rng.integers(0, 1500, size=40)].copy()
    q[:, L // 2] = 4
    store = smafa_amd.SubjectStore(L, 0)
    store.push(s)
    assert store.info().planes == 2
    for D in (1, 4):
        assert store.scan(q, max_divergence=D).tobytes() == oracle.scan_codes(s, q, D).tobytes()
    store.close()


# ------------------------------------------------------------------- CLI edge cases vs the oracle CLI
def both_cli(*args):
    got = cli(*args)
    want = oracle.run_cli(*args)
    return got, want


def test_cli_edge_cases_match_oracle(tmp_path):
    import gzip

    d = tmp_path
    # multi-line FASTA with CRLF, lowercase and IUPAC letters, '-' gaps; last record without trailing newline
    (d / "s.fna").write_bytes(b">s0 first\r\nACGTAC\r\nGTAC\r\n>s1\nacgtrygtac\n>s2\nAC-TACGTAN\n>s3\nTTTTTTTTTT\n>s4 dup of s0\nACGTACGTAC")
    # gzip FASTQ queries
    with gzip.open(d / "q.fq.gz", "wb") as f:
        f.write(b"@q0\nACGTACGTAC\n+\nIIIIIIIIII\n@q1\nACGTNNGTAC\n+\nIIIIIIIIII\n@q2\nTTTTTTTTTA\n+\nIIIIIIIIII\n")
    db = str(d / "db")
    assert cli("makedb", "-i", str(d / "s.fna"), "-d", db).returncode == 0
    db2 = str(d / "db2")
    assert oracle.run_cli("makedb", "-i", str(d / "s.fna"), "-d", db2).returncode == 0
    assert open(db, "rb").read() == open(db2, "rb").read()
    for flags in ([], ["--max-divergence", "1"], ["--max-num-hits", "3"], ["--max-num-hits", "2", "--limit-per-sequence", "1"],
                  ["--max-divergence", "0"], ["--max-num-hits", "100"]):
        got, want = both_cli("query", "-d", db, "-q", str(d / "q.fq.gz"), *flags)
        assert got.returncode == want.returncode == 0, (flags, got.stderr, want.stderr)
        assert got.stdout == want.stdout, flags
    # a store with zero windows serialises to 3 bytes (02 00 00), and the reference slices &buffer[0..4]
    # before anything else (src/lib.rs:214): that panic, not the empty-vector unwrap, is what a user sees
    smafa_amd.write_db(str(d / "empty.db"), np.zeros((0, 10), dtype=np.uint8))
    assert open(d / "empty.db", "rb").read() == bytes([2, 0, 0])
    for flags in ([], ["--max-num-hits", "5"]):
        got, want = both_cli("query", "-d", str(d / "empty.db"), "-q", str(d / "q.fq.gz"), *flags)
        assert got.returncode == want.returncode == 101 and got.stdout == want.stdout == ""
        assert "range end index 4 out of range for slice of length 3" in got.stderr
    # an empty record among the queries: length mismatch after the rows of the earlier queries
    (d / "q2.fna").write_bytes(b">a\nACGTACGTAC\n>empty\n\n>c\nACGTACGTAC\n")
    got, want = both_cli("query", "-d", db, "-q", str(d / "q2.fna"))
    assert got.returncode == want.returncode == 101 and got.stdout == want.stdout and got.stdout.startswith("0\t0\t0\t")
    assert "Cannot compute distances between seq of length 0 and windows of lengths 10" in got.stderr
    # single-subject store, k larger than the store, max-num-hits 0 (index underflow in the reference)
    (d / "one.fna").write_bytes(b">only\nACGTACGTAC\n")
    assert cli("makedb", "-i", str(d / "one.fna"), "-d", str(d / "one.db")).returncode == 0
    got, want = both_cli("query", "-d", str(d / "one.db"), "-q", str(d / "q.fq.gz"), "--max-num-hits", "7")
    assert got.returncode == 0 and got.stdout == want.stdout
    got, want = both_cli("query", "-d", str(d / "one.db"), "-q", str(d / "q.fq.gz"), "--max-num-hits", "0")
    assert got.returncode == want.returncode == 101 and "index out of bounds" in got.stderr
    # cluster: duplicates differing only by case / IUPAC class collapse to one record; raw bytes echoed
    (d / "c.fna").write_bytes(b">1\nACGTACGTAC\n>2\nacgtacgtac\n>3\nACGTACGTAR\n>4\nACGTACGTAY\n>5\nTTTTTTTTTT\n")
    for D in ("0", "1"):
        got, want = both_cli("cluster", "-i", str(d / "c.fna"), "-d", D)
        assert got.returncode == want.returncode == 0 and got.stdout == want.stdout, D
    # cluster: ragged input fails after the lines already due
    (d / "c2.fna").write_bytes(b">1\nACGTACGTAC\n>2\nACGTACGTAA\n>3\nACGT\n>4\nACGTACGTAC\n")
    got, want = both_cli("cluster", "-i", str(d / "c2.fna"), "-d", "1")
    assert got.returncode == want.returncode == 101 and got.stdout == want.stdout and got.stdout.count("\n") == 2


# ------------------------------------------------------------------ randomized differential sweep
def test_randomized_cli_sweep_vs_oracle(tmp_path):
    """40 random (length, store size, mutation load, flag set) draws through `smafa makedb|query|cluster`
    against the oracle CLI: same stdout, same exit status"""
    rng = np.random.default_rng(2024)
    letters = np.frombuffer(b"ACGTN", dtype=np.uint8)
    for trial in range(40):
        L = int(rng.choice([3, 9, 12, 13, 31, 32, 33, 47, 60, 64, 65, 100, 128, 131]))
        n = int(rng.choice([1, 2, 5, 63, 64, 65, 300, 1500]))
        nq = int(rng.integers(1, 40))
        s = letters[rng.integers(0, 4, size=(n, L))]
        if rng.random() < 0.5:
            s[rng.random(size=s.shape) < 0.03] = ord("N")
        if n > 4 and rng.random() < 0.7:  # duplicates and near-duplicates
            s[n // 2] = s[0]
            s[n - 1] = s[0]
            s[n - 2, : L // 2] = s[0, : L // 2]
        q = s[rng.integers(0, n, size=nq)].copy()
        for r in q:
            for _ in range(int(rng.integers(0, max(2, L // 4)))):
                r[rng.integers(0, L)] = letters[rng.integers(0, 5)]
        flags = []
        if rng.random() < 0.6:
            flags += ["--max-divergence", str(int(rng.integers(0, L + 2)))]
        k = None
        if rng.random() < 0.6:
            k = int(rng.choice([1, 2, 3, 7, 50, 5000]))
            flags += ["--max-num-hits", str(k)]
        if k is not None and k > 1 and rng.random() < 0.4:
            flags += ["--limit-per-sequence", str(int(rng.integers(1, 4)))]
        d = tmp_path / f"t{trial}"
        d.mkdir()
        sf, qf, db = str(d / "s.fna"), str(d / "q.fna"), str(d / "db")
        oracle.write_fasta(sf, [bytes(r) for r in s])
        oracle.write_fasta(qf, [bytes(r) for r in q])
        assert cli("makedb", "-i", sf, "-d", db).returncode == 0
        got, want = both_cli("query", "-d", db, "-q", qf, *flags)
        assert got.returncode == want.returncode, (trial, L, n, flags, got.stderr, want.stderr)
        assert got.stdout == want.stdout, (trial, L, n, nq, flags)
        D = str(int(rng.integers(0, max(2, L // 3))))
        got, want = both_cli("cluster", "-i", sf, "-d", D)
        assert got.returncode == want.returncode == 0 and got.stdout == want.stdout, (trial, L, n, D)


def test_randomized_api_sweep_vs_oracle_aa():
    """amino-acid stores through the C ABI: random lengths / sizes / bounds / k against the code-byte oracle"""
    rng = np.random.default_rng(777)
    for trial in range(30):
        L = int(rng.choice([5, 20, 31, 33, 60, 64, 70, 96, 127, 128, 160]))
        n = int(rng.choice([1, 100, 255, 257, 1024, 4097, 9000]))
        nq = int(rng.integers(1, 60))
        n_letters = int(rng.choice([2, 4, 20, 28]))
        s, q = planted(rng, n, L, n_letters, nq, min(L, 9))
        store = smafa_amd.SubjectStore(L, smafa_amd.ALPHABET_AA)
        half = n // 2
        store.push(s[:half])
        store.push(s[half:])
        D = None if rng.random() < 0.3 else int(rng.integers(0, L + 1))
        k = None if rng.random() < 0.5 else int(rng.choice([1, 2, 4, 9, 100]))
        got = store.scan(q, max_divergence=D, max_num_hits=k)
        full = oracle.scan_codes(s, q, L if D is None else D)
        want = full if k is None else expected_with_k(full, k)
        assert got.tobytes() == want.tobytes(), (trial, L, n, nq, n_letters, D, k)
        store.close()


def test_big_query_file_takes_the_threaded_loader_and_matches_oracle(tmp_path):
    """query files of 32 MB and more are parsed and encoded by several threads before the scans: same rows, and the
    same failure at the same record (bad byte, wrong length, wrong first length) after the same rows"""
    rng = np.random.default_rng(99)
    L, n, nq = 60, 400, 560_000  # 560k x (60 + header) bytes = 38 MB of FASTA
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    s = letters[rng.integers(0, 4, size=(n, L))]
    q = s[rng.integers(0, n, size=nq)].copy()
    sub = rng.random(size=q.shape) < 0.03
    q[sub] = letters[rng.integers(0, 4, size=int(sub.sum()))]
    sf, db = str(tmp_path / "s.fna"), str(tmp_path / "db")
    oracle.write_fasta(sf, [bytes(r) for r in s])
    assert cli("makedb", "-i", sf, "-d", db).returncode == 0

    def write_queries(path, rows, patch=None):
        body = bytearray()
        for i, r in enumerate(rows):
            body += b">q%d\n" % i
            body += (patch[i] if patch and i in patch else bytes(r)) + b"\n"
        open(path, "wb").write(body)
        assert len(body) >= 32 << 20

    qf = str(tmp_path / "q.fna")
    write_queries(qf, q)
    for flags in (["--max-divergence", "2"], []):
        got, want = both_cli("query", "-d", db, "-q", qf, *flags)
        assert got.returncode == want.returncode == 0 and got.stdout == want.stdout and len(got.stdout) > 1_000_000, flags
    # a byte outside the alphabet deep in the file; a record of another length; a first record of another length
    for patch in ({401_234: bytes(q[401_234][:30]) + b"E" + bytes(q[401_234][31:])}, {333_333: bytes(q[333_333][:59])},
                  {0: bytes(q[0]) + b"A", 7: b"ACGTE"}):
        write_queries(qf, q, patch)
        got, want = both_cli("query", "-d", db, "-q", qf, "--max-divergence", "1")
        assert got.returncode == want.returncode == 101, patch.keys()
        assert got.stdout == want.stdout
        assert got.stderr.strip().splitlines()[-1] == want.stderr.strip().splitlines()[-1]


def test_big_fastq_and_gzip_query_files_take_the_threaded_loader(tmp_path):
    """>= 32 MB .fq and .fq.gz query files: parsed by several threads (gzip: one inflating thread feeding them), the same
    rows as the oracle CLI, and the same failure at the same record after the same rows"""
    import gzip

    rng = np.random.default_rng(123)
    L, n, nq = 60, 300, 330_000  # 330k x (60 + 60 + header + 3) bytes = 44 MB of FASTQ
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    s = letters[rng.integers(0, 4, size=(n, L))]
    q = s[rng.integers(0, n, size=nq)].copy()
    sub = rng.random(size=q.shape) < 0.03
    q[sub] = letters[rng.integers(0, 4, size=int(sub.sum()))]
    sf, db = str(tmp_path / "s.fna"), str(tmp_path / "db")
    oracle.write_fasta(sf, [bytes(r) for r in s])
    assert cli("makedb", "-i", sf, "-d", db).returncode == 0

    def body(patch=None):
        out = bytearray()
        for i, r in enumerate(q):
            seq = patch[i] if patch and i in patch else bytes(r)
            out += b"@q%d\n" % i + seq + b"\n+\n" + (b"@" if i % 3 == 0 else b"F") * len(seq) + b"\n"
        assert len(out) >= 32 << 20
        return bytes(out)

    fq, gz = str(tmp_path / "q.fq"), str(tmp_path / "q.fq.gz")
    good = body()
    open(fq, "wb").write(good)
    open(gz, "wb").write(gzip.compress(good, 1))
    want = oracle.run_cli("query", "-d", db, "-q", fq, "--max-divergence", "2")
    assert want.returncode == 0 and len(want.stdout) > 1_000_000
    for path, note in ((fq, "parsed by"), (gz, "while one thread inflated")):
        got = cli("query", "-d", db, "-q", path, "--max-divergence", "2", "-v")
        assert got.returncode == 0, got.stderr[-2000:]
        assert got.stdout == want.stdout, path
        assert note in got.stderr, got.stderr[-2000:]
    bad = body({250_001: bytes(q[250_001][:20]) + b"E" + bytes(q[250_001][21:]), 123_456: bytes(q[123_456][:58])})
    open(fq, "wb").write(bad)
    open(gz, "wb").write(gzip.compress(bad, 1))
    want = oracle.run_cli("query", "-d", db, "-q", fq, "--max-divergence", "1")
    for path in (fq, gz):
        got = cli("query", "-d", db, "-q", path, "--max-divergence", "1")
        assert got.returncode == want.returncode == 101, path
        assert got.stdout == want.stdout
        assert got.stderr.strip().splitlines()[-1] == want.stderr.strip().splitlines()[-1]


@pytest.mark.parametrize("alphabet,n_letters", [(0, 4), (1, 20)])
def test_near_hit_ladder_planned_from_a_sample(alphabet, n_letters):
    """the k-th-distance modes without a tight bound on a BIG query batch: after the first bounded step the later steps are
    chosen from a sample of the open queries (engine.hip plan_later_steps) — near, middling (8..20 columns away) and
    unrelated queries mixed; every mix against the exhaustive answer (all distances, numpy, the k-th rule of
    src/lib.rs:250-262 applied to them)"""
    rng = np.random.default_rng(77 + alphabet)
    n, L, nq = 6_000, 60, 4_500
    s = rng.integers(0, n_letters, size=(n, L), dtype=np.uint8)
    s[500:520] = s[499]

    def planted(count, lo, hi):
        q = s[rng.integers(0, n, size=count)].copy()
        for r in q:
            for c in rng.choice(L, size=int(rng.integers(lo, hi + 1)), replace=False):
                r[c] = (r[c] + 1 + rng.integers(0, n_letters - 1)) % n_letters
        return q

    modes = ((None, 1), (None, 3), (40, 1), (25, 2))

    def exhaustive(q):  # -> {mode: rows}; the distance matrix is formed once per chunk of queries
        out = {m: [] for m in modes}
        for lo in range(0, len(q), 250):
            d = (q[lo:lo + 250, None, :] != s[None, :, :]).sum(axis=2).astype(np.uint32)
            for D, k in modes:
                kth = np.partition(d, k - 1, axis=1)[:, k - 1]
                thr = np.minimum(kth, np.uint32(L if D is None else D))
                qi, sj = np.nonzero(d <= thr[:, None])
                dd = d[qi, sj]
                order = np.lexsort((sj, dd, qi))
                rows = np.zeros(len(order), dtype=smafa_amd.HIT_DTYPE)
                rows["query"], rows["subject"], rows["dist"] = qi[order] + lo, sj[order], dd[order]
                out[(D, k)].append(rows)
        return {m: np.concatenate(v) for m, v in out.items()}

    near, mid = planted(nq // 3, 0, 4), planted(nq // 3, 8, 20)
    far = rng.integers(0, n_letters, size=(nq - len(near) - len(mid), L), dtype=np.uint8)
    mixes = {"all kinds": np.concatenate([near, mid, far]), "middling only": np.concatenate([mid, mid[::-1], mid]),
             "unrelated only": np.concatenate([far, far[::-1], far]), "near, then unrelated": np.concatenate([near, far, far])}
    store = smafa_amd.SubjectStore(L, alphabet)
    store.push(s)
    for name, q in mixes.items():
        q = q[rng.permutation(len(q))]
        want = exhaustive(q)
        for D, k in modes:
            got = store.scan(q, max_divergence=D, max_num_hits=k)
            assert got.tobytes() == want[(D, k)].tobytes(), (name, D, k, len(got), len(want[(D, k)]))
    store.close()
