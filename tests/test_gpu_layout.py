"""GPU parity of the round-2 store layout: per-column re-coding and column order chosen from the data, subjects kept
sorted by their filter words (positions != subject indices), the zone level of the filter-plane-resident kernel, and
the single-launch row append (LDS-staged rows, one counter, exact totals).  Everything against the oracle, bit-exact,
through the C ABI."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle
import smafa_amd

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def _built():
    smafa_amd.build()
    assert smafa_amd.device_count() >= 1


def skewed_store(rng, n, L, n_letters, conserved_frac=0.5, families=0):
    """columns with very different letter distributions: a share of them nearly constant (one dominant letter per
    column), the rest uniform — what the layout's column order and re-coding react to; optionally family structure"""
    s = rng.integers(0, n_letters, size=(n, L), dtype=np.uint8)
    cons = rng.random(L) < conserved_frac
    dom = rng.integers(0, n_letters, size=L, dtype=np.uint8)
    keep = rng.random((n, L)) < 0.93
    s[:, cons] = np.where(keep[:, cons], dom[cons][None, :], s[:, cons])
    if families:
        roots = s[rng.integers(0, n, size=families)]
        member = rng.integers(0, families, size=n)
        mut = rng.random((n, L)) < 0.15
        s = np.where(mut, s, roots[member])
    return np.ascontiguousarray(s)


def queries_from(rng, s, q, n_letters, max_subs):
    out = s[rng.integers(0, len(s), size=q)].copy()
    for r in out:
        for _ in range(rng.integers(0, max_subs + 1)):
            r[rng.integers(0, s.shape[1])] = rng.integers(0, n_letters)
    return out


def expected_with_k(all_hits, k):
    out, i = [], 0
    while i < len(all_hits):
        j = i
        while j < len(all_hits) and all_hits[j]["query"] == all_hits[i]["query"]:
            j += 1
        grp = all_hits[i:j]
        kth = grp[k - 1]["dist"] if len(grp) >= k else 0xFFFFFFFF
        out.append(grp[grp["dist"] <= kth])
        i = j
    return np.concatenate(out) if out else all_hits[:0]


@pytest.fixture
def zone_env():
    """SMAFA_ZONE / SMAFA_SORT are read when a handle is created"""
    old = {k: os.environ.get(k) for k in ("SMAFA_ZONE", "SMAFA_SORT", "SMAFA_ZONE_DIRECT")}
    yield os.environ
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


@pytest.mark.parametrize("alphabet,n_letters", [(0, 4), (0, 5), (1, 20), (1, 28)])
@pytest.mark.parametrize("L", [20, 60, 100, 150])
@pytest.mark.parametrize("zone", ["0", "1", "2"])
def test_skewed_columns_sorted_store_all_modes(zone_env, alphabet, n_letters, L, zone):
    """stores big enough to be sorted (>= 4096 rows per append), skewed columns, families: rows must not depend on the
    layout, on the order of positions, or on the zone level (off / automatic / forced)"""
    zone_env["SMAFA_ZONE"] = zone
    rng = np.random.default_rng(L * 31 + n_letters + 7 * int(zone))
    zone_env["SMAFA_ZONE_DIRECT"] = str(int(rng.integers(0, 2)))  # both forms of the fixed-bound zone kernel across the cases
    n = 9000
    s = skewed_store(rng, n, L, n_letters, families=40 if L >= 60 else 0)
    q = queries_from(rng, s, 150, n_letters, 7)
    store = smafa_amd.SubjectStore(L, alphabet)
    store.push(s)
    for D in (0, 2, 5, min(9, L), L):
        got = store.scan(q, max_divergence=D)
        assert got.tobytes() == oracle.scan_codes(s, q, D).tobytes(), (alphabet, L, D, zone)
    for D, k in ((5, 1), (None, 1), (4, 3), (None, 6)):
        got = store.scan(q, max_divergence=D, max_num_hits=k)
        want = expected_with_k(oracle.scan_codes(s, q, L if D is None else D), k)
        assert got.tobytes() == want.tobytes(), (alphabet, L, D, k, zone)
    d = store.get_distances(q[0])
    assert (d == oracle.distances_codes(s, q[0])).all()
    store.close()


@pytest.mark.parametrize("direct", ["1", "0"])
def test_zone_kernel_is_the_one_that_runs_when_forced(zone_env, direct):
    """... in its unstaged form (fixed bound: heads straight from the record array, no barrier) and in the staged one"""
    zone_env["SMAFA_ZONE"] = "2"
    zone_env["SMAFA_ZONE_DIRECT"] = direct
    rng = np.random.default_rng(5)
    s = rng.integers(0, 20, size=(20000, 60), dtype=np.uint8)
    q = queries_from(rng, s, 300, 20, 8)
    store = smafa_amd.SubjectStore(60, 1)
    store.push(s)
    got = store.scan(q, max_divergence=5)
    assert got.tobytes() == oracle.scan_codes(s, q, 5).tobytes()
    assert store.last_scan_kernel() == "smafa::scan_zone_kernel<5, 5, 2, true, %s>" % ("true" if direct == "1" else "false")
    for few in (1, 3, 64):  # a handful of queries per pass: tiles fetched on demand
        one = store.scan(q[:few], max_divergence=5)
        assert one.tobytes() == oracle.scan_codes(s, q[:few], 5).tobytes()
        assert store.last_scan_kernel() == "smafa::scan_zone_few_kernel<5, 5, 2>"
    store.close()
    zone_env["SMAFA_ZONE"] = "0"
    store = smafa_amd.SubjectStore(60, 1)
    store.push(s)
    assert store.scan(q, max_divergence=5).tobytes() == got.tobytes()
    assert store.last_scan_kernel() == "smafa::scan_lazy_kernel<5, 5, 2, 4, false, false>"
    store.close()


@pytest.mark.parametrize("alphabet,n_letters", [(0, 4), (1, 24)])
def test_appends_in_pieces_sorted_and_unsorted(zone_env, alphabet, n_letters):
    """big (sorted) and small (unsorted) appends interleaved, starting mid-tile: positions of later appends follow
    the earlier ones, subject indices stay the append order; an N arriving late re-planes a sorted 2-bit store"""
    zone_env["SMAFA_ZONE"] = "2"
    rng = np.random.default_rng(99 + alphabet)
    L = 60
    parts = [rng.integers(0, n_letters, size=(m, L), dtype=np.uint8) for m in (5000, 3, 4100, 700, 1, 6000)]
    if alphabet == 0:
        parts[3][5, 7] = 4  # the first N: the store gains its third plane after two sorted appends
    s = np.concatenate(parts)
    q = queries_from(rng, s, 120, n_letters, 6)
    whole = smafa_amd.SubjectStore(L, alphabet)
    whole.push(s)
    pieces = smafa_amd.SubjectStore(L, alphabet)
    for p in parts:
        pieces.push(p)
    assert len(pieces) == len(s)
    for D in (0, 3, 6, 30):
        want = oracle.scan_codes(s, q, D)
        assert whole.scan(q, max_divergence=D).tobytes() == want.tobytes(), D
        assert pieces.scan(q, max_divergence=D).tobytes() == want.tobytes(), D
    want1 = expected_with_k(oracle.scan_codes(s, q, L), 1)
    assert pieces.scan(q, max_num_hits=1).tobytes() == want1.tobytes()
    assert (pieces.get_distances(q[3]) == oracle.distances_codes(s, q[3])).all()
    whole.close()
    pieces.close()


def besthit_rows(s, q):
    """every subject at each query's minimum distance, ordered (query, subject) — src/lib.rs:296-313 — from the oracle's
    per-query distance vectors (materialising every pair as a row costs ten seconds at these sizes)"""
    out = []
    for i in range(len(q)):
        d = oracle.distances_codes(s, q[i])
        j = np.nonzero(d == d.min())[0]
        rows = np.zeros(len(j), dtype=smafa_amd.HIT_DTYPE)
        rows["query"], rows["subject"], rows["dist"] = i, j, d.min()
        out.append(rows)
    return np.concatenate(out)


@pytest.mark.parametrize("alphabet,n_letters,L,D", [(1, 20, 60, 3), (0, 4, 60, 2), (0, 5, 100, 3), (1, 24, 20, 2)])
def test_store_grown_by_small_appends_is_sorted_again(zone_env, tmp_path, alphabet, n_letters, L, D):
    """a patchwork of appends (cluster's centroid set, a DB loaded in pieces) is sorted again on the device before a scan
    once it has grown by a quarter since its last full sort: the zone kernel then runs on it, rows unchanged; SMAFA_RESORT=0
    keeps the patchwork (and the kernel for it)"""
    zone_env.pop("SMAFA_ZONE", None)
    rng = np.random.default_rng(1000 + L + alphabet)
    sizes = [3000] * 20 + [1, 255, 4500, 700]  # small (unsorted) and big (sorted one by one) appends
    parts = [rng.integers(0, n_letters, size=(m, L), dtype=np.uint8) for m in sizes]
    if alphabet == 0 and n_letters == 5:
        parts = [np.minimum(p, 3) for p in parts]  # N comes later
    s = np.concatenate(parts)
    q = queries_from(rng, s, 300, min(n_letters, 4) if alphabet == 0 else n_letters, 5)
    want = oracle.scan_codes(s, q, D)
    os.environ["SMAFA_RESORT"] = "0"
    try:
        plain = smafa_amd.SubjectStore(L, alphabet)
    finally:
        os.environ.pop("SMAFA_RESORT")
    store = smafa_amd.SubjectStore(L, alphabet)
    for p_ in parts:
        plain.push(p_)
        store.push(p_)
    assert plain.scan(q, max_divergence=D).tobytes() == want.tobytes()
    assert not plain.last_scan_kernel().startswith("smafa::scan_zone")
    assert store.scan(q, max_divergence=D).tobytes() == want.tobytes()
    assert store.last_scan_kernel().startswith("smafa::scan_zone_kernel"), store.last_scan_kernel()
    assert store.scan(q[:5], max_divergence=D).tobytes() == oracle.scan_codes(s, q[:5], D).tobytes()
    assert store.scan(q, max_num_hits=1).tobytes() == besthit_rows(s, q).tobytes()
    assert (store.get_distances(q[1]) == oracle.distances_codes(s, q[1])).all()
    # grows by less than a quarter: the tail stays a patch; then by more: sorted again.  Rows right either way.
    more = [rng.integers(0, n_letters, size=(m, L), dtype=np.uint8) for m in (2000, 9000, 9000, 5)]
    for i, m_ in enumerate(more):
        store.push(m_)
        s = np.concatenate([s, m_])
        assert store.scan(q, max_divergence=D).tobytes() == oracle.scan_codes(s, q, D).tobytes(), i
    path = str(tmp_path / "grown.packed")
    store.save(path)
    store.close()
    plain.close()
    back = smafa_amd.SubjectStore.load(path)
    assert back.scan(q, max_divergence=D).tobytes() == oracle.scan_codes(s, q, D).tobytes()
    back.close()
    alphabet_back, codes = smafa_amd.read_db(path)
    assert alphabet_back == alphabet and codes.tobytes() == s.tobytes()


@pytest.mark.parametrize("alphabet,n_letters,L", [(1, 20, 200), (0, 4, 150), (0, 5, 300), (1, 28, 129)])
def test_zone_level_of_the_wide_kernel(zone_env, alphabet, n_letters, L):
    """more than 128 columns: scan_wide_kernel applies the zone level itself when the store is sorted well enough"""
    rng = np.random.default_rng(7 * L + alphabet)
    s = rng.integers(0, n_letters, size=(40000, L), dtype=np.uint8)
    s[100:200] = s[0:100]
    q = queries_from(rng, s, 333, n_letters, 6)
    rows = {}
    for zone in ("1", "0", "2"):
        zone_env["SMAFA_ZONE"] = zone
        store = smafa_amd.SubjectStore(L, alphabet)
        store.push(s[:30000])
        store.push(s[30000:])
        for D in (0, 3, 5):
            got = store.scan(q, max_divergence=D)
            assert got.tobytes() == oracle.scan_codes(s, q, D).tobytes(), (zone, D)
            name = store.last_scan_kernel()
            assert name.startswith("smafa::scan_wide_kernel"), name
            if zone != "1" or D == 0:  # automatic: on where the tiles' shared bits exclude most queries (here: D = 0)
                assert name.endswith("(zone level on)") == (zone != "0"), (zone, D, name)
        rows[zone] = store.scan(q, max_num_hits=2, max_divergence=4).tobytes()
        one = store.scan(q[:1], max_divergence=3)
        assert one.tobytes() == oracle.scan_codes(s, q[:1], 3).tobytes()
        store.close()
    assert len(set(rows.values())) == 1
    assert rows["1"] == expected_with_k(oracle.scan_codes(s, q, 4), 2).tobytes()


def test_sorted_and_unsorted_layouts_agree(zone_env):
    rng = np.random.default_rng(17)
    s = skewed_store(rng, 30000, 60, 20, families=100)
    q = queries_from(rng, s, 200, 20, 8)
    rows = []
    for sort, zone in (("1", "2"), ("0", "2"), ("0", "0"), ("1", "1")):
        zone_env["SMAFA_SORT"], zone_env["SMAFA_ZONE"] = sort, zone
        store = smafa_amd.SubjectStore(60, 1)
        store.push(s)
        rows.append(store.scan(q, max_divergence=6).tobytes())
        store.close()
    assert len(set(rows)) == 1 and rows[0] == oracle.scan_codes(s, q, 6).tobytes()


def test_device_launch_counts_are_exact_at_any_capacity():
    """smafa_scan_launch's contract: *d_count = number of qualifying rows, even past cap; the first cap are stored.
    A dense store where every pair qualifies, cap == the exact row count, cap below it, cap == 0 rows stored."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "device_capacity_worker.py")], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "device capacity ok" in r.stdout


def test_dense_rows_through_the_lds_stage_and_spill():
    """far more rows per workgroup and chunk than the LDS stage holds: the spill path, exact totals, every row once"""
    rng = np.random.default_rng(3)
    L, n = 60, 6000
    base = rng.integers(0, 20, size=L, dtype=np.uint8)
    s = np.tile(base, (n, 1))
    for r in s:
        for _ in range(rng.integers(0, 4)):
            r[rng.integers(0, L)] = rng.integers(0, 20)
    q = s[rng.integers(0, n, size=300)].copy()
    store = smafa_amd.SubjectStore(L, 1)
    store.push(s)
    for D in (2, 6, L):
        got = store.scan(q, max_divergence=D)
        want = oracle.scan_codes(s, q, D)
        assert len(got) == len(want) and got.tobytes() == want.tobytes(), D
    store.close()


# ------------------------------------------------------------- one process, several handles (smafa_query_multi)
def _write_fasta(path, ascii_rows):
    with open(path, "wb") as f:
        for i, r in enumerate(ascii_rows):
            f.write(b">r%d\n" % i + bytes(r) + b"\n")


@pytest.mark.parametrize("flags", [[], ["--max-divergence", "4"], ["--max-num-hits", "5"],
                                   ["--max-num-hits", "4", "--limit-per-sequence", "1", "--max-divergence", "20"]])
def test_query_multi_handles_equal_oracle_cli(tmp_path, flags):
    """`smafa query --devices 0,0,..`: N handles on GPU 0, one host thread each, contiguous query blocks — stdout must be
    byte-identical for N = 1, 2, 3 and equal to the oracle CLI's (src/lib.rs:232-318 carries no state across queries)."""
    from smafa_amd import _lib

    rng = np.random.default_rng(31)
    letters = np.frombuffer(b"ACGTN", dtype=np.uint8)
    s = letters[rng.integers(0, 4, size=(6000, 60))]
    s[rng.random(size=s.shape) < 0.005] = ord("N")
    s[100:140] = s[7]
    q = s[rng.integers(0, len(s), size=211)].copy()
    for r in q:
        for _ in range(rng.integers(0, 7)):
            r[rng.integers(0, 60)] = letters[rng.integers(0, 5)]
    sf, qf, db = str(tmp_path / "s.fna"), str(tmp_path / "q.fna"), str(tmp_path / "db")
    _write_fasta(sf, s)
    _write_fasta(qf, q)
    assert subprocess.run([_lib.CLI_PATH, "makedb", "-i", sf, "-d", db]).returncode == 0
    want = oracle.run_cli("query", "-d", db, "-q", qf, *flags)
    assert want.returncode == 0 and len(want.stdout) > 0
    for devs in ("0", "0,0", "0,0,0"):
        got = subprocess.run([_lib.CLI_PATH, "query", "-d", db, "-q", qf, "--devices", devs, *flags], capture_output=True, text=True)
        assert got.returncode == 0, got.stderr
        assert got.stdout == want.stdout, devs
    got = subprocess.run([_lib.CLI_PATH, "query", "-d", db, "-q", qf, "--gpus", "1", *flags], capture_output=True, text=True)
    assert got.returncode == 0 and got.stdout == want.stdout
    # through the C ABI / Python mirror as well
    out = str(tmp_path / "out.tsv")
    fd = os.open(out, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    kw = {}
    for i, f in enumerate(flags):
        if f.startswith("--"):
            kw[f[2:].replace("-", "_")] = int(flags[i + 1])
    smafa_amd.query(db, qf, out_fd=fd, devices=[0, 0], **kw)
    os.close(fd)
    assert open(out).read() == want.stdout


def test_query_multi_errors_and_partial_output(tmp_path):
    from smafa_amd import _lib

    sf, qf, db = str(tmp_path / "s.fna"), str(tmp_path / "q.fna"), str(tmp_path / "db")
    open(sf, "wb").write(b">a\nACGTACGT\n>b\nTTTTACGT\n")
    open(qf, "wb").write(b">q0\nACGTACGT\n>q1\nTTTTACGT\n>q2\nACGTAEGT\n>q3\nTTTTACGT\n")
    assert subprocess.run([_lib.CLI_PATH, "makedb", "-i", sf, "-d", db]).returncode == 0
    want = oracle.run_cli("query", "-d", db, "-q", qf)
    got = subprocess.run([_lib.CLI_PATH, "query", "-d", db, "-q", qf, "--devices", "0,0"], capture_output=True, text=True)
    assert got.returncode == want.returncode == 101
    assert got.stdout == want.stdout == "0\t0\t0\tACGTACGT\n1\t1\t0\tTTTTACGT\n"
    assert 'Byte 69 cannot be interpreted as nucleotide, in sequence "q2" at position 5' in got.stderr
    bad = subprocess.run([_lib.CLI_PATH, "query", "-d", db, "-q", qf, "--devices", "0,99"], capture_output=True, text=True)
    assert bad.returncode == 1 and "device 99 out of range" in bad.stderr


# ------------------------------------------------------------------ packed store file (makedb --packed)
@pytest.mark.parametrize("with_n", [False, True])
def test_packed_store_cli_round_trip(tmp_path, with_n):
    """makedb --packed -> query: same bytes as the version-2 path and as the oracle CLI; the reference-side version gate
    rejects the file with its own text; smafa_dbfile_read recovers the code rows from the bit-planes"""
    from smafa_amd import _lib

    rng = np.random.default_rng(71 + with_n)
    letters = np.frombuffer(b"ACGTN", dtype=np.uint8)
    s = letters[rng.integers(0, 4, size=(7000, 60))]
    if with_n:
        s[rng.random(size=s.shape) < 0.004] = ord("N")
    s[50:90] = s[3]
    q = s[rng.integers(0, len(s), size=160)].copy()
    for r in q:
        for _ in range(rng.integers(0, 7)):
            r[rng.integers(0, 60)] = letters[rng.integers(0, 5)]
    sf, qf = str(tmp_path / "s.fna"), str(tmp_path / "q.fna")
    v2, pk = str(tmp_path / "v2.db"), str(tmp_path / "packed.db")
    _write_fasta(sf, s)
    _write_fasta(qf, q)
    run = lambda *a: subprocess.run([_lib.CLI_PATH, *a], capture_output=True, text=True)
    assert run("makedb", "-i", sf, "-d", v2).returncode == 0
    r = run("makedb", "-i", sf, "-d", pk, "--packed")
    assert r.returncode == 0, r.stderr
    assert open(pk, "rb").read(8) == b"\x03\x02SMAFA\x01"
    for flags in ([], ["--max-divergence", "5"], ["--max-num-hits", "4"], ["--max-num-hits", "5", "--limit-per-sequence", "1"]):
        want = oracle.run_cli("query", "-d", v2, "-q", qf, *flags)
        a, b = run("query", "-d", v2, "-q", qf, *flags), run("query", "-d", pk, "-q", qf, *flags)
        assert a.returncode == b.returncode == want.returncode == 0, (a.stderr, b.stderr)
        assert a.stdout == b.stdout == want.stdout and len(want.stdout) > 0, flags
        c = run("query", "-d", pk, "-q", qf, "--devices", "0,0", *flags)
        assert c.returncode == 0 and c.stdout == want.stdout
    ref = oracle.run_cli("query", "-d", pk, "-q", qf)  # the reference's version gate, src/lib.rs:214-217
    assert ref.returncode != 0 and "Unsupported db file version: 3." in ref.stderr
    alphabet, codes = smafa_amd.read_db(pk)
    _, codes2 = smafa_amd.read_db(v2)
    assert alphabet == 0 and codes.tobytes() == codes2.tobytes()
    info = smafa_amd.SubjectStore.load(pk).info()
    assert info.planes == (3 if with_n else 2) and info.n_subjects == len(s)


def test_packed_store_save_load_aa_and_appends(tmp_path):
    rng = np.random.default_rng(12)
    L = 60
    parts = [rng.integers(0, 24, size=(m, L), dtype=np.uint8) for m in (9000, 100, 5000)]
    s = np.concatenate(parts)
    q = queries_from(rng, s, 130, 24, 7)
    store = smafa_amd.SubjectStore(L, 1)
    for p in parts:
        store.push(p)
    path = str(tmp_path / "aa.packed")
    store.save(path)
    want = oracle.scan_codes(s, q, 6)
    assert store.scan(q, max_divergence=6).tobytes() == want.tobytes()
    store.close()
    back = smafa_amd.SubjectStore.load(path)
    assert len(back) == len(s) and back.alphabet == 1 and back.seq_len == L
    assert back.scan(q, max_divergence=6).tobytes() == want.tobytes()
    assert back.scan(q, max_num_hits=1).tobytes() == expected_with_k(oracle.scan_codes(s, q, L), 1).tobytes()
    extra = rng.integers(0, 24, size=(300, L), dtype=np.uint8)
    back.push(extra)  # a loaded store is an ordinary store
    s2 = np.concatenate([s, extra])
    assert back.scan(q, max_divergence=6).tobytes() == oracle.scan_codes(s2, q, 6).tobytes()
    assert (back.get_distances(q[0]) == oracle.distances_codes(s2, q[0])).all()
    back.close()
    alphabet, codes = smafa_amd.read_db(path)
    assert alphabet == 1 and codes.tobytes() == s.tobytes()


def test_packed_store_rejects_damaged_files(tmp_path):
    rng = np.random.default_rng(2)
    s = rng.integers(0, 4, size=(5000, 60), dtype=np.uint8)
    store = smafa_amd.SubjectStore(60, 0)
    store.push(s)
    path = str(tmp_path / "ok.packed")
    store.save(path)
    store.close()
    good = open(path, "rb").read()
    import struct

    def damaged(edit, name):
        b = bytearray(good)
        edit(b)
        p = str(tmp_path / name)
        open(p, "wb").write(bytes(b))
        return p

    cases = {
        "trunc": lambda b: b.__delitem__(slice(len(b) - 4096, len(b))),
        "rows": lambda b: b.__setitem__(slice(8 + 16, 8 + 24), struct.pack("<Q", 10**9)),     # n
        "planes": lambda b: b.__setitem__(slice(8 + 8, 8 + 12), struct.pack("<I", 7)),        # planes
        "perm": lambda b: b.__setitem__(slice(4096, 4098), struct.pack("<H", 60)),            # column 60 of 60
        "tab": lambda b: b.__setitem__(slice(8192, 8194), b"\x00\x00"),                       # two codes -> one stored code
        "offset": lambda b: b.__setitem__(slice(8 + 40 + 48, 8 + 40 + 56), struct.pack("<Q", len(good))),  # planes offset at the end
    }
    for name, edit in cases.items():
        with pytest.raises(smafa_amd.SmafaError):
            smafa_amd.SubjectStore.load(damaged(edit, name))
        with pytest.raises(smafa_amd.SmafaError):
            smafa_amd.read_db(damaged(edit, name))
    smafa_amd.SubjectStore.load(damaged(lambda b: None, "same")).close()
    # zone words are pruning metadata: a loader that believed a damaged copy would lose rows silently.  They are
    # recomputed from the planes at load time, so garbage there changes nothing.
    off_zone = struct.unpack_from("<Q", good, 8 + 40 + 40)[0]
    n_tiles = struct.unpack_from("<Q", good, 8 + 24)[0]
    q = queries_from(rng, s, 100, 4, 4)
    os.environ["SMAFA_ZONE"] = "2"
    try:
        back = smafa_amd.SubjectStore.load(damaged(lambda b: b.__setitem__(slice(off_zone, off_zone + 16 * n_tiles), b"\xff" * (16 * n_tiles)), "zone"))
    finally:
        os.environ.pop("SMAFA_ZONE")
    assert back.scan(q, max_divergence=3).tobytes() == oracle.scan_codes(s, q, 3).tobytes()
    assert back.last_scan_kernel().startswith("smafa::scan_zone_kernel")
    back.close()


@pytest.mark.parametrize("alphabet,n_letters,n,L", [(0, 4, 3000, 60), (0, 4, 30000, 60), (0, 5, 12000, 60), (1, 20, 30000, 60),
                                                    (1, 28, 5000, 60), (1, 20, 9000, 20), (0, 4, 9000, 40), (1, 24, 6000, 150)])
def test_host_packed_file_equals_device_saved_file(tmp_path, alphabet, n_letters, n, L):
    """the host restatement of the device's packing (layout, stable sort by filter words, ballot bit-planes, order, zone
    words) must produce the SAME BYTES as smafa_db_save of a store packed by the kernels — an independent check of
    pack_rows_kernel, row_keys_kernel, the radix sort's stability and zone_kernel"""
    from smafa_amd import synth

    rng = np.random.default_rng(n + n_letters)
    s = skewed_store(rng, n, L, n_letters, families=30)
    fa = str(tmp_path / "s.fa")
    synth.write_fasta(fa, s, alphabet)
    host, dev = str(tmp_path / "host.packed"), str(tmp_path / "dev.packed")
    smafa_amd.makedb_packed(fa, host, alphabet, device=-1)
    smafa_amd.makedb_packed(fa, dev, alphabet, device=0)
    a, b = open(host, "rb").read(), open(dev, "rb").read()
    assert len(a) == len(b)
    assert a == b
    q = queries_from(rng, s, 64, n_letters, 6)
    store = smafa_amd.SubjectStore.load(host)
    assert store.scan(q, max_divergence=5).tobytes() == oracle.scan_codes(s, q, 5).tobytes()
    store.close()
