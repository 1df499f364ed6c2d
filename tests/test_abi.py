"""CPU-side tests of the product: the C-ABI library loads and exports every declared symbol, the host
logic (encoding tables, DB file, FASTX reader, selection rules, CLI plumbing) matches the oracle and the
reference's golden vectors, and GPU entry points fail loudly without a device.  No compute calls here."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import oracle
import smafa_amd
from smafa_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def _built():
    smafa_amd.build()


def cli(*args):
    return subprocess.run([_lib.CLI_PATH, *args], capture_output=True, text=True)


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "smafa_amd.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(smafa_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    l = _lib.lib()
    for name in declared:
        assert hasattr(l, name), name


def test_encode_tables_match_oracle():
    for alphabet in (smafa_amd.ALPHABET_NT, smafa_amd.ALPHABET_AA):
        for b in range(256):
            want = oracle.lib().orc_code(alphabet, b)
            if want == 255:
                with pytest.raises(smafa_amd.SmafaPanic):
                    smafa_amd.encode(bytes([b]), alphabet)
            else:
                assert smafa_amd.encode(bytes([b]), alphabet)[0] == want
    assert smafa_amd.decode(smafa_amd.encode(b"ACGTURYKM-nacgtu"), smafa_amd.ALPHABET_NT) == b"ACGTTNNNNNNACGTT"
    assert smafa_amd.decode(smafa_amd.encode(b"acdeXZ*-", 1), 1) == b"ACDEXZ*-"


@pytest.mark.parametrize("name", ["random_3_2.fna", "random_3_2_one_repeated.fna"])
def test_makedb_bytes_match_reference_db(golden, tmp_path, name):
    db = tmp_path / "t.db"
    smafa_amd.makedb(os.path.join(golden, name), str(db))
    assert db.read_bytes() == open(os.path.join(golden, name + ".smafadb"), "rb").read()


def test_makedb_subjects_unit(golden, tmp_path):
    # src/lib.rs:334-367
    db = tmp_path / "s.db"
    smafa_amd.makedb(os.path.join(golden, "subjects.fa"), str(db))
    assert db.read_bytes() == bytes([2, 5, 1, 16, 1, 8, 1, 4, 1, 2, 1, 1, 1, 1])
    alphabet, codes = smafa_amd.read_db(str(db))
    assert alphabet == 0 and codes.tolist() == [[0], [1], [2], [3], [4]]


def test_makedb_equals_oracle_makedb_on_random_fasta(tmp_path):
    rng = np.random.default_rng(5)
    letters = np.frombuffer(b"ACGTUNRYKMacgtn-", dtype=np.uint8)
    for L in (1, 12, 13, 60, 61, 130):
        rows = letters[rng.integers(0, len(letters), size=(50, L))]
        f = tmp_path / f"r{L}.fna"
        with open(f, "wb") as fh:  # multi-line FASTA with CRLF on some lines
            for i, r in enumerate(rows):
                fh.write(b">s%d some description\r\n" % i)
                b = bytes(r)
                fh.write(b[: L // 2] + b"\r\n" + b[L // 2:] + b"\n")
        a, b = tmp_path / "a.db", tmp_path / "b.db"
        smafa_amd.makedb(str(f), str(a))
        assert oracle.run_cli("makedb", "-i", str(f), "-d", str(b)).returncode == 0
        assert a.read_bytes() == b.read_bytes()
        alphabet, codes = smafa_amd.read_db(str(a))
        assert (codes == oracle.codes_from_ascii(rows, 0)).all()


def test_dbfile_roundtrip_aa(tmp_path):
    rng = np.random.default_rng(6)
    codes = rng.integers(0, 28, size=(100, 60), dtype=np.uint8)
    p = tmp_path / "aa.db"
    smafa_amd.write_db(str(p), codes, smafa_amd.ALPHABET_AA)
    alphabet, back = smafa_amd.read_db(str(p))
    assert alphabet == 1 and (back == codes).all()
    # the reference's version gate rejects it, as the oracle restates (src/lib.rs:215-217)
    r = oracle.run_cli("query", "-d", str(p), "-q", os.path.join(ROOT, "tests/golden/random_3_2.fna"))
    assert r.returncode != 0 and "Unsupported db file version: 3." in r.stderr


def test_old_db_version_message(golden):
    # tests/test_cmdline.rs:28-41 — fails before any device work
    r = cli("query", "-d", os.path.join(golden, "random_3_2.fna.v1.smafadb"), "-q", os.path.join(golden, "random_3_2.fna"))
    assert r.returncode == 101 and "Unsupported db file version: 1." in r.stderr and r.stdout == ""


def test_count_cli(golden):
    # tests/test_cmdline.rs:184-201
    for name, reads, bases in (("random_3_2.fna", 2, 6), ("random_30_4.fq.gz", 4, 120)):
        p = os.path.join(golden, name)
        r = cli("count", "-i", p)
        assert r.returncode == 0
        assert r.stdout == '[{"path":"%s","num_reads":%d,"num_bases":%d}]\n' % (p, reads, bases)


def test_bad_input_messages(tmp_path):
    f = tmp_path / "bad.fna"
    f.write_bytes(b">seq one\nACGTACGTACGTAE\n")
    r = cli("makedb", "-i", str(f), "-d", str(tmp_path / "x.db"))
    assert r.returncode == 101
    assert 'Byte 69 cannot be interpreted as nucleotide, in sequence "seq one" at position 13' in r.stderr
    f.write_bytes(b">a\nACGT\n>b\nACG\n")
    r = cli("makedb", "-i", str(f), "-d", str(tmp_path / "x.db"))
    assert r.returncode == 101 and "WindowSet seq length is 4, got a new sequence of length 3" in r.stderr
    r = cli("cluster", "-i", str(f))
    assert r.returncode == 101  # src/main.rs:43 unwraps --max-divergence


def _oracle_rows(text):
    return [tuple(l.split("\t")) for l in text.splitlines()]


@pytest.mark.parametrize("flags", [
    (None, None, None), (5, None, None), (None, 3, None), (4, 3, None), (None, 99999, None), (6, 5, 1), (None, 7, 2),
    (0, None, None), (0, 2, None), (60, 4, 1),
])
def test_select_rows_equals_oracle_query(flags):
    """product selection (host) on oracle-made hit lists == oracle's full query output"""
    max_div, max_hits, limit = flags
    rng = np.random.default_rng(hash(flags) % 1000)
    L, n, q = 24, 400, 40
    codes = rng.integers(0, 4, size=(n, L), dtype=np.uint8)
    codes[50:80] = codes[10]                     # many identical subjects (ties, limit-per-sequence)
    codes[100:120, :20] = codes[10, :20]         # near copies
    qcodes = codes[rng.integers(0, n, size=q)].copy()
    for r in qcodes:
        for _ in range(rng.integers(0, 6)):
            r[rng.integers(0, L)] = rng.integers(0, 4)
    ascii_s = np.frombuffer(b"ACGT", dtype=np.uint8)[codes]
    ascii_q = np.frombuffer(b"ACGT", dtype=np.uint8)[qcodes]
    args = []
    if max_div is not None:
        args += ["--max-divergence", str(max_div)]
    if max_hits is not None:
        args += ["--max-num-hits", str(max_hits)]
    if limit is not None:
        args += ["--limit-per-sequence", str(limit)]
    want = oracle.query_text([bytes(r) for r in ascii_s], [bytes(r) for r in ascii_q], *args)
    # the hit list a scan would deliver: everything within max_div (the k-th bound is applied by select)
    hits = oracle.scan_codes(codes, qcodes, L if max_div is None else max_div)
    rows = smafa_amd.select_rows(hits, q, n, codes, L, max_div, max_hits, limit)
    got = "".join("%d\t%d\t%d\t%s\n" % (r["query"], r["subject"], r["dist"], smafa_amd.decode(codes[r["subject"]]).decode())
                  for r in rows)
    assert got == want


def test_select_rows_panics():
    hits = np.zeros(1, dtype=smafa_amd.HIT_DTYPE)
    codes = np.zeros((1, 4), dtype=np.uint8)
    with pytest.raises(smafa_amd.SmafaPanic, match="limit_per_sequence is implemented unless"):
        smafa_amd.select_rows(hits, 1, 1, codes, 4, None, None, 1)          # src/lib.rs:301-303
    with pytest.raises(smafa_amd.SmafaPanic, match="index out of bounds"):
        smafa_amd.select_rows(hits, 1, 1, codes, 4, None, 0, None)          # src/lib.rs:255
    with pytest.raises(smafa_amd.SmafaPanic, match="Option::unwrap"):
        smafa_amd.select_rows(hits[:0], 1, 0, None, 4, None, None, None)    # src/lib.rs:298 on an empty store


def test_scan_fails_loudly_without_gpu(golden):
    if smafa_amd.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(smafa_amd.SmafaError, match="no CPU fallback"):
        smafa_amd.SubjectStore(60)
    r = cli("query", "-d", os.path.join(golden, "random_3_2.fna.smafadb"), "-q", os.path.join(golden, "random_3_2.fna"))
    assert r.returncode == 1 and "no CPU fallback" in r.stderr and r.stdout == ""


def test_synth_queries_have_planted_distance():
    from smafa_amd import synth
    for alphabet, ms in ((1, 10), (0, 6)):
        s = synth.subjects(2000, 60, alphabet, seed=1)
        q, rows, subs = synth.queries(s, 200, alphabet, seed=3, max_subs=ms)
        d = (s[rows] != q).sum(axis=1)
        assert (d == subs).all()


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """what a C or Rust host does: include the header as C99, link the shared object, call the ABI"""
    exe = tmp_path / "abi_c_check"
    lib_dir = os.path.dirname(_lib.LIB_PATH)
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tests", "abi_c_check.c"), "-o", str(exe), "-L", lib_dir, "-lsmafa_amd",
                        "-Wl,-rpath," + lib_dir], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe), str(tmp_path / "t.db")], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "abi ok" in r.stdout


def test_parallel_ingest_paths_match_oracle(tmp_path):
    """inputs big enough for the multi-threaded FASTA loader (>= 32 MB) and DB decode/serialise (>= 2^20 rows):
    DB bytes equal the oracle's sequential makedb, decode round-trips, first-error semantics are the oracle's"""
    from smafa_amd import synth
    n, L = 1_150_000, 29                      # 29: not a multiple of 12, so rows are compacted after decode
    codes = synth.subjects(n, L, 0, seed=9, n_frac=0.01)
    f = str(tmp_path / "big.fna")
    synth.write_fasta(f, codes, 0)
    assert os.path.getsize(f) >= 32 << 20
    a, b = str(tmp_path / "a.db"), str(tmp_path / "b.db")
    smafa_amd.makedb(f, a)
    assert oracle.run_cli("makedb", "-i", f, "-d", b).returncode == 0
    assert open(a, "rb").read() == open(b, "rb").read()
    alphabet, back = smafa_amd.read_db(a)
    assert alphabet == 0 and (back == codes).all()
    assert (smafa_amd.load_fastx(f, 0) == codes).all()
    # a bad byte and, later, a short record: the FIRST offending record in file order is reported
    raw = open(f, "rb").read()
    i = raw.index(b">r900000\n") + len(b">r900000\n")
    j = raw.index(b">r1100000\n") + len(b">r1100000\n")
    broken = raw[:i + 3] + b"!" + raw[i + 4:j] + raw[j + 2:]
    g = str(tmp_path / "broken.fna")
    open(g, "wb").write(broken)
    r = cli("makedb", "-i", g, "-d", str(tmp_path / "x.db"))
    o = oracle.run_cli("makedb", "-i", g, "-d", str(tmp_path / "y.db"))
    assert r.returncode == o.returncode == 101
    assert 'Byte 33 cannot be interpreted as nucleotide, in sequence "r900000" at position 3' in r.stderr
    assert r.stderr.strip().splitlines()[-1] == o.stderr.strip().splitlines()[-1]


def test_cluster_sharded_argument_checks(tmp_path):
    """rank/world/transport are validated before anything touches a device"""
    import ctypes as C

    l = _lib.lib()
    path = os.fsencode(str(tmp_path / "none.fna"))
    null_cb = C.cast(None, _lib.ALLGATHER_FN)
    assert l.smafa_cluster_sharded(path, 3, 1, 0, 0, 2, 2, null_cb, None) == -1
    assert b"rank 2 outside world of 2" in l.smafa_last_error()
    assert l.smafa_cluster_sharded(path, 3, 1, 0, 0, 0, 0, null_cb, None) == -1
    assert l.smafa_cluster_sharded(path, 3, 1, 0, 0, 1, 2, null_cb, None) == -1
    assert b"NULL allgather" in l.smafa_last_error()
    # an empty file is a parse error (needletail refuses it) before the transport or a device is touched
    empty = str(tmp_path / "empty.fna")
    open(empty, "w").close()
    calls = []
    cb = _lib.ALLGATHER_FN(lambda *a: calls.append(a) or 1)
    assert l.smafa_cluster_sharded(os.fsencode(empty), 3, 1, 0, 0, 1, 2, cb, None) == -6 and not calls  # .expect(..), src/cluster.rs:28
    assert b"valid path/file of input fasta" in l.smafa_last_error()


def test_cluster_multi_argument_checks(tmp_path):
    import ctypes as C

    l = _lib.lib()
    path = os.fsencode(str(tmp_path / "none.fna"))
    assert l.smafa_cluster_multi(path, 3, 1, None, 2, 0) == -1
    devs = (C.c_int * 2)(0, 0)
    assert l.smafa_cluster_multi(path, 3, 1, devs, 0, 0) == -1
    assert l.smafa_cluster_multi(path, 3, 1, devs, 2, 0) == -6  # the input is looked at first: .expect(..), src/cluster.rs:28
    assert b"valid path/file of input fasta" in l.smafa_last_error()


def test_dbfile_property_any_shape(tmp_path):
    """hypothesis over shapes and letters: makedb bytes == the oracle's makedb bytes, and read -> write is the
    identity on the file (the v2 wire format of src/lib.rs:54-60,161-162 for every window count and varint width)"""
    from hypothesis import given, settings, strategies as st

    valid = b"ACGTUNRYKMSWBDHVacgtunrykmswbdhv-"
    counter = [0]

    @settings(max_examples=30, deadline=None, derandomize=True, database=None)
    @given(st.integers(1, 140), st.integers(1, 40), st.integers(0, 2**32 - 1))
    def check(L, n, seed):
        counter[0] += 1
        rng = np.random.default_rng(seed)
        letters = np.frombuffer(valid, dtype=np.uint8)
        # skew towards one letter now and then: long runs of equal windows, small and large varints
        p = rng.dirichlet(np.ones(len(letters)) * (0.05 if seed % 3 == 0 else 1.0))
        rows = letters[rng.choice(len(letters), size=(n, L), p=p)]
        f, a, b, c = (str(tmp_path / ("%s%d" % (x, counter[0]))) for x in "fabc")
        oracle.write_fasta(f, [bytes(r) for r in rows])
        smafa_amd.makedb(f, a)
        assert oracle.run_cli("makedb", "-i", f, "-d", b).returncode == 0
        assert open(a, "rb").read() == open(b, "rb").read()
        alphabet, codes = smafa_amd.read_db(a)
        assert alphabet == 0 and (codes == oracle.codes_from_ascii(rows, 0)).all()
        smafa_amd.write_db(c, codes, 0)
        assert open(c, "rb").read() == open(a, "rb").read()

    check()


def test_query_multi_needs_devices_and_a_gpu(golden, tmp_path):
    """smafa_query_multi: argument checks on any host; with no GPU the scan entry fails with SMAFA_ERR_DEVICE (no fallback)"""
    import ctypes as C

    l = _lib.lib()
    db = os.fsencode(os.path.join(golden, "random_3_2.fna.smafadb"))
    q = os.fsencode(os.path.join(golden, "random_3_2.fna"))
    devs = (C.c_int * 2)(0, 0)
    assert l.smafa_query_multi(db, q, _lib.NONE, _lib.NONE, _lib.NONE, 1, devs, 0) == _lib.ERR_INVALID
    assert l.smafa_query_multi(db, q, _lib.NONE, _lib.NONE, _lib.NONE, 1, None, 2) == _lib.ERR_INVALID
    if smafa_amd.device_count() == 0:
        assert l.smafa_query_multi(db, q, _lib.NONE, _lib.NONE, _lib.NONE, 1, devs, 2) == _lib.ERR_DEVICE
        assert b"no CPU fallback" in l.smafa_last_error()


def test_parallel_fastq_and_gzip_ingest_match_oracle(tmp_path):
    """>= 32 MB of FASTQ, FASTA.gz and FASTQ.gz take the threaded loader (gzip: one inflating thread feeding the parsers);
    rows equal the generator's, quality lines that begin with '@' do not confuse the chunking, and a damaged record
    fails at the same record with the same text as the oracle's one-thread reader; multi-member gzip falls back."""
    import gzip

    from smafa_amd import synth
    n, L = 620_000, 60
    codes = synth.subjects(n, L, 0, seed=13, n_frac=0.01)
    asc = np.frombuffer(b"ACGTN", dtype=np.uint8)[codes]

    def fastq_bytes(patch=None):
        parts = []
        for i in range(n):
            seq = patch[i] if patch and i in patch else asc[i].tobytes()
            qual = (b"@" if i % 5 == 0 else b"I") * len(seq)
            parts.append(b"@r%d\n" % i + seq + b"\n+\n" + qual + b"\n")
        return b"".join(parts)

    def load_log(path):  # which loader ran: the drivers' debug line on stderr
        r = subprocess.run([sys.executable, "-c",
                            "import sys; sys.path.insert(0, %r); import smafa_amd; from smafa_amd import _lib; "
                            "_lib.lib().smafa_set_verbosity(2); a = smafa_amd.load_fastx(%r, 0); print(a.shape[0])" % (ROOT, path)],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        return r.stderr

    body = fastq_bytes()
    assert len(body) >= 64 << 20
    fq, fqgz, fagz, multi = (str(tmp_path / x) for x in ("a.fq", "a.fq.gz", "a.fna.gz", "multi.fq.gz"))
    open(fq, "wb").write(body)
    with gzip.open(fqgz, "wb", compresslevel=1) as f:
        f.write(body)
    fa = b"".join(b">r%d\n" % i + asc[i].tobytes() + b"\n" for i in range(n))
    with gzip.open(fagz, "wb", compresslevel=1) as f:
        f.write(fa)
    with open(multi, "wb") as f:  # two gzip members: the size trailer describes only the last one
        half = body[: body.index(b"@r300000\n")]
        f.write(gzip.compress(half, 1) + gzip.compress(body[len(half):], 1))
    for path, expect in ((fq, "parsed by"), (fqgz, "while one thread inflated"), (fagz, "while one thread inflated"),
                         (multi, "one-thread reader")):
        assert (smafa_amd.load_fastx(path, 0) == codes).all(), path
        assert expect in load_log(path), path
    # damaged records: a byte outside the alphabet deep in the file, then (earlier in the file) a short record
    for patch in ({500_123: asc[500_123].tobytes()[:17] + b"!" + asc[500_123].tobytes()[18:]},
                  {500_123: asc[500_123].tobytes()[:17] + b"!" + asc[500_123].tobytes()[18:], 222_222: asc[222_222].tobytes()[:59]}):
        bad = fastq_bytes(patch)
        for path, writer in ((str(tmp_path / "bad.fq"), lambda p: open(p, "wb").write(bad)),
                             (str(tmp_path / "bad.fq.gz"), lambda p: open(p, "wb").write(gzip.compress(bad, 1)))):
            writer(path)
            r = cli("makedb", "-i", path, "-d", str(tmp_path / "x.db"))
            o = oracle.run_cli("makedb", "-i", path, "-d", str(tmp_path / "y.db"))
            assert r.returncode == o.returncode == 101, path
            assert r.stderr.strip().splitlines()[-1] == o.stderr.strip().splitlines()[-1], path


def test_packed_store_of_very_long_sequences(tmp_path):
    """more than 65 535 columns: the column-order table holds 32-bit entries (a 16-bit table wrapped around and packed
    duplicated columns); the host packer's file decodes back to the rows"""
    from smafa_amd import synth

    L, n = 70000, 6
    rng = np.random.default_rng(11)
    s = rng.integers(0, 4, size=(n, L), dtype=np.uint8)
    s[:, 65536:] = np.where(rng.random((n, L - 65536)) < 0.5, 0, s[:, 65536:])  # columns past 2^16 differ in information
    fa, pk = str(tmp_path / "long.fa"), str(tmp_path / "long.packed")
    synth.write_fasta(fa, s, 0)
    r = cli("makedb", "-i", fa, "-d", pk, "--packed", "--no-gpu")
    assert r.returncode == 0, r.stderr
    a, codes = smafa_amd.read_db(pk)
    assert a == 0 and codes.shape == (n, L) and codes.tobytes() == s.tobytes()


def test_db_with_a_non_one_hot_group_is_rejected_at_load(golden, tmp_path):
    """deliberate restriction: the v2 loader refuses windows the reference's makedb cannot produce,
    with its own text (not the reference's print-time panic) — pinned on a mutated copy of the reference's fixture"""
    good = open(os.path.join(golden, "random_3_2.fna.smafadb"), "rb").read()
    assert good == bytes.fromhex("020201c810019021" "0103")
    # window 0 = varint(2120) = C | T<<5 | T<<10; make its first group 3 (two bits set): 2115 = c3 10
    bad = good.replace(bytes.fromhex("c810"), bytes.fromhex("c310"), 1)
    p = str(tmp_path / "bad.smafadb")
    open(p, "wb").write(bad)
    with pytest.raises(smafa_amd.SmafaError) as e:
        smafa_amd.read_db(p)
    assert e.value.code == _lib.ERR_FORMAT and "not a one-hot nucleotide code" in str(e.value)
    assert not isinstance(e.value, smafa_amd.SmafaPanic)
    r = cli("query", "-d", p, "-q", os.path.join(golden, "random_3_2.fna"))
    assert r.returncode == 1 and r.stdout == "" and "damaged store file" in r.stderr
    # an empty group inside the sequence (column 1 of window 0): 8 | 0<<5 | 2<<10 = 2056 = 88 10
    empty = good.replace(bytes.fromhex("c810"), bytes.fromhex("8810"), 1)
    open(p, "wb").write(empty)
    with pytest.raises(smafa_amd.SmafaError):
        smafa_amd.read_db(p)


@pytest.mark.parametrize("alphabet,n_letters,n", [(0, 4, 3000), (0, 5, 6000), (1, 24, 9000)])
def test_packed_store_file_written_without_a_gpu(tmp_path, alphabet, n_letters, n):
    """`smafa makedb --packed --no-gpu`: the packed store file from host threads (host/layout.cpp).  It decodes back to the
    code rows, the reference-side version gate rejects it with its own text, and damaged copies are refused at open time."""
    import struct

    rng = np.random.default_rng(n)
    L = 60
    s = rng.integers(0, n_letters, size=(n, L), dtype=np.uint8)
    s[:, :20] = np.where(rng.random((n, 20)) < 0.9, s[0, :20], s[:, :20])  # conserved columns: the layout reorders them
    from smafa_amd import synth
    fa, pk = str(tmp_path / "s.fa"), str(tmp_path / "s.packed")
    synth.write_fasta(fa, s, alphabet)
    r = cli("makedb", "-i", fa, "-d", pk, "--packed", "--no-gpu", *(["--alphabet", "aa"] if alphabet else []))
    assert r.returncode == 0, r.stderr
    good = open(pk, "rb").read()
    assert good[:8] == b"\x03\x02SMAFA\x01"  # version 3, kind 2, layout revision 1
    a, codes = smafa_amd.read_db(pk)
    assert a == alphabet and codes.tobytes() == s.tobytes()
    planes = struct.unpack_from("<I", good, 8 + 8)[0]
    assert planes == (5 if alphabet else (3 if n_letters == 5 else 2))
    ref = oracle.run_cli("query", "-d", pk, "-q", fa)  # src/lib.rs:214-217
    assert ref.returncode != 0 and "Unsupported db file version: 3." in ref.stderr

    def damaged(edit, name):
        b = bytearray(good)
        edit(b)
        p = str(tmp_path / name)
        open(p, "wb").write(bytes(b))
        return p

    cases = {
        "trunc": lambda b: b.__delitem__(slice(len(b) - 4096, len(b))),
        "rows": lambda b: b.__setitem__(slice(8 + 16, 8 + 24), struct.pack("<Q", 10**9)),
        "planes": lambda b: b.__setitem__(slice(8 + 8, 8 + 12), struct.pack("<I", 7)),
        "perm": lambda b: b.__setitem__(slice(4096, 4098), struct.pack("<H", 60)),
        "tab": lambda b: b.__setitem__(slice(8192, 8194), b"\x00\x00"),
        "offset": lambda b: b.__setitem__(slice(8 + 40 + 48, 8 + 40 + 56), struct.pack("<Q", len(good))),
        # the subject order (position -> subject: what the device indexes its output with) and its inverse
        "order_dup": lambda b: b.__setitem__(slice(off_order, off_order + 4), b[off_order + 4:off_order + 8]),
        "order_big": lambda b: b.__setitem__(slice(off_order + 8, off_order + 12), struct.pack("<I", n + 5)),
        "inv": lambda b: b.__setitem__(slice(off_inv, off_inv + 4), struct.pack("<I", (struct.unpack_from("<I", b, off_inv)[0] + 1) % n)),
    }
    off_inv, off_order = struct.unpack_from("<QQ", good, 8 + 40 + 24)
    for name, edit in cases.items():
        with pytest.raises(smafa_amd.SmafaError):
            smafa_amd.read_db(damaged(edit, name))
    # a file of an earlier layout revision (16-bit column table) is refused by name, whatever its bytes would have decoded to
    with pytest.raises(smafa_amd.SmafaError, match="older build.*re-run makedb --packed"):
        smafa_amd.read_db(damaged(lambda b: b.__setitem__(7, 0), "old_revision"))


@pytest.mark.parametrize("kind", ["fasta", "fasta_multiline_crlf", "fastq", "fastq_at_quality"])
def test_query_file_parts_partition_the_records(tmp_path, kind):
    """smafa_fastx_load_part (what one process per GPU reads of a query file): for any number of parts the parts'
    records, concatenated in part order, are the whole file's records; gzip input is refused (usable = False)"""
    import gzip

    rng = np.random.default_rng(3)
    n, L = 997, 37
    s = rng.integers(0, 4, size=(n, L), dtype=np.uint8)
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    recs = []
    for i, r in enumerate(letters[s]):
        seq = r.tobytes()
        if kind == "fasta":
            recs.append(b">r%d some words\n%s\n" % (i, seq))
        elif kind == "fasta_multiline_crlf":
            recs.append(b">r%d\r\n%s\r\n%s\r\n" % (i, seq[:20], seq[20:]))
        else:
            qual = bytes(rng.integers(33, 74, size=L, dtype=np.uint8))
            if kind == "fastq_at_quality":
                qual = b"@" + qual[1:]  # a quality line that begins like a header: the cut rule must not take it for one
            recs.append(b"@r%d\n%s\n+\n%s\n" % (i, seq, qual))
    path = str(tmp_path / "q.fx")
    open(path, "wb").write(b"".join(recs))
    whole = smafa_amd.load_fastx(path)
    assert whole.tobytes() == s.tobytes()
    for parts in (1, 2, 3, 7, 64, 2000):
        got, counts = [], []
        for p in range(parts):
            rows, err, usable = smafa_amd.load_fastx_part(path, p, parts)
            assert usable and err is None
            counts.append(len(rows))
            if len(rows):
                got.append(rows)
        assert np.concatenate(got).tobytes() == s.tobytes(), (kind, parts)
        if parts <= 64:
            assert max(counts) - min(counts) <= 2 + n // parts // 4, counts  # shares of nearly equal size
    # a bad byte in the middle: the part that holds it reports it, with the rows in front of it
    bad = b"".join(recs[:500]) + recs[500].replace(b"A", b"E", 1) + b"".join(recs[501:])
    open(path, "wb").write(bad)
    rows, err, usable = smafa_amd.load_fastx_part(path, 1, 2)
    first = len(smafa_amd.load_fastx_part(path, 0, 2)[0])
    assert usable and isinstance(err, smafa_amd.SmafaPanic) and "cannot be interpreted as nucleotide" in str(err)
    assert first + len(rows) == 500
    gz = str(tmp_path / "q.fx.gz")
    with gzip.open(gz, "wb") as g:
        g.write(b"".join(recs))
    assert smafa_amd.load_fastx_part(gz, 0, 2)[2] is False


@pytest.mark.parametrize("kind", ["bz2", "bz2_two_streams", "xz", "xz_two_streams"])
def test_bzip2_and_xz_inputs(tmp_path, kind):
    """needletail reads bzip2 and xz input as well as gzip (Cargo.toml:27; call sites src/lib.rs:144,221,381): makedb, count and
    the loaders give the bytes of the uncompressed file; a truncated stream is an error, not a short read"""
    import bz2
    import lzma

    rng = np.random.default_rng(8)
    n, L = 5000, 60
    s = rng.integers(0, 5, size=(n, L), dtype=np.uint8)
    from smafa_amd import synth
    plain = str(tmp_path / "s.fna")
    synth.write_fasta(plain, s, 0)
    raw = open(plain, "rb").read()
    half = raw.index(b">", len(raw) // 2)
    if kind == "bz2":
        blob = bz2.compress(raw)
    elif kind == "bz2_two_streams":
        blob = bz2.compress(raw[:half]) + bz2.compress(raw[half:])
    elif kind == "xz":
        blob = lzma.compress(raw, format=lzma.FORMAT_XZ)
    else:
        blob = lzma.compress(raw[:half], format=lzma.FORMAT_XZ) + lzma.compress(raw[half:], format=lzma.FORMAT_XZ)
    packed = str(tmp_path / ("s.fna." + kind.split("_")[0]))
    open(packed, "wb").write(blob)
    assert smafa_amd.load_fastx(packed).tobytes() == s.tobytes()
    db_a, db_b = str(tmp_path / "a.db"), str(tmp_path / "b.db")
    assert cli("makedb", "-i", plain, "-d", db_a).returncode == 0
    r = cli("makedb", "-i", packed, "-d", db_b)
    assert r.returncode == 0, r.stderr
    assert open(db_a, "rb").read() == open(db_b, "rb").read()
    want = oracle.run_cli("count", "-i", plain).stdout.replace(plain, packed)
    assert cli("count", "-i", packed).stdout == want
    rows, err, usable = smafa_amd.load_fastx_part(packed, 0, 2)
    assert usable is False  # a compressed stream cannot be taken in byte ranges
    open(packed, "wb").write(blob[: len(blob) - 40])
    r = cli("makedb", "-i", packed, "-d", db_b)
    assert r.returncode == 101 and ("truncated" in r.stderr or "damaged" in r.stderr), r.stderr  # .expect(..): src/lib.rs:144
    assert cli("count", "-i", packed).returncode == 1  # `?`: src/lib.rs:381


def test_allocation_failure_is_an_error_code_not_an_exception_through_the_c_abi(tmp_path):
    """every int-returning entry point is a function-try-block: std::bad_alloc inside the call comes back as
    SMAFA_ERR_NOMEM with a message, the process lives, and the same call succeeds once memory is there.  (A subprocess,
    because the address-space limit that provokes the failure cannot be taken back.)"""
    import subprocess
    import sys

    prog = r"""
import ctypes as C, os, resource, sys
import numpy as np
sys.path.insert(0, %r)
from smafa_amd import _lib
lib = _lib.lib()
path = os.path.join(%r, "s.fna")
rng = np.random.default_rng(1)
rec = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(300000, 60))]
with open(path, "wb") as f:
    for i in range(0, len(rec), 10000):
        f.write(b"".join(b">r\n" + bytes(r) + b"\n" for r in rec[i:i + 10000]))
vm = int([l for l in open("/proc/self/status") if l.startswith("VmSize")][0].split()[1]) * 1024
out = []
for extra in (4 << 20, 1 << 30):
    resource.setrlimit(resource.RLIMIT_AS, (vm + extra, resource.RLIM_INFINITY))
    codes, n, L = C.c_void_p(), C.c_uint64(), C.c_uint32()
    rc = lib.smafa_fastx_load(path.encode(), 0, C.byref(codes), C.byref(n), C.byref(L))
    out.append((rc, n.value, lib.smafa_last_error().decode() if rc else ""))
print(out)
""" % (ROOT, str(tmp_path))
    r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    got = eval(r.stdout.strip().splitlines()[-1])
    assert got[0][0] == _lib.ERR_NOMEM == -7 and "out of host memory" in got[0][2] and got[0][1] == 0
    assert got[1][0] == 0 and got[1][1] == 300000
