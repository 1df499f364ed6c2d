"""Pins the CPU oracle against every live golden vector of the reference.

Each case cites the reference test it restates (file:line relative to /root/reference);
the fixture files under tests/golden/ are the reference's own tests/data files, the
expected strings are the reference's expected outputs.
"""
import os

import numpy as np
import pytest

import oracle

T = "\t"


def rows(*r):
    return "".join(T.join(map(str, x)) + "\n" for x in r)


def run_ok(*args):
    r = oracle.run_cli(*args)
    assert r.returncode == 0, r.stderr
    return r.stdout


# ---------------------------------------------------------------- encoding / makedb
def test_lut_classes():
    # src/lib.rs:171-178
    expect = {}
    for ch in "Aa":
        expect[ord(ch)] = 0b10000
    for ch in "Cc":
        expect[ord(ch)] = 0b01000
    for ch in "Gg":
        expect[ord(ch)] = 0b00100
    for ch in "TtUu":
        expect[ord(ch)] = 0b00010
    for ch in "NWSMKRYBDHV-nwsmkrybdhv":
        expect[ord(ch)] = 0b00001
    for b in range(256):
        assert oracle.lut_nt(b) == expect.get(b, 0), b


def test_makedb_subjects_unit(golden, tmp_path):
    # src/lib.rs:334-367 test_makedb: 5 windows [[16],[8],[4],[2],[1]]
    db = tmp_path / "s.db"
    run_ok("makedb", "-i", os.path.join(golden, "subjects.fa"), "-d", str(db))
    assert db.read_bytes() == bytes([2, 5, 1, 16, 1, 8, 1, 4, 1, 2, 1, 1, 1, 1])
    assert [oracle.encode_onehot(c) for c in (b"A", b"C", b"G", b"T", b"N")] == [[16], [8], [4], [2], [1]]


@pytest.mark.parametrize("name", ["random_3_2.fna", "random_3_2_one_repeated.fna"])
def test_makedb_bytes_match_reference_db(golden, tmp_path, name):
    # DB byte images shipped in tests/data (SURVEY.md §4 last row)
    db = tmp_path / "t.db"
    run_ok("makedb", "-i", os.path.join(golden, name), "-d", str(db))
    assert db.read_bytes() == open(os.path.join(golden, name + ".smafadb"), "rb").read()


def test_encode_multiword_and_error():
    # 13 symbols -> 2 words; symbol i at bit 5*(i%12) of word i/12 (src/lib.rs:32,44)
    enc = oracle.encode_onehot(b"ACGTNACGTNACG")
    assert len(enc) == 2
    assert enc[1] == 0b00100
    assert enc[0] & 31 == 16 and (enc[0] >> 55) & 31 == 8
    with pytest.raises(ValueError) as e:
        oracle.encode_onehot(b"ACGTE")
    assert e.value.args[0] == 4


def test_bad_byte_message(tmp_path):
    # src/lib.rs:38-41
    f = tmp_path / "bad.fna"
    f.write_bytes(b">seq one\nACGTACGTACGTAE\n")
    r = oracle.run_cli("makedb", "-i", str(f), "-d", str(tmp_path / "x.db"))
    assert r.returncode != 0
    assert 'Byte 69 cannot be interpreted as nucleotide, in sequence "seq one" at position 13' in r.stderr


def test_unequal_lengths_message(tmp_path):
    # src/lib.rs:92-101
    f = tmp_path / "ragged.fna"
    f.write_bytes(b">a\nACGT\n>b\nACG\n")
    r = oracle.run_cli("makedb", "-i", str(f), "-d", str(tmp_path / "x.db"))
    assert r.returncode != 0
    assert "WindowSet seq length is 4, got a new sequence of length 3" in r.stderr


# ------------------------------------------------------------------------- query
def test_dna_makedb_and_query(golden, tmp_path):
    # tests/test_cmdline.rs:10-25
    db = str(tmp_path / "t.db")
    q = os.path.join(golden, "random_3_2.fna")
    run_ok("makedb", "-i", q, "-d", db)
    assert run_ok("query", "-d", db, "-q", q) == rows((0, 0, 0, "CTT"), (1, 1, 0, "AGG"))


def test_old_db_version(golden):
    # tests/test_cmdline.rs:28-41
    r = oracle.run_cli("query", "-d", os.path.join(golden, "random_3_2.fna.v1.smafadb"), "-q",
                       os.path.join(golden, "random_3_2.fna"))
    assert r.returncode != 0
    assert "Unsupported db file version: 1." in r.stderr


def test_degenerate_makedb_and_query(golden, tmp_path):
    # tests/test_cmdline.rs:44-74
    db = str(tmp_path / "t.db")
    f = os.path.join(golden, "degenerate.fna")
    run_ok("makedb", "-i", f, "-d", db)
    assert run_ok("query", "-d", db, "-q", f, "--max-num-hits", "99") == rows(
        (0, 0, 0, "CTTNGG"), (0, 1, 5, "AGGTGA"), (0, 2, 6, "NACTTT"),
        (1, 1, 0, "AGGTGA"), (1, 0, 5, "CTTNGG"), (1, 2, 5, "NACTTT"),
        (2, 2, 0, "NACTTT"), (2, 1, 5, "AGGTGA"), (2, 0, 6, "CTTNGG"))


FOUR = rows((0, 0, 0, "CTT"), (0, 1, 3, "AGG"), (1, 1, 0, "AGG"), (1, 0, 3, "CTT"))
TWO = rows((0, 0, 0, "CTT"), (1, 1, 0, "AGG"))


@pytest.mark.parametrize("flags,expected", [
    (["--max-divergence", "99", "--max-num-hits", "99"], FOUR),  # test_cmdline.rs:77-97
    (["--max-divergence", "2", "--max-num-hits", "99"], TWO),    # :100-118
    (["--max-divergence", "3", "--max-num-hits", "99"], FOUR),   # :121-141 (<= is inclusive)
    (["--max-num-hits", "1"], TWO),                              # :144-160 (k==1 == default)
    (["--max-num-hits", "99"], FOUR),                            # :163-181 (k > N => all)
])
def test_query_prebuilt_db(golden, flags, expected):
    out = run_ok("query", "-d", os.path.join(golden, "random_3_2.fna.smafadb"), "-q",
                 os.path.join(golden, "random_3_2.fna"), *flags)
    assert out == expected


def test_limit_per_sequence(golden):
    # tests/test_cmdline.rs:204-247
    db = os.path.join(golden, "random_3_2_one_repeated.fna.smafadb")
    q = os.path.join(golden, "random_3_2.fna")
    assert run_ok("query", "-d", db, "-q", q, "--max-num-hits", "99") == rows(
        (0, 0, 0, "CTT"), (0, 1, 3, "AGG"), (0, 2, 3, "AGG"), (1, 1, 0, "AGG"), (1, 2, 0, "AGG"), (1, 0, 3, "CTT"))
    assert run_ok("query", "-d", db, "-q", q, "--max-num-hits", "99", "--limit-per-sequence", "1") == FOUR


def test_limit_per_sequence_besthit_panics(golden):
    # src/lib.rs:301-303
    r = oracle.run_cli("query", "-d", os.path.join(golden, "random_3_2.fna.smafadb"), "-q",
                       os.path.join(golden, "random_3_2.fna"), "--limit-per-sequence", "1")
    assert r.returncode != 0 and "limit_per_sequence is implemented unless max_num_hits > 1" in r.stderr


# ------------------------------------------------------------------------- count
def test_counts(golden):
    # tests/test_cmdline.rs:184-201 (also pins gzip FASTQ parsing)
    for name, reads, bases in (("random_3_2.fna", 2, 6), ("random_30_4.fq.gz", 4, 120)):
        p = os.path.join(golden, name)
        assert run_ok("count", "-i", p) == '[{"path":"%s","num_reads":%d,"num_bases":%d}]\n' % (p, reads, bases)


# ----------------------------------------------------------------------- cluster
def test_cluster_simple(golden):
    # src/cluster.rs:102-112
    assert run_ok("cluster", "-i", os.path.join(golden, "cluster_dummy1.fna"), "-d", "1") == \
        "ATGC\tATGC\nATGG\tATGC\nAAAA\tAAAA\n"


@pytest.mark.parametrize("name", ["cluster_bug1.fna", "cluster_best_hit_changes.fna"])
def test_cluster_bugs(golden, name):
    # src/cluster.rs:115-143 (centroids only; exact duplicates emit nothing)
    assert run_ok("cluster", "-i", os.path.join(golden, name), "-d", "2") == \
        "ATGCAAAAA\tATGCAAAAA\nATAAAAAAA\tATGCAAAAA\nTTAAAAAAA\tTTAAAAAAA\n"


# ------------------------------------------- cross-pin: code-byte path == one-hot path
def test_codes_scan_equals_onehot_scan_on_nucleotides():
    rng = np.random.default_rng(7)
    letters = np.frombuffer(b"ACGTUNRYacgtn-", dtype=np.uint8)
    for L in (1, 11, 12, 13, 60, 64, 65):
        subj = letters[rng.integers(0, len(letters), size=(300, L))]
        subj[17] = subj[3]
        qry = np.concatenate([subj[:20].copy(), letters[rng.integers(0, len(letters), size=(10, L))]])
        qry[1, 0] = ord("A") if qry[1, 0] != ord("A") else ord("C")
        for D in (0, 2, L):
            a = oracle.scan_onehot(subj, qry, D)
            b = oracle.scan_codes(oracle.codes_from_ascii(subj, oracle.ALPHABET_NT),
                                  oracle.codes_from_ascii(qry, oracle.ALPHABET_NT), D)
            assert a.tobytes() == b.tobytes(), (L, D)
            assert len(a) >= 20 if D == 0 else True


def test_cluster_codes_equals_cluster_cli():
    rng = np.random.default_rng(11)
    roots = rng.integers(0, 4, size=(12, 20))
    recs = []
    for i in range(200):
        r = roots[rng.integers(0, 12)].copy()
        for _ in range(rng.integers(0, 4)):
            r[rng.integers(0, 20)] = rng.integers(0, 4)
        recs.append(r)
    codes = np.array(recs, dtype=np.uint8)
    ascii_rows = np.frombuffer(b"ACGT", dtype=np.uint8)[codes]
    text = oracle.cluster_text([bytes(r) for r in ascii_rows], 3)
    assigned = oracle.cluster_codes(codes, 3)
    # rebuild the CLI text from the assignment vector
    cents, lines = [], []
    for i, a in enumerate(assigned):
        if a == 0xFFFFFFFF:
            continue
        if a == len(cents):
            cents.append(bytes(ascii_rows[i]))
        lines.append(bytes(ascii_rows[i]) + b"\t" + cents[a] + b"\n")
    assert text.encode() == b"".join(lines)
