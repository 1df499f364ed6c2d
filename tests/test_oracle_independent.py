"""A second, independently structured restatement of the reference's arithmetic, in Python big integers, checked
against the C oracle at MULTI-WORD lengths (the reference's own golden vectors stop at 9 columns = one u64).

The two restatements share an author but nothing else: the oracle packs u64 words in C loops; this file builds ONE Python
integer per sequence, derives the u64 words from it, counts bits with int.bit_count-style arithmetic, writes postcard's
LEB128 by hand, and cross-checks both against the *meaning* of the arithmetic (the number of columns whose collapsed
symbols differ).  Reference text followed: src/lib.rs:29-52 (from_bytes: 12 symbols per u64, symbol i of a chunk at bit 5*i),
:71-89 (sum of popcounts of the XORed words, halved once at the end), :171-178 (the byte classes), :161-162 + serde's
derive order for WindowSet {version, windows, len} (the DB bytes)."""
import numpy as np
import pytest

import oracle

ONE_HOT = {}
for letters, code in (("Aa", 16), ("Cc", 8), ("Gg", 4), ("TtUu", 2), ("NnWwSsMmKkRrYyBbDdHhVv-", 1)):  # src/lib.rs:171-178
    for ch in letters:
        ONE_HOT[ord(ch)] = code
COLLAPSED = {16: "A", 8: "C", 4: "G", 2: "T", 1: "N"}


def big_int_of(seq: bytes) -> int:
    """the whole sequence as one integer: symbol i lives at bit 64 * (i // 12) + 5 * (i % 12)"""
    v = 0
    for i, b in enumerate(seq):
        v |= ONE_HOT[b] << (64 * (i // 12) + 5 * (i % 12))
    return v


def words_of(seq: bytes):
    v, n = big_int_of(seq), (len(seq) + 11) // 12
    return [(v >> (64 * w)) & 0xFFFFFFFFFFFFFFFF for w in range(n)]


def distance(a: bytes, b: bytes) -> int:
    return bin(big_int_of(a) ^ big_int_of(b)).count("1") // 2  # popcounts of the words add up to the popcount of the whole


def leb128(v: int) -> bytes:
    out = bytearray()
    while True:
        byte, v = v & 0x7F, v >> 7
        out.append(byte | (0x80 if v else 0))
        if not v:
            return bytes(out)


def db_bytes(seqs) -> bytes:
    """postcard of WindowSet { version: u32 = 2, windows: Vec<SeqEncoding(Vec<u64>)>, len: Option<NonZeroUsize> }"""
    out = bytearray(leb128(2) + leb128(len(seqs)))
    for s in seqs:
        w = words_of(s)
        out += leb128(len(w))
        for x in w:
            out += leb128(x)
    out += (b"\x01" + leb128(len(seqs[0]))) if seqs else b"\x00"
    return bytes(out)


ALPHABET = np.frombuffer(b"ACGTNacgtuURYKMSWBDHV-", dtype=np.uint8)


def random_rows(rng, n, L):
    return ALPHABET[rng.integers(0, len(ALPHABET), size=(n, L))]


@pytest.mark.parametrize("L", [1, 11, 12, 13, 24, 25, 48, 59, 60, 61, 72, 120, 121, 255])
def test_words_and_distances_at_multiword_lengths(L):
    rng = np.random.default_rng(L)
    subj, qry = random_rows(rng, 120, L), random_rows(rng, 9, L)
    subj[7] = qry[0]  # an exact copy after collapsing is distance 0
    for row in list(subj[:20]) + list(qry):
        assert oracle.encode_onehot(row.tobytes()) == words_of(row.tobytes())
    got = oracle.scan_onehot(subj, qry, L)  # every pair: (query, subject, dist), ordered (query, dist, subject)
    assert len(got) == len(subj) * len(qry)
    want = {}
    for qi, q in enumerate(qry):
        for sj, s_ in enumerate(subj):
            d = distance(q.tobytes(), s_.tobytes())
            # the meaning of that arithmetic: columns whose collapsed symbols differ
            assert d == sum(COLLAPSED[ONE_HOT[x]] != COLLAPSED[ONE_HOT[y]] for x, y in zip(q.tobytes(), s_.tobytes()))
            want[(qi, sj)] = d
    assert {(int(r["query"]), int(r["subject"])): int(r["dist"]) for r in got} == want
    keys = [(int(r["query"]), int(r["dist"]), int(r["subject"])) for r in got]
    assert keys == sorted(keys)


@pytest.mark.parametrize("L", [3, 12, 13, 60, 61])
def test_db_bytes_at_multiword_lengths(tmp_path, L):
    rng = np.random.default_rng(100 + L)
    rows = random_rows(rng, 37, L)
    fa = str(tmp_path / "s.fna")
    oracle.write_fasta(fa, [r.tobytes() for r in rows])
    db = str(tmp_path / "s.db")
    r = oracle.run_cli("makedb", "-i", fa, "-d", db)
    assert r.returncode == 0, r.stderr
    assert open(db, "rb").read() == db_bytes([r_.tobytes() for r_ in rows])


def test_the_two_reference_db_images(golden):
    """the hand-written serialiser reproduces the reference's own files (tests/data/*.smafadb) — so it is itself pinned"""
    import os

    assert db_bytes([b"CTT", b"AGG"]) == open(os.path.join(golden, "random_3_2.fna.smafadb"), "rb").read()
    assert db_bytes([b"CTT", b"AGG", b"AGG"]) == open(os.path.join(golden, "random_3_2_one_repeated.fna.smafadb"), "rb").read()


def test_query_selection_at_60_columns_against_a_python_restatement(tmp_path):
    """`query` end to end at 60 columns (5 words per sequence), every mode, against the selection rules written out in Python
    (src/lib.rs:241-315): k-th smallest with ties, k == 1 == default, k > N, inclusive max-divergence, limit-per-sequence on
    adjacent equal strings"""
    rng = np.random.default_rng(9)
    L, n, nq = 60, 150, 25
    base = np.frombuffer(b"ACGT", dtype=np.uint8)
    subj = base[rng.integers(0, 4, size=(n, L))]
    subj[10:14] = subj[9]          # equal strings: ties and limit-per-sequence
    subj[20, 5] = ord("N")
    qry = subj[rng.integers(0, n, size=nq)].copy()
    for r in qry:
        for _ in range(rng.integers(0, 7)):
            r[rng.integers(0, L)] = base[rng.integers(0, 4)]
    sf, qf, db = str(tmp_path / "s.fna"), str(tmp_path / "q.fna"), str(tmp_path / "s.db")
    oracle.write_fasta(sf, [r.tobytes() for r in subj])
    oracle.write_fasta(qf, [r.tobytes() for r in qry])
    assert oracle.run_cli("makedb", "-i", sf, "-d", db).returncode == 0
    collapsed = ["".join(COLLAPSED[ONE_HOT[b]] for b in r.tobytes()) for r in subj]

    def expected(max_div, k, limit):
        lines = []
        for qi, q in enumerate(qry):
            d = [distance(q.tobytes(), s_.tobytes()) for s_ in subj]
            if k is None or k == 1:  # :296-313
                m = min(d)
                if max_div is None or m <= max_div:
                    lines += ["%d\t%d\t%d\t%s" % (qi, j, m, collapsed[j]) for j in range(n) if d[j] == m]
                continue
            order = sorted((d[j], j) for j in range(n))  # :243-250
            thr = max(d) if k > n else order[k - 1][0]  # :253-256
            last, count = None, 0
            for dist, j in order:
                if dist > thr or (max_div is not None and dist > max_div):
                    continue
                if limit is not None:  # :269-289
                    if collapsed[j] == last:
                        if count >= limit:
                            continue
                        count += 1
                    else:
                        last, count = collapsed[j], 1
                lines.append("%d\t%d\t%d\t%s" % (qi, j, dist, collapsed[j]))
        return "".join(x + "\n" for x in lines)

    for max_div, k, limit in ((None, None, None), (4, None, None), (None, 1, None), (None, 3, None), (5, 3, None), (None, 500, None),
                              (2, 500, None), (None, 6, 1), (3, 6, 2)):
        flags = []
        if max_div is not None:
            flags += ["--max-divergence", str(max_div)]
        if k is not None:
            flags += ["--max-num-hits", str(k)]
        if limit is not None:
            flags += ["--limit-per-sequence", str(limit)]
        r = oracle.run_cli("query", "-d", db, "-q", qf, *flags)
        assert r.returncode == 0, r.stderr
        assert r.stdout == expected(max_div, k, limit), flags


def test_cluster_at_60_columns_against_a_python_restatement(tmp_path):
    """`cluster` at 60 columns against the loop of src/cluster.rs:35-85 written out in Python over big integers: duplicates
    (after collapsing) skipped silently, distances to the centroids so far, first minimum wins, a new centroid otherwise; column 1
    is the RAW record (case and IUPAC letters kept), column 2 the collapsed centroid"""
    rng = np.random.default_rng(31)
    L, roots, members = 60, 25, 12
    base = np.frombuffer(b"ACGT", dtype=np.uint8)
    r = base[rng.integers(0, 4, size=(roots, L))]
    recs = np.repeat(r, members, axis=0)
    for row in recs:
        for _ in range(rng.integers(0, 5)):
            row[rng.integers(0, L)] = ALPHABET[rng.integers(0, len(ALPHABET))]
    recs = recs[rng.permutation(len(recs))]
    recs[40:46] = recs[3]                      # exact duplicates
    recs[50] = np.frombuffer(recs[3].tobytes().lower(), dtype=np.uint8)  # the same sequence after collapsing: also a duplicate
    path = str(tmp_path / "c.fna")
    oracle.write_fasta(path, [x.tobytes() for x in recs])
    for D in (0, 2, 5, 9):
        seen, centroids, lines = set(), [], []
        for x in recs:
            raw = x.tobytes()
            key = big_int_of(raw)
            if key in seen:
                continue
            seen.add(key)
            d = [distance(raw, c) for c in centroids]
            m = min(d) if d else 2 * D + 2
            if m <= D:
                assigned = d.index(m)
            else:
                assigned = len(centroids)
                centroids.append(raw)
            lines.append(raw.decode() + "\t" + "".join(COLLAPSED[ONE_HOT[b]] for b in centroids[assigned]) + "\n")
        got = oracle.run_cli("cluster", "-i", path, "-d", str(D))
        assert got.returncode == 0, got.stderr
        assert got.stdout == "".join(lines), D
