"""BASELINE.json's full sizes, checked through size-independent properties (the oracle cannot brute-force
1e11..1e12 pairs in test time):

  * every reported row's distance is recomputed on the host from the code bytes and must match; rows are
    strictly ordered by (query, dist, subject) — the reference's print order;
  * every planted (query, subject, substitutions <= D) pair is present (recall of known answers);
  * completeness on a random sample of queries against the oracle's exhaustive scan;
  * batch-split invariance: scanning the query batch in two halves gives the same rows;
  * cluster: every member is within D of its centroid, centroids are pairwise > D apart in creation order
    (sampled), a record's centroid is the nearest earlier centroid with the lowest index on ties (sampled),
    and a 60k-record prefix run equals the oracle's sequential greedy loop record for record.
"""
import os
import subprocess

import numpy as np
import pytest

import oracle
import smafa_amd
from smafa_amd import _lib, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _built():
    smafa_amd.build()
    assert smafa_amd.device_count() >= 1


def check_rows(rows, subj, qry, D):
    d = (subj[rows["subject"]] != qry[rows["query"]]).sum(axis=1)
    assert (d == rows["dist"]).all()
    assert (rows["dist"] <= D).all()
    key = (rows["query"].astype(np.int64) << 40) | (rows["dist"].astype(np.int64) << 32) | rows["subject"].astype(np.int64)
    assert (np.diff(key) > 0).all(), "rows must be strictly ordered by (query, dist, subject)"


def check_planted(rows, planted_row, planted_subs, D):
    have = set(zip(rows["query"].tolist(), rows["subject"].tolist()))
    want = [(i, int(planted_row[i])) for i in range(len(planted_row)) if planted_subs[i] <= D]
    missing = [w for w in want if w not in have]
    assert not missing, missing[:5]
    return len(want)


def sample_vs_oracle(rows, subj, qry, D, rng, k=12):
    pick = np.sort(rng.choice(len(qry), size=k, replace=False))
    want = oracle.scan_codes(subj, qry[pick], D)
    want = want.copy()
    want["query"] = pick[want["query"]]
    got = rows[np.isin(rows["query"], pick)]
    assert got.tobytes() == want.tobytes()


@pytest.mark.parametrize("n,q,alphabet,D,max_subs,n_frac", [
    (1_000_000, 10_000, 1, 5, 10, 0.0),     # configs[1]: 1M x 60 aa, 10k queries, max-divergence 5
    (10_000_000, 10_000, 1, 5, 10, 0.0),    # the metric's point: 10M x 60 aa, d <= 5
    (10_000_000, 100_000, 0, 3, 6, 0.0),    # configs[2]: 10M x 60 nt, 100k queries, max-divergence 3 (N-free)
    (10_000_000, 20_000, 0, 3, 6, 0.001),   # configs[2] variant B: each column N with probability 0.001
])
def test_query_scan_full_size(n, q, alphabet, D, max_subs, n_frac):
    rng = np.random.default_rng(99)
    subj = synth.subjects(n, 60, alphabet, seed=1 if alphabet else 2, n_frac=n_frac)
    qry, planted_row, planted_subs = synth.queries(subj, q, alphabet, seed=3, max_subs=max_subs)
    store = smafa_amd.SubjectStore(60, alphabet)
    store.push(subj)
    rows = store.scan(qry, max_divergence=D)
    check_rows(rows, subj, qry, D)
    n_planted = check_planted(rows, planted_row, planted_subs, D)
    assert n_planted > q // 3
    sample_vs_oracle(rows, subj, qry, D, rng)
    # batch-split invariance
    half = q // 2
    a = store.scan(qry[:half], max_divergence=D)
    b = store.scan(qry[half:], max_divergence=D).copy()
    b["query"] += half
    assert np.concatenate([a, b]).tobytes() == rows.tobytes()
    # best-hit rows (k = 1, no max-divergence) on a slice: the minimum must be the planted distance or lower
    best = store.scan(qry[:200], max_divergence=None, max_num_hits=1)
    check_rows(best, subj, qry[:200], 60)
    first = {}
    for r in best:
        first.setdefault(int(r["query"]), int(r["dist"]))
    assert len(first) == 200 and all(first[i] <= planted_subs[i] for i in range(200))
    want_best = oracle.scan_codes(subj, qry[:4], 60)
    for i in range(4):
        grp = want_best[want_best["query"] == i]
        dmin = grp["dist"].min()
        assert best[best["query"] == i].tobytes() == grp[grp["dist"] == dmin].tobytes()
    if n == 10_000_000 and q <= 100_000 and n_frac == 0.0:
        kth_modes_full_size(store, subj, qry, alphabet)
    # the block index (smafa_db_build_index): the same call answered by D + 1 probes per query — the same bytes
    info = store.build_index(D)
    assert info["current"] == 1 and info["max_div_served"] == D and info["longest_run"] < 64, info
    assert store.scan(qry, max_divergence=D).tobytes() == rows.tobytes()
    assert "index_probe" in store.last_scan_kernel()
    if D > 1:
        lower = store.scan(qry, max_divergence=D - 2)
        assert "index_probe" in store.last_scan_kernel()
        assert lower.tobytes() == rows[rows["dist"] <= D - 2].tobytes()
    store.close()


def kth_modes_full_size(store, subj, qry, alphabet):
    """The K branch (src/lib.rs:242-295) at BASELINE size, no bound: k in {2, 5, 50} on 1 500 planted queries + 100 uniform-random
    ones (no near subject at all).  Every row's distance recomputed; the k-th rule's invariants for every query (at least k
    rows, fewer than k rows strictly below the largest reported distance, ordered by (query, dist, subject)); the complete
    row lists of 12 sampled queries == the oracle's 10M distances put through the rule (bench.expected_with_k)."""
    import bench

    far = synth.subjects(100, 60, alphabet, seed=9, dup_frac=0.0)
    kq = np.concatenate([qry[:1500], far])
    pick = np.concatenate([np.arange(0, 1500, 150), [1500, 1599]])  # 10 planted + 2 far
    dist = {int(qi): oracle.distances_codes(subj, kq[qi]) for qi in pick}
    for k in (2, 5, 50):
        rows = store.scan(kq, max_divergence=None, max_num_hits=k)
        check_rows(rows, subj, kq, 60)
        r3 = bench.rows3(rows)
        assert bench.kth_rows_ok(np, subj, kq, r3, k, None, [])  # the invariants, every query
        for qi in pick:
            want = bench.expected_with_k(np, dist[int(qi)], k, None)
            assert r3[r3[:, 0] == qi][:, 1:].tobytes() == want.tobytes(), (k, int(qi))
    # with a bound the K branch prints only what is within both (src/lib.rs:261-264)
    rows = store.scan(kq, max_divergence=4, max_num_hits=5)
    r3 = bench.rows3(rows)
    assert bench.kth_rows_ok(np, subj, kq, r3, 5, 4, [])
    for qi in pick:
        assert r3[r3[:, 0] == qi][:, 1:].tobytes() == bench.expected_with_k(np, dist[int(qi)], 5, 4).tobytes()


def test_query_scan_50m_store_one_rank_share():
    """configs[3]: 50M x 60 aa store replicated on 8 GPUs, 1M queries sharded — this is ONE rank's share:
    the whole 50M store (2 GB of bit-planes) and a contiguous block of 125 000 queries"""
    n, q_total, world, rank, D = 50_000_000, 1_000_000, 8, 3, 5
    subj = synth.subjects(n, 60, 1, seed=1)
    rng = np.random.default_rng(17)
    from smafa_amd import dist as sdist
    lo, hi = sdist.shard_bounds(q_total, world, rank)
    assert hi - lo == 125_000
    # the rank's block of a 1M-query batch: planted queries drawn with the block's own seed
    qry, planted_row, planted_subs = synth.queries(subj, hi - lo, 1, seed=1000 + rank, max_subs=10)
    store = smafa_amd.SubjectStore(60, 1)
    store.push(subj)
    assert store.info().hbm_bytes == (n + 255) // 256 * 256 * 40
    rows = store.scan(qry, max_divergence=D)
    check_rows(rows, subj, qry, D)
    assert check_planted(rows, planted_row, planted_subs, D) > 60_000
    sample_vs_oracle(rows, subj, qry, D, rng, k=4)
    assert store.build_index(D)["max_div_served"] == D  # 6 blocks x 50M (key, position) pairs: 2.4 GB
    assert store.scan(qry, max_divergence=D).tobytes() == rows.tobytes()
    assert "index_probe" in store.last_scan_kernel()
    store.close()


def rows_as_void(a):
    a = np.ascontiguousarray(a)
    return a.view(np.dtype((np.void, a.shape[1]))).ravel()


def run_cluster_cli(path, D, alphabet="aa"):
    r = subprocess.run([_lib.CLI_PATH, "cluster", "-i", path, "-d", str(D), "--alphabet", alphabet],
                       capture_output=True)
    assert r.returncode == 0, r.stderr[-500:]
    return r.stdout


def test_cluster_prefix_equals_oracle_sequential(tmp_path):
    recs = synth.cluster_records(2_000, 30, 60, 1, seed=4, max_subs=4)  # 60k records, ~8k centroids
    f = str(tmp_path / "c.faa")
    synth.write_fasta(f, recs, 1)
    out = run_cluster_cli(f, 5)
    assigned = oracle.cluster_codes(recs, 5, oracle.ALPHABET_AA)
    letters = np.array([ord("A") + i for i in range(26)] + [ord("*"), ord("-")], dtype=np.uint8)
    ascii_rows = letters[recs]
    cents, lines = [], []
    for i, a in enumerate(assigned):
        if a == 0xFFFFFFFF:
            continue
        if a == len(cents):
            cents.append(ascii_rows[i].tobytes())
        lines.append(ascii_rows[i].tobytes() + b"\t" + cents[a] + b"\n")
    assert out == b"".join(lines)
    assert len(cents) > 2000


def test_cluster_full_size_properties(tmp_path):
    """configs[4]: 5M x 60 aa (100k roots x 50 members, 0..4 substitutions, shuffled), max-divergence 5"""
    D = 5
    recs = synth.cluster_records(100_000, 50, 60, 1, seed=4, max_subs=4)
    f = str(tmp_path / "c.faa")
    synth.write_fasta(f, recs, 1)
    out = run_cluster_cli(f, D)
    del_letters = np.full(256, 255, dtype=np.uint8)
    for c in range(26):
        del_letters[ord("A") + c] = c
    del_letters[ord("*")], del_letters[ord("-")] = 26, 27
    raw = np.frombuffer(out, dtype=np.uint8).reshape(-1, 122)  # 60 + tab + 60 + newline
    assert (raw[:, 60] == 9).all() and (raw[:, 121] == 10).all()
    member = del_letters[raw[:, :60]]
    centroid = del_letters[raw[:, 61:121]]
    # duplicates are skipped: output rows = first occurrences, in input order
    _, first_idx = np.unique(rows_as_void(recs), return_index=True)
    first_idx.sort()
    assert len(member) == len(first_idx)
    assert (member == recs[first_idx]).all()
    # (a) every member within D of its centroid
    d = (member != centroid).sum(axis=1)
    assert d.max() <= D
    # centroids in creation order = rows where member == centroid
    is_cent = d == 0
    cent_rows = member[is_cent]
    cent_pos = np.nonzero(is_cent)[0]
    assert len(np.unique(rows_as_void(cent_rows))) == len(cent_rows)
    # every distinct centroid string in column 2 is one of those rows
    assert len(np.unique(rows_as_void(centroid))) == len(cent_rows)
    rng = np.random.default_rng(5)
    # (b) sampled centroids are > D from every EARLIER centroid
    for ci in rng.choice(len(cent_rows), size=60, replace=False):
        if ci == 0:
            continue
        dd = (cent_rows[:ci] != cent_rows[ci]).sum(axis=1)
        assert dd.min() > D
    # (c) sampled members: centroid = nearest among centroids created before the member, lowest index on ties
    for ri in rng.choice(len(member), size=60, replace=False):
        n_before = int(np.searchsorted(cent_pos, ri, side="left"))  # centroids created strictly before row ri
        if is_cent[ri]:
            if n_before:
                assert ((cent_rows[:n_before] != member[ri]).sum(axis=1)).min() > D
            continue
        dd = (cent_rows[:n_before] != member[ri]).sum(axis=1)
        best = int(np.argmin(dd))  # argmin returns the first (lowest index) minimum
        assert dd[best] <= D
        assert (cent_rows[best] == centroid[ri]).all()
