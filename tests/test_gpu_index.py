"""GPU parity of the block index (smafa_db_build_index, smafa_amd/csrc/index.hip.h): a fixed tight bound answered by
bound + 1 probes per query must give the rows of the scan kernels and of the oracle, byte for byte — whatever the store
looks like (duplicates, families, letters the store has never seen, several appends) — and must step aside by itself
where the store's blocks are not selective or the index is stale.  Everything through the C ABI."""
import os
import subprocess

import numpy as np
import pytest

import oracle
import smafa_amd
from smafa_amd import _lib
from test_gpu_layout import expected_with_k, queries_from, skewed_store

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _built():
    smafa_amd.build()
    assert smafa_amd.device_count() >= 1


@pytest.fixture
def index_env():
    """SMAFA_INDEX* are read when a handle is created"""
    keys = ("SMAFA_INDEX", "SMAFA_INDEX_MAX_RUN", "SMAFA_INDEX_CAND", "SMAFA_NT_PLANES")
    old = {k: os.environ.get(k) for k in keys}
    yield os.environ
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def probe_name(store):
    info = store.info()
    pq = 5 if store.alphabet == 1 else 3
    return "smafa::index_probe_kernel<%d, %d, %d>" % (info.planes, pq, info.words_per_plane)


@pytest.mark.parametrize("alphabet,n_letters,L", [(1, 20, 60), (1, 25, 20), (1, 20, 100), (1, 20, 128), (0, 4, 60), (0, 5, 60),
                                                   (0, 4, 31), (0, 5, 97)])
def test_index_rows_equal_scan_rows_and_oracle(alphabet, n_letters, L):
    rng = np.random.default_rng(1000 * alphabet + 10 * L + n_letters)
    n = 70000
    s = rng.integers(0, n_letters, size=(n, L), dtype=np.uint8)
    s[5000:5040] = s[4999]  # identical subjects: one run of 41 equal keys in every block
    q = np.concatenate([queries_from(rng, s, 250, n_letters, 7), rng.integers(0, n_letters, size=(50, L), dtype=np.uint8),
                        s[4999:5000]])
    store = smafa_amd.SubjectStore(L, alphabet)
    store.push(s)
    info = store.build_index(6)
    assert info["current"] == 1 and info["blocks"] == 7 and info["longest_run"] >= 41
    # short rows have short blocks: few distinct keys, long runs — the index then serves only the bounds whose probes stay cheap
    served = info["max_div_served"]
    if L >= 60:
        assert served == 6 and info["usable_blocks"] == 7, info
    probes = 0
    for D in (0, 3, 5, 6):
        want = oracle.scan_codes(s, q, D)
        got = store.scan(q, max_divergence=D)
        by_index = store.last_scan_kernel() == probe_name(store)
        if L >= 60:  # (short rows: which bounds the short blocks serve depends on the scan kernels' form at that bound)
            assert by_index, (D, served, store.last_scan_kernel())
        assert by_index or "index_probe" not in store.last_scan_kernel()
        assert got.tobytes() == want.tobytes(), (alphabet, L, D)
        # --max-num-hits with a bound: the fixed-bound rows first, selection afterwards (collect_range)
        got_k = store.scan(q, max_divergence=D, max_num_hits=2)
        assert got_k.tobytes() == expected_with_k(want, 2).tobytes()
        probes += 1 if by_index else 0
    # without a bound (`smafa query`'s default, best hit; the K branch): the index answers the ladder's first step — every pair
    # within the largest bound it serves — and the scan kernels take the queries that step leaves open
    few = q[:90]
    full = oracle.scan_codes(s, few, L)
    first_step = (min(32, L) - 1) // 6  # (the index takes the ladder's first step only where it serves at least that bound)
    for k in (1, 3):
        before = store.index_info()["probe_launches"]
        got_k = store.scan(few, max_num_hits=k)
        assert got_k.tobytes() == expected_with_k(full, k).tobytes(), (alphabet, L, k)
        if L >= 60:
            assert store.index_info()["probe_launches"] > before
        elif served is None or served < first_step:
            assert store.index_info()["probe_launches"] == before
    # bounds past the index, a handful of queries, the prefilter-off audit launch and mode 0: the scan kernels
    assert store.scan(q, max_divergence=7).tobytes() == oracle.scan_codes(s, q, 7).tobytes()
    assert "index_probe" not in store.last_scan_kernel()
    assert store.scan(q[:40], max_divergence=3).tobytes() == oracle.scan_codes(s, q[:40], 3).tobytes()
    assert "index_probe" not in store.last_scan_kernel()
    store.set_prefilter(False)
    assert store.scan(q, max_divergence=3).tobytes() == oracle.scan_codes(s, q, 3).tobytes()
    assert "index_probe" not in store.last_scan_kernel()
    store.set_prefilter(True)
    store.set_index(0)
    assert store.scan(q, max_divergence=3).tobytes() == oracle.scan_codes(s, q, 3).tobytes()
    assert "index_probe" not in store.last_scan_kernel()
    store.set_index(1)
    assert store.index_info()["probe_launches"] >= probes
    store.close()


def test_stale_index_steps_aside_and_rebuilds(index_env):
    rng = np.random.default_rng(77)
    s = rng.integers(0, 20, size=(50000, 60), dtype=np.uint8)
    more = rng.integers(0, 20, size=(30000, 60), dtype=np.uint8)
    q = queries_from(rng, np.concatenate([s, more]), 300, 20, 6)
    store = smafa_amd.SubjectStore(60, 1)
    store.push(s)
    store.build_index(5)
    assert store.scan(q, max_divergence=5).tobytes() == oracle.scan_codes(s, q, 5).tobytes()
    assert "index_probe" in store.last_scan_kernel()
    store.push(more)  # the positions change (an append, then the re-sort the quarter rule asks for): the index is stale
    both = np.concatenate([s, more])
    assert store.index_info()["current"] == 0
    assert store.scan(q, max_divergence=5).tobytes() == oracle.scan_codes(both, q, 5).tobytes()
    assert "index_probe" not in store.last_scan_kernel()
    assert store.build_index(5)["current"] == 1
    assert store.scan(q, max_divergence=5).tobytes() == oracle.scan_codes(both, q, 5).tobytes()
    assert "index_probe" in store.last_scan_kernel()
    store.drop_index()
    assert store.index_info()["current"] == 0
    assert store.scan(q, max_divergence=5).tobytes() == oracle.scan_codes(both, q, 5).tobytes()
    assert "index_probe" not in store.last_scan_kernel()
    store.close()
    # mode 2: the first big fixed-bound scan builds it (stores of >= 65 536 subjects)
    index_env["SMAFA_INDEX"] = "2"
    store = smafa_amd.SubjectStore(60, 1)
    store.push(both)
    assert store.scan(q, max_divergence=4).tobytes() == oracle.scan_codes(both, q, 4).tobytes()
    assert "index_probe" in store.last_scan_kernel() and store.index_info()["blocks"] == 5
    assert store.scan(q, max_divergence=5).tobytes() == oracle.scan_codes(both, q, 5).tobytes()  # a wider bound: built again
    assert "index_probe" in store.last_scan_kernel() and store.index_info()["blocks"] == 6
    assert store.scan(q, max_divergence=2).tobytes() == oracle.scan_codes(both, q, 2).tobytes()  # served by the six blocks
    assert "index_probe" in store.last_scan_kernel() and store.index_info()["blocks"] == 6
    store.close()
    # mode 3: rent or buy — the scan kernels until their (estimated) cost since the last change of the store reaches the build's
    index_env["SMAFA_INDEX"] = "3"
    store = smafa_amd.SubjectStore(60, 1)
    store.push(both)
    want = oracle.scan_codes(both, q, 5).tobytes()
    launches = 0
    while "index_probe" not in store.last_scan_kernel():
        assert store.scan(q, max_divergence=5).tobytes() == want
        launches += 1
        assert launches < 4000
    # 300 queries x 80 000 subjects x 10 vectors x 1.7e-12 ms = 4.1e-4 ms per scan against 0.3 + 6 x 80 000 x 1e-7 = 0.348 ms
    assert 800 < launches < 900, launches
    store.close()


def test_rent_or_buy_without_a_bound(index_env):
    """mode 3 (what `smafa query` sets) for the reference's default mode — best hit, no bound: the ladder's scans are charged at the
    scan kernels' rate and an index for the ladder's first steps is built once they have cost what it would; rows never change"""
    index_env["SMAFA_INDEX"] = "3"
    index_env["SMAFA_INDEX_MIN_ROWS"] = "1"
    rng = np.random.default_rng(5)
    s = rng.integers(0, 20, size=(80000, 60), dtype=np.uint8)
    q = queries_from(rng, s, 300, 20, 9)
    full = oracle.scan_codes(s, q, 60)
    want = {k: expected_with_k(full, k).tobytes() for k in (1, 2)}
    store = smafa_amd.SubjectStore(60, 1)
    store.push(s)
    calls = 0
    while store.index_info()["probe_launches"] == 0:
        k = 1 + calls % 2
        assert store.scan(q, max_num_hits=k).tobytes() == want[k], (calls, k)
        calls += 1
        assert calls < 400
    # 300 queries x 80 000 subjects x 10 vectors x 1.6e-11 ms = 3.8e-3 ms per call against 0.3 + 20 blocks x 80 000 x 1e-7 = 0.46 ms
    info = store.index_info()
    assert 110 < calls < 130 and info["blocks"] == 20 and info["current"] == 1, (calls, info)
    for k in (1, 2):
        assert store.scan(q, max_num_hits=k).tobytes() == want[k]
    assert store.index_info()["probe_launches"] >= 3
    store.close()


def test_dense_and_low_complexity_stores_are_left_to_the_scan_kernels_unless_forced(index_env):
    rng = np.random.default_rng(99)
    L = 60
    # families of near-identical members, the last 24 columns gap-only: long runs of equal keys
    s = skewed_store(rng, 80000, L, 20, conserved_frac=0.3, families=60)
    s[:, 36:60] = 27
    q = queries_from(rng, s, 200, 20, 5)
    want = oracle.scan_codes(s, q, 4)
    store = smafa_amd.SubjectStore(L, 1)
    store.push(s)
    info = store.build_index(4)
    # the two gap-only blocks are never probed (one run of 80 000 equal keys each); three blocks are left for five probes
    assert info["current"] == 1 and info["longest_run"] > 4096 and info["usable_blocks"] == 3, info
    assert info["max_div_served"] is None or info["max_div_served"] <= 2, info
    assert store.scan(q, max_divergence=4).tobytes() == want.tobytes()
    assert "index_probe" not in store.last_scan_kernel()
    store.close()
    # forced (no limit on runs or candidates): slow, but the rows are the same — every subject of a run is compared in full
    index_env["SMAFA_INDEX_MAX_RUN"] = "100000000"
    index_env["SMAFA_INDEX_CAND"] = "100"
    store = smafa_amd.SubjectStore(L, 1)
    store.push(s)
    assert store.build_index(4)["max_div_served"] == 4
    assert store.scan(q, max_divergence=4).tobytes() == want.tobytes()
    assert "index_probe" in store.last_scan_kernel()
    assert store.scan(q, max_divergence=0).tobytes() == oracle.scan_codes(s, q, 0).tobytes()
    store.close()


@pytest.mark.parametrize("planes", ["2", "3"])
def test_query_letters_the_store_has_never_seen(index_env, planes):
    """a 2-plane nucleotide store (no N among the subjects) probed by queries with N: a block holding an N matches nobody"""
    if planes == "3":
        index_env["SMAFA_NT_PLANES"] = "3"
    rng = np.random.default_rng(5)
    s = rng.integers(0, 4, size=(66000, 60), dtype=np.uint8)
    q = queries_from(rng, s, 300, 4, 3)
    q[rng.random(q.shape) < 0.02] = 4
    store = smafa_amd.SubjectStore(60, 0)
    store.push(s)
    assert store.info().planes == int(planes)
    store.build_index(3)
    for D in (0, 2, 3):
        assert store.scan(q, max_divergence=D).tobytes() == oracle.scan_codes(s, q, D).tobytes()
        assert store.last_scan_kernel() == "smafa::index_probe_kernel<%s, 3, 2>" % planes
    store.close()


def test_index_arguments():
    store = smafa_amd.SubjectStore(150, 1)
    store.push(np.zeros((10, 150), dtype=np.uint8))
    with pytest.raises(smafa_amd.SmafaError):
        store.build_index(3)  # 150 columns: five words per plane
    store.close()
    store = smafa_amd.SubjectStore(20, 1)
    with pytest.raises(smafa_amd.SmafaError):
        store.build_index(3)  # empty store
    store.push(np.zeros((10, 20), dtype=np.uint8))
    with pytest.raises(smafa_amd.SmafaError):
        store.build_index(20)  # more blocks than columns
    with pytest.raises(smafa_amd.SmafaError):
        store.build_index(40)
    assert store.build_index(19)["blocks"] == 20
    store.close()


def test_product_cli_answers_from_the_index_when_told_to(index_env, tmp_path):
    """`smafa query` lets each store decide when its block index has paid for itself (rent or buy: never at this size); an explicit
    SMAFA_INDEX wins — here mode 2, an index at the first chunk of queries: the same bytes as the oracle CLI, and the log says so"""
    rng = np.random.default_rng(11)
    letters = np.frombuffer(b"ACGTN", dtype=np.uint8)
    s = letters[rng.integers(0, 4, size=(30000, 60))]
    s[7] = s[3]
    q = s[rng.integers(0, len(s), size=400)].copy()
    for r in q:
        for _ in range(int(rng.integers(0, 6))):
            r[rng.integers(0, 60)] = letters[rng.integers(0, 5)]
    sf, qf, db = str(tmp_path / "s.fna"), str(tmp_path / "q.fna"), str(tmp_path / "db")
    oracle.write_fasta(sf, [bytes(r) for r in s])
    oracle.write_fasta(qf, [bytes(r) for r in q])
    assert subprocess.run([_lib.CLI_PATH, "makedb", "-i", sf, "-d", db], capture_output=True).returncode == 0
    for flags in (["--max-divergence", "3"], ["--max-divergence", "2", "--max-num-hits", "4"]):
        want = oracle.run_cli("query", "-d", db, "-q", qf, *flags)
        plain = subprocess.run([_lib.CLI_PATH, "query", "-d", db, "-q", qf, "-v", *flags], capture_output=True, text=True)
        assert plain.returncode == want.returncode == 0 and plain.stdout == want.stdout
        assert "block index" not in plain.stderr  # rent or buy: 400 queries x 30 000 subjects never pay for one
        env = dict(os.environ, SMAFA_INDEX="2", SMAFA_INDEX_MIN_ROWS="1")
        forced = subprocess.run([_lib.CLI_PATH, "query", "-d", db, "-q", qf, "-v", *flags], capture_output=True, text=True, env=env)
        assert forced.returncode == 0 and forced.stdout == want.stdout, flags
        assert "block index of 30000 rows" in forced.stderr, forced.stderr[-400:]
