"""N>1 path on CPU: world_size-2 gloo run of the sharded query, compared with the oracle CLI."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import oracle
import smafa_amd
from smafa_amd import dist as sdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def make_inputs(tmp_path, n=600, q=45, L=30):
    rng = np.random.default_rng(21)
    letters = np.frombuffer(b"ACGTN", dtype=np.uint8)
    s = letters[rng.integers(0, 4, size=(n, L))]
    s[40:60] = s[3]
    qs = s[rng.integers(0, n, size=q)].copy()
    for r in qs:
        for _ in range(rng.integers(0, 5)):
            r[rng.integers(0, L)] = letters[rng.integers(0, 5)]
    sf, qf, db = str(tmp_path / "s.fna"), str(tmp_path / "q.fna"), str(tmp_path / "db")
    oracle.write_fasta(sf, [bytes(r) for r in s])
    oracle.write_fasta(qf, [bytes(r) for r in qs])
    smafa_amd.build()
    smafa_amd.makedb(sf, db)
    return db, qf


def launch(world, worker_args):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "tests", "dist_worker.py"), *worker_args]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    return subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=300)


def exit_code_of_ranks(r):
    """torch.distributed.run exits 1 whatever its workers returned; their own status is in its failure report"""
    import re

    codes = set(int(m) for m in re.findall(r"exitcode\s*:\s*(-?\d+)", r.stderr))
    codes.discard(-15)  # a rank the launcher terminated after the first one had failed
    return codes.pop() if len(codes) == 1 else (0 if r.returncode == 0 else sorted(codes))


def run_world(world, db, qf, out, flags, hip=False):
    return launch(world, ["--db", db, "--queries", qf, "--out", out, *flags] + (["--hip"] if hip else []))


def test_shard_bounds_cover_everything():
    for q in (0, 1, 7, 100, 1001):
        for world in (1, 2, 3, 8):
            b = [sdist.shard_bounds(q, world, r) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == q
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))


@pytest.mark.parametrize("flags", [
    [], ["--max-divergence", "3"], ["--max-num-hits", "5"], ["--max-num-hits", "4", "--limit-per-sequence", "1"],
])
def test_sharded_query_world2_gloo_equals_oracle(tmp_path, flags):
    db, qf = make_inputs(tmp_path)
    want = oracle.run_cli("query", "-d", db, "-q", qf, *flags)
    assert want.returncode == 0
    outs = []
    for world in (1, 2):
        out = str(tmp_path / f"out{world}.tsv")
        r = run_world(world, db, qf, out, flags)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(open(out).read())
    assert outs[0] == outs[1] == want.stdout
    assert len(want.stdout) > 0


def test_sharded_query_panics_on_every_rank_without_hanging(tmp_path):
    """fewer queries than ranks + an input the reference panics on (--limit-per-sequence without --max-num-hits > 1,
    src/lib.rs:301-303): the rank with the empty shard must fail like the others instead of waiting in the gather"""
    db, qf = make_inputs(tmp_path, q=1)
    out = str(tmp_path / "out.tsv")
    r = run_world(2, db, qf, out, ["--limit-per-sequence", "1"])
    assert r.returncode != 0
    assert r.stderr.count("limit_per_sequence is implemented unless max_num_hits > 1") >= 2, r.stderr[-3000:]
    assert "Timeout" not in r.stderr and "timed out" not in r.stderr


def test_sharded_query_prints_rows_in_front_of_a_bad_record(tmp_path):
    """a query record with a byte outside the alphabet: the rows of the queries before it are written, then every rank
    fails with the reference's panic text — the same bytes and exit as the oracle CLI (src/lib.rs:232-318)"""
    db, qf = make_inputs(tmp_path, q=30)
    raw = open(qf, "rb").read().split(b">")
    raw[21] = raw[21][:-5] + b"E" + raw[21][-4:]  # record 20 gets a bad byte
    open(qf, "wb").write(b">".join(raw))
    want = oracle.run_cli("query", "-d", db, "-q", qf)
    assert want.returncode == 101 and len(want.stdout) > 0
    out = str(tmp_path / "out.tsv")
    r = run_world(2, db, qf, out, [])
    assert r.returncode != 0
    assert open(out).read() == want.stdout
    assert want.stderr.strip().splitlines()[-1].split("panicked")[-1][-60:] in r.stderr or "cannot be interpreted as nucleotide" in r.stderr


@pytest.mark.gpu
def test_sharded_query_hip_scanner_equals_oracle(tmp_path):
    """same driver, product (HIP) scanner, 2 ranks sharing GPU 0 (gloo for the row gather)"""
    db, qf = make_inputs(tmp_path, n=3000, q=200, L=60)
    flags = ["--max-divergence", "6", "--max-num-hits", "3"]
    want = oracle.run_cli("query", "-d", db, "-q", qf, *flags)
    out = str(tmp_path / "out.tsv")
    r = run_world(2, db, qf, out, flags, hip=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert open(out).read() == want.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_native_sharded_query_parts_packed_store_gzip_and_bad_records(tmp_path, world):
    """the product path of smafa_amd.dist (smafa_qsession_*): every rank opens the DB itself (version-2 file, then the packed
    store file: no host code rows), parses only its byte range of the query file (a gzip file: everybody falls back to the
    whole file), and rank 0 prints — stdout and exit status of the oracle CLI in every case, incl. a bad record in the
    first and in the last rank's share (rows behind a bad record are never printed)"""
    import gzip

    db, qf = make_inputs(tmp_path, n=3000, q=211, L=60)
    flags = ["--max-divergence", "6", "--max-num-hits", "3"]
    want = oracle.run_cli("query", "-d", db, "-q", qf, *flags)
    assert want.returncode == 0 and want.stdout
    out = str(tmp_path / "out.tsv")
    r = run_world(world, db, qf, out, flags, hip=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert open(out).read() == want.stdout
    # the packed store file (what `makedb --packed` writes)
    packed = str(tmp_path / "db.packed")
    alphabet, codes = smafa_amd.read_db(db)
    st = smafa_amd.SubjectStore(60, alphabet, 0)
    st.push(codes)
    st.save(packed)
    st.close()
    for f in (flags, [], ["--max-num-hits", "4", "--limit-per-sequence", "1"])[: 3 if world == 2 else 1]:
        w = oracle.run_cli("query", "-d", db, "-q", qf, *f)
        r = run_world(world, packed, qf, out, f, hip=True)
        assert r.returncode == 0, r.stderr[-2000:]
        assert open(out).read() == w.stdout, f
    # gzip queries: no byte ranges — every rank takes the whole file and its block by count
    gz = str(tmp_path / "q.fna.gz")
    with gzip.open(gz, "wb") as g:
        g.write(open(qf, "rb").read())
    r = run_world(world, packed, gz, out, flags, hip=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert open(out).read() == want.stdout
    # a bad byte in the first share / in the last share
    raw = open(qf, "rb").read().split(b">")
    for victim in (20, 200) if world == 2 else (200,):
        bad = list(raw)
        bad[victim + 1] = bad[victim + 1][:-5] + b"E" + bad[victim + 1][-4:]
        bf = str(tmp_path / ("bad%d.fna" % victim))
        open(bf, "wb").write(b">".join(bad))
        w = oracle.run_cli("query", "-d", db, "-q", bf, *flags)
        assert w.returncode == 101 and w.stdout
        r = run_world(world, packed, bf, out, flags, hip=True)
        assert exit_code_of_ranks(r) == 101, (r.returncode, r.stderr[-2000:])
        assert open(out).read() == w.stdout, victim
        assert r.stderr.count("cannot be interpreted as nucleotide") >= world, r.stderr[-2000:]
    # an input-independent panic with an empty share on one rank: every rank fails alike, nobody waits in the gather
    one = str(tmp_path / "one.fna")
    open(one, "wb").write(b">" + raw[1])
    r = run_world(world, packed, one, out, ["--limit-per-sequence", "1"], hip=True)
    assert exit_code_of_ranks(r) == 101 and r.stderr.count("limit_per_sequence is implemented unless max_num_hits > 1") >= world, r.stderr[-3000:]
    if world > 2:  # (every launch of torch.distributed.run costs ~2 s: the remaining cases run with two ranks only)
        return
    # a missing query file is a panic (exit 101), a missing DB an Err (exit 1) — on every rank
    r = run_world(world, packed, str(tmp_path / "nope.fna"), out, flags, hip=True)
    assert exit_code_of_ranks(r) == 101, r.stderr[-1000:]
    r = run_world(world, str(tmp_path / "nope.db"), qf, out, flags, hip=True)
    assert exit_code_of_ranks(r) == 1, r.stderr[-1000:]


@pytest.mark.parametrize("world", [2, 3])
def test_allgather_bytes_ragged_blocks_gloo(tmp_path, world):
    """the transport under smafa_cluster_sharded: blocks of different sizes (one empty) in rank order"""
    out = str(tmp_path / "g")
    r = launch(world, ["--mode", "gather", "--out", out])
    assert r.returncode == 0, r.stderr[-2000:]
    want = []
    for rnd in range(3):
        for rank in range(world):
            size = [rank * 5 + 1, 0 if rank == 0 else 3, 70001][rnd]
            want.append(((np.arange(size) * 7 + rank * 13 + rnd) % 251).astype(np.uint8))
    want = np.concatenate(want)
    for rank in range(world):
        assert np.array_equal(np.load(out + ".rank%d.npy" % rank), want)


def cluster_input(tmp_path, roots=700, members=9, L=60, seed=5):
    rng = np.random.default_rng(seed)
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    base = letters[rng.integers(0, 4, size=(roots, L))]
    recs = np.repeat(base, members, axis=0)
    for r in recs:
        for _ in range(rng.integers(0, 4)):
            r[rng.integers(0, L)] = letters[rng.integers(0, 4)]
    recs = recs[rng.permutation(len(recs))]
    recs[100:130] = recs[7]  # exact duplicates (skipped by the reference, src/cluster.rs:46-48)
    path = str(tmp_path / "c.fna")
    oracle.write_fasta(path, [bytes(r) for r in recs])
    return path


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_cluster_equals_oracle_and_single_rank(tmp_path, world):
    """smafa_cluster_sharded: slices of every batch on different ranks (all on GPU 0 here), two all-gathers per
    batch, identical sequential pass on every rank — the bytes of the reference's cluster, for any world size"""
    path = cluster_input(tmp_path)  # 6300 records: several batches (the first holds 1024)
    want = oracle.run_cli("cluster", "-i", path, "-d", "4")
    assert want.returncode == 0 and len(want.stdout) > 0
    out = str(tmp_path / "out.tsv")
    r = launch(world, ["--mode", "cluster", "--input", path, "--max-divergence", "4", "--out", out])
    assert r.returncode == 0, r.stderr[-2000:]
    assert open(out).read() == want.stdout
    single = str(tmp_path / "single.tsv")
    fd = os.open(single, os.O_WRONLY | os.O_CREAT | os.O_TRUNC)
    try:
        smafa_amd.cluster(path, 4, out_fd=fd)
    finally:
        os.close(fd)
    assert open(single).read() == want.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("ndev", [1, 2, 3])
def test_cluster_multi_one_process_equals_oracle(tmp_path, ndev):
    """smafa_cluster_multi / `smafa cluster --devices ..`: one process, one host thread per handle playing a rank of the sharded
    cluster, exchanges through memory — the bytes of the oracle's cluster for any number of handles (all on GPU 0 here), also
    with a bad record behind good ones (same stdout, exit 101)"""
    path = cluster_input(tmp_path)
    want = oracle.run_cli("cluster", "-i", path, "-d", "4")
    assert want.returncode == 0 and len(want.stdout) > 0
    r = subprocess.run([smafa_amd._lib.CLI_PATH, "cluster", "-i", path, "-d", "4", "--devices", ",".join(["0"] * ndev)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-1000:]
    assert r.stdout == want.stdout
    # every rank's handle lives on its own entry of --devices and its thread launched there (debug log of each rank): with
    # more GPUs visible the entries spread over them, so a run where every replica silently lands on device 0 fails here
    import re

    devs = [d % smafa_amd.device_count() for d in range(ndev)]
    rv = subprocess.run([smafa_amd._lib.CLI_PATH, "cluster", "-i", path, "-d", "4", "--devices", ",".join(map(str, devs)), "-v"],
                        capture_output=True, text=True)
    assert rv.returncode == 0 and rv.stdout == want.stdout
    seen = sorted((int(a), int(b), int(c), int(d)) for a, b, c, d in re.findall(
        r"rank (\d+) of \d+: handle on device (\d+), scan launches issued with device (-?\d+) current, (\d+) off", rv.stderr))
    assert seen == [(r_, devs[r_], devs[r_], 0) for r_ in range(ndev)], (seen, rv.stderr[-1500:])
    out = str(tmp_path / "api.tsv")
    fd = os.open(out, os.O_WRONLY | os.O_CREAT | os.O_TRUNC)
    try:
        smafa_amd.cluster(path, 4, out_fd=fd, devices=[0] * ndev)
    finally:
        os.close(fd)
    assert open(out).read() == want.stdout
    raw = open(path, "rb").read().split(b">")
    raw[3001] = raw[3001][:-5] + b"E" + raw[3001][-4:]
    bad = str(tmp_path / "bad.fna")
    open(bad, "wb").write(b">".join(raw))
    w = oracle.run_cli("cluster", "-i", bad, "-d", "4")
    r = subprocess.run([smafa_amd._lib.CLI_PATH, "cluster", "-i", bad, "-d", "4", "--devices", ",".join(["0"] * ndev)],
                       capture_output=True, text=True)
    assert w.returncode == 101 and r.returncode == 101 and r.stdout == w.stdout and len(w.stdout) > 0


# ---- bench.py --gpus N from a plain shell: the process starts its own ranks
def _bench(*flags, timeout=600):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True,
                          env=env, timeout=timeout)


def test_bench_self_launch_spawns_ranks_on_cpu_only_host():
    """No GPU here: the parent must still have spawned torch.distributed.run with 2 ranks, each of which refuses to
    run without a HIP device (the engine has no CPU fallback) — the parent relays the failure."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("host has a GPU: covered by the gpu test below")
    r = _bench("--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "0", "--db-rows", "1000", "--queries", "8",
               "--no-cpu-baseline", "--no-stream")
    assert r.returncode != 0
    assert r.stderr.count("bench.py needs a HIP device") >= 2, r.stderr[-2000:]
    assert "WORLD_SIZE" not in r.stderr  # the old refusal ("launch with torch.distributed.run") is gone


@pytest.mark.gpu
def test_bench_line_contract_on_a_small_workload(tmp_path):
    """the JSON line the driver reads: the LAST line of stdout, at most 4096 characters with EVERY leg present (the BASELINE
    configs block runs at a hundredth of its sizes here), the contract's fields, `roofline` and `cpu_baseline` objects; the
    full record (every leg in detail) in the file the line names, every side leg verified"""
    import json

    full_path = str(tmp_path / "bench_full.json")
    r = _bench("--steps", "3", "--warmup", "1", "--db-rows", "300000", "--queries", "2048", "--cpu-seconds", "0.3", "--no-related",
               "--configs-scale", "0.01", "--full-record", full_path)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = r.stdout.splitlines()
    assert lines and lines[-1].startswith("{") and len([l for l in lines if l.startswith("{")]) == 1
    assert len(lines[-1]) <= 4096, len(lines[-1])
    out = json.loads(lines[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in out, key
    assert out["n_gpus"] == 1 and out["steps"] == 3 and out["warmup"] == 1 and out["verified"] is True and out["vs_baseline"] is None
    assert out["unit"] == "query seqs/s" and out["higher_is_better"] is True and out["dtype"] == "u32" and "workload" in out["config"]
    assert abs(out["value"] - 2048 / (out["ms_per_step"] * 1e-3)) / out["value"] < 1e-4
    assert out["roofline"]["bound"] == "valu" and out["roofline"]["kernel"].startswith("smafa::scan_")
    assert 0 < out["roofline"]["hbm_stream"]["frac_wall"] < 1.0 and out["roofline"]["hbm_stream"]["rows_identical"] is True
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and cb["unit"] == "query seqs/s" and cb["sample"]
    assert cb["kmode"]["value"] > 0 and cb["kmode"]["max_num_hits"] == 5
    legs = out["legs"]
    for name in ("unfiltered", "bound8", "bound14", "bound24", "besthit_mixed", "besthit_far", "host_api", "kth5", "kth50", "kth5_d5",
                 "kth50_d5", "cfg1", "cfg2", "cfg2N", "cfg3", "cfg4"):
        assert name in legs and legs[name][0] > 0, (name, legs.get(name))
    for name in ("bound8", "bound14", "bound24"):
        assert legs[name][2] is True
    for name in ("besthit_mixed", "kth5", "kth50", "kth5_d5", "kth50_d5", "host_api", "cfg1", "cfg2", "cfg2N", "cfg3", "cfg4"):
        assert legs[name][3] is True, (name, legs[name])
    assert out["full_record"] == full_path
    full = json.load(open(full_path))
    assert full["value"] == pytest.approx(out["value"], rel=1e-6) and full["verified"] is True
    st = full["stream"]
    assert st["roofline"]["bound"] == "hbm" and st["metric_store"]["rows_identical"] is True
    assert 0 < st["metric_store"]["streaming"]["frac_wall_streamed"] < 1.0
    assert all(x["verified"] for x in full["loose_bounds"]) and full["besthit_unbounded"]["verified"] is True
    assert full["host_api"]["rows_identical_to_device_launch"] is True and full["unfiltered"]["kernel_ms"] > 0
    assert all(v["verified"] for k, v in full["kth"].items() if isinstance(v, dict))
    assert all(v["verified"] for v in full["configs"].values())


@pytest.mark.gpu
def test_bench_self_launch_two_ranks_one_gpu_gloo(tmp_path):
    import json

    full_path = str(tmp_path / "bench_full.json")
    r = _bench("--gpus", "2", "--single-device", "--backend", "gloo", "--steps", "2", "--warmup", "1", "--db-rows", "200000",
               "--queries", "512", "--no-cpu-baseline", "--no-related", "--full-record", full_path)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) <= 4096, r.stdout[-2000:]
    line = json.loads(lines[0])
    out = json.load(open(full_path))
    assert line["n_gpus"] == 2 and line["verified"] is True and line["gathered_bytes_per_rank_per_step"] == out["gathered_bytes_per_rank_per_step"]
    assert out["n_gpus"] == 2 and out["verified"] is True and out["steps"] == 2
    frac = out["roofline"]["frac"]  # claimed only where profiles/ holds counters of this workload and kernel
    assert out["roofline"]["bound"] == "valu" and (frac is None or 0 < frac <= 1.0)
    # what crosses a link per rank and step is [count | the rows found (+ headroom)], not the capacity buffer:
    # 16 bytes of header + 12 bytes per row
    rows = out["checks"]["gathered_rows_total"] / 2
    assert out["checks"]["rows_fit_gathered_width"] is True
    assert 16 + 12 * rows <= out["gathered_bytes_per_rank_per_step"] <= 16 + 12 * (2.0 * rows + 128), out["gathered_bytes_per_rank_per_step"]


@pytest.mark.gpu
def test_rccl_rehearsal_world_of_one(tmp_path):
    """Nothing in this suite can put two ranks on two GPUs, so the RCCL side of the N>1 path is rehearsed over a world of
    ONE rank on the one GPU: `python -m smafa_amd.dist query|cluster --backend nccl` (device tensors through all_gather /
    gather / all_reduce of RCCL, no shortcut for a world of one) must print the oracle CLI's bytes, and `bench.py --gpus 1
    --rehearse-collectives` runs the timed N>1 loop (two buffers, gather on its own stream, widths by all_reduce MAX)
    through RCCL and verifies what the gather delivered."""
    import json

    db, qf = make_inputs(tmp_path, n=3000, q=211, L=60)
    env = dict(os.environ, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")

    def run(*args):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), "-m", "smafa_amd.dist", *args]
        return subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=300, cwd=ROOT)

    out = str(tmp_path / "o.tsv")
    for flags in (["--max-divergence", "6", "--max-num-hits", "3"], []):
        want = oracle.run_cli("query", "-d", db, "-q", qf, *flags)
        r = run("query", "-d", db, "-q", qf, *flags, "--backend", "nccl", "-o", out)
        assert r.returncode == 0, r.stderr[-3000:]
        assert open(out).read() == want.stdout and want.stdout
    path = cluster_input(tmp_path)
    want = oracle.run_cli("cluster", "-i", path, "-d", "4")
    r = run("cluster", "-i", path, "-d", "4", "--backend", "nccl", "-o", out)
    assert r.returncode == 0, r.stderr[-3000:]
    assert open(out).read() == want.stdout and want.stdout
    for collective in ("gather", "all_gather"):
        full_path = str(tmp_path / ("bench_%s.json" % collective))
        r = _bench("--gpus", "1", "--rehearse-collectives", "--collective", collective, "--steps", "3", "--warmup", "2",
                   "--db-rows", "200000", "--queries", "512", "--full-record", full_path)
        assert r.returncode == 0, r.stderr[-3000:]
        assert len(r.stdout.splitlines()[-1]) <= 4096
        line = json.load(open(full_path))
        assert line["verified"] is True and line["n_gpus"] == 1 and line["steps"] == 3
        c = line["checks"]
        assert c["gather_block_is_own_buffer"] is True and c["rows_fit_gathered_width"] is True and c["gathered_rows_total"] == line["rows_per_step"]
