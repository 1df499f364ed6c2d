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


def run_world(world, db, qf, out, flags, hip=False):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "tests", "dist_worker.py"), "--db", db, "--queries", qf, "--out", out, *flags]
    if hip:
        cmd.append("--hip")
    env = dict(os.environ, OMP_NUM_THREADS="1")
    return subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=300)


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
