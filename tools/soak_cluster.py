"""Randomised differential soak of `cluster` on the GPU box: random family-structured inputs (duplicates, N / ambiguity
letters, lower case), random max-divergence, product vs the oracle CLI, byte for byte, for SOAK_SECONDS."""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle, smafa_amd

seed0 = int(os.environ.get("SOAK_SEED", str(int(time.time()))))
budget = float(os.environ.get("SOAK_SECONDS", "120"))
oracle.build()
tmp = tempfile.mkdtemp()
t_end, rounds = time.time() + budget, 0
print("cluster soak seed", seed0, flush=True)
while time.time() < t_end:
    rng = np.random.default_rng(seed0 + rounds)
    rounds += 1
    L = int(rng.choice([1, 5, 20, 32, 33, 60, 60, 64, 100, 150]))
    n = int(rng.choice([1, 2, 50, 1023, 1024, 1025, 3000, 8000]))
    letters = np.frombuffer(rng.choice([b"ACGT", b"ACGTN", b"ACGTacgtNRY-", b"AC"]), dtype=np.uint8)
    roots = letters[rng.integers(0, len(letters), size=(max(1, n // int(rng.choice([3, 30, 300]))), L))]
    recs = roots[rng.integers(0, len(roots), size=n)].copy()
    mut = rng.random(size=recs.shape) < float(rng.choice([0.0, 0.03, 0.1, 0.3]))
    recs[mut] = letters[rng.integers(0, len(letters), size=int(mut.sum()))]
    D = int(rng.integers(0, min(L, 10) + 1))
    path = os.path.join(tmp, "c.fna")
    oracle.write_fasta(path, [bytes(r) for r in recs])
    want = oracle.run_cli("cluster", "-i", path, "-d", str(D))
    out = os.path.join(tmp, "o.tsv")
    fd = os.open(out, os.O_WRONLY | os.O_CREAT | os.O_TRUNC)
    try:
        smafa_amd.cluster(path, D, out_fd=fd)
    finally:
        os.close(fd)
    if want.returncode != 0 or open(out).read() != want.stdout:
        print("MISMATCH round", rounds - 1, "seed", seed0, dict(L=L, n=n, D=D, letters=bytes(letters)), flush=True)
        sys.exit(1)
print("cluster soak ok: %d inputs in %.0f s" % (rounds, budget), flush=True)
