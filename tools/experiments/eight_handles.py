import sys, os, subprocess, time
sys.path.insert(0, os.getcwd())
import numpy as np, oracle, smafa_amd
from smafa_amd import synth, _lib
n, L, nq = 200_000, 60, 4001
subj = synth.subjects(n, L, 1, seed=21); qry, _, _ = synth.queries(subj, nq, 1, seed=22, max_subs=8)
want = oracle.scan_codes(subj, qry, 5)
g = smafa_amd.SubjectGroup(L, 1, devices=[0] * 8); g.push(subj[:150_000]); g.push(subj[150_000:])
for rep in range(3):
    got = g.scan(qry, max_divergence=5)
    assert got.tobytes() == want.tobytes()
best = g.scan(qry[:512], max_num_hits=1)
one = smafa_amd.SubjectStore(L, 1, 0); one.push(subj)
assert best.tobytes() == one.scan(qry[:512], max_num_hits=1).tobytes()
g.close(); one.close()
print("group x8 ok")
recs = synth.cluster_records(3000, 20, 60, 1, seed=4, max_subs=4)
synth.write_fasta("/tmp/c8.faa", recs, 1)
a = subprocess.run([_lib.CLI_PATH, "cluster", "-i", "/tmp/c8.faa", "-d", "5", "--alphabet", "aa"], capture_output=True)
t = time.time()
b = subprocess.run([_lib.CLI_PATH, "cluster", "-i", "/tmp/c8.faa", "-d", "5", "--alphabet", "aa", "--devices", "0,0,0,0,0,0,0,0"], capture_output=True)
print("cluster x8", a.returncode, b.returncode, a.stdout == b.stdout, len(a.stdout), round(time.time() - t, 2))
q8 = subprocess.run([_lib.CLI_PATH, "query", "-d", "/nonexistent", "-q", "/tmp/c8.faa", "--gpus", "8"], capture_output=True)
print("query --gpus 8 on a 1-GPU box:", q8.returncode, q8.stderr[-120:])
