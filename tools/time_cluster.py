"""Time `smafa cluster` on the SURVEY §8(d) cluster workload (run on the GPU box)."""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smafa_amd import synth, _lib
n_roots = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
members = int(sys.argv[2]) if len(sys.argv) > 2 else 50
t = time.time(); recs = synth.cluster_records(n_roots, members, 60, 1, seed=4, max_subs=4); print("gen %.1fs" % (time.time() - t), flush=True)
t = time.time(); synth.write_fasta("/tmp/cluster.faa", recs, 1); print("write %.1fs" % (time.time() - t), flush=True)
t = time.time()
r = subprocess.run([_lib.CLI_PATH, "cluster", "-i", "/tmp/cluster.faa", "-d", "5", "--alphabet", "aa"], stdout=open("/tmp/cluster.out", "wb"), stderr=subprocess.PIPE)
dt = time.time() - t
lines = sum(1 for _ in open("/tmp/cluster.out", "rb"))
cents = len(set(l.split(b"\t")[1] for l in open("/tmp/cluster.out", "rb")))
print("cluster rc=%d %.1fs records=%d lines=%d centroids=%d stderr=%s" % (r.returncode, dt, len(recs), lines, cents, r.stderr[-300:]))
