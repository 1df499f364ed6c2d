"""End-to-end CLI timing at scale (run on the GPU box): makedb on an N-row nucleotide FASTA, then query."""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smafa_amd import synth, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
q = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
t = time.time(); subj = synth.subjects(n, 60, 0, seed=2); qry, _, _ = synth.queries(subj, q, 0, seed=3, max_subs=6); print("gen %.1fs" % (time.time() - t), flush=True)
t = time.time(); synth.write_fasta("/tmp/s.fna", subj, 0); synth.write_fasta("/tmp/q.fna", qry, 0); print("write fasta %.1fs (%.0f MB)" % (time.time() - t, os.path.getsize("/tmp/s.fna") / 1e6), flush=True)
def run(*args, out=None):
    t = time.time()
    r = subprocess.run([_lib.CLI_PATH, *args], stdout=open(out, "wb") if out else subprocess.DEVNULL, stderr=subprocess.PIPE)
    print("%-60s rc=%d %.2fs %s" % (" ".join(args)[:60], r.returncode, time.time() - t, r.stderr[-200:].decode()), flush=True)
run("makedb", "-i", "/tmp/s.fna", "-d", "/tmp/s.db")
print("db file %.0f MB" % (os.path.getsize("/tmp/s.db") / 1e6))
run("query", "-d", "/tmp/s.db", "-q", "/tmp/q.fna", "--max-divergence", "3", out="/tmp/o1.tsv")
run("query", "-d", "/tmp/s.db", "-q", "/tmp/q.fna", out="/tmp/o2.tsv")
run("query", "-d", "/tmp/s.db", "-q", "/tmp/q.fna", "--max-divergence", "3", "--max-num-hits", "5", out="/tmp/o3.tsv")
for f in ("/tmp/o1.tsv", "/tmp/o2.tsv", "/tmp/o3.tsv"):
    print(f, sum(1 for _ in open(f, "rb")), "rows")
