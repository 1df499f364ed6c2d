"""Start-up cost of `smafa query` against a 10M x 60 nucleotide store: the reference's version-2 file (postcard varints,
decoded by all threads, then packed on the device) against the packed store file (mapped, copied).  Run on the GPU box:
    python3 tools/time_startup.py [rows]
Prints the drivers' own stage timings (-v) and the wall time of each run."""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import smafa_amd
from smafa_amd import _lib, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
td = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
subj = synth.subjects(n, 60, 0, seed=2)
q, _, _ = synth.queries(subj, 1000, 0, seed=3, max_subs=6)
sf, qf, v2, pk = (os.path.join(td, x) for x in ("s.fna", "q.fna", "v2.db", "packed.db"))
synth.write_fasta(sf, subj, 0)
synth.write_fasta(qf, q, 0)
def run(*a):
    t = time.time(); r = subprocess.run([_lib.CLI_PATH, *a], capture_output=True, text=True); dt = time.time() - t
    assert r.returncode == 0, r.stderr[-2000:]
    return dt, r
for name, flags, out in (("makedb (version 2)", [], v2), ("makedb --packed", ["--packed"], pk)):
    dt, r = run("makedb", "-i", sf, "-d", out, "-v", *flags)
    print("%-22s %6.2f s  file %4d MB" % (name, dt, os.path.getsize(out) >> 20), flush=True)
outs = []
for name, db in (("query, version-2 file", v2), ("query, packed file", pk), ("query, version-2 file", v2), ("query, packed file", pk)):
    dt, r = run("query", "-d", db, "-q", qf, "--max-divergence", "3", "-v")
    outs.append(r.stdout)
    stages = [l.split("smafa] ")[1] for l in r.stderr.splitlines() if "db " in l or "packed into HBM" in l or "queries:" in l]
    print("%-22s %6.2f s wall | %s" % (name, dt, " | ".join(stages)), flush=True)
assert len(set(outs)) == 1 and len(outs[0]) > 0
print("rows identical:", len(outs[0].splitlines()))
for f in (sf, qf, v2, pk):
    os.remove(f)
os.rmdir(td)
