"""End-to-end CLI timings at scale on the GPU box (`-v` stage lines on stderr):
  makedb on a 10M-record FASTA, query with 100 000 queries against it (version-2 and packed store), --gpus 1 vs
  --devices 0,0 (two handles on one GPU: the multi-device driver's overhead), a 1M-query FASTQ.gz through the threaded loader."""
import gzip, os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from smafa_amd import _lib, synth

td = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
subj = synth.subjects(n, 60, 0, seed=2)
q, _, _ = synth.queries(subj, 1_000_000, 0, seed=3, max_subs=6)
sf, qf, qbig, v2, pk = (os.path.join(td, x) for x in ("s.fna", "q100k.fna", "q1m.fq.gz", "v2.db", "packed.db"))
synth.write_fasta(sf, subj, 0)
synth.write_fasta(qf, q[:100_000], 0)
letters = np.frombuffer(b"ACGTN", dtype=np.uint8)
asc = letters[q]
with gzip.open(qbig, "wb", compresslevel=1) as f:
    f.write(b"".join(b"@q%d\n" % i + asc[i].tobytes() + b"\n+\n" + b"I" * 60 + b"\n" for i in range(len(q))))
def run(label, *a, out=None):
    t = time.time()
    r = subprocess.run([_lib.CLI_PATH, *a], stdout=open(out, "wb") if out else subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
    dt = time.time() - t
    assert r.returncode == 0, r.stderr[-2000:]
    stages = [l.split("smafa] ")[1] for l in r.stderr.splitlines() if "DEBUG" in l]
    print("%-46s %6.2f s | %s" % (label, dt, " | ".join(stages)), flush=True)
run("makedb 10M records (version 2)", "makedb", "-i", sf, "-d", v2, "-v")
run("makedb --packed", "makedb", "-i", sf, "-d", pk, "--packed", "-v")
outs = []
for label, flags in (("query 100k, v2 file", ["-d", v2]), ("query 100k, packed file", ["-d", pk]), ("query 100k, packed, --devices 0,0", ["-d", pk, "--devices", "0,0"]),
                     ("query 100k, packed, best hit (no bound)", ["-d", pk, "NOBOUND"])):
    o = os.path.join(td, "out%d.tsv" % len(outs))
    nob = "NOBOUND" in flags
    flags = [f for f in flags if f != "NOBOUND"]
    run(label, "query", *flags, "-q", qf, *([] if nob else ["--max-divergence", "3"]), "-v", out=o)
    outs.append(open(o, "rb").read())
assert outs[0] == outs[1] == outs[2] and len(outs[0]) > 0
run("query 1M FASTQ.gz (threaded loader), packed", "query", "-d", pk, "-q", qbig, "--max-divergence", "3", "-v", out=os.path.join(td, "o.tsv"))
for f in os.listdir(td):
    os.remove(os.path.join(td, f))
os.rmdir(td)
