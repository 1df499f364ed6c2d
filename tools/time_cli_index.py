"""On the GPU box: `smafa query --max-divergence 5` on a 10M x 60 aa packed store with 1M queries, the scan kernels only (SMAFA_INDEX=1:
nothing builds an index) against the CLI's default (rent or buy: each store builds its block index once the chunks scanned so far
have cost what the build would).  Same bytes; wall time and the -v stage lines."""
import hashlib, os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smafa_amd import _lib, synth

td = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
n, nq = int(os.environ.get("ROWS", 10_000_000)), int(os.environ.get("QUERIES", 1_000_000))
subj = synth.subjects(n, 60, 1, seed=1)
q, _, _ = synth.queries(subj, nq, 1, seed=3, max_subs=10)
sf, qf, pk = (os.path.join(td, x) for x in ("s.faa", "q.faa", "packed.db"))
synth.write_fasta(sf, subj, 1)
synth.write_fasta(qf, q, 1)
r = subprocess.run([_lib.CLI_PATH, "makedb", "-i", sf, "-d", pk, "--packed", "--alphabet", "aa"], capture_output=True, text=True)
assert r.returncode == 0, r.stderr[-1000:]
for what, flags in (("--max-divergence 5", ["--max-divergence", "5"]), ("no bound (best hit, the reference's default)", [])):
    print("== smafa query, %d queries x %d aa subjects, %s" % (nq, n, what), flush=True)
    ref = None
    for rep in range(2):
        for label, env in (("scan kernels only (SMAFA_INDEX=1)", {"SMAFA_INDEX": "1"}), ("default (rent or buy)", {})):
            o = os.path.join(td, "out.tsv")
            t = time.time()
            r = subprocess.run([_lib.CLI_PATH, "query", "-d", pk, "-q", qf, *flags, "-v"], stdout=open(o, "wb"),
                               stderr=subprocess.PIPE, text=True, env=dict(os.environ, **env))
            dt = time.time() - t
            assert r.returncode == 0, r.stderr[-2000:]
            h = hashlib.sha256(open(o, "rb").read()).hexdigest()
            ref = ref or h
            stages = [l.split("smafa] ")[1] for l in r.stderr.splitlines() if "DEBUG" in l and any(k in l for k in ("scans +", "block index"))]
            print("%-36s %6.3f s  identical=%s | %s" % (label, dt, h == ref, " | ".join(stages)), flush=True)
for f in os.listdir(td):
    os.remove(os.path.join(td, f))
