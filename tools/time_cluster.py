"""Time `smafa cluster` on the SURVEY §8(d) cluster workload (run on the GPU box)."""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smafa_amd import synth, _lib
n_roots = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
members = int(sys.argv[2]) if len(sys.argv) > 2 else 50
t = time.time(); recs = synth.cluster_records(n_roots, members, 60, 1, seed=4, max_subs=4); print("gen %.1fs" % (time.time() - t), flush=True)
t = time.time(); synth.write_fasta("/tmp/cluster.faa", recs, 1); print("write %.1fs" % (time.time() - t), flush=True)
for rep in range(3):  # wall of the whole CLI run, stdout to a file (the first run also pays the page cache of the input)
    t = time.time()
    r = subprocess.run([_lib.CLI_PATH, "cluster", "-i", "/tmp/cluster.faa", "-d", "5", "--alphabet", "aa", "-v"], stdout=open("/tmp/cluster.out", "wb"), stderr=subprocess.PIPE)
    dt = time.time() - t
    print("run %d: %.3f s  | %s" % (rep, dt, " | ".join(l.split("smafa] ")[1] for l in r.stderr.decode(errors="replace").splitlines() if any(k in l for k in ("parsed", "bring-up", "batches", "scan kernels", "lines written", "from the start", "inside main")))), flush=True)
lines = sum(1 for _ in open("/tmp/cluster.out", "rb"))
cents = len(set(l.split(b"\t")[1] for l in open("/tmp/cluster.out", "rb")))
print("cluster rc=%d %.3fs records=%d lines=%d centroids=%d stderr=%s" % (r.returncode, dt, len(recs), lines, cents, r.stderr[-300:]))
# sharded form: W ranks over gloo, all on GPU 0 of this box (a rehearsal of the exchange, not a speed-up:
# the ranks share one GPU); the bytes must equal the single-process output
import hashlib
want = hashlib.sha256(open("/tmp/cluster.out", "rb").read()).hexdigest()
for world in [int(w) for w in os.environ.get("SMAFA_WORLDS", "2,4").split(",") if w]:
    t = time.time()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world,
                        "--master-addr", "127.0.0.1", "--master-port", str(29610 + world), "-m", "smafa_amd.dist",
                        "cluster", "-i", "/tmp/cluster.faa", "-d", "5", "--alphabet", "aa", "--backend", "gloo",
                        "--single-device", "-v", "-o", "/tmp/cluster.w.out"], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    dt = time.time() - t
    got = hashlib.sha256(open("/tmp/cluster.w.out", "rb").read()).hexdigest()  # -o: gloo's banner goes to stdout
    print("world=%d rc=%d %.1fs (incl. %d x python+torch start-up) identical=%s %s" % (world, r.returncode, dt, world, got == want, r.stderr[-420:].decode(errors="replace")), flush=True)
# one process, several handles (smafa_cluster_multi: host threads as ranks, exchanges through memory) — all on GPU 0 here
for devs in ("0,0", "0,0,0,0"):
    t = time.time()
    r = subprocess.run([_lib.CLI_PATH, "cluster", "-i", "/tmp/cluster.faa", "-d", "5", "--alphabet", "aa", "--devices", devs, "-v"],
                       stdout=open("/tmp/cluster.m.out", "wb"), stderr=subprocess.PIPE)
    dt = time.time() - t
    got = hashlib.sha256(open("/tmp/cluster.m.out", "rb").read()).hexdigest()
    print("--devices %s rc=%d %.2fs identical=%s %s" % (devs, r.returncode, dt, got == want, " | ".join(l.split("smafa] ")[1] for l in r.stderr.decode(errors="replace").splitlines() if "batches" in l or "scan kernels" in l)), flush=True)
