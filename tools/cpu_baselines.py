"""BASELINE.md §2 CPU baselines on the GPU box's host, with the oracle (test infrastructure) as the CPU port:
  B1   reference-mirroring: 5-bit one-hot u64 x5 per 60-nt subject, per query fill N distances + min pass + equality
       pass (src/lib.rs:238,298,307), ONE thread, gcc -O3 baseline x86-64 (no POPCNT, like `cargo build --release`)
  B1n  same, -march=native
  B2   B1n over all the cores this job may use (query shards in worker processes; store shared copy-on-write)
  aa   the code-byte port used by bench.py's cpu_baseline (60 B per subject), one thread
"""
import multiprocessing as mp, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from smafa_amd import synth

N, L, D = 10_000_000, 60, 3
NT = np.frombuffer(b"ACGTN", dtype=np.uint8)
subj = synth.subjects(N, L, 0, seed=2)
qry, _, _ = synth.queries(subj, 512, 0, seed=3, max_subs=6)
sa, qa = NT[subj], NT[qry]
print("host: %d logical cpus visible, %s" % (os.cpu_count(), open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t")), flush=True)

def timed(fn, nq):
    t = time.perf_counter(); rows = fn(nq); dt = time.perf_counter() - t
    return nq / dt, rows

for name, native in (("B1  one-hot, 1 thread, -O3 (no POPCNT)", False), ("B1n one-hot, 1 thread, -O3 -march=native", True)):
    db = oracle.OnehotDB(sa, native=native)
    enc = db.encode_queries(qa)
    db.bench_besthit(enc[:2], D)
    rate, rows = timed(lambda nq: db.bench_besthit(enc[:nq], D), 40 if not native else 120)
    print("%-44s %8.2f query seqs/s" % (name, rate), flush=True)
    if native:
        workers = min(16, os.cpu_count() or 1)
        shards = np.array_split(np.arange(len(enc)), workers)
        def work(idx):
            return db.bench_besthit(np.ascontiguousarray(enc[idx]), D)
        ctx = mp.get_context("fork")
        t = time.perf_counter()
        with ctx.Pool(workers) as pool:
            pool.map(work, shards)
        dt = time.perf_counter() - t
        print("%-44s %8.2f query seqs/s  (%d worker processes)" % ("B2  one-hot, -march=native, all workers", len(enc) / dt, workers), flush=True)
    db.close()

subj_aa = synth.subjects(N, L, 1, seed=1)
q_aa, _, _ = synth.queries(subj_aa, 200, 1, seed=3, max_subs=10)
oracle.bench_besthit_codes(subj_aa, q_aa[:2], 5)
t = time.perf_counter(); oracle.bench_besthit_codes(subj_aa, q_aa, 5); dt = time.perf_counter() - t
print("%-44s %8.2f query seqs/s" % ("aa  code bytes, 1 thread, -O3", len(q_aa) / dt), flush=True)
