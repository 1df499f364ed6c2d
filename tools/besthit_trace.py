"""Best-hit mode without --max-divergence (the reference's default `smafa query`): running-minimum bounds,
seed launch + growing segments.  Prints host-API time, device scan time and launches per batch size (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import smafa_amd
from smafa_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
alphabet = int(sys.argv[2]) if len(sys.argv) > 2 else 1
max_subs = int(sys.argv[3]) if len(sys.argv) > 3 else (10 if alphabet else 6)  # substitutions per query: 0..max_subs
subj = synth.subjects(n, 60, alphabet, seed=1 if alphabet else 2)
store = smafa_amd.SubjectStore(60, alphabet); store.push(subj)
for nq in (2000, 10_000, 100_000):
    q, _, _ = synth.queries(subj, nq, alphabet, seed=3, max_subs=max_subs)
    store.scan(q[:8], max_divergence=None, max_num_hits=1)
    for D in (None, 5 if alphabet else 3):
        best = None
        for rep in range(3):
            t = time.perf_counter(); rows = store.scan(q, max_divergence=D, max_num_hits=1); dt = time.perf_counter() - t
            ms, launches = store.last_scan_ms()
            if best is None or dt < best[0]: best = (dt, ms, launches, len(rows))
        dt, ms, launches, nrows = best
        print("best-hit alphabet=%d N=%d Q=%-6d max_div=%-4s rows=%-6d host-api %.2f ms (%.0f q/s)  scan kernels %.2f ms in %d launches  plan=%s"
              % (alphabet, n, nq, D, nrows, dt * 1e3, nq / dt, ms, launches, store.last_scan_plan()), flush=True)
