import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import smafa_amd
from smafa_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
subj = synth.subjects(n, 60, 1, seed=1)
q, _, _ = synth.queries(subj, 2000, 1, seed=3, max_subs=10)
store = smafa_amd.SubjectStore(60, 1); store.push(subj)
store.scan(q[:8], max_divergence=None, max_num_hits=1)
for rep in range(3):
    t = time.perf_counter(); rows = store.scan(q, max_divergence=None, max_num_hits=1); dt = time.perf_counter() - t
    print("best-hit N=%d Q=%d rows=%d %.2f ms" % (n, len(q), len(rows), dt * 1e3), flush=True)
