import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, smafa_amd
from smafa_amd import synth
alphabet = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = 10_000_000
subj = synth.subjects(n, 60, alphabet, seed=1 if alphabet else 2)
store = smafa_amd.SubjectStore(60, alphabet); store.push(subj)
NQ = int(os.environ.get('KTH_Q', '10000'))
q, _, _ = synth.queries(subj, NQ, alphabet, seed=3, max_subs=10 if alphabet else 6)
for D, k in ((None, 2), (None, 5), (5 if alphabet else 3, 5), (None, 50), (12, 5)):
    best = None
    for rep in range(3):
        t = time.perf_counter(); rows = store.scan(q, max_divergence=D, max_num_hits=k); dt = time.perf_counter() - t
        ms, nl = store.last_scan_ms()
        best = dt if best is None else min(best, dt)
    print("alphabet=%d Q=%d max_div=%s k=%d rows=%d host-api %.1f ms (last scan %.1f ms, %d launches) plan=%s" % (alphabet, NQ, D, k, len(rows), best * 1e3, ms, nl, store.last_scan_plan()), flush=True)
