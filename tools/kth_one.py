import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, smafa_amd
from smafa_amd import synth
k = int(sys.argv[1])
far = len(sys.argv) > 2 and sys.argv[2] == "far"  # uniform-random queries: no near subject at all
subj = synth.subjects(10_000_000, 60, 1, seed=1)
q, _, _ = synth.queries(subj, 10_000, 1, seed=3, max_subs=10)
if far:
    q = synth.subjects(5_000, 60, 1, seed=9, dup_frac=0.0)
if len(sys.argv) > 2 and sys.argv[2] == "mixed":  # bench.py's besthit_unbounded leg: half planted, half uniform random
    q[5_000:] = synth.subjects(5_000, 60, 1, seed=9, dup_frac=0.0)
store = smafa_amd.SubjectStore(60, 1); store.push(subj)
store.scan(q, max_num_hits=k)
store.scan(q, max_num_hits=k)
