import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, smafa_amd
from smafa_amd import synth, _lib
_lib.lib().smafa_set_verbosity(2)
subj = synth.cluster_records(1000, 1000, 60, 1, seed=7, max_subs=4)
rng = np.random.default_rng(1); q = subj[rng.integers(0, len(subj), size=2000)]
store = smafa_amd.SubjectStore(60, 1); store.push(subj)
for i in range(3):
    t = time.perf_counter(); rows = store.scan(q, max_divergence=5); print("host-api %.1f ms rows=%d" % ((time.perf_counter() - t) * 1e3, len(rows)), flush=True)
