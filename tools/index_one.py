#!/usr/bin/env python3
"""For rocprofv3: the metric's launch answered from the block index, 20 times (after the scan kernels' launch once).
    rocprofv3 --kernel-trace --stats -d gpurun_out/index_prof -- python3 tools/index_one.py [aa|nt] [queries] [bound]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import smafa_amd
from smafa_amd import synth

alphabet = 0 if (len(sys.argv) > 1 and sys.argv[1] == "nt") else 1
Q = int(sys.argv[2]) if len(sys.argv) > 2 else (100_000 if alphabet == 0 else 10_000)
D = int(sys.argv[3]) if len(sys.argv) > 3 else (3 if alphabet == 0 else 5)
N = int(os.environ.get("INDEX_AB_ROWS", 10_000_000))
subj = synth.subjects(N, 60, alphabet, seed=1 if alphabet else 2)
q, _, _ = synth.queries(subj, Q, alphabet, seed=3, max_subs=10 if alphabet else 6)
store = smafa_amd.SubjectStore(60, alphabet)
store.push(subj)
rows0 = store.scan(q, max_divergence=D)
store.build_index(D)
for _ in range(20):
    rows = store.scan(q, max_divergence=D)
print(store.last_scan_kernel(), store.last_call_stats(), len(rows), rows.tobytes() == rows0.tobytes())
