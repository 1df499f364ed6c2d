"""Which kernel for which bound?  10 000 planted queries against 10M x 60 subjects at bounds 5..18, fixed-bound scans and
best-hit scans that start at that bound (the near-hit probe's form), under the automatic dispatch, with the zone kernel
forced / off, and with the filter-plane-resident kernels allowed at any bound (SMAFA_PRUNE_P=1).  GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.cuda.init()  # (before the library opens the device, as in bench.py)
import smafa_amd
from smafa_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
alphabet = int(sys.argv[2]) if len(sys.argv) > 2 else 1
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 10_000
L = int(sys.argv[4]) if len(sys.argv) > 4 else 60
subj = synth.subjects(n, L, alphabet, seed=1 if alphabet else 2)
q, _, _ = synth.queries(subj, nq, alphabet, seed=3, max_subs=10 if alphabet else 6)
for env in ({}, {"SMAFA_ZONE": "2"}, {"SMAFA_ZONE": "0"}, {"SMAFA_ZONE": "0", "SMAFA_PRUNE_P": "1"}, {"SMAFA_ZONE": "2", "SMAFA_PRUNE_P": "1"}):
    for k in ("SMAFA_ZONE", "SMAFA_PRUNE_P"): os.environ.pop(k, None)
    os.environ.update(env)
    store = smafa_amd.SubjectStore(L, alphabet); store.push(subj)
    qs = smafa_amd.QuerySet(store, q)
    hits = torch.empty(3 * (1 << 24), dtype=torch.int32, device="cuda"); cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    for k in (None, 1):
        line = []
        for D in (5, 6, 7, 8, 9, 10, 12, 14, 16, 18, 20, 24, 28, 30, 34):
            best = 1e9
            for rep in range(3):
                store.scan_launch(qs, D, k, hits.data_ptr(), 1 << 24, cnt.data_ptr()); store.sync()
                ms, launches = store.last_scan_ms(); best = min(best, ms)
            line.append("%d:%.1f" % (D, best))
        print("%-44s k=%-4s ms by bound  %s   [%s]" % (env or "automatic", k, "  ".join(line), store.last_scan_kernel()), flush=True)
    qs.close(); store.close()
