#!/usr/bin/env python3
"""Same-box A/B of the k-th modes without a bound (`smafa query --max-num-hits k`, src/lib.rs:242-295) on the metric's store:
SMAFA_KTH_SAMPLE = 0 (count every pair first, append in a second pass) against 1/div samples (engine.hip scan_range).
    python3 tools/kth_ab.py [aa|nt] [queries]   -> one line per (div, k): wall / kernel ms of smafa_scan_hits, rows; rows must agree"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import smafa_amd
from smafa_amd import synth

alphabet = 0 if (len(sys.argv) > 1 and sys.argv[1] == "nt") else 1
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
subj = synth.subjects(10_000_000, 60, alphabet, seed=1 if alphabet else 2)
q, _, _ = synth.queries(subj, Q, alphabet, seed=3, max_subs=10 if alphabet else 6)
ref = {}
for div in (0, 16, 8, 32):
    os.environ["SMAFA_KTH_SAMPLE"] = str(div)
    store = smafa_amd.SubjectStore(60, alphabet)
    store.push(subj)
    for k in (5, 50, 3):
        store.scan(q[:256], max_num_hits=k)
        best_w, best_k, st = None, None, None
        for _ in range(3):
            t = time.perf_counter()
            rows = store.scan(q, max_num_hits=k)
            w = (time.perf_counter() - t) * 1e3
            st = store.last_call_stats()
            if best_w is None or w < best_w:
                best_w, best_k = w, st["kernel_ms"]
        same = ref.setdefault(k, rows.tobytes()) == rows.tobytes()
        print("%s Q=%d sample 1/%-2d k=%-2d wall %7.2f ms  kernels %7.2f ms  %3d launches %d scans  rows %8d  same rows %s"
              % ("aa" if alphabet else "nt", Q, div, k, best_w, best_k, st["launches"], st["scans"], len(rows), same), flush=True)
    store.close()
