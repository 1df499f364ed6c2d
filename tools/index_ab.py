#!/usr/bin/env python3
"""Same-box A/B: the fixed-bound batched scan through the scan kernels against the same call answered from the block index
(smafa_db_build_index, index.hip.h), on the BASELINE shapes.  Rows must be byte-identical (both lists sorted).
    python3 tools/index_ab.py [aa|nt|aa50m] [queries] [bound]
One line per (bound, path): smafa_scan_hits wall / kernel ms (best of 5), rows, the index's own numbers."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import smafa_amd
from smafa_amd import synth

shape = sys.argv[1] if len(sys.argv) > 1 else "aa"
alphabet = 0 if shape == "nt" else 1
N = 50_000_000 if shape == "aa50m" else int(os.environ.get("INDEX_AB_ROWS", 10_000_000))
Q = int(sys.argv[2]) if len(sys.argv) > 2 else (100_000 if alphabet == 0 else 10_000)
bounds = [int(sys.argv[3])] if len(sys.argv) > 3 else ([3, 1] if alphabet == 0 else [5, 3, 8, 11])
subj = synth.subjects(N, 60, alphabet, seed=1 if alphabet else 2)
q, _, _ = synth.queries(subj, Q, alphabet, seed=3, max_subs=10 if alphabet else 6)
store = smafa_amd.SubjectStore(60, alphabet)
store.push(subj)


def timed(D):
    store.scan(q[:256], max_divergence=D)
    best = None
    for _ in range(5):
        t = time.perf_counter()
        rows = store.scan(q, max_divergence=D)
        w = (time.perf_counter() - t) * 1e3
        k = store.last_call_stats()["kernel_ms"]
        if best is None or k < best[1]:
            best = (w, k)
    return rows, best, store.last_scan_kernel()


for D in bounds:
    store.set_index(0)
    rows0, (w0, k0), name0 = timed(D)
    t = time.perf_counter()
    info = store.build_index(D)
    build_wall = (time.perf_counter() - t) * 1e3
    store.set_index(1)
    rows1, (w1, k1), name1 = timed(D)
    print("%s N=%d Q=%d bound %d  scan: wall %8.3f ms kernel %8.3f ms (%s)" % (shape, N, Q, D, w0, k0, name0), flush=True)
    print("%s N=%d Q=%d bound %d index: wall %8.3f ms kernel %8.3f ms (%s)  x%.1f  rows %d  identical %s" %
          (shape, N, Q, D, w1, k1, name1, k0 / max(k1, 1e-6), len(rows1), rows0.tobytes() == rows1.tobytes()), flush=True)
    print("    index: %d blocks, %.0f MB, built in %.1f ms (call %.1f ms), longest run %d, candidates/query %.2f, served up to %s"
          % (info["blocks"], info["bytes"] / 1e6, info["build_ms"], build_wall, info["longest_run"], info["candidates_per_query"],
             info["max_div_served"]), flush=True)
# the reference's default mode (best hit, no bound) and the K branch with an index built for bounds up to 11 (aa) / 5 (nt): the
# index answers the ladder's first step
if os.environ.get("INDEX_AB_BESTHIT", "1") != "0" and N <= 10_000_000:
    far = synth.subjects(Q // 2, 60, alphabet, seed=9, dup_frac=0.0)
    mixed = np.concatenate([q[: Q - Q // 2], far])
    for label, qq in (("planted", q), ("half unrelated", mixed)):
        for k in (1, 5):
            res = {}
            for mode in (0, 1):
                if mode:
                    info = store.build_index(11 if alphabet else 5)
                else:
                    store.drop_index()
                store.scan(qq[:512], max_num_hits=k)
                best = None
                for _ in range(3):
                    t = time.perf_counter()
                    rows = store.scan(qq, max_num_hits=k)
                    w = (time.perf_counter() - t) * 1e3
                    st = store.last_call_stats()
                    if best is None or w < best[0]:
                        best = (w, st["kernel_ms"], st["scans"])
                res[mode] = (rows.tobytes(), best)
            print("%s N=%d Q=%d no bound, k=%d, %s queries: scan kernels wall %7.2f ms (kernels %7.2f, %d scans); with the index (%d blocks, "
                  "served up to %s) wall %7.2f ms (kernels %7.2f, %d scans)  identical %s" %
                  (shape, N, len(qq), k, label, res[0][1][0], res[0][1][1], res[0][1][2], info["blocks"], info["max_div_served"],
                   res[1][1][0], res[1][1][1], res[1][1][2], res[0][0] == res[1][0]), flush=True)
store.close()
