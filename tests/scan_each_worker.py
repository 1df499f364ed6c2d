"""Worker for tests/test_gpu_group.py::test_scan_each_equals_the_batch_scan.

smafa_scan_each: one store pass per query, all enqueued back to back by one call (optionally replayed as a HIP graph).
Query i's rows and exact count must equal its rows in the oracle's scan, with the zone level off (the pass streams the
prefilter's plane), automatic and forced, and on every repetition (the graph is captured by the first call and replayed
by the others).  torch supplies the device buffers (as bench.py does) and is imported first."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
torch.cuda.init()

import oracle  # noqa: E402
import smafa_amd  # noqa: E402
from smafa_amd import synth  # noqa: E402


def rows3(a):
    return np.stack([a["query"], a["subject"], a["dist"]], axis=1).astype(np.uint32)


def main():
    oracle.build()
    n, L, nq, D, cap = 300_000, 60, 24, 5, 64
    subj = synth.subjects(n, L, 1, seed=31)
    subj[1000:1040] = subj[999]  # a dense spot: one query with 41 rows
    qry, _, _ = synth.queries(subj, nq, 1, seed=32, max_subs=7)
    qry[3] = subj[999]
    want = oracle.scan_codes(subj, qry, D)
    dev = torch.device("cuda", 0)
    store = smafa_amd.SubjectStore(L, 1, 0)
    store.push(subj)
    qs = smafa_amd.QuerySet(store, qry)
    hits = torch.zeros(nq * cap * 3, dtype=torch.int32, device=dev)
    counts = torch.full((nq,), -1, dtype=torch.int64, device=dev)
    kernels = set()
    for zone_level in (0, 1, 2):
        store.set_zone_level(zone_level)
        for use_graph in (False, True):
            for rep in range(3):
                counts.fill_(-1)
                torch.cuda.synchronize()
                store.scan_each(qs, D, hits.data_ptr(), cap, counts.data_ptr(), use_graph=use_graph)
                store.sync()
                c = counts.cpu().numpy()
                h = hits.cpu().numpy().view(np.uint32).reshape(nq, cap, 3)
                for q in range(nq):
                    w = want[want["query"] == q]
                    assert c[q] == len(w), (zone_level, use_graph, rep, q, c[q], len(w))
                    r = h[q, : c[q]]  # rows carry the query's index in the set; within a pass they arrive in any order
                    r = r[np.lexsort((r[:, 1], r[:, 2], r[:, 0]))]
                    assert r.tobytes() == rows3(w).tobytes(), (zone_level, use_graph, rep, q)
            kernels.add(store.last_scan_kernel())
    # a smaller capacity than a query's rows: the count stays exact, the first `cap` rows are stored
    counts.fill_(-1)
    store.scan_each(qs, D, hits.data_ptr(), 8, counts.data_ptr())
    store.sync()
    assert int(counts[3].item()) == int((want["query"] == 3).sum()) and int(counts[3].item()) > 8
    assert any("scan_zone_few_kernel" in k for k in kernels) and any("scan_lazy_kernel" in k or "scan_kernel" in k for k in kernels), kernels
    # a query set destroyed and another one of the SAME size created (the allocator may hand out the same host address and
    # the same device buffer): the graph captured for the first must not be replayed for the second
    store.set_zone_level(1)
    store.scan_each(qs, D, hits.data_ptr(), cap, counts.data_ptr(), use_graph=True)
    store.sync()
    qs.close()
    qry_b, _, _ = synth.queries(subj, nq, 1, seed=33, max_subs=4)
    want_b = oracle.scan_codes(subj, qry_b, D)
    assert want_b.tobytes() != want.tobytes()
    qs_b = smafa_amd.QuerySet(store, qry_b)
    for rep in range(2):
        counts.fill_(-1)
        torch.cuda.synchronize()
        store.scan_each(qs_b, D, hits.data_ptr(), cap, counts.data_ptr(), use_graph=True)
        store.sync()
        c = counts.cpu().numpy()
        h = hits.cpu().numpy().view(np.uint32).reshape(nq, cap, 3)
        for q in range(nq):
            w = want_b[want_b["query"] == q]
            assert c[q] == len(w), ("recreated set", rep, q, c[q], len(w))
            r = h[q, : c[q]]
            r = r[np.lexsort((r[:, 1], r[:, 2], r[:, 0]))]
            assert r.tobytes() == rows3(w).tobytes(), ("recreated set", rep, q)
    qs_b.close()
    store.close()
    print("scan_each ok", sorted(kernels))


if __name__ == "__main__":
    main()
