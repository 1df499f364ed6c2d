#!/usr/bin/env python3
"""One query per store pass, three ways of enqueueing K passes (wall time per pass on the stream, from HIP events):
a Python loop of smafa_scan_launch (round 2's stream leg), smafa_scan_each (one C call, K launches) and the same
captured as a HIP graph.  Zone level off: every pass streams the prefilter's bit-plane.
    python3 tools/stream_probe.py [--db-rows N] [--passes K]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--db-rows", type=int, default=10_000_000)
    ap.add_argument("--passes", type=int, default=200)
    ap.add_argument("--alphabet", default="aa")
    a = ap.parse_args()
    import torch

    import smafa_amd
    from smafa_amd import synth

    alphabet = 1 if a.alphabet == "aa" else 0
    t = time.time()
    subj = synth.subjects(a.db_rows, 60, alphabet, seed=1 if alphabet else 2)
    q, _, _ = synth.queries(subj, a.passes, alphabet, seed=3, max_subs=10 if alphabet else 6)
    t_gen = time.time() - t
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    store = smafa_amd.SubjectStore(60, alphabet, 0)
    t = time.time()
    store.push(subj)
    t_push = time.time() - t
    store.set_stream(stream.cuda_stream)
    info = store.info()
    K, cap = a.passes, 256
    hits = torch.zeros(K * cap * 3, dtype=torch.int32, device=dev)
    counts = torch.zeros(K, dtype=torch.int64, device=dev)
    qs = smafa_amd.QuerySet(store, q)
    ones = [smafa_amd.QuerySet(store, q[i:i + 1]) for i in range(8)]
    out = {"db_rows": a.db_rows, "passes": K, "generate_s": t_gen, "push_s": t_push}
    for zone in (0, 1):
        store.set_zone_level(zone)
        res = {}

        def timed(fn, reps=3):
            best = None
            for _ in range(reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                e0.record(stream)
                fn()
                e1.record(stream)
                t_host = time.perf_counter() - t0
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / K
                if best is None or ms < best[0]:
                    best = (ms, t_host / K * 1e3)
            return {"wall_ms_per_pass": best[0], "host_enqueue_ms_per_pass": best[1]}

        def py_loop():
            for i in range(K):
                store.scan_launch(ones[i % 8], 5 if alphabet else 3, None, hits.data_ptr(), cap, counts.data_ptr())

        D = 5 if alphabet else 3
        res["python_loop_scan_launch"] = timed(py_loop)
        res["scan_each"] = timed(lambda: store.scan_each(qs, D, hits.data_ptr(), cap, counts.data_ptr(), False))
        store.scan_each(qs, D, hits.data_ptr(), cap, counts.data_ptr(), True)  # capture
        torch.cuda.synchronize()
        res["scan_each_graph"] = timed(lambda: store.scan_each(qs, D, hits.data_ptr(), cap, counts.data_ptr(), True))
        if zone == 0:
            # the same K passes as ONE launch: query blocks of one query (smafa_set_query_block 1) — the grid walks the query
            # list, workgroup after workgroup, with no kernel boundary between two passes (rows of all queries in one list)
            big = torch.zeros(K * cap * 3, dtype=torch.int32, device=dev)
            total = torch.zeros(1, dtype=torch.int64, device=dev)
            store.set_query_block(1)
            res["one_launch_blocks_of_one"] = timed(lambda: store.scan_launch(qs, D, None, big.data_ptr(), K * cap, total.data_ptr()))
            res["one_launch_blocks_of_one"]["kernel"] = store.last_scan_kernel()
            res["one_launch_blocks_of_one"]["query_blocks"] = store.last_scan_plan()["query_blocks"]
            res["one_launch_blocks_of_one"]["rows"] = int(total.item())
            store.set_query_block(0)
        k_ms = []
        for i in range(20):
            store.scan_launch(ones[i % 8], D, None, hits.data_ptr(), cap, counts.data_ptr())
            k_ms.append(store.last_scan_ms()[0])
        res["kernel_ms_median"] = float(np.median(k_ms))
        res["kernel"] = store.last_scan_kernel()
        plane = info.hbm_bytes * (info.words_per_plane * 4) / info.bytes_per_subject
        res["filter_plane_bytes"] = plane
        if zone == 0:
            for k in ("python_loop_scan_launch", "scan_each", "scan_each_graph", "one_launch_blocks_of_one"):
                res[k]["frac_of_8TBs"] = plane / res[k]["wall_ms_per_pass"] / 1e6 / 8000.0
            res["kernel_frac_of_8TBs"] = plane / res["kernel_ms_median"] / 1e6 / 8000.0
        out["zone_level_%d" % zone] = res
    out["read_probe_8GiB_GBs"] = smafa_amd.hbm_read_probe(0, 8 << 30)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
