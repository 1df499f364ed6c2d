"""Worker for tests/test_gpu_layout.py::test_device_launch_counts_are_exact_at_any_capacity (torch supplies the device
buffers, as in bench.py, and is imported first)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
torch.cuda.init()

import oracle  # noqa: E402
import smafa_amd  # noqa: E402


def main():
    oracle.build()
    rng = np.random.default_rng(8)
    L, n = 40, 4096
    s = rng.integers(0, 4, size=(n, L), dtype=np.uint8)
    q = s[:3].copy()
    store = smafa_amd.SubjectStore(L, 0)
    store.push(s)
    d_count = torch.zeros(1, dtype=torch.int64, device="cuda")
    for nq in (1, 3):
        qset = smafa_amd.QuerySet(store, q[:nq])
        total = n * nq  # max_div = L: every pair qualifies
        want = oracle.scan_codes(s, q[:nq], L)
        for cap in (total, total + 1, total - 1, 1000, 1):
            d_hits = torch.full((max(cap, 1) * 3 + 3,), -1, dtype=torch.int32, device="cuda")
            for _ in range(2):  # twice: the kernel leaves its internal counters at zero for the next launch
                store.scan_launch(qset, L, None, d_hits.data_ptr(), cap, d_count.data_ptr())
                store.sync()
                assert int(d_count.item()) == total, (nq, cap, int(d_count.item()))
            rows = d_hits[: 3 * min(cap, total)].cpu().numpy().view(np.uint32).reshape(-1, 3)
            assert (d_hits[3 * min(cap, total):].cpu().numpy() == -1).all(), "wrote past the capacity"
            order = np.lexsort((rows[:, 1], rows[:, 2], rows[:, 0]))
            got = np.ascontiguousarray(rows[order]).view(smafa_amd.HIT_DTYPE).reshape(-1)
            if cap >= total:
                assert got.tobytes() == want.tobytes(), (nq, cap)
            else:  # any cap rows of the answer, each at most once
                keys = set(zip(want["query"].tolist(), want["subject"].tolist(), want["dist"].tolist()))
                mine = list(zip(got["query"].tolist(), got["subject"].tolist(), got["dist"].tolist()))
                assert len(set(mine)) == len(mine) == cap and set(mine) <= keys, (nq, cap)
        store.scan_launch(qset, L, None, 0, 0, d_count.data_ptr())  # count only: no buffer at all
        store.sync()
        assert int(d_count.item()) == total
        # tightening mode on the same store: k = 1 keeps the rows at the minimum (the query itself and its copies)
        store.scan_launch(qset, None, 1, d_hits.data_ptr(), 1, d_count.data_ptr())
        store.sync()
        assert int(d_count.item()) >= nq
        qset.close()
    store.close()
    # the same contract when the launch is answered from the block index: 128 distinct rows x 64 copies, 100 of them as queries —
    # 6 400 rows, far more per workgroup than its LDS stage parks (the rest goes straight to the list)
    os.environ["SMAFA_INDEX_CAND"] = "1"  # (runs of 64 equal keys: forced past the limit on expected candidates)
    distinct = rng.integers(0, 4, size=(128, L), dtype=np.uint8)
    s = np.repeat(distinct, 64, axis=0)[rng.permutation(128 * 64)]
    q = distinct[:100].copy()
    store = smafa_amd.SubjectStore(L, 0)
    store.push(s)
    assert store.build_index(2)["max_div_served"] == 2
    qset = smafa_amd.QuerySet(store, q)
    want = oracle.scan_codes(s, q, 2)
    total = len(want)
    assert total >= 6400
    keys = set(zip(want["query"].tolist(), want["subject"].tolist(), want["dist"].tolist()))
    for cap in (total, total - 1, 700, 1, 0):
        d_hits = torch.full((max(cap, 1) * 3 + 3,), -1, dtype=torch.int32, device="cuda")
        for _ in range(2):
            store.scan_launch(qset, 2, None, d_hits.data_ptr() if cap else 0, cap, d_count.data_ptr())
            store.sync()
            assert "index_probe" in store.last_scan_kernel(), store.last_scan_kernel()
            assert int(d_count.item()) == total, (cap, int(d_count.item()), total)
        rows = d_hits[: 3 * cap].cpu().numpy().view(np.uint32).reshape(-1, 3)
        assert (d_hits[3 * cap:].cpu().numpy() == -1).all(), "wrote past the capacity"
        mine = list(zip(rows[:, 0].tolist(), rows[:, 1].tolist(), rows[:, 2].tolist()))
        assert len(set(mine)) == len(mine) == cap and set(mine) <= keys, cap
    qset.close()
    store.close()
    print("device capacity ok")


if __name__ == "__main__":
    main()
