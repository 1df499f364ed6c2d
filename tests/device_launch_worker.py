"""Worker for tests/test_gpu_parity.py::test_device_resident_launch_all_modes.

smafa_scan_launch leaves rows and count in HBM.  Fixed bound = the oracle's rows; k = 1 = exactly the rows at each
query's minimum; k >= 2 = a superset bounded by the device's final bound, which the k-th rule reduces to the
oracle's rows.  torch supplies the device buffers (as bench.py does) and is imported first."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
torch.cuda.init()

import oracle  # noqa: E402
import smafa_amd  # noqa: E402


def expected_with_k(all_hits, k):
    out, i = [], 0
    while i < len(all_hits):
        j = i
        while j < len(all_hits) and all_hits[j]["query"] == all_hits[i]["query"]:
            j += 1
        grp = all_hits[i:j]
        kth = grp[k - 1]["dist"] if len(grp) >= k else 0xFFFFFFFF
        out.append(grp[grp["dist"] <= kth])
        i = j
    return np.concatenate(out) if out else all_hits[:0]


def main():
    oracle.build()
    rng = np.random.default_rng(41)
    L, n = 60, 30000
    s = rng.integers(0, 4, size=(n, L), dtype=np.uint8)
    s[200:260] = s[9]
    q = s[rng.integers(0, n, size=120)].copy()
    for r in q:
        for _ in range(rng.integers(0, 9)):
            r[rng.integers(0, L)] = rng.integers(0, 4)
    store = smafa_amd.SubjectStore(L, 0)
    store.push(s)
    qset = smafa_amd.QuerySet(store, q)
    cap = 1 << 22  # the tightening modes append far more rows than they keep: the scratch block is 2 x cap
    d_hits = torch.zeros(cap * 3, dtype=torch.int32, device="cuda")
    d_count = torch.zeros(1, dtype=torch.int64, device="cuda")

    def launch(D, k):
        store.scan_launch(qset, D, k, d_hits.data_ptr(), cap, d_count.data_ptr())
        store.sync()
        cnt = int(d_count.item())
        assert cnt <= cap, (D, k, cnt)
        rows = d_hits[: 3 * cnt].cpu().numpy().view(np.uint32).reshape(-1, 3)
        order = np.lexsort((rows[:, 1], rows[:, 2], rows[:, 0]))
        return np.ascontiguousarray(rows[order]).view(smafa_amd.HIT_DTYPE).reshape(-1)

    for D in (3, 9, None):
        full = oracle.scan_codes(s, q, L if D is None else D)
        if D is not None:
            assert launch(D, None).tobytes() == full.tobytes(), D
        assert launch(D, 1).tobytes() == expected_with_k(full, 1).tobytes(), D
        for k in (2, 5, 70):
            got = launch(D, k)
            want = expected_with_k(full, k)
            assert len(got) >= len(want) and expected_with_k(got, k).tobytes() == want.tobytes(), (D, k)
    store.close()
    print("device launch modes ok")


if __name__ == "__main__":
    main()
