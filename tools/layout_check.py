"""What the data-driven store layout buys on stores whose columns differ (run on the GPU box).
A 10M x 60 store in which half of the columns are conserved (one letter in 90 % of the rows) and the conserved ones come
FIRST in the file — the worst case for a prefilter that looks at the first 32 columns as they are — scanned with the
layout chosen from the data (default) and with SMAFA_LAYOUT=0 (columns in file order, default code split)."""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

if len(sys.argv) > 1 and sys.argv[1] == "--worker":
    import torch
    import smafa_amd
    from smafa_amd import synth
    alphabet = int(sys.argv[2])
    n, L, Q, D = int(os.environ.get("LAYOUT_ROWS", "10000000")), 60, 10000, 5 if alphabet else 3
    rng = np.random.default_rng(5)
    lc = synth.letter_codes(alphabet)
    s = lc[rng.integers(0, len(lc), size=(n, L), dtype=np.uint8)]
    dom = lc[rng.integers(0, len(lc), size=30)]
    keep = rng.random(size=(n, 30)) < 0.90
    s[:, :30] = np.where(keep, dom[None, :], s[:, :30])   # columns 0..29 conserved, 30..59 uniform
    q, _, _ = synth.queries(s, Q, alphabet, seed=3, max_subs=10 if alphabet else 6)
    store = smafa_amd.SubjectStore(L, alphabet)
    store.push(s)
    qs = smafa_amd.QuerySet(store, q)
    cap = 1 << 22
    d_hits = torch.zeros(cap * 3, dtype=torch.int32, device="cuda")
    d_count = torch.zeros(1, dtype=torch.int64, device="cuda")
    ms = []
    for _ in range(8):
        store.scan_launch(qs, D, None, d_hits.data_ptr(), cap, d_count.data_ptr())
        ms.append(store.last_scan_ms()[0])
    rows = d_hits[: 3 * int(d_count.item())].cpu().numpy().view(np.uint32).reshape(-1, 3)
    order = np.lexsort((rows[:, 1], rows[:, 2], rows[:, 0]))
    import hashlib
    print("alphabet=%s layout=%s  %-36s %8.3f ms/launch  rows=%d  sha=%s" % (
        "aa" if alphabet else "nt", os.environ.get("SMAFA_LAYOUT", "1"), store.last_scan_kernel(), float(np.median(ms[2:])),
        len(rows), hashlib.sha256(np.ascontiguousarray(rows[order]).tobytes()).hexdigest()[:12]), flush=True)
    sys.exit(0)

for alphabet in (1, 0):
    for layout in ("1", "0"):
        env = dict(os.environ, SMAFA_LAYOUT=layout)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", str(alphabet)], env=env, check=True)
