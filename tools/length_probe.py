"""Scan throughput and parity at sequence lengths other than 60 (run on the GPU box)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, smafa_amd, oracle
from smafa_amd import synth
import os
alpha = int(sys.argv[1]) if len(sys.argv) > 1 else 1
D = int(sys.argv[2]) if len(sys.argv) > 2 else 5
for L in [int(x) for x in os.environ.get("LENGTHS", "20,30,60,90,120,150,250,600").split(",")]:
    s = synth.subjects(2_000_000, L, alpha, seed=1); q, rows_p, subs = synth.queries(s, 4000, alpha, seed=3, max_subs=min(10, L // 3))
    st = smafa_amd.SubjectStore(L, alpha); st.push(s); st.scan(q[:8], D)
    r = st.scan(q, D)
    ms, _ = st.last_scan_ms(); plan = st.last_scan_plan()
    d = (s[r["subject"]] != q[r["query"]]).sum(axis=1)
    have = set(zip(r["query"].tolist(), r["subject"].tolist()))
    ok = bool((d == r["dist"]).all()) and all((i, int(rows_p[i])) in have for i in range(4000) if subs[i] <= D)
    w = oracle.scan_codes(s, q[:6], D); ok = ok and r[r["query"] < 6].tobytes() == w.tobytes()
    print("alphabet=%d D=%d L=%-3d rows=%-6d kernel %.3f ms  %.2e pairs/s  plan=%s ok=%s" % (alpha, D, L, len(r), ms, 2e6 * 4000 / (ms * 1e-3), plan, ok), flush=True)
    st.close()
