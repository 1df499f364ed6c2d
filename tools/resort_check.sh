#!/bin/bash
# On the GPU box: `smafa cluster` (5M records) with and without the re-sort of the growing centroid store; the outputs
# must be the same bytes.  Then a query against a store loaded in 40 pieces (python API), both ways.
cd "$(dirname "$0")/.."
SMAFA_WORLDS="" python3 tools/time_cluster.py 100000 50 | tail -1 | cut -c1-120
for r in 0 1; do
  SMAFA_RESORT=$r smafa_amd/bin/smafa cluster -i /tmp/cluster.faa -d 5 --alphabet aa -v > /tmp/cluster_r$r.out 2> /tmp/cluster_r$r.err
  echo "SMAFA_RESORT=$r rc=$?"; grep -E "batches|sorted again" /tmp/cluster_r$r.err | tail -8
done
cmp /tmp/cluster_r0.out /tmp/cluster_r1.out && echo "cluster outputs identical"
python3 - <<'PY'
import os, time, numpy as np, smafa_amd
rng = np.random.default_rng(3)
n, L = 4_000_000, 60
s = rng.integers(0, 20, size=(n, L), dtype=np.uint8)
q = s[rng.integers(0, n, size=10000)].copy()
for r in q:
    for _ in range(rng.integers(0, 8)): r[rng.integers(0, L)] = rng.integers(0, 20)
rows = {}
for resort in ("0", "1"):
    os.environ["SMAFA_RESORT"] = resort
    st = smafa_amd.SubjectStore(L, 1)
    for a in range(0, n, 100_000): st.push(s[a:a + 100_000])
    t = time.time(); got = st.scan(q, max_divergence=5); first = time.time() - t
    ms = []
    for _ in range(5):
        t = time.time(); st.scan(q, max_divergence=5); ms.append((time.time() - t) * 1e3)
    rows[resort] = got.tobytes()
    print("store of %d rows appended in 40 pieces, SMAFA_RESORT=%s: first scan %.1f ms, then %.2f ms per 10 000-query scan (host API), %s"
          % (n, resort, first * 1e3, min(ms), st.last_scan_kernel()))
    st.close()
print("rows identical:", rows["0"] == rows["1"])
PY
