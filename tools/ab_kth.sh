#!/bin/bash
# On the GPU box: tools/kth_ab.py-style timing of the k-th modes for the default library and every smafa_amd/lib_v*/ build
cd "$(dirname "$0")/.."
for lib in "" smafa_amd/lib_v*/libsmafa_amd.so; do
  [ -z "$lib" ] || [ -f "$lib" ] || continue
  echo "== ${lib:-default library}"
  SMAFA_AMD_LIB=${lib:+$PWD/$lib} python3 - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, smafa_amd
from smafa_amd import synth
subj = synth.subjects(10_000_000, 60, 1, seed=1)
q, _, _ = synth.queries(subj, 10_000, 1, seed=3, max_subs=10)
store = smafa_amd.SubjectStore(60, 1); store.push(subj)
for k in (5, 50):
    store.scan(q[:256], max_num_hits=k)
    best = None
    for _ in range(3):
        t = time.perf_counter(); rows = store.scan(q, max_num_hits=k); w = (time.perf_counter() - t) * 1e3
        st = store.last_call_stats()
        if best is None or w < best[0]: best = (w, st["kernel_ms"])
    print("k=%-2d wall %7.2f ms kernels %7.2f ms rows %d" % (k, best[0], best[1], len(rows)), flush=True)
PY
done
