"""Scan behaviour outside the sparse-hit regime of the headline bench (run on the GPU box).
  dense:   1000 roots x 1000 members (0..4 substitutions): every query has ~hundreds of rows within D=5
  related: members differ from their root in ~25 % of the columns: few rows, but low lower bounds
Prints rows, wall ms per batch through the host API, with the prefilter on and off."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import smafa_amd
from smafa_amd import synth

L = int(os.environ.get("DENSE_L", "60"))
ALPHA = int(os.environ.get("DENSE_ALPHABET", "1"))
NLET = 20 if ALPHA else 4

def related(n_roots, members, frac, seed):
    rng = np.random.default_rng(seed)
    lc = synth.letter_codes(ALPHA)
    roots = lc[rng.integers(0, NLET, size=(n_roots, L))]
    recs = np.repeat(roots, members, axis=0)
    mut = rng.random(size=recs.shape) < frac
    recs[mut] = lc[rng.integers(0, NLET, size=int(mut.sum()))]
    rng.shuffle(recs, axis=0)
    return recs

for name, subj in (("dense", synth.cluster_records(1000, 1000, L, ALPHA, seed=7, max_subs=4)),
                   ("related25", related(1000, 1000, 0.25, 8)),
                   ("related10", related(1000, 1000, 0.10, 9))):
    rng = np.random.default_rng(1)
    q = subj[rng.integers(0, len(subj), size=2000)]
    store = smafa_amd.SubjectStore(L, ALPHA)
    store.push(subj)
    store.scan(q[:16], max_divergence=5)
    t = time.perf_counter(); rows = store.scan(q, max_divergence=5); dt = time.perf_counter() - t
    ms, nl = store.last_scan_ms()
    t = time.perf_counter(); best = store.scan(q, max_divergence=None, max_num_hits=1); dtb = time.perf_counter() - t
    t = time.perf_counter(); best = store.scan(q, max_divergence=None, max_num_hits=1); dtb = min(dtb, time.perf_counter() - t)
    bms, bnl = store.last_scan_ms()
    print("alphabet=%d L=%d %-10s filter=%s  N=%d Q=%d  rows=%d  host-api %.1f ms  (last scan kernel %.2f ms, %d launches)  best-hit rows=%d %.1f ms (last scan %.2f ms, %d launches)"
          % (ALPHA, L, name, os.environ.get("SMAFA_FILTER", "1"), len(subj), len(q), len(rows), dt * 1e3, ms, nl, len(best), dtb * 1e3, bms, bnl), flush=True)
    store.close()
