"""Randomised differential soak on the GPU box: random shapes, alphabets, letter skews, family structure, bounds,
k-th modes AND kernel forms (the SMAFA_* switches are drawn per store), HIP scan vs the oracle, for SOAK_SECONDS.
Prints the failing configuration and exits 1 on the first difference.  (Checker only: the oracle is test infrastructure.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle, smafa_amd

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

seed0 = int(os.environ.get("SOAK_SEED", str(int(time.time()))))
budget = float(os.environ.get("SOAK_SECONDS", "120"))
oracle.build()
t_end, rounds, scans = time.time() + budget, 0, 0
SWITCHES = {"SMAFA_FILTER": ["1", "1", "1", "0"], "SMAFA_LAZY": ["1", "1", "0"], "SMAFA_TILES": ["", "1", "2", "4"],
            "SMAFA_NT_PLANES": ["", "", "3"], "SMAFA_WIDE_FROM": ["5", "5", "3"], "SMAFA_WIDE_ONE": ["1", "1", "0"],
            "SMAFA_TWO_PHASE": ["1", "1", "0"], "SMAFA_COUNT_FIRST_K": ["3", "3", "2", "1000000"],
            "SMAFA_ZONE": ["1", "1", "2", "2", "0"], "SMAFA_SORT": ["1", "1", "0"], "SMAFA_LAYOUT": ["1", "1", "0"],
            "SMAFA_RESORT": ["1", "1", "1", "0"], "SMAFA_RESORT_MIN": ["", "2", "300", "5000"], "SMAFA_FOLD3": ["1", "1", "0"], "SMAFA_STREAM_NT": ["1", "1", "0"],
            # round 4: the k-th modes' counting pass over a sample of the tiles (forced onto small stores), or over everything
            "SMAFA_ZONE_DIRECT": ["1", "1", "0"], "SMAFA_LADDER_PROBE": ["1", "1", "0"], "SMAFA_LAZY_FOLD": ["1", "1", "0"], "SMAFA_KTH_SAMPLE": ["16", "2", "4", "0"], "SMAFA_KTH_HIST_SEED": ["1", "1", "0"], "SMAFA_KTH_SAMPLE_MIN_TILES": ["2", "8", "4096"],
            # the block index: built by the first big fixed-bound scan (mode 2) or on request below; forced onto dense stores too
            "SMAFA_INDEX": ["1", "1", "2", "2", "0"], "SMAFA_INDEX_MIN_ROWS": ["1", "1000"], "SMAFA_INDEX_MAX_RUN": ["", "100000000"],
            "SMAFA_INDEX_CAND": ["", "100"]}
print("soak seed", seed0, flush=True)
t_note = time.time()
while time.time() < t_end:
    rng = np.random.default_rng(seed0 + rounds)
    rounds += 1
    alphabet = int(rng.integers(0, 2))
    n_letters = int(rng.choice([2, 4, 5] if alphabet == 0 else [2, 4, 20, 28]))
    L = int(rng.choice([1, 2, 7, 12, 20, 31, 32, 33, 60, 60, 60, 64, 65, 90, 96, 128, 129, 150, 200, 257]))
    n = int(rng.choice([1, 3, 255, 256, 257, 1000, 4097, 9000, 20000]))
    nq = int(rng.choice([1, 2, 17, 64, 65, 300, 300]))
    if n <= 4097 and rng.random() < 0.12:
        nq = int(rng.choice([2100, 2600]))  # enough open queries for the near-hit ladder to be planned from a sample
    env = {k: str(rng.choice(v)) for k, v in SWITCHES.items()}
    for k, v in env.items():
        if v: os.environ[k] = v
        else: os.environ.pop(k, None)
    # subjects: unrelated, or families around a few roots (dense / related neighbourhoods), plus exact copies
    mode = int(rng.integers(0, 3))
    if mode == 0:
        s = rng.integers(0, n_letters, size=(n, L), dtype=np.uint8)
    else:
        roots = rng.integers(0, n_letters, size=(max(1, n // int(rng.choice([5, 50, 500]))), L), dtype=np.uint8)
        s = roots[rng.integers(0, len(roots), size=n)].copy()
        mut = rng.random(size=s.shape) < float(rng.choice([0.02, 0.1, 0.25]))
        s[mut] = rng.integers(0, n_letters, size=int(mut.sum()), dtype=np.uint8)
    if n > 4:
        s[n // 2] = s[1]; s[n - 1] = s[1]
    q = s[rng.integers(0, n, size=nq)].copy()
    for r in q:
        for _ in range(int(rng.integers(0, min(L, 9) + 1))):
            r[rng.integers(0, L)] = rng.integers(0, n_letters)
    if nq > 3:
        q[0] = rng.integers(0, n_letters, size=L, dtype=np.uint8)
    handles = int(rng.choice([0, 0, 0, 0, 2, 3]))  # >0: the same store behind a group of handles (all on GPU 0)
    store = smafa_amd.SubjectGroup(L, alphabet, devices=[0] * handles) if handles else smafa_amd.SubjectStore(L, alphabet)
    cuts = sorted(set([0, n] + [int(x) for x in rng.integers(0, n + 1, size=int(rng.integers(0, 3)))]))
    for a, b in zip(cuts[:-1], cuts[1:]):
        store.push(s[a:b])
    for scan_no in range(4):
        if scan_no == 3:  # the store grows between scans (a re-sort of the whole store may follow)
            if rng.random() < 0.4:
                break
            extra = s[rng.integers(0, n, size=int(rng.choice([1, 300, max(1, n // 2), n])))].copy()
            flip = rng.random(size=extra.shape) < 0.05
            extra[flip] = rng.integers(0, n_letters, size=int(flip.sum()), dtype=np.uint8)
            store.push(extra)
            s = np.concatenate([s, extra])
        D = None if rng.random() < 0.25 else int(rng.integers(0, L + 1)) if rng.random() < 0.3 else int(rng.integers(0, min(L, 9) + 1))
        k = None if rng.random() < 0.4 else int(rng.choice([1, 1, 2, 3, 10, 400]))
        if D is None and k is None:
            D = int(rng.integers(0, min(L, 6) + 1))
        if not handles and (D is None or D < min(L, 32)) and L <= 128 and rng.random() < 0.3:
            # (a later append leaves it stale: the scan kernels answer then; without a bound the index answers the ladder's first step)
            store.build_index(int(rng.integers(0 if D is None else D, min(L, 32))))
        got = store.scan(q, max_divergence=D, max_num_hits=k)
        full = oracle.scan_codes(s, q, L if D is None else D)
        want = full if k is None else expected_with_k(full, k)
        scans += 1
        if got.tobytes() != want.tobytes():
            print("MISMATCH round", rounds - 1, "seed", seed0, dict(alphabet=alphabet, n_letters=n_letters, L=L, n=n, nq=nq, mode=mode, D=D, k=k, handles=handles,
                  env=env, got=len(got), want=len(want), plan=None if handles else store.last_scan_plan()), flush=True)
            sys.exit(1)
    store.close()
    if time.time() - t_note > 60:  # a line a minute: a silent run is taken for a hung one on the GPU box
        t_note = time.time()
        print("  %d stores, %d scans so far" % (rounds, scans), flush=True)
print("soak ok: %d stores, %d scans in %.0f s" % (rounds, scans, budget), flush=True)
