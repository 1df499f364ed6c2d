"""Synthetic workloads of SURVEY.md §8(d) (fixed seeds; numpy PCG64).

Subjects: n rows of `seq_len` symbols uniform over the alphabet's letters, 1 % of the rows
overwritten with copies of earlier rows (ties / multiple minima).  Queries: a uniformly drawn subject
with s substitutions at distinct positions, each to a different letter, s uniform in 0..max_subs — so a
query has a hit within divergence D iff s <= D (random 60-mers are far apart).
Returns CODE bytes (what crosses the C ABI) and the letters used.
"""
from __future__ import annotations

import numpy as np

AA_LETTERS = b"ACDEFGHIKLMNPQRSTVWY"
NT_LETTERS = b"ACGT"


def letter_codes(alphabet: int) -> np.ndarray:
    """code bytes of the letters the generators draw from (AA: 'A'..'Z' -> 0..25; NT: ACGT -> 0..3)"""
    if alphabet == 1:
        return np.frombuffer(AA_LETTERS, dtype=np.uint8) - ord("A")
    return np.arange(4, dtype=np.uint8)


def subjects(n: int, seq_len: int = 60, alphabet: int = 1, seed: int = 1, dup_frac: float = 0.01,
             n_frac: float = 0.0) -> np.ndarray:
    rng = np.random.default_rng(seed)
    lc = letter_codes(alphabet)
    out = lc[rng.integers(0, len(lc), size=(n, seq_len), dtype=np.uint8)]
    if n_frac > 0 and alphabet == 0:  # nucleotide variant B: each column -> N with probability n_frac
        out[rng.random(size=out.shape) < n_frac] = 4
    n_dup = int(n * dup_frac)
    if n_dup and n > 1:
        dst = rng.integers(1, n, size=n_dup)
        src = (rng.random(size=n_dup) * dst).astype(np.int64)  # an EARLIER row
        out[dst] = out[src]
    return out


def related_subjects(n_families: int, members: int = 100, seq_len: int = 60, alphabet: int = 1, seed: int = 7,
                     div_lo: float = 0.10, div_hi: float = 0.25) -> np.ndarray:
    """A store of homologous sequences: n_families uniform roots, `members` members each; a member is its root
    with every column substituted (to a different letter) with the member's own probability, uniform in
    [div_lo, div_hi]; rows shuffled.  Two members of a family then differ in roughly 20-45 % of their columns —
    close enough that a one-plane lower bound cannot separate them cheaply, far enough that few are hits."""
    rng = np.random.default_rng(seed)
    lc = letter_codes(alphabet)
    roots = rng.integers(0, len(lc), size=(n_families, seq_len), dtype=np.uint8)  # indices into lc
    n = n_families * members
    out = np.empty((n, seq_len), dtype=np.uint8)
    step = max(1, (1 << 20) // members)  # families per chunk (~1M rows)
    for f0 in range(0, n_families, step):
        f1 = min(n_families, f0 + step)
        idx = np.repeat(roots[f0:f1], members, axis=0)
        rows = len(idx)
        div = rng.uniform(div_lo, div_hi, size=rows)
        hit = rng.integers(0, 1 << 16, size=(rows, seq_len), dtype=np.uint16) < (div * 65536.0).astype(np.uint32)[:, None]
        shift = rng.integers(1, len(lc), size=(rows, seq_len), dtype=np.uint8)
        idx = np.where(hit, (idx + shift) % len(lc), idx).astype(np.uint8)
        out[f0 * members:f1 * members] = lc[idx]
    rng.shuffle(out, axis=0)
    return out


def queries(subject_codes: np.ndarray, q: int, alphabet: int = 1, seed: int = 3, max_subs: int = 10):
    """-> (query codes, planted subject row, number of substitutions)"""
    rng = np.random.default_rng(seed)
    n, L = subject_codes.shape
    lc = letter_codes(alphabet)
    rows = rng.integers(0, n, size=q)
    out = subject_codes[rows].copy()
    subs = rng.integers(0, max_subs + 1, size=q)
    # distinct positions: the first s entries of a random permutation per query
    perm = np.argsort(rng.random(size=(q, L)), axis=1)
    shift = rng.integers(1, len(lc), size=(q, L))  # change to a DIFFERENT letter of the generator's set
    inv = np.full(256, 0, dtype=np.int64)
    inv[lc] = np.arange(len(lc))
    for i in range(q):
        pos = perm[i, : subs[i]]
        cur = out[i, pos]
        ok = cur < 255
        idx = inv[cur]
        new = lc[(idx + shift[i, pos]) % len(lc)]
        # a column holding a symbol outside the generator's set (an N) is replaced by a letter: still a change
        out[i, pos] = np.where(np.isin(cur, lc) & ok, new, lc[shift[i, pos] % len(lc)])
    return out, rows, subs


def cluster_records(n_roots: int, members: int, seq_len: int = 60, alphabet: int = 1, seed: int = 4,
                    max_subs: int = 4) -> np.ndarray:
    """SURVEY.md §8(d) cluster workload: n_roots uniform roots x `members` members, each member = its root
    with s in 0..max_subs substitutions, Fisher-Yates shuffled.  Returns code rows."""
    rng = np.random.default_rng(seed)
    lc = letter_codes(alphabet)
    roots = lc[rng.integers(0, len(lc), size=(n_roots, seq_len), dtype=np.uint8)]
    recs = np.repeat(roots, members, axis=0)
    n = len(recs)
    subs = rng.integers(0, max_subs + 1, size=n)
    for s in range(1, max_subs + 1):  # s-th substitution for every record that has at least s
        rows = np.nonzero(subs >= s)[0]
        cols = rng.integers(0, seq_len, size=len(rows))
        recs[rows, cols] = lc[rng.integers(0, len(lc), size=len(rows))]
    rng.shuffle(recs, axis=0)
    return recs


def write_fasta(path: str, code_rows: np.ndarray, alphabet: int = 1, prefix: str = "r") -> None:
    """code rows -> single-line FASTA (vectorised; fine for millions of rows)"""
    if alphabet == 1:
        letters = np.array([ord("A") + i for i in range(26)] + [ord("*"), ord("-")], dtype=np.uint8)
    else:
        letters = np.frombuffer(b"ACGTN", dtype=np.uint8)
    n, L = code_rows.shape
    ascii_rows = letters[code_rows]
    with open(path, "wb") as f:
        step = 200_000
        for lo in range(0, n, step):
            hi = min(n, lo + step)
            parts = []
            for i in range(lo, hi):
                parts.append(b">%s%d\n" % (prefix.encode(), i))
                parts.append(ascii_rows[i].tobytes())
                parts.append(b"\n")
            f.write(b"".join(parts))
