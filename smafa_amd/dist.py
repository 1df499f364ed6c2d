"""Multi-GPU `query`: one process per GPU over torch.distributed (backend "nccl" = RCCL on ROCm).

The query loop of the reference carries no state between records except the running query number
(/root/reference/src/lib.rs:232-318), so queries shard with NO data-path collective: the subject store is
replicated on every GPU, rank g takes the contiguous block [g*Q/G, (g+1)*Q/G) of the query file, scans and
selects its rows locally, and the only exchange is one gather of the finished row lists on rank 0
(counts, then rows padded to the longest list — RCCL has no gatherv).  Blocks are contiguous and in rank
order, so the concatenation is already in the reference's print order; the output is byte-identical for
any number of ranks.

Launch:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
             -m smafa_amd.dist query -d DB -q QUERIES [--max-divergence D] [--max-num-hits K] ...
"""
from __future__ import annotations

import os
import sys
from typing import Callable, Optional

import numpy as np

from . import api

ScanFn = Callable[[np.ndarray, np.ndarray, Optional[int], Optional[int]], np.ndarray]


def shard_bounds(n_queries: int, world: int, rank: int) -> tuple[int, int]:
    """contiguous block of rank `rank`: [rank*Q/G, (rank+1)*Q/G)"""
    return (rank * n_queries) // world, ((rank + 1) * n_queries) // world


class HipScanner:
    """The product scanner: the subject store is packed into HBM on `device` at first use and every call
    runs the HIP scan kernels through the C ABI."""

    def __init__(self, alphabet: int, device: int):
        self.alphabet, self.device, self.store = alphabet, device, None

    def __call__(self, subject_codes, query_codes, max_divergence, max_num_hits):
        if self.store is None:
            self.store = api.SubjectStore(subject_codes.shape[1], self.alphabet, self.device)
            self.store.push(subject_codes)
        return self.store.scan(query_codes, max_divergence, max_num_hits)


def gather_rows(rows: np.ndarray, dist, device=None) -> Optional[np.ndarray]:
    """Gather per-rank row lists (structured HIT_DTYPE arrays) on rank 0, in rank order."""
    import torch

    world, rank = dist.get_world_size(), dist.get_rank()
    count = torch.tensor([len(rows)], dtype=torch.int64, device=device)
    counts = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(counts, count)
    counts = [int(c.item()) for c in counts]
    width = max(max(counts), 1)
    padded = np.zeros((width, 3), dtype=np.int32)
    if len(rows):
        padded[: len(rows)] = rows.view(np.uint32).reshape(-1, 3).view(np.int32)
    mine = torch.from_numpy(padded).to(device) if device is not None else torch.from_numpy(padded)
    parts = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    if rank != 0:
        return None
    out = [p.cpu().numpy().view(np.uint32).reshape(-1, 3)[:c] for p, c in zip(parts, counts)]
    return np.ascontiguousarray(np.concatenate(out, axis=0)).view(api.HIT_DTYPE).reshape(-1)


def query_sharded(db_path: str, query_fasta: str, max_divergence: Optional[int] = None,
                  max_num_hits: Optional[int] = None, limit_per_sequence: Optional[int] = None, out_fd: int = 1,
                  scan_fn: Optional[ScanFn] = None, dist=None, device=None) -> None:
    """`smafa query` across the ranks of an initialised process group (src/lib.rs:198-325 semantics).

    `scan_fn(subject_codes, query_codes, max_divergence, k)` must return every row within the bounds ordered
    by (query, dist, subject); the default is the HIP scanner on this rank's GPU.  (Tests on CPU-only hosts
    inject a checker here; the product never does.)
    """
    if dist is None:
        import torch.distributed as dist  # type: ignore[no-redef]
    world, rank = dist.get_world_size(), dist.get_rank()
    alphabet, subj = api.read_db(db_path)
    queries = api.load_fastx(query_fasta, alphabet)
    n, L = subj.shape
    if len(queries) and n and queries.shape[1] != L:
        raise api.SmafaPanic(-6, "Cannot compute distances between seq of length %d and windows of lengths %d"
                             % (queries.shape[1], L))
    if scan_fn is None:
        scan_fn = HipScanner(alphabet, int(os.environ.get("LOCAL_RANK", rank)))
    lo, hi = shard_bounds(len(queries), world, rank)
    kmode = max_num_hits is not None and max_num_hits != 1
    dev_k = 1 if not kmode else (None if (max_num_hits == 0 or max_num_hits > n) else max_num_hits)
    mine = queries[lo:hi]
    if n and len(mine):
        hits = scan_fn(subj, mine, max_divergence, dev_k)
    else:
        hits = np.zeros(0, dtype=api.HIT_DTYPE)
    rows = api.select_rows(hits, len(mine), n, subj, L, max_divergence, max_num_hits, limit_per_sequence)
    rows = rows.copy()
    rows["query"] += lo  # global query numbers
    all_rows = gather_rows(rows, dist, device)
    if rank == 0:
        letters = np.array([ord(api.decode(np.array([c], dtype=np.uint8), alphabet)) for c in range(28 if alphabet else 5)],
                           dtype=np.uint8)
        chunks = []
        for r in all_rows:
            chunks.append(b"%d\t%d\t%d\t" % (r["query"], r["subject"], r["dist"]) + letters[subj[r["subject"]]].tobytes() + b"\n")
        os.write(out_fd, b"".join(chunks)) if chunks else None


def _main(argv) -> int:
    import argparse

    import torch
    import torch.distributed as dist

    ap = argparse.ArgumentParser(prog="smafa_amd.dist")
    ap.add_argument("command", choices=["query"])
    ap.add_argument("-d", "--database", required=True)
    ap.add_argument("-q", "--query", required=True)
    ap.add_argument("--max-divergence", type=int)
    ap.add_argument("--max-num-hits", type=int)
    ap.add_argument("--limit-per-sequence", type=int)
    ap.add_argument("--backend", default="nccl")
    a = ap.parse_args(argv)
    local = int(os.environ.get("LOCAL_RANK", "0"))
    device = None
    if a.backend == "nccl":
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
        dist.init_process_group("nccl", device_id=device)
    else:
        dist.init_process_group(a.backend)
    try:
        query_sharded(a.database, a.query, a.max_divergence, a.max_num_hits, a.limit_per_sequence, 1, dist=dist, device=device)
    except api.SmafaError as e:
        sys.stderr.write(str(e) + "\n")
        return 101 if isinstance(e, api.SmafaPanic) else 1
    finally:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(_main(sys.argv[1:]))
