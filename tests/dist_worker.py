"""Worker for tests/test_dist.py: one rank of a sharded `query` / `cluster` over torch.distributed.

Backend gloo on CPU: the per-rank scanner is the ORACLE (tests may call it; the product's default
scanner is HIP-only).  What is under test is the N>1 plumbing of smafa_amd.dist — contiguous shards,
global query numbers, host-side selection, gather in rank order, byte-identical output.
With --hip the product scanner is used instead (GPU box).
--mode gather checks the variable-size all-gather under the sharded cluster; --mode cluster runs
smafa_cluster_sharded itself (HIP only: the cluster driver has no injectable scanner)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch.distributed as dist  # noqa: E402

import smafa_amd  # noqa: E402
from smafa_amd import dist as sdist  # noqa: E402


def oracle_scan(subject_codes, query_codes, max_divergence, max_num_hits):
    import oracle

    L = subject_codes.shape[1]
    rows = oracle.scan_codes(subject_codes, query_codes, L if max_divergence is None else min(max_divergence, L))
    if max_num_hits is None:
        return rows
    keep, i = [], 0
    while i < len(rows):
        j = i
        while j < len(rows) and rows[j]["query"] == rows[i]["query"]:
            j += 1
        grp = rows[i:j]
        kth = grp[max_num_hits - 1]["dist"] if len(grp) >= max_num_hits else 0xFFFFFFFF
        keep.append(grp[grp["dist"] <= kth])
        i = j
    return np.concatenate(keep) if keep else rows[:0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", choices=["query", "gather", "cluster"], default="query")
    ap.add_argument("--db")
    ap.add_argument("--queries")
    ap.add_argument("--input")
    ap.add_argument("--alphabet", type=int, default=0)
    ap.add_argument("--out", required=True)
    ap.add_argument("--max-divergence", type=int)
    ap.add_argument("--max-num-hits", type=int)
    ap.add_argument("--limit-per-sequence", type=int)
    ap.add_argument("--hip", action="store_true")
    a = ap.parse_args()
    dist.init_process_group("gloo")
    if a.mode == "gather":
        # rank r contributes r*5+1 bytes in round 0, nothing from rank 0 in round 1, 70 001 bytes in round 2
        rank, world = dist.get_rank(), dist.get_world_size()
        got = []
        for rnd, size in enumerate([rank * 5 + 1, 0 if rank == 0 else 3, 70001]):
            mine = ((np.arange(size) * 7 + rank * 13 + rnd) % 251).astype(np.uint8)
            got.append(sdist.allgather_bytes(mine, dist))
        np.save(a.out + ".rank%d.npy" % rank, np.concatenate(got))
        dist.destroy_process_group()
        return
    if a.mode == "cluster":
        fd = os.open(a.out, os.O_WRONLY | os.O_CREAT | os.O_TRUNC) if dist.get_rank() == 0 else -1
        try:  # every rank shares GPU 0 of the one-GPU test box
            sdist.cluster_sharded(a.input, a.max_divergence, fd if fd >= 0 else 1, a.alphabet, dist=dist, gpu=0)
        finally:
            if fd >= 0:
                os.close(fd)
            dist.destroy_process_group()
        return
    fd = os.open(a.out, os.O_WRONLY | os.O_CREAT | os.O_TRUNC) if dist.get_rank() == 0 else -1
    scan_fn = oracle_scan
    if a.hip:  # the product path (smafa_qsession_*); every rank shares GPU 0 of the one-GPU test box
        scan_fn = None
    try:
        sdist.query_sharded(a.db, a.queries, a.max_divergence, a.max_num_hits, a.limit_per_sequence,
                            out_fd=fd if fd >= 0 else 1, scan_fn=scan_fn, dist=dist, gpu=0)
    except smafa_amd.SmafaPanic as e:  # the reference's panic: message on stderr, exit 101 (like the CLI)
        sys.stderr.write(str(e) + "\n")
        sys.exit(101)
    except smafa_amd.SmafaError as e:  # an Err out of main: exit 1
        sys.stderr.write(str(e) + "\n")
        sys.exit(1)
    finally:
        if fd >= 0:
            os.close(fd)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
