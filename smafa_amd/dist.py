"""Multi-GPU `query` and `cluster`: one process per GPU over torch.distributed (backend "nccl" = RCCL on ROCm).

The query loop of the reference carries no state between records except the running query number
(/root/reference/src/lib.rs:232-318), so queries shard with NO data-path collective: the subject store is
replicated on every GPU, rank g takes the contiguous block [g*Q/G, (g+1)*Q/G) of the query file, scans and
selects its rows locally, and the only exchange is one gather of the finished row lists to rank 0
(counts, then rows padded to the longest list — RCCL has no gatherv).  Blocks are contiguous and in rank
order, so the concatenation is already in the reference's print order; the output is byte-identical for
any number of ranks.

`cluster` is sequential across records (/root/reference/src/cluster.rs:35-85) but exact in batches: every rank
keeps a replica of the centroid store, scans its slice of each batch, and two small all-gathers per batch
(nearest old centroid per record; in-range (record, candidate) rows) give every rank what it needs to resolve
the batch identically — see smafa_cluster_sharded in include/smafa_amd.h.  The batch logic is the C++ driver's;
this module only supplies the transport.

Launch:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
             -m smafa_amd.dist query -d DB -q QUERIES [--max-divergence D] [--max-num-hits K] ...
         python -m torch.distributed.run ... -m smafa_amd.dist cluster -i INPUT -d D [--alphabet aa]
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


def _lib_err_other_rank() -> int:
    from . import _lib

    return _lib.ERR_DEVICE


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
    """Gather per-rank row lists (structured HIT_DTYPE arrays) on rank 0, in rank order: the row counts first
    (one tiny all_gather, so that every rank knows the padded width), then ONE gather to rank 0 of the lists padded
    to the longest (RCCL has no gatherv) — each rank's rows cross its own xGMI link to the root once, nothing goes
    to the other ranks."""
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
    parts = [torch.zeros_like(mine) for _ in range(world)] if rank == 0 else None
    dist.gather(mine, parts, dst=0)
    if rank != 0:
        return None
    out = [p.cpu().numpy().view(np.uint32).reshape(-1, 3)[:c] for p, c in zip(parts, counts)]
    return np.ascontiguousarray(np.concatenate(out, axis=0)).view(api.HIT_DTYPE).reshape(-1)


def allgather_bytes(mine: np.ndarray, dist, device=None) -> np.ndarray:
    """Blocks of bytes of different sizes, one per rank -> their concatenation in rank order, on every rank
    (sizes first, then the blocks padded to the longest: RCCL has no allgatherv)."""
    import torch

    world = dist.get_world_size()
    mine = np.ascontiguousarray(mine, dtype=np.uint8).reshape(-1)
    count = torch.tensor([len(mine)], dtype=torch.int64, device=device)
    counts = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(counts, count)
    counts = [int(c.item()) for c in counts]
    width = max(max(counts), 1)
    padded = np.zeros(width, dtype=np.uint8)
    padded[: len(mine)] = mine
    block = torch.from_numpy(padded).to(device) if device is not None else torch.from_numpy(padded)
    parts = [torch.zeros_like(block) for _ in range(world)]
    dist.all_gather(parts, block)
    return np.concatenate([p.cpu().numpy()[:c] for p, c in zip(parts, counts)])


def cluster_sharded(input_fasta: str, max_divergence: int, out_fd: int = 1, alphabet: int = api.ALPHABET_NT,
                    dist=None, device=None, gpu: Optional[int] = None) -> None:
    """`smafa cluster` across the ranks of an initialised process group (src/cluster.rs:13-94 semantics; output
    bytes do not depend on the number of ranks).  Rank 0 writes to out_fd.  `gpu` = HIP device of this rank
    (default LOCAL_RANK)."""
    import ctypes as C

    from . import _lib

    if dist is None:
        import torch.distributed as dist  # type: ignore[no-redef]
    world, rank = dist.get_world_size(), dist.get_rank()
    hold = {}

    def gather(ctx, send, n, recv, recv_n):
        try:
            mine = np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), shape=(int(n),)) if n else np.zeros(0, np.uint8)
            hold["buf"] = out = allgather_bytes(mine, dist, device)  # stays alive until the next call
            recv[0] = out.ctypes.data if len(out) else None
            recv_n[0] = len(out)
            return 0
        except Exception as e:  # never unwind through the C frames
            sys.stderr.write("allgather failed on rank %d: %r\n" % (rank, e))
            return 1

    cb = _lib.ALLGATHER_FN(gather)
    if gpu is None:
        gpu = int(os.environ.get("LOCAL_RANK", rank))
    api.check(_lib.lib().smafa_cluster_sharded(os.fsencode(input_fasta), int(max_divergence), out_fd, gpu, alphabet,
                                                rank, world, cb, None))


class QuerySession:
    """smafa_qsession_*: the DB opened once by this process (a packed store file is mapped and copied to HBM — no decode,
    no host code rows), shares of a query file answered, rows printed by rank 0."""

    def __init__(self, db_path: str, device: int):
        import ctypes as C

        from . import _lib

        self._lib, self._C = _lib, C
        self._h = C.c_void_p()
        api.check(_lib.lib().smafa_qsession_open(C.byref(self._h), os.fsencode(db_path), int(device)))

    def scan_part(self, query_fasta, max_divergence, max_num_hits, limit_per_sequence, part, parts, whole_file):
        """-> (rows numbered from 0 within the share, records answered, records in front or None, pending error or None,
        retry_whole)"""
        C, l = self._C, self._lib.lib()
        rows_p, n_rows, n_q, n_before = C.c_void_p(), C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        pending, retry = C.c_int(0), C.c_int(0)
        opt = lambda v: self._lib.NONE if v is None else int(v)
        api.check(l.smafa_qsession_scan_part(self._h, os.fsencode(query_fasta), opt(max_divergence), opt(max_num_hits),
                                             opt(limit_per_sequence), part, parts, 1 if whole_file else 0, C.byref(rows_p),
                                             C.byref(n_rows), C.byref(n_q), C.byref(n_before), C.byref(pending), C.byref(retry)))
        err = None
        if pending.value != self._lib.OK:
            err = (pending.value, l.smafa_last_error().decode(errors="replace"))
        try:
            n = n_rows.value
            rows = np.zeros(n, dtype=api.HIT_DTYPE)
            if n:
                C.memmove(rows.ctypes.data, rows_p, n * 12)
        finally:
            l.smafa_free(rows_p)
        before = None if n_before.value == 0xFFFFFFFFFFFFFFFF else int(n_before.value)
        return rows, int(n_q.value), before, err, bool(retry.value)

    def write(self, rows: np.ndarray, out_fd: int) -> None:
        rows = np.ascontiguousarray(rows, dtype=api.HIT_DTYPE)
        api.check(self._lib.lib().smafa_qsession_write(self._h, rows.ctypes.data, len(rows), out_fd))

    def close(self):
        if self._h:
            self._lib.lib().smafa_qsession_close(self._h)
            self._h = self._C.c_void_p()


def _raise(code: int, msg: str):
    from . import _lib

    raise (api.SmafaPanic if code == _lib.ERR_PANIC else api.SmafaError)(code, msg)


def _query_sharded_native(db_path, query_fasta, max_divergence, max_num_hits, limit_per_sequence, out_fd, dist, device, gpu):
    """The product path: every rank opens the DB itself (a packed store file: mmap + copies), parses ONLY its byte range of
    the query file, scans and selects on its GPU; what is exchanged is one small state vector per rank, then one gather of
    the finished row lists to rank 0, which prints them (subject strings decoded for the hit rows only)."""
    import torch

    world, rank = dist.get_world_size(), dist.get_rank()

    def exchange(failure, retry, nq, pending):
        """-> per-rank [failed, retry, records answered, pending code]; any failure's text is passed around and raised on
        EVERY rank (nobody is left waiting in the gather)"""
        mine = torch.tensor([1 if failure else 0, 1 if retry else 0, nq, pending[0] if pending else 0], dtype=torch.int64, device=device)
        parts = [torch.zeros(4, dtype=torch.int64, device=device) for _ in range(world)]
        dist.all_gather(parts, mine)
        state = [[int(x) for x in p.cpu().tolist()] for p in parts]
        if any(st[0] for st in state):
            text = (str(failure).encode() if failure else b"")[:2000]
            code = np.array([failure.code if failure else 0], dtype=np.int32).tobytes()
            blob = allgather_bytes(np.frombuffer(code + len(text).to_bytes(4, "little") + text, dtype=np.uint8), dist, device)
            off = 0
            for st in state:  # blocks in rank order; the first failing rank's error is everybody's
                c = int(np.frombuffer(blob[off:off + 4].tobytes(), dtype=np.int32)[0])
                n = int.from_bytes(blob[off + 4:off + 8].tobytes(), "little")
                if st[0]:
                    _raise(c, blob[off + 8:off + 8 + n].tobytes().decode(errors="replace"))
                off += 8 + n
        return state

    sess, failure = None, None
    rows, nq, before, pending, retry = np.zeros(0, dtype=api.HIT_DTYPE), 0, None, None, False
    try:
        sess = QuerySession(db_path, gpu)
        rows, nq, before, pending, retry = sess.scan_part(query_fasta, max_divergence, max_num_hits, limit_per_sequence,
                                                          rank, world, False)
    except api.SmafaError as e:
        failure = e
    try:
        state = exchange(failure, retry, nq, pending)
        whole = any(st[1] for st in state)
        if whole:  # gzip input, or a cut that did not hold: everybody parses the whole file and takes its block by count
            try:
                rows, nq, before, pending, retry = sess.scan_part(query_fasta, max_divergence, max_num_hits,
                                                                  limit_per_sequence, rank, world, True)
            except api.SmafaError as e:
                failure = e
            state = exchange(failure, False, nq, pending)
        if whole:
            offset, first_bad = before, None  # the pending error is the same on every rank; every share is in front of it
        else:
            offset = sum(st[2] for st in state[:rank])
            bad = [r for r, st in enumerate(state) if st[3]]
            first_bad = bad[0] if bad else None
            if first_bad is not None and rank > first_bad:  # the reference never got this far (src/lib.rs:232-318)
                rows = rows[:0]
        rows = rows.copy()
        rows["query"] += offset
        all_rows = gather_rows(rows, dist, device)
        if rank == 0:
            sess.write(all_rows, out_fd)
        # the record the reference's loop fails on, reported after the rows in front of it — by every rank
        if whole and pending is not None:
            _raise(*pending)
        if first_bad is not None:
            text = pending[1].encode()[:2000] if (pending and rank == first_bad) else b""
            blob = allgather_bytes(np.frombuffer(text, dtype=np.uint8), dist, device)
            _raise(state[first_bad][3], blob.tobytes().decode(errors="replace"))
    finally:
        if sess is not None:
            sess.close()


def query_sharded(db_path: str, query_fasta: str, max_divergence: Optional[int] = None,
                  max_num_hits: Optional[int] = None, limit_per_sequence: Optional[int] = None, out_fd: int = 1,
                  scan_fn: Optional[ScanFn] = None, dist=None, device=None, gpu: Optional[int] = None) -> None:
    """`smafa query` across the ranks of an initialised process group (src/lib.rs:198-325 semantics).

    Default (scan_fn None): the product path — smafa_qsession_* on this rank's GPU: the DB opened once per rank without
    host code rows for a packed store, only this rank's byte range of the query file parsed, rows gathered on rank 0.
    `scan_fn(subject_codes, query_codes, max_divergence, k)` (every row within the bounds ordered by (query, dist,
    subject)) replaces the device scan for tests on CPU-only hosts, which then exercise the sharding and the gather over
    host code rows; the product never passes one.
    """
    if dist is None:
        import torch.distributed as dist  # type: ignore[no-redef]
    world, rank = dist.get_world_size(), dist.get_rank()
    if scan_fn is None:  # the product path
        return _query_sharded_native(db_path, query_fasta, max_divergence, max_num_hits, limit_per_sequence, out_fd, dist,
                                     device, int(os.environ.get("LOCAL_RANK", rank)) if gpu is None else gpu)
    # ---- test scaffolding (CPU-only hosts): the same sharding and gather with an injected scanner over host code rows
    alphabet, subj = api.read_db(db_path)
    # the queries in front of a bad record are answered before the failure is reported, as the reference's loop does
    queries, pending = api.load_fastx_partial(query_fasta, alphabet)
    n, L = subj.shape
    if len(queries) and n and queries.shape[1] != L:  # the FIRST query already fails the length check: nothing is printed
        raise api.SmafaPanic(-6, "Cannot compute distances between seq of length %d and windows of lengths %d"
                             % (queries.shape[1], L))
    if pending is not None and "seq of length" in str(pending) and n:  # a later record of another length: say the store's
        got = str(pending).split("seq of length ")[1].split(" ")[0]
        pending = api.SmafaPanic(-6, "Cannot compute distances between seq of length %s and windows of lengths %d" % (got, L))
    if scan_fn is None:
        scan_fn = HipScanner(alphabet, int(os.environ.get("LOCAL_RANK", rank)) if gpu is None else gpu)
    # The reference's input-independent panics (empty store, k = 0, --limit-per-sequence without k > 1;
    # src/lib.rs:254-255,298,301-303) are raised HERE, on every rank alike and before anything is sharded: a rank whose
    # shard is empty would otherwise sail past them into the gather while the others have already left the group.
    needs_panic = n == 0 or max_num_hits == 0 or (limit_per_sequence is not None and max_num_hits in (None, 1))
    if len(queries) and needs_panic:  # raises the reference's panic, identically on every rank
        api.select_rows(np.zeros(0, dtype=api.HIT_DTYPE), 1, n, subj, L, max_divergence, max_num_hits, limit_per_sequence)
    lo, hi = shard_bounds(len(queries), world, rank)
    kmode = max_num_hits is not None and max_num_hits != 1
    dev_k = 1 if not kmode else (None if (max_num_hits == 0 or max_num_hits > n) else max_num_hits)
    mine = queries[lo:hi]
    failure = None
    try:
        if n and len(mine):
            hits = scan_fn(subj, mine, max_divergence, dev_k)
        else:
            hits = np.zeros(0, dtype=api.HIT_DTYPE)
        rows = api.select_rows(hits, len(mine), n, subj, L, max_divergence, max_num_hits, limit_per_sequence)
        rows = rows.copy()
        rows["query"] += lo  # global query numbers
    except api.SmafaError as e:  # a failure only this rank sees (device trouble ...): tell the others before the gather
        failure = e
        rows = np.zeros(0, dtype=api.HIT_DTYPE)
    import torch

    flag = torch.tensor([1 if failure else 0], dtype=torch.int64, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    if failure:
        raise failure
    if int(flag.item()):
        raise api.SmafaError(_lib_err_other_rank(), "another rank failed; no rows written")
    all_rows = gather_rows(rows, dist, device)
    if rank == 0:
        api.write_rows(all_rows, subj, alphabet, out_fd)
    if pending is not None:  # identical on every rank (same file, same loader)
        raise pending


def _main(argv) -> int:
    import argparse

    import torch
    import torch.distributed as dist

    ap = argparse.ArgumentParser(prog="smafa_amd.dist")
    sub = ap.add_subparsers(dest="command", required=True)
    q = sub.add_parser("query")  # flags of src/main.rs:64-95
    q.add_argument("-d", "--database", required=True)
    q.add_argument("-q", "--query", required=True)
    q.add_argument("--max-divergence", type=int)
    q.add_argument("--max-num-hits", type=int)
    q.add_argument("--limit-per-sequence", type=int)
    c = sub.add_parser("cluster")  # flags of src/main.rs:96-108
    c.add_argument("-i", "--input", required=True)
    c.add_argument("-d", "--max-divergence", type=int, required=True)
    c.add_argument("--alphabet", choices=["nt", "aa"], default="nt")
    for p in (q, c):
        p.add_argument("--backend", default="nccl")
        p.add_argument("-o", "--output", help="rank 0 writes the rows here instead of stdout")
        p.add_argument("--single-device", action="store_true", help="every rank uses GPU 0 (rehearsals on a 1-GPU box)")
        p.add_argument("-v", "--verbose", action="store_true", help="rank 0 logs the drivers' debug lines to stderr")
    a = ap.parse_args(argv)
    if a.verbose and int(os.environ.get("RANK", "0")) == 0:
        from . import _lib
        _lib.lib().smafa_set_verbosity(2)
    local = int(os.environ.get("LOCAL_RANK", "0"))
    device = None
    if a.backend == "nccl":
        local = 0 if a.single_device else local
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
        dist.init_process_group("nccl", device_id=device)
    else:
        dist.init_process_group(a.backend)
    fd = 1
    if a.output and int(os.environ.get("RANK", "0")) == 0:
        fd = os.open(a.output, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    try:
        if a.command == "cluster":
            cluster_sharded(a.input, a.max_divergence, fd, api.ALPHABET_AA if a.alphabet == "aa" else api.ALPHABET_NT,
                            dist=dist, device=device, gpu=0 if a.single_device else local)
        else:
            query_sharded(a.database, a.query, a.max_divergence, a.max_num_hits, a.limit_per_sequence, fd, dist=dist,
                          device=device, gpu=0 if a.single_device else local)
    except api.SmafaError as e:
        sys.stderr.write(str(e) + "\n")
        return 101 if isinstance(e, api.SmafaPanic) else 1
    finally:
        if fd != 1:
            os.close(fd)
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(_main(sys.argv[1:]))
