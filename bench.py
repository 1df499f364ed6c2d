#!/usr/bin/env python3
"""bench.py — headline benchmark of the smafa scan engine on MI355X.

Metric (BASELINE.json): query seqs/sec (+ DB residues/sec) vs the HBM roofline, 10M x 60-aa DB, d <= 5.
A "step" = one pass of the hot path over one batch: every query of the batch (default 10 000 per GPU)
scanned against the whole resident subject store, qualifying rows appended on the device, and — when
more than one GPU takes part — the per-rank row lists gathered on rank 0 over RCCL.  The packed subject
block and the packed query batch are resident in HBM before the timed region starts.

    python bench.py [--gpus N] [--steps K] [--warmup W]            # N = 1
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  Scaling is weak: the DB is replicated, every rank scans its own query
shard of fixed size.  `roofline` is for the dominant kernel (scan_kernel), timed with HIP events on the
launch stream; `cpu_baseline` is the oracle's single-thread port of the reference's per-query loop on a
bounded sample (the only place bench.py touches oracle/, besides the post-run result check).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
VALU_PEAK_LANE_OPS = 256 * 4 * 32 * 2.4e9  # 256 CUs x 4 SIMD-32 x 2.4 GHz (one wave64 VALU op = 2 cycles)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--db-rows", type=int, default=10_000_000)
    ap.add_argument("--queries", type=int, default=10_000, help="queries per GPU per step")
    ap.add_argument("--seq-len", type=int, default=60)
    ap.add_argument("--alphabet", choices=["aa", "nt"], default="aa")
    ap.add_argument("--max-div", type=int, default=5)
    ap.add_argument("--n-frac", type=float, default=0.0,
                    help="nt only: each column becomes N with this probability (SURVEY 8d variant B: 0.001, 3-plane store)")
    ap.add_argument("--query-block", type=int, default=0, help="queries per workgroup pass (0 = automatic)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stream", action="store_true", help="skip the one-query-per-pass leg (clean rocprof stats)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (tests use gloo)")
    ap.add_argument("--single-device", action="store_true", help="testing only: every rank uses GPU 0")
    args = ap.parse_args()

    import torch

    import smafa_amd
    from smafa_amd import synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available() or smafa_amd.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device: the scan engine has no CPU fallback")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist

        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    alphabet = smafa_amd.ALPHABET_AA if args.alphabet == "aa" else smafa_amd.ALPHABET_NT
    L, N, Q, D = args.seq_len, args.db_rows, args.queries, args.max_div
    max_subs = 10 if alphabet == smafa_amd.ALPHABET_AA else 6

    # ---- synthetic workload: identical DB on every rank, disjoint query shards
    t_gen = time.time()
    subj = synth.subjects(N, L, alphabet, seed=1 if alphabet else 2, n_frac=args.n_frac)  # seeds of SURVEY 8d
    all_q, planted_row, planted_subs = synth.queries(subj, Q * world, alphabet, seed=3, max_subs=max_subs)
    q_lo = rank * Q
    my_q = all_q[q_lo:q_lo + Q]
    t_gen = time.time() - t_gen

    # ---- residency: pack + upload once
    t_up = time.time()
    store = smafa_amd.SubjectStore(L, alphabet, local_rank)
    store.push(subj)
    qset = smafa_amd.QuerySet(store, my_q)
    t_up = time.time() - t_up
    info = store.info()
    if args.query_block:
        store.set_query_block(args.query_block)
    stream = torch.cuda.Stream(device=dev)  # a real (non-null) HIP stream shared by torch and the library
    torch.cuda.set_stream(stream)
    store.set_stream(stream.cuda_stream)  # launches go to torch's stream: torch events see them

    cap = max(4 * Q, 1 << 16)
    # one buffer per step parity: [count (u64) | pad | rows], so that ONE collective moves count and rows together
    HEAD = 4  # int32 words before the rows (the count lives in the first two)
    bufs = [torch.zeros(HEAD + cap * 3, dtype=torch.int32, device=dev) for _ in range(2 if world > 1 else 1)]
    d_hits, d_count = bufs[0][HEAD:], bufs[0][:2].view(torch.int64)
    if world > 1:
        # The gather of step i runs on its own stream while the scan of step i+1 runs on the main one (two buffers);
        # all_gather keeps every rank symmetric, rank 0 is the reader.
        comm = torch.cuda.Stream(device=dev)
        gathered = [torch.zeros(world * (HEAD + cap * 3), dtype=torch.int32, device=dev) for _ in range(2)]
        scan_done = [torch.cuda.Event() for _ in range(2)]
        gather_done = [torch.cuda.Event() for _ in range(2)]

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    it = [0]  # steps issued so far (warm-up included): picks the buffer

    def step(i_timed: int | None) -> None:
        b = it[0] % len(bufs)
        if world > 1 and it[0] >= 2:
            stream.wait_event(gather_done[b])  # the gather that read this buffer two steps ago has finished
        if i_timed is not None:
            ev[i_timed][0].record(stream)
        store.scan_launch(qset, D, None, bufs[b].data_ptr() + 4 * HEAD, cap, bufs[b].data_ptr())
        if i_timed is not None:
            ev[i_timed][1].record(stream)
        if world > 1:  # RCCL gather of the row lists
            scan_done[b].record(stream)
            with torch.cuda.stream(comm):
                comm.wait_event(scan_done[b])
                dist.all_gather_into_tensor(gathered[b], bufs[b])
                gather_done[b].record(comm)
        it[0] += 1

    def fence() -> None:
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(None)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    kernel_ms = [a.elapsed_time(b) for a, b in ev]
    kernel_ms_avg = float(np.mean(kernel_ms)) if kernel_ms else float("nan")
    plan = store.last_scan_plan()  # which kernel form the timed launches used

    # ---- result check (outside the timed region): planted rows present, every row's distance recomputed
    last = (it[0] - 1) % len(bufs)
    d_hits, d_count = bufs[last][HEAD:], bufs[last][:2].view(torch.int64)
    n_rows = int(d_count.item())
    rows = d_hits[: 3 * min(n_rows, cap)].cpu().numpy().view(np.uint32).reshape(-1, 3)
    ok = n_rows <= cap
    if world > 1:  # what the gather delivered: this rank's block must be its own buffer, every count within capacity
        g = gathered[last].view(world, HEAD + cap * 3)
        ok = ok and bool(torch.equal(g[rank], bufs[last]))
        ok = ok and all(0 <= int(g[r][:2].view(torch.int64).item()) <= cap for r in range(world))
    recomputed = (subj[rows[:, 1]] != my_q[rows[:, 0]]).sum(axis=1)
    ok = ok and bool((recomputed == rows[:, 2]).all()) and bool((rows[:, 2] <= D).all())
    have = set(zip(rows[:, 0].tolist(), rows[:, 1].tolist()))
    for qi in range(Q):
        if planted_subs[q_lo + qi] <= D and (qi, int(planted_row[q_lo + qi])) not in have:
            ok = False
            break
    if world > 1:
        flag = torch.tensor([1 if ok else 0], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = bool(flag.item())

    # ---- stream mode: ONE query per pass — the HBM-bound form of the same kernel (north_star's literal
    #      "broadcast each query against all subjects"); reported beside the batched roofline
    stream_info = None
    if rank == 0 and not args.no_stream:
        one = smafa_amd.QuerySet(store, my_q[:1])
        for _ in range(3):
            store.scan_launch(one, D, None, d_hits.data_ptr(), cap, d_count.data_ptr())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 50
        e0.record(stream)
        for _ in range(reps):
            store.scan_launch(one, D, None, d_hits.data_ptr(), cap, d_count.data_ptr())
        e1.record(stream)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps  # includes the counter-reset launch and launch gaps
        k_ms = []
        for _ in range(20):  # kernel-only time: HIP events recorded by the library right around the scan kernel
            store.scan_launch(one, D, None, d_hits.data_ptr(), cap, d_count.data_ptr())
            k_ms.append(store.last_scan_ms()[0])
        k_med = float(np.median(k_ms))
        splan = store.last_scan_plan()
        # bytes one pass streams: the whole block, or only the prefilter's plane when that is all that is resident
        sb = info.words_per_plane * 4 if splan["filter_plane_resident"] else info.bytes_per_subject
        streamed = info.hbm_bytes * sb / info.bytes_per_subject
        stream_info = {
            "ms_per_query": ms,
            "kernel_ms_median": k_med,
            "streamed_bytes_per_subject": int(sb),
            "kernel_streamed_GBs": streamed / k_med / 1e6,
            "kernel_frac_of_peak": streamed / k_med / 1e6 / HBM_PEAK_GBS,
            "kernel_algorithmic_GBs": N * L * (8 if args.alphabet == "aa" else int(info.planes)) / 8 / k_med / 1e6,
            "plan": splan,
            "note": "one query per DB pass; streamed = bit-plane bytes the kernel actually reads per subject (the "
                    "prefilter's plane only when the other planes are fetched on demand); algorithmic = %g B/subject"
                    % (L * (8 if args.alphabet == "aa" else int(info.planes)) / 8),
        }
        one.close()

    # ---- the same launch with the prefilter off: every pair gets the full comparison (results identical)
    unfiltered = None
    if rank == 0 and not args.no_stream:
        store.set_prefilter(False)
        store.scan_launch(qset, D, None, d_hits.data_ptr(), cap, d_count.data_ptr())
        torch.cuda.synchronize()
        u_ms = []
        for _ in range(5):
            store.scan_launch(qset, D, None, d_hits.data_ptr(), cap, d_count.data_ptr())
            u_ms.append(store.last_scan_ms()[0])
        store.set_prefilter(True)
        u_med = float(np.median(u_ms))
        unfiltered = {"kernel_ms": u_med, "queries_per_s": Q / (u_med * 1e-3), "rows": int(d_count.item()),
                      "note": "SAME kernel, prefilter disabled (smafa_set_prefilter 0): all planes, all words, "
                              "for every pair"}

    # ---- host-buffer API (PCIe-inclusive): queries uploaded + packed, rows copied back and ordered on the host
    host_api = None
    if rank == 0 and world == 1 and not args.no_stream:
        store.scan(my_q[:64], max_divergence=D)
        reps = 3
        tq = time.perf_counter()
        for _ in range(reps):
            rows_h = store.scan(my_q, max_divergence=D)
        dt = (time.perf_counter() - tq) / reps
        host_api = {"queries_per_s": Q / dt, "ms_per_batch": dt * 1e3, "rows": int(len(rows_h)),
                    "note": "smafa_scan_hits: host code bytes in, ordered rows out (upload, pack, scan, copy back, sort)"}

    # ---- CPU baseline: the oracle's port of the reference's per-query loop, one thread, bounded sample
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle

        oracle.build()
        tq = time.perf_counter()
        oracle.bench_besthit_codes(subj, my_q[:2], D)
        per_q = (time.perf_counter() - tq) / 2
        n_sample = int(max(4, min(Q, args.cpu_seconds / max(per_q, 1e-6))))
        tq = time.perf_counter()
        oracle.bench_besthit_codes(subj, my_q[:n_sample], D)
        dt = time.perf_counter() - tq
        cpu = {
            "value": n_sample / dt, "unit": "query seqs/s", "cores": 1, "kind": "port",
            "sample": "first %d queries of the same batch against the same %d x %d %s store; oracle C port of "
                      "src/lib.rs:238 (distances) + :298 (min) + :307 (equality pass), gcc -O3, 1 thread, "
                      "contiguous store" % (n_sample, N, L, args.alphabet),
            "host_cores_available": os.cpu_count(),
        }

    if rank == 0:
        q_total = Q * world
        value = q_total * args.steps / elapsed
        pairs_per_launch = Q * N
        # B_s = L x bits per symbol / 8 — aa 8 bits (60 B), nt 2 bits (15 B), nt with N 3 planes (22.5 B)
        sym_bits = 8 if args.alphabet == "aa" else int(info.planes)
        alg_bytes = pairs_per_launch * L * sym_bits // 8
        achieved = alg_bytes / (kernel_ms_avg * 1e-3) / 1e9
        filt = os.environ.get("SMAFA_FILTER", "1") != "0"  # the timed steps run with the library default
        W_, P_ = info.words_per_plane, info.planes
        # VALU wave-instructions per (query, subject) pair = lane-ops per pair.  Prefilter fast path, per lane and
        # query: 4 subjects x W xor/bitop3 + 2 and + 2 popcounts (one per two subjects) + or + compare + LDS
        # address add = 4W + 7 per 4 pairs.  Full comparison: 4 subjects x (P*W xor/bitop3 + W popcounts) + 4 compares.
        ops_per_pair = (W_ + 1.75) if filt else (P_ * W_ + W_ + 1.0)
        if filt and plan["filter_plane_resident"]:
            # level-1 bound only, 4T subjects per lane: 4T xor + 4T popcounts + 2T or3 + compare + LDS address
            T_ = plan["tiles_per_wave"]
            ops_per_pair = (10.0 * T_ + 2.0) / (4.0 * T_)
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(pmc_path) and (N, L, Q, D, args.alphabet) == (10_000_000, 60, 10_000, 5, "aa"):
            traffic = json.load(open(pmc_path))["batched_launch"]["hbm_bytes"]  # measured per launch, same workload
        out = {
            "metric": "query seqs/sec (DB residues/sec in `residues_per_s`) vs HBM roofline, %dM x %d%s DB, d<=%d"
                      % (N // 1_000_000, L, args.alphabet, D),
            "value": value,
            "unit": "query seqs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {
                "workload": "%d x %d %s subject store (uniform letters, 1%% duplicate rows, seed %d%s) replicated per GPU; "
                            "%d planted queries per GPU per step (0..%d substitutions, seed 3); max-divergence %d"
                            % (N, L, args.alphabet, 1 if alphabet else 2,
                               ", N with probability %g per column" % args.n_frac if args.n_frac else "", Q, max_subs, D),
                "db_rows": N, "seq_len": L, "alphabet": args.alphabet, "queries_per_gpu": Q, "max_divergence": D,
                "parallelism": "query shards x%d, DB replicated, RCCL all_gather of row lists" % world,
            },
            "residues_per_s": value * N * L,
            "pairs_per_s": value * N,
            "rows_per_step": n_rows,
            "verified": ok,
            "roofline": {
                "bound": "hbm",
                # the instantiation rocprofv3 lists (profiles/r01_d_kernel_stats.csv): <PS, PQ, W, T, SEED>
                "kernel": "smafa::%s<%d, %d, %d, %d, false>" % (
                    "scan_lazy_kernel" if plan["filter_plane_resident"] else "scan_kernel", info.planes,
                    5 if args.alphabet == "aa" else 3, info.words_per_plane, plan["tiles_per_wave"]),
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": "profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 gfx950 correction + "
                                  "WRITE_SIZE, bytes per launch)" if traffic else None,
                "kernel_ms_avg": kernel_ms_avg,
                "algorithmic_bytes_per_launch": alg_bytes,
                "note": "algorithmic bytes = queries x subjects x %g B (SURVEY 8d). The kernel keeps a 1024-subject "
                        "tile in registers and walks a whole query block over it, so the store is streamed from HBM "
                        "once per query block, not once per query (see `traffic`): frac > 1 is register-level reuse "
                        "plus the exact lower-bound prefilter, NOT HBM efficiency. The real ceiling of this kernel "
                        "is VALU issue (see `valu`); the HBM-bound form (one query per pass) is in `stream`." % (L * sym_bits / 8),
                "valu": {
                    "prefilter": filt,
                    "plan": plan,
                    "lane_ops_per_pair": ops_per_pair,
                    "achieved_lane_ops": pairs_per_launch * ops_per_pair / (kernel_ms_avg * 1e-3),
                    "peak_lane_ops": VALU_PEAK_LANE_OPS,
                    "frac": pairs_per_launch * ops_per_pair / (kernel_ms_avg * 1e-3) / VALU_PEAK_LANE_OPS,
                    "note": "peak = 256 CU x 4 SIMD x 32 lanes x 2.4 GHz; measured issue rates on this chip "
                            "(profiles/r01_ubench_valu*.txt): all-VGPR xor/bitop3/add ~60e12, v_bcnt/v_cmp/v_min and "
                            "any op with an SGPR source ~37e12 lane-ops/s",
                },
                "stored_bytes_per_subject": int(info.bytes_per_subject),
                "streamed_bytes_per_subject": int(info.words_per_plane * 4 if plan["filter_plane_resident"] else info.bytes_per_subject),
            },
            "stream": stream_info,
            "unfiltered": unfiltered,
            "host_api": host_api,
            "cpu_baseline": cpu,
            "setup_s": {"generate": t_gen, "pack_upload": t_up},
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
