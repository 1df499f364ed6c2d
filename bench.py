#!/usr/bin/env python3
"""bench.py — headline benchmark of the smafa scan engine on MI355X.

Metric (BASELINE.json): query seqs/sec (+ DB residues/sec) vs the roofline, 10M x 60-aa DB, d <= 5.
A "step" = one pass of the hot path over one batch: every query of the batch (default 10 000 per GPU)
scanned against the whole resident subject store, qualifying rows appended on the device, and — when
more than one GPU takes part — the per-rank row lists gathered on rank 0 over RCCL.  The packed subject block and the
packed query batch are resident in HBM before the timed region starts.

    python bench.py [--gpus N] [--steps K] [--warmup W]

With --gpus N > 1 and no torch.distributed environment, this process only spawns
`python -m torch.distributed.run --nproc-per-node N bench.py ...` (before touching torch or the GPU), relays
rank 0's JSON line and exits with the child's code; under torch.distributed.run it is one of the N ranks.

Rank 0 prints ONE JSON line.  Scaling is weak: the store is replicated, every rank scans its own query shard
of fixed size.  Besides the contract's fields the line carries (N = 1 only, all outside the timed region):

  roofline           dominant kernel of the timed launch, live HIP-event time; VALU-bound: frac = VALU lane-ops/s over the
                     chip's nominal issue peak, instruction count from the committed rocprofv3 counter record of this same
                     command (profiles/r04_pmc.json, tools/collect_pmc.py; `insts_source_is_this_build` says whether the
                     record belongs to the binary being timed)
  stream             ONE query per store pass (north_star's "each query is broadcast against all subjects"): the HBM-bound
                     form, on a plane LARGER than the 256 MB Infinity Cache (the 50M-row store's 400 MB filter plane),
                     passes enqueued back to back by smafa_scan_each; fractions of the 8 TB/s peak by kernel and wall time
  unfiltered / loose_bounds / besthit_unbounded
                     the regimes the prefilter cannot help: the same launch with the prefilter off, fixed bounds of 14
                     and 24, and `smafa query` without --max-divergence on queries half of which have no near subject —
                     each with its own roofline block
  configs            BASELINE.json configs[1..4]: 1M aa, 10M nt (N-free and 0.1 % N), one rank's share of 50M x 1M,
                     `smafa cluster` on 5M records — each verified, each with kernel / wall times
  cpu_baseline       the oracle's port of the reference's loops on bounded samples (the only place bench.py touches
                     oracle/, besides the post-run result checks); query: B1 / B1n / B2 / aa; cluster: a prefix
"""
from __future__ import annotations

import argparse
import json
import os
import re
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
VALU_PEAK_LANE_OPS = 256 * 4 * 32 * 2.4e9  # 256 CUs x 4 SIMD-32 x 2.4 GHz (one wave64 VALU op = 2 cycles)
PMC_JSONS = [os.path.join(ROOT, "profiles", "r%02d_pmc.json" % r) for r in (4, 3, 2)]  # first file with a matching record wins
STREAM_PMC_JSONS = [os.path.join(ROOT, "profiles", "r%02d_stream_pmc.json" % r) for r in (4, 3)]
T_START = time.time()


def parse_args(argv=None):
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
    ap.add_argument("--store", choices=["uniform", "related"], default="uniform",
                    help="related: the TIMED workload is the related-families store (default: uniform letters; the related "
                         "store is then a side leg)")
    ap.add_argument("--mode", choices=["scan", "besthit", "kth"], default="scan",
                    help="besthit: the TIMED step is smafa_scan_hits with max_num_hits = 1 and no bound (`smafa query` "
                         "without --max-divergence); --far-frac of the queries are uniform random (no near subject).  "
                         "kth: the TIMED step is smafa_scan_hits with max_num_hits = --kth-k (the K branch, src/lib.rs:242-295) "
                         "on the planted queries, without a bound unless --kth-bounded")
    ap.add_argument("--far-frac", type=float, default=0.5)
    ap.add_argument("--kth-k", type=int, default=5, help="--mode kth: max_num_hits")
    ap.add_argument("--kth-bounded", action="store_true", help="--mode kth: also pass --max-div as max_divergence")
    ap.add_argument("--prefilter", type=int, choices=[0, 1], default=1, help="0: the TIMED launches run with the prefilter off")
    ap.add_argument("--query-block", type=int, default=0, help="queries per workgroup pass (0 = automatic)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stream", action="store_true", help="skip every side leg (clean rocprof stats of the timed workload)")
    ap.add_argument("--no-related", action="store_true", help="skip the related-store leg")
    ap.add_argument("--no-configs", action="store_true", help="skip the BASELINE configs block")
    ap.add_argument("--cpu-seconds", type=float, default=4.0, help="CPU time budget per CPU baseline")
    ap.add_argument("--time-budget", type=float, default=400.0,
                    help="seconds after start beyond which optional legs are skipped (the driver allows 600 s)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (tests use gloo)")
    ap.add_argument("--rehearse-collectives", action="store_true",
                    help="with --gpus 1: initialise the process group anyway and run every collective of the N>1 path over a world of one (an RCCL rehearsal on a one-GPU box)")
    ap.add_argument("--collective", choices=["gather", "all_gather"], default="gather",
                    help="how the per-rank row lists reach rank 0 each step (N > 1)")
    ap.add_argument("--single-device", action="store_true", help="testing only: every rank uses GPU 0")
    ap.add_argument("--full-record", default=os.path.join(ROOT, "gpurun_out", "bench_full.json"),
                    help="where the full record (every leg in detail, ~30 KB) is written; stdout carries the compact line only")
    ap.add_argument("--configs-scale", type=float, default=1.0,
                    help="testing only: the BASELINE configs block with every size multiplied by this (and run whatever the timed "
                         "workload is), so that the line's length can be checked with every leg present")
    ap.add_argument("--no-kth", action="store_true", help="skip the k-th-mode leg (--max-num-hits 5 / 50)")
    ap.add_argument("--no-index", action="store_true", help="skip the block-index legs (smafa_db_build_index)")
    return ap.parse_args(argv)


def find_free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args) -> int:
    """--gpus N > 1 from a plain shell: start the N ranks as a child job.  Nothing in this process has touched
    torch or the GPU, and it never execs: it waits, and passes the child's output and exit code on."""
    port = find_free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "4")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def sorted_rows(rows):
    """(n, 3) uint32 rows [query, subject, dist] -> ordered by (query, dist, subject), the reference's print order"""
    import numpy as np

    order = np.lexsort((rows[:, 1], rows[:, 2], rows[:, 0]))
    return np.ascontiguousarray(rows[order])


def rows3(a):
    """structured hit array -> (n, 3) uint32"""
    import numpy as np

    return np.stack([a["query"], a["subject"], a["dist"]], axis=1).astype(np.uint32)


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def elapsed() -> float:
    return time.time() - T_START


# ------------------------------------------------------------------------------------------ counter records
_pmc_cache = None


def pmc_records():
    global _pmc_cache
    if _pmc_cache is None:
        _pmc_cache = []
        for path in PMC_JSONS:
            if os.path.exists(path):
                for rec in json.load(open(path)).get("records", []):
                    rec["_file"] = os.path.relpath(path, ROOT)
                    _pmc_cache.append(rec)
    return _pmc_cache


def pmc_lookup(cfg: dict, kernel: str):
    """the committed counter record of this workload and kernel (first file that has one wins: r04, r03, r02)"""
    def norm(c):
        return (c.get("db_rows"), c.get("seq_len"), c.get("queries"), c.get("max_div"), c.get("alphabet"),
                c.get("store", "uniform"), float(c.get("n_frac", 0.0)), int(c.get("prefilter", 1)), c.get("mode", "scan"),
                float(c.get("far_frac", 0.5)) if c.get("mode", "scan") == "besthit" else 0.0,
                (int(c.get("kth_k", 0)), bool(c.get("kth_bounded", False))) if c.get("mode", "scan") == "kth" else (0, False))
    for rec in pmc_records():
        if norm(rec.get("config", {})) == norm(cfg) and rec.get("kernel") == kernel.split(" (")[0]:
            return rec
    return None


def roofline_block(cfg: dict, kernel: str, kernel_ms: float, pairs: float, alg_bytes: float, build_id: str, extra=None):
    """VALU-issue roofline of one launch (or, mode besthit, of all scan kernels of one call): achieved = recorded
    SQ_INSTS_VALU x 64 lanes over the LIVE kernel time"""
    rec = pmc_lookup(cfg, kernel)
    k_s = kernel_ms * 1e-3
    insts = float(rec["per_launch"]["SQ_INSTS_VALU"]) if rec else None
    lane_ops = insts * 64.0 / k_s if insts and k_s > 0 else None
    traffic = float(rec["per_launch"]["hbm_bytes"]) if rec and "hbm_bytes" in rec["per_launch"] else None
    out = {
        "bound": "valu",
        "kernel": kernel,
        "achieved": lane_ops / 1e12 if lane_ops else None,
        "peak": VALU_PEAK_LANE_OPS / 1e12,
        "unit": "Tlane-op/s",
        "frac": lane_ops / VALU_PEAK_LANE_OPS if lane_ops else None,
        "kernel_ms_avg": kernel_ms,
        "valu_insts_per_launch": insts,
        "valu_insts_per_1024_pairs": insts / (pairs / 1024.0) if insts and pairs else None,
        "insts_source": ("%s: rocprofv3 --pmc passes of `bench.py %s` (tools/collect_pmc.py), build %s; this run is build %s"
                         % (rec["_file"], rec.get("command", "").split("bench.py")[-1].strip(), rec.get("build_id"), build_id))
        if rec else "no counter record of this workload and kernel under profiles/: achieved / frac not claimed",
        "insts_source_is_this_build": bool(rec and rec.get("build_id") == build_id),
        "kernel_ms_under_profiler": (rec.get("kernel_ms_under_profiler") or {}).get("sq_a") if rec else None,
        "traffic": traffic,
        "traffic_source": ("recorded (FETCH_SIZE x 2 by the guide's gfx950 rule + WRITE_SIZE, separate --pmc passes), NOT "
                           "measured in this run") if traffic else None,
        "hbm_frac_from_recorded_traffic": (traffic / k_s / 1e9 / HBM_PEAK_GBS) if traffic and k_s > 0 else None,
        "algorithmic_bytes_per_launch": alg_bytes,
        "algorithmic_reuse_x": alg_bytes / k_s / 1e9 / HBM_PEAK_GBS if k_s > 0 else None,
    }
    if extra:
        out.update(extra)
    return out


# ------------------------------------------------------------------------------------------ CPU baselines
def cpu_baselines(N, L, D, alphabet_name, subj_codes, q_codes, budget_s):
    """SURVEY 8(d) / BASELINE.md §2: the reference's per-query loop (src/lib.rs:238 distances, :298 min, :307 equality
    pass) restated by the oracle and timed on this host, each on a bounded sample.
      B1   5-bit one-hot u64 x ceil(L/12) per subject (the reference's own arithmetic, src/lib.rs:71-89), ONE thread,
           gcc -O3 for baseline x86-64 (no POPCNT: what `cargo build --release` gives)
      B1n  the same, -march=native
      B2   B1n over min(16, nproc) worker threads (query shards; the reference has no threads: a courtesy baseline)
      aa   the code-byte port (one byte per column): the only CPU form that can hold amino-acid letters
    The reference's cost does not depend on the letters (same words per subject), so for the amino-acid metric B1..B2
    run on a nucleotide store of the same shape (seed 2) — the reference itself rejects amino-acid input."""
    import numpy as np

    import oracle
    from smafa_amd import synth

    oracle.build()
    out = {"cpu_model": cpu_model(), "nproc": os.cpu_count()}
    NT = np.frombuffer(b"ACGTN", dtype=np.uint8)
    if alphabet_name == "nt":
        s_nt, q_nt = subj_codes, q_codes[:512]
    else:
        s_nt = synth.subjects(N, L, 0, seed=2)
        q_nt, _, _ = synth.queries(s_nt, 512, 0, seed=3, max_subs=6)
    sa, qa = NT[s_nt], NT[q_nt]

    def sample_size(per_q, cap):
        return int(max(4, min(cap, budget_s / max(per_q, 1e-6))))

    for key, native in (("b1", False), ("b1n", True)):
        db = oracle.OnehotDB(sa, native=native)
        enc = db.encode_queries(qa)
        t = time.perf_counter()
        db.bench_besthit(enc[:2], D)
        n_s = sample_size((time.perf_counter() - t) / 2, len(enc))
        t = time.perf_counter()
        db.bench_besthit(enc[:n_s], D)
        dt = time.perf_counter() - t
        out[key] = {"value": n_s / dt, "unit": "query seqs/s", "cores": 1, "queries": n_s,
                    "build": "gcc -O3 -march=native" if native else "gcc -O3, baseline x86-64 (no POPCNT)"}
        if not native:
            # the K branch (src/lib.rs:242-295): per query an N x 16 B tuple vector and a full sort of it, on top of the distances
            t = time.perf_counter()
            db.bench_kmode(enc[:1], oracle.NO_LIMIT, 5)
            n_k = sample_size(time.perf_counter() - t, 64)
            t = time.perf_counter()
            db.bench_kmode(enc[:n_k], oracle.NO_LIMIT, 5)
            dtk = time.perf_counter() - t
            out["kmode"] = {"value": n_k / dtk, "unit": "query seqs/s", "cores": 1, "queries": n_k, "max_num_hits": 5,
                            "sample": "first %d queries of the B1 batch, max_num_hits 5, no bound: oracle C port of src/lib.rs:238 + "
                                      ":243-250 (tuple vector + full merge sort per query) + :253-293, gcc -O3 without POPCNT, 1 thread"
                                      % n_k}
        if native:
            workers = min(16, os.cpu_count() or 1)
            per_q = dt / n_s
            n_b2 = int(max(workers, min(len(enc), workers * 0.25 * budget_s / per_q)))  # a pass = a quarter of the budget
            shards = np.array_split(np.arange(n_b2), workers)

            # worker THREADS: the oracle's C loop runs without the interpreter lock (ctypes releases it) and allocates its
            # distance buffer per call; no process is forked from one that has initialised the GPU
            from concurrent.futures import ThreadPoolExecutor

            with ThreadPoolExecutor(workers) as pool:
                run = lambda idx: db.bench_besthit(np.ascontiguousarray(enc[idx]), D)
                # the first second or two of a threaded burst can run far below the steady rate (CPU wake-up / scheduler
                # quota): passes over the same shards until the budget is used, the fastest pass is reported
                dt2, t_all = float("inf"), time.perf_counter()
                while True:
                    t = time.perf_counter()
                    list(pool.map(run, shards))
                    dt2 = min(dt2, time.perf_counter() - t)
                    if time.perf_counter() - t_all > budget_s:
                        break
            out["b2"] = {"value": n_b2 / dt2, "unit": "query seqs/s", "cores": workers, "queries": n_b2,
                         "build": "gcc -O3 -march=native, %d worker threads, fastest pass of the budget" % workers}
        db.close()
    if alphabet_name == "aa":
        t = time.perf_counter()
        oracle.bench_besthit_codes(subj_codes, q_codes[:2], D)
        n_s = sample_size((time.perf_counter() - t) / 2, len(q_codes))
        t = time.perf_counter()
        oracle.bench_besthit_codes(subj_codes, q_codes[:n_s], D)
        dt = time.perf_counter() - t
        out["aa_code_bytes"] = {"value": n_s / dt, "unit": "query seqs/s", "cores": 1, "queries": n_s,
                                "build": "gcc -O3, one byte per column, the same store and queries as the GPU"}
    return out


# ------------------------------------------------------------------------------------------ the line on stdout
LINE_LIMIT = 4096  # the driver keeps a tail of stdout: the ONE line it parses must fit it with room to spare


def sig(x, n=4):
    """numbers to n significant digits (the line is read by a parser and a judge, not by a plotter)"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    try:
        return float("%.*g" % (n, float(x)))
    except (TypeError, ValueError):
        return None


def compact_line(full: dict, full_path=None) -> dict:
    """The one JSON line of stdout, <= LINE_LIMIT characters: the contract's fields, the headline roofline (+ the HBM-bound
    one-query pass as roofline.hbm_stream), cpu_baseline, and one short array per side leg.  Every detail (notes, sources,
    per-leg roofline blocks) stays in the full record (`full_record`)."""
    g = lambda d, *ks: (g(d.get(ks[0]), *ks[1:]) if len(ks) > 1 else d.get(ks[0])) if isinstance(d, dict) else None
    cfg = dict(full.get("config") or {})
    cfg["workload"] = str(cfg.get("workload", ""))[:200]
    if cfg.get("mode") != "besthit":
        cfg.pop("far_frac", None)
    if cfg.get("mode") != "kth":
        cfg.pop("kth_k", None)
        cfg.pop("kth_bounded", None)
    cfg["parallelism"] = str(cfg.get("parallelism", ""))[:80]
    r = full.get("roofline") or {}
    roof = {"bound": r.get("bound"), "kernel": r.get("kernel"), "achieved": sig(r.get("achieved")), "peak": sig(r.get("peak"), 6),
            "unit": r.get("unit"), "frac": sig(r.get("frac")), "kernel_ms_avg": sig(r.get("kernel_ms_avg")),
            "traffic": sig(r.get("traffic")), "hbm_frac": sig(r.get("hbm_frac_from_recorded_traffic")),
            "valu_per_1024_pairs": sig(r.get("valu_insts_per_1024_pairs")),
            "algorithmic_bytes": r.get("algorithmic_bytes_per_launch"), "algorithmic_reuse_x": sig(r.get("algorithmic_reuse_x")),
            "insts_source_is_this_build": r.get("insts_source_is_this_build")}
    st = full.get("stream") or {}
    best = st.get("hbm_store") or st.get("metric_store")
    if best:
        sr = st.get("roofline") or {}
        roof["hbm_stream"] = {
            "store": best.get("store"), "bytes_per_pass": best.get("streamed_bytes_per_pass"),
            "served_by": "hbm" if (best.get("streamed_bytes_per_pass") or 0) > (256 << 20) else "cache",
            "us_wall": sig(1e3 * (g(best, "streaming", "ms_per_query_wall") or 0)),
            "us_kernel": sig(1e3 * (g(best, "streaming", "kernel_ms_median") or 0)),
            "frac_wall": sig(g(best, "streaming", "frac_wall_streamed")), "frac_kernel": sig(g(best, "streaming", "frac_kernel_streamed")),
            "fetched_over_plane": sig(st.get("fetched_over_plane"), 5),
            "every_plane_metric_store": [sig(1e3 * (g(st, "metric_store", "streaming_every_plane", "ms_per_query_wall") or 0)),
                                         sig(g(st, "metric_store", "streaming_every_plane", "frac_wall_streamed"))],
            "read_ceiling_frac": sig(st.get("empirical_read_ceiling_frac_of_peak")), "rows_identical": best.get("rows_identical"),
            "peak": sr.get("peak"), "unit": sr.get("unit")}
    c = full.get("cpu_baseline")
    cpu = None
    if c:
        cpu = {"value": sig(c.get("value")), "unit": c.get("unit"), "cores": c.get("cores"), "kind": c.get("kind"),
               "sample": str(c.get("sample", ""))[:160], "cpu_model": c.get("cpu_model"),
               "b1n": sig(g(c, "b1n", "value")), "b2": [sig(g(c, "b2", "value")), g(c, "b2", "cores")],
               "aa_code_bytes": sig(g(c, "aa_code_bytes", "value"))}
        if c.get("kmode"):
            cpu["kmode"] = {k: sig(v) if not isinstance(v, (str, list)) else v for k, v in c["kmode"].items() if k != "sample"}
    legs = {}
    u = full.get("unfiltered")
    if u:
        legs["unfiltered"] = [sig(u.get("kernel_ms")), sig(g(u, "roofline", "frac"))]
    for b in full.get("loose_bounds") or []:
        legs["bound%d" % b.get("max_divergence", 0)] = [sig(b.get("kernel_ms")), sig(g(b, "roofline", "frac")), b.get("verified")]
    bh = full.get("besthit_unbounded")
    if bh:
        legs["besthit_mixed"] = [sig(bh.get("wall_ms")), sig(bh.get("queries_per_s_wall")), sig(g(bh, "roofline", "frac")), bh.get("verified")]
        legs["besthit_far"] = [sig(g(bh, "far_queries_only", "wall_ms")), sig(g(bh, "far_queries_only", "queries_per_s_wall"))]
    rel = full.get("related")
    if rel:
        legs["related"] = [sig(rel.get("kernel_ms")), sig(rel.get("queries_per_s")), sig(g(rel, "roofline", "frac")), rel.get("verified")]
        nm = rel.get("besthit_novel_members")
        if nm:
            legs["besthit_novel"] = [sig(nm.get("wall_ms")), sig(nm.get("queries_per_s_wall")), None, nm.get("verified")]
            if nm.get("indexed"):  # [wall ms scan kernels, wall ms with a 15-block index, verified]
                legs["idx_novel"] = [sig(g(nm, "indexed", "scan_kernels", "wall_ms")), sig(g(nm, "indexed", "with_index", "wall_ms")),
                                     nm["indexed"].get("verified")]
    for name, leg in (full.get("kth") or {}).items():
        if isinstance(leg, dict):
            legs[name] = [sig(leg.get("wall_ms")), sig(leg.get("queries_per_s_wall")), sig(g(leg, "roofline", "frac")),
                          leg.get("verified"), sig(leg.get("kernel_ms"))]
    def idx(leg):  # [kernel ms, query seqs/s, x the scan kernels, verified, build ms]
        return [sig(leg.get("kernel_ms")), sig(leg.get("queries_per_s")), sig(leg.get("times_the_scan_kernels"), 3), leg.get("verified"),
                sig(g(leg, "index", "build_ms_call"), 3)]

    if full.get("indexed"):
        legs["idx"] = idx(full["indexed"])
        bp = full["indexed"].get("besthit_planted")
        if bp:  # best hit without a bound, planted queries: [wall ms scan kernels, wall ms with the index, verified]
            legs["idx_besthit"] = [sig(g(bp, "scan_kernels", "wall_ms")), sig(g(bp, "with_index", "wall_ms")), bp.get("verified")]
    ha = full.get("host_api")
    if ha:
        legs["host_api"] = [sig(ha.get("ms_per_batch")), sig(ha.get("queries_per_s")), None, ha.get("rows_identical_to_device_launch")]
    short = {"configs[1] 1M aa": "cfg1", "configs[2] 10M nt, N-free (2-bit store)": "cfg2",
             "configs[2] 10M nt, 0.1 % N (3-plane store)": "cfg2N",
             "configs[3] one rank's share: 50M aa x 125k of 1M queries": "cfg3", "configs[4] cluster 5M aa": "cfg4"}
    for name, leg in (full.get("configs") or {}).items():
        key = short.get(name, name[:12])
        if "wall_s" in leg:  # cluster: seconds, records/s
            legs[key] = [sig(leg.get("wall_s")), sig(leg.get("records_per_s")), sig(g(leg, "roofline", "frac")), leg.get("verified"),
                         sig(g(leg, "stages", "scan_kernels_ms")), sig(g(leg, "cpu_baseline", "extrapolated_full_run_s"))]
        else:
            legs[key] = [sig(leg.get("kernel_ms")), sig(leg.get("queries_per_s")), sig(g(leg, "roofline", "frac")), leg.get("verified")]
            if leg.get("indexed"):
                legs[key + "i"] = idx(leg["indexed"])
    out = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                    "scaling", "vs_baseline", "dtype", "data")}
    out["value"], out["ms_per_step"] = sig(out["value"], 7), sig(out["ms_per_step"], 6)
    out.update({"config": cfg, "residues_per_s": sig(full.get("residues_per_s")), "rows_per_step": full.get("rows_per_step"),
                "verified": full.get("verified"), "build_id": full.get("build_id"), "roofline": roof, "cpu_baseline": cpu,
                "legs": legs,
                "legs_key": "[ms (kernel; wall for besthit*/kth*/host_api; cfg4: wall s), query seqs/s (cfg4: records/s), "
                            "roofline frac, verified, ..]; unfiltered/boundN: [kernel ms, frac, verified]; kth*: 5th = kernel ms; "
                            "cfg4: 5th = scan kernel ms, 6th = CPU s extrapolated by pairs from a prefix; idx/cfgNi (opt-in block index on the same "
                            "store): [kernel ms, query seqs/s, x the scan kernels, verified, build ms]; idx_besthit / idx_novel (no bound; planted queries / novel family "
                            "members): [wall ms scan kernels, wall ms with the index, verified]",
                "gathered_bytes_per_rank_per_step": full.get("gathered_bytes_per_rank_per_step"),
                "run_s": full.get("run_s"), "skipped_for_time": [s_.get("leg") for s_ in full.get("skipped_for_time") or []],
                "full_record": full_path})
    # the limit is a contract: shed the least important blocks rather than overrun it
    for drop in ("legs_key", "skipped_for_time", "residues_per_s", "legs"):
        if len(json.dumps(out)) <= LINE_LIMIT:
            break
        out.pop(drop, None)
    return out


def emit(full: dict, full_path) -> None:
    """full record -> file (best effort), compact line -> the LAST line of stdout"""
    written = None
    if full_path:
        try:
            os.makedirs(os.path.dirname(os.path.abspath(full_path)), exist_ok=True)
            with open(full_path, "w") as f:
                json.dump(full, f)
                f.write("\n")
            written = os.path.relpath(full_path, ROOT) if os.path.abspath(full_path).startswith(ROOT) else full_path
        except OSError as e:
            print("bench.py: full record not written (%s)" % e, file=sys.stderr)
    line = json.dumps(compact_line(full, written))
    assert len(line) <= LINE_LIMIT, len(line)
    sys.stdout.flush()
    print(line, flush=True)



# ------------------------------------------------------------------------------------------ helpers for the side legs
class Bench:
    """one device, one stream, one shared row buffer: what every side leg launches through"""

    def __init__(self, torch, dev, stream, local_rank, cap):
        import numpy as np

        self.torch, self.dev, self.stream, self.local_rank, self.cap = torch, dev, stream, local_rank, cap
        self.buf = torch.zeros(4 + cap * 3, dtype=torch.int32, device=dev)
        self.d_hits, self.d_count = self.buf[4:], self.buf[:2].view(torch.int64)
        self.np = np

    def ensure(self, cap):
        if cap > self.cap:
            self.cap = cap
            self.buf = self.torch.zeros(4 + cap * 3, dtype=self.torch.int32, device=self.dev)
            self.d_hits, self.d_count = self.buf[4:], self.buf[:2].view(self.torch.int64)

    def launch_rows(self, store, qset, D):
        """one launch outside any timed region -> (count, rows ordered)"""
        store.scan_launch(qset, D, None, self.d_hits.data_ptr(), self.cap, self.d_count.data_ptr())
        self.torch.cuda.synchronize()
        n = int(self.d_count.item())
        r = self.d_hits[: 3 * min(n, self.cap)].cpu().numpy().view(self.np.uint32).reshape(-1, 3)
        return n, sorted_rows(r)

    def kernel_ms(self, store, qset, D, reps):
        """median HIP-event time of the scan kernel over `reps` launches + wall time per launch of the last K back to back"""
        ms = []
        for _ in range(reps):
            store.scan_launch(qset, D, None, self.d_hits.data_ptr(), self.cap, self.d_count.data_ptr())
            ms.append(store.last_scan_ms()[0])
        e0, e1 = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
        e0.record(self.stream)
        for _ in range(reps):
            store.scan_launch(qset, D, None, self.d_hits.data_ptr(), self.cap, self.d_count.data_ptr())
        e1.record(self.stream)
        self.torch.cuda.synchronize()
        return float(self.np.median(ms)), e0.elapsed_time(e1) / reps


    def indexed_leg(self, store, qset, D, Q, scan_kernel_ms, reps=20):
        """the same fixed-bound launch answered from the store's block index (smafa_db_build_index: D + 1 probes per query
        instead of a pass over every subject) — an opt-in for callers that scan one resident store many times.  Rows must
        be those of the scan kernels' launch, byte for byte once both lists are ordered."""
        n0, rows0 = self.launch_rows(store, qset, D)
        t = time.perf_counter()
        info = store.build_index(D)
        build_call_ms = (time.perf_counter() - t) * 1e3
        n1, rows1 = self.launch_rows(store, qset, D)
        name = store.last_scan_kernel()
        out = {"max_divergence": D, "served": "index_probe" in name, "kernel": name,
               "index": {"blocks": info["blocks"], "bytes": info["bytes"], "build_ms_device": info["build_ms"],
                         "build_ms_call": build_call_ms, "longest_run": info["longest_run"],
                         "candidates_per_query": info["candidates_per_query"], "max_div_served": info["max_div_served"]},
               "rows": n1, "verified": bool(n0 == n1 and rows0.tobytes() == rows1.tobytes()),
               "checks": "ordered rows byte-identical to the scan kernels' launch on the same resident store and query set"}
        if out["served"]:
            k_ms, w_ms = self.kernel_ms(store, qset, D, reps)
            out.update({"kernel_ms": k_ms, "wall_ms_per_launch": w_ms, "queries_per_s": Q / (k_ms * 1e-3),
                        "queries_per_s_wall": Q / (w_ms * 1e-3), "scan_kernel_ms": scan_kernel_ms,
                        "times_the_scan_kernels": scan_kernel_ms / k_ms if k_ms else None,
                        "launches_to_repay_the_build": build_call_ms / max(scan_kernel_ms - k_ms, 1e-9)})
        store.drop_index()
        return out

    def indexed_besthit(self, store, queries, wide=11):
        """`smafa query` without --max-divergence (best hit) on the planted queries, host code bytes in, ordered rows out: the scan
        kernels' ladder against the same call with a block index built for bounds up to `wide` (it answers the ladder's first step)"""
        out = {}
        for name in ("scan_kernels", "with_index"):
            info = store.build_index(wide) if name == "with_index" else None
            store.scan(queries[:512], max_num_hits=1)
            walls, kms, rows = [], [], None
            for _ in range(3):
                t = time.perf_counter()
                rows = store.scan(queries, max_num_hits=1)
                walls.append((time.perf_counter() - t) * 1e3)
                kms.append(store.last_call_stats()["kernel_ms"])
            out[name] = {"wall_ms": float(self.np.median(walls)), "kernel_ms": float(self.np.median(kms)),
                         "queries_per_s_wall": len(queries) / (float(self.np.median(walls)) * 1e-3), "rows": int(len(rows))}
            if info:
                out[name]["index"] = {"blocks": info["blocks"], "bytes": info["bytes"], "build_ms_device": info["build_ms"],
                                      "max_div_served": info["max_div_served"]}
                out["verified"] = bool(rows.tobytes() == ref)
            else:
                ref = rows.tobytes()
        store.drop_index()
        return out


def verify_rows(np, subj, qry, rows, D, planted_row=None, planted_subs=None):
    """soundness (every distance recomputed from the code bytes) + recall of the planted rows"""
    rec = (subj[rows[:, 1]] != qry[rows[:, 0]]).sum(axis=1)
    ok = bool((rec == rows[:, 2]).all()) and bool((rows[:, 2] <= D).all())
    if planted_row is not None:
        key = rows[:, 0].astype(np.int64) << 32 | rows[:, 1].astype(np.int64)
        want_q = np.nonzero(planted_subs <= D)[0]
        want = want_q.astype(np.int64) << 32 | planted_row[want_q].astype(np.int64)
        ok = ok and bool(np.isin(want, key).all())
    return ok


def oracle_sample(np, subj, qry, rows, D, pick):
    """complete row lists of the sampled queries against the whole store, bit for bit (oracle: code-byte scan)"""
    import oracle

    want = oracle.scan_codes(subj, qry[pick], D)  # ordered (query, dist, subject), query = index into `pick`
    want_a = rows3(want)
    got = rows[np.isin(rows[:, 0], pick)].copy()
    remap = np.full(int(pick.max()) + 1, -1, dtype=np.int64)
    remap[pick] = np.arange(len(pick))
    got[:, 0] = remap[got[:, 0]]
    return sorted_rows(got).tobytes() == want_a.tobytes()


def expected_with_k(np, d, k, bound):
    """rows (subject, dist) the reference prints for ONE query given all its N distances (src/lib.rs:242-313): k = 1 or None:
    every subject at the minimum; k >= 2: every subject within the k-th smallest distance (all of them when k > N), both
    capped by `bound` (max_divergence) when there is one; in (dist, subject) order"""
    if not k or k == 1:
        thr = int(d.min())
    else:
        thr = int(np.partition(d, k - 1)[k - 1]) if k <= len(d) else int(d.max())
    if bound is not None and (not k or k == 1) and thr > bound:
        return np.zeros((0, 2), dtype=np.uint32)
    if bound is not None:
        thr = min(thr, int(bound))
    idx = np.nonzero(d <= thr)[0]
    idx = idx[np.argsort(d[idx], kind="stable")]
    return np.stack([idx, d[idx]], axis=1).astype(np.uint32)


def kth_rows_ok(np, subj, qry, rows, k, bound, pick):
    """a k-th-mode row list (n, 3) [query, subject, dist] ordered (query, dist, subject): every distance recomputed from the
    code bytes; per query the k-th rule's own invariant (fewer than k rows strictly below the largest reported distance, and
    without a bound at least min(k, N) rows); the sampled queries' complete lists against the oracle's distances"""
    import oracle

    rec = (subj[rows[:, 1]] != qry[rows[:, 0]]).sum(axis=1)
    ok = bool((rec == rows[:, 2]).all())
    Q = len(qry)
    n_rows = np.bincount(rows[:, 0], minlength=Q)
    last = np.zeros(Q, dtype=np.int64)
    np.maximum.at(last, rows[:, 0], rows[:, 2])
    below = np.bincount(rows[:, 0], weights=(rows[:, 2] < last[rows[:, 0]]), minlength=Q)
    ok = ok and bool((below < max(k, 1)).all())
    if bound is None:
        ok = ok and bool((n_rows >= min(max(k, 1), len(subj))).all())
    else:
        ok = ok and bool((rows[:, 2] <= bound).all())
    for qi in pick:
        want = expected_with_k(np, oracle.distances_codes(subj, qry[qi]), k, bound)
        mine = rows[rows[:, 0] == qi][:, 1:]
        ok = ok and mine.tobytes() == want.tobytes()
    return ok


def main() -> int:
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        return self_launch(args)

    import numpy as np
    import torch

    import smafa_amd
    from smafa_amd import synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available() or smafa_amd.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device: the scan engine has no CPU fallback")
    if args.mode != "scan" and world > 1:
        raise SystemExit("--mode %s is a one-GPU leg" % args.mode)
    call_mode = args.mode != "scan"  # the TIMED step is one host-buffer call (smafa_scan_hits), not one device launch
    call_k = 1 if args.mode == "besthit" else args.kth_k
    call_bound = args.max_div if (args.mode == "kth" and args.kth_bounded) else None
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    multi = world > 1 or args.rehearse_collectives  # the N>1 code path (collectives, two buffers); a world of one may rehearse it
    if multi:
        import torch.distributed as dist

        if "MASTER_ADDR" not in os.environ:  # a rehearsal started without torch.distributed.run
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(find_free_port()), RANK="0", WORLD_SIZE="1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    alphabet = smafa_amd.ALPHABET_AA if args.alphabet == "aa" else smafa_amd.ALPHABET_NT
    L, N, Q, D = args.seq_len, args.db_rows, args.queries, args.max_div
    max_subs = 10 if alphabet == smafa_amd.ALPHABET_AA else 6
    side_legs = rank == 0 and not multi and not args.no_stream
    build_id = smafa_amd.build_id()
    skipped = []

    def in_budget(name, need_s):
        if elapsed() + need_s <= args.time_budget:
            return True
        skipped.append({"leg": name, "at_s": round(elapsed(), 1), "needs_s": need_s})
        return False

    # ---- synthetic workload: identical store on every rank, disjoint query shards
    t_gen = time.time()
    if args.store == "related":
        subj = synth.related_subjects(N // 100, 100, L, alphabet, seed=7)
        N = len(subj)
    else:
        subj = synth.subjects(N, L, alphabet, seed=1 if alphabet else 2, n_frac=args.n_frac)  # seeds of SURVEY 8d
    all_q, planted_row, planted_subs = synth.queries(subj, Q * world, alphabet, seed=3, max_subs=max_subs)
    if args.mode == "besthit":  # a share of the queries has no near subject at all: uniform letters (seed 9)
        n_far = int(Q * args.far_frac)
        all_q[Q - n_far:] = synth.subjects(n_far, L, alphabet, seed=9, dup_frac=0.0)
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
    if not args.prefilter:
        store.set_prefilter(False)

    cap = max(4 * Q, 1 << 16)
    # one buffer per step parity: [count (u64) | pad | rows], so that ONE collective moves count and rows together
    HEAD = 4  # int32 words before the rows (the count lives in the first two)
    bufs = [torch.zeros(HEAD + cap * 3, dtype=torch.int32, device=dev) for _ in range(2 if multi else 1)]
    d_hits, d_count = bufs[0][HEAD:], bufs[0][:2].view(torch.int64)
    gathered_bytes = None
    if multi:
        # The gather of step i runs on its own stream while the scan of step i+1 runs on the main one (two buffers).
        # north_star: "a final RCCL gather over xGMI of the hit lists" — a gather to rank 0.  What crosses a link per step is
        # [count | the rows this rank found], not the whole capacity buffer: the width every rank sends is the largest row
        # count of the WARM-UP steps plus a quarter of headroom (counts are stable for a fixed query shard; the check
        # below fails the run if a timed step ever exceeded it), so no collective waits for a count.
        comm = torch.cuda.Stream(device=dev)
        use_gather = args.collective == "gather"
        scan_done = [torch.cuda.Event() for _ in range(2)]
        gather_done = [torch.cuda.Event() for _ in range(2)]
        width = [HEAD + cap * 3]  # int32 words gathered per rank and step (set after the warm-up)
        gathered = [None, None]

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    it = [0]  # steps issued so far (warm-up included): picks the buffer
    besthit_stats = []

    def step(i_timed: int | None) -> None:
        b = it[0] % len(bufs)
        if multi and it[0] >= 2:
            stream.wait_event(gather_done[b])  # the gather that read this buffer two steps ago has finished
        if i_timed is not None:
            ev[i_timed][0].record(stream)
        if call_mode:
            step.rows = store.scan(my_q, max_divergence=call_bound, max_num_hits=call_k)
            besthit_stats.append(store.last_call_stats())
        else:
            store.scan_launch(qset, D, None, bufs[b].data_ptr() + 4 * HEAD, cap, bufs[b].data_ptr())
        if i_timed is not None:
            ev[i_timed][1].record(stream)
        if multi:  # RCCL gather of the row lists
            scan_done[b].record(stream)
            with torch.cuda.stream(comm):
                comm.wait_event(scan_done[b])
                send = bufs[b][: width[0]]
                if gathered[b] is None or gathered[b].numel() != world * width[0]:
                    gathered[b] = torch.zeros(world * width[0], dtype=torch.int32, device=dev) if (rank == 0 or not use_gather) else None
                if use_gather:
                    dist.gather(send, list(gathered[b].view(world, width[0]).unbind(0)) if rank == 0 else None, dst=0)
                else:
                    dist.all_gather_into_tensor(gathered[b], send)
                gather_done[b].record(comm)
        it[0] += 1

    def fence() -> None:
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(None)
    fence()
    if multi:
        # size the gathered block from what the warm-up steps found (all ranks agree on ONE width: MAX over ranks)
        seen = max(int(b[:2].view(torch.int64).item()) for b in bufs) if args.warmup else cap
        w = torch.tensor([seen], dtype=torch.int64, device=dev)
        dist.all_reduce(w, op=dist.ReduceOp.MAX)
        rows_w = min(cap, int(w.item()) + int(w.item()) // 4 + 64)
        width[0] = HEAD + rows_w * 3
        gathered_bytes = width[0] * 4
        for b in range(2):  # the receive buffers of the timed steps exist before the clock starts
            gathered[b] = torch.zeros(world * width[0], dtype=torch.int32, device=dev) if (rank == 0 or not use_gather) else None
        fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    elapsed_s = time.perf_counter() - t0
    if multi:
        t = torch.tensor([elapsed_s], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed_s = float(t.item())

    if call_mode:
        kernel_ms = [s["kernel_ms"] for s in besthit_stats[args.warmup:]]
        wall_ms = [a.elapsed_time(b) for a, b in ev]
    else:
        kernel_ms = [a.elapsed_time(b) for a, b in ev]
        wall_ms = None
    kernel_ms_avg = float(np.mean(kernel_ms)) if kernel_ms else float("nan")
    plan = store.last_scan_plan()  # which kernel form the timed launches used
    kernel_name = store.last_scan_kernel()
    cfg = {"db_rows": N, "seq_len": L, "queries": Q, "max_div": D, "alphabet": args.alphabet, "store": args.store,
           "n_frac": args.n_frac, "prefilter": args.prefilter, "mode": args.mode, "far_frac": args.far_frac,
           "kth_k": args.kth_k, "kth_bounded": bool(args.kth_bounded)}

    B = Bench(torch, dev, stream, local_rank, cap)

    # ---- result checks (outside the timed region)
    checks = {}
    if args.mode == "kth":
        rows = rows3(step.rows)
        n_rows = len(rows)
        if rank == 0:
            import oracle

            oracle.build()
        checks["kth_rows_ok_incl_oracle_sample"] = kth_rows_ok(np, subj, my_q, rows, call_k, call_bound, np.array([0, Q // 2, Q - 1]))
        checks["oracle_sample_queries"] = 3
    elif args.mode == "besthit":
        rows = rows3(step.rows)
        n_rows = len(rows)
        rec = (subj[rows[:, 1]] != my_q[rows[:, 0]]).sum(axis=1)
        checks["distances_recomputed"] = bool((rec == rows[:, 2]).all())
        checks["every_query_has_a_best_hit"] = len(np.unique(rows[:, 0])) == Q
        near = np.nonzero(planted_subs[q_lo:q_lo + Q - int(Q * args.far_frac)] <= 10)[0]
        # a planted query's best hit is at most its number of substitutions away
        first = np.full(Q, 1 << 30, dtype=np.int64)
        np.minimum.at(first, rows[:, 0], rows[:, 2])
        checks["planted_bound_holds"] = bool((first[near] <= planted_subs[q_lo + near]).all())
        if rank == 0:
            import oracle

            oracle.build()
            pick = np.unique(np.concatenate([np.arange(3), np.arange(Q - 3, Q)]))  # three near, three far
            everything = oracle.scan_codes(subj, my_q[pick], L)
            want = []
            for i in range(len(pick)):
                r = everything[everything["query"] == i]
                want.append(r[r["dist"] == r["dist"][0]])
            want_a = rows3(np.concatenate(want))
            got = rows[np.isin(rows[:, 0], pick)].copy()
            remap = np.full(Q, -1, dtype=np.int64)
            remap[pick] = np.arange(len(pick))
            got[:, 0] = remap[got[:, 0]]
            checks["oracle_sample_identical"] = sorted_rows(got).tobytes() == want_a.tobytes()
            checks["oracle_sample_queries"] = int(len(pick))
    else:
        last = (it[0] - 1) % len(bufs)
        d_hits, d_count = bufs[last][HEAD:], bufs[last][:2].view(torch.int64)
        n_rows = int(d_count.item())
        rows = sorted_rows(d_hits[: 3 * min(n_rows, cap)].cpu().numpy().view(np.uint32).reshape(-1, 3))
        checks["rows_fit"] = n_rows <= cap
        if multi:
            checks["rows_fit_gathered_width"] = n_rows * 3 + HEAD <= width[0]
            if gathered[last] is not None:
                # what the gather delivered (on the root; on every rank with all_gather): this rank's block must be its own
                # buffer's head, every rank's count within the gathered width
                g = gathered[last].view(world, width[0])
                checks["gather_block_is_own_buffer"] = bool(torch.equal(g[rank], bufs[last][: width[0]]))
                counts_g = [int(g[r][:2].view(torch.int64).item()) for r in range(world)]
                checks["gather_counts_in_range"] = all(0 <= c and c * 3 + HEAD <= width[0] for c in counts_g)
                checks["gathered_rows_total"] = int(sum(counts_g))
        # (1) soundness + (2) recall of the planted rows
        checks["distances_recomputed_and_planted_rows_present"] = verify_rows(
            np, subj, my_q, rows, D, planted_row[q_lo:q_lo + Q], planted_subs[q_lo:q_lo + Q])
        # (3) the prefilter is an exact early-out: the same launch with it switched the other way must give the same BYTES
        store.set_prefilter(not args.prefilter)
        n_off, rows_off = B.launch_rows(store, qset, D)
        other_kernel = store.last_scan_kernel()
        u_ms = []
        if side_legs and args.prefilter:
            for _ in range(5):
                store.scan_launch(qset, D, None, B.d_hits.data_ptr(), B.cap, B.d_count.data_ptr())
                u_ms.append(store.last_scan_ms()[0])
        store.set_prefilter(bool(args.prefilter))
        checks["filter_on_off_rows_identical"] = n_off == n_rows and rows_off.tobytes() == rows.tobytes()
        # (4) an oracle scan of 8 sampled queries against the whole store: complete row lists, bit for bit
        if rank == 0:
            import oracle

            oracle.build()
            pick = np.unique(np.concatenate([np.nonzero(planted_subs[q_lo:q_lo + Q] <= D)[0][:4],
                                             np.random.default_rng(5).integers(0, Q, size=4)]))[:8]
            checks["oracle_sample_identical"] = oracle_sample(np, subj, my_q, rows, D, pick)
            checks["oracle_sample_queries"] = int(len(pick))
    ok = all(v for k, v in checks.items() if isinstance(v, bool))
    if multi:
        flag = torch.tensor([1 if ok else 0], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = bool(flag.item())

    sym_bits = 8 if args.alphabet == "aa" else int(info.planes)
    # B_s = L x bits per symbol / 8 — aa 8 bits (60 B), nt 2 bits (15 B), nt with N 3 planes (22.5 B)
    alg_bytes_main = Q * N * L * sym_bits // 8

    unfiltered = None
    if side_legs and args.mode == "scan" and args.prefilter and u_ms:
        u_med = float(np.median(u_ms))
        unfiltered = {"kernel": other_kernel, "kernel_ms": u_med, "queries_per_s": Q / (u_med * 1e-3), "rows": n_off,
                      "roofline": roofline_block(dict(cfg, prefilter=0), other_kernel, u_med, Q * N, alg_bytes_main, build_id),
                      "note": "the same launch with the prefilter disabled (smafa_set_prefilter 0): all planes, all "
                              "words, for every pair; rows byte-identical (checks.filter_on_off_rows_identical)"}

    # ---- bounds level 1 cannot prune at: fixed bounds of 14 and 24 (scan_kernel's FOLD 1 / FOLD 2 forms at 60 columns)
    loose = None
    if side_legs and args.mode == "scan" and args.store == "uniform" and in_budget("loose_bounds", 5):
        loose = []
        for bound in (8, 14, 24):
            B.ensure(1 << 22)
            n_b, rows_b = B.launch_rows(store, qset, bound)
            k_ms, w_ms = B.kernel_ms(store, qset, bound, 5)
            kname = store.last_scan_kernel()
            v = n_b <= B.cap and verify_rows(np, subj, my_q, rows_b, bound, planted_row[q_lo:q_lo + Q], planted_subs[q_lo:q_lo + Q])
            loose.append({"max_divergence": bound, "kernel": kname, "kernel_ms": k_ms, "queries_per_s": Q / (k_ms * 1e-3),
                          "rows": n_b, "verified": bool(v),
                          "roofline": roofline_block(dict(cfg, max_div=bound), kname, k_ms, Q * N, alg_bytes_main, build_id)})
            ok = ok and bool(v)

    # ---- `smafa query` without --max-divergence (the reference's default mode, src/lib.rs:296-313) on queries half of
    #      which have NO near subject: the near-hit ladder finishes the near half, the far half pays the full comparison
    besthit = None
    if side_legs and args.mode == "scan" and args.store == "uniform" and in_budget("besthit_unbounded", 8):
        far_frac = 0.5
        bq = my_q.copy()
        n_far = int(Q * far_frac)
        bq[Q - n_far:] = synth.subjects(n_far, L, alphabet, seed=9, dup_frac=0.0)
        store.scan(bq[:256], max_divergence=None, max_num_hits=1)
        walls, kms, st = [], [], None
        for _ in range(3):
            tq = time.perf_counter()
            r_b = store.scan(bq, max_divergence=None, max_num_hits=1)
            walls.append((time.perf_counter() - tq) * 1e3)
            st = store.last_call_stats()
            kms.append(st["kernel_ms"])
        rb = rows3(r_b)
        rec = (subj[rb[:, 1]] != bq[rb[:, 0]]).sum(axis=1)
        v = bool((rec == rb[:, 2]).all()) and len(np.unique(rb[:, 0])) == Q
        # exactness of the minimum on a sample of far queries (no lower bound helps them: the oracle scans everything)
        import oracle

        pick = np.arange(Q - 2, Q)
        ev_all = oracle.scan_codes(subj, bq[pick], L)
        for i, qi in enumerate(pick):
            r = ev_all[ev_all["query"] == i]
            mine = rb[rb[:, 0] == qi]
            v = v and len(mine) == int((r["dist"] == r["dist"][0]).sum()) and int(mine[0, 2]) == int(r["dist"][0])
        k_med, w_med = float(np.median(kms)), float(np.median(walls))
        # the far half alone, as its own call: the rate a batch of novel sequences sees
        tq = time.perf_counter()
        store.scan(bq[Q - n_far:], max_divergence=None, max_num_hits=1)
        far_wall = (time.perf_counter() - tq) * 1e3
        far_st = store.last_call_stats()
        besthit = {"queries": Q, "far_fraction": far_frac, "wall_ms": w_med, "kernel_ms": k_med, "scans": st["scans"],
                   "launches": st["launches"], "queries_per_s_wall": Q / (w_med * 1e-3), "queries_per_s_kernel": Q / (k_med * 1e-3),
                   "far_queries_only": {"queries": n_far, "wall_ms": far_wall, "kernel_ms": far_st["kernel_ms"],
                                        "queries_per_s_wall": n_far / (far_wall * 1e-3),
                                        "queries_per_s_kernel": n_far / (far_st["kernel_ms"] * 1e-3) if far_st["kernel_ms"] else None},
                   "rows": int(len(rb)), "verified": bool(v),
                   "roofline": roofline_block(dict(cfg, mode="besthit", far_frac=far_frac, max_div=D), "scan kernels of one call",
                                              k_med, Q * N, alg_bytes_main, build_id),
                   "note": "smafa_scan_hits(max_num_hits = 1, no bound) — host code bytes in, ordered rows out; kernel_ms = "
                           "all scan kernels of the call (smafa_last_call_stats); the far half = uniform random letters "
                           "(seed 9): their nearest subject is ~45 columns away, every lower bound passes, every pair "
                           "gets the full comparison"}
        ok = ok and bool(v)

    # ---- the K branch (`smafa query --max-num-hits k`, k >= 2: src/lib.rs:242-295 — the reference's most expensive mode: an
    #      N x 16 B tuple vector and a full sort per query): the planted queries with k = 5 and 50, without a bound and with
    #      --max-divergence D.  Without a bound every query's k-th nearest subject is ~40 columns away on this store (uniform
    #      letters: one planted neighbour, the rest unrelated), i.e. the loose-bound path: count first, append second.
    kth = None
    if side_legs and args.mode == "scan" and args.store == "uniform" and not args.no_kth and in_budget("kth", 12):
        kth = {"note": "smafa_scan_hits(max_num_hits = k[, max_divergence]) on the %d planted queries: host code bytes in, ordered rows "
                       "out; kernel_ms = all scan kernels of the call; verified = every distance recomputed, the k-th rule's "
                       "invariants for every query, complete lists of 3 sampled queries == oracle distances + the rule" % Q}
        for k in (5, 50):
            for bound in (None, D):
                name = "kth%d" % k + ("_d%d" % bound if bound is not None else "")
                store.scan(my_q[:256], max_divergence=bound, max_num_hits=k)
                walls, kms, st = [], [], None
                for _ in range(3):
                    tq = time.perf_counter()
                    r_k = store.scan(my_q, max_divergence=bound, max_num_hits=k)
                    walls.append((time.perf_counter() - tq) * 1e3)
                    st = store.last_call_stats()
                    kms.append(st["kernel_ms"])
                rk = rows3(r_k)
                v = kth_rows_ok(np, subj, my_q, rk, k, bound, np.array([1, Q // 3, Q - 2]))
                k_med, w_med = float(np.median(kms)), float(np.median(walls))
                kth[name] = {"max_num_hits": k, "max_divergence": bound, "queries": Q, "wall_ms": w_med, "kernel_ms": k_med,
                             "queries_per_s_wall": Q / (w_med * 1e-3), "queries_per_s_kernel": Q / (k_med * 1e-3) if k_med else None,
                             "scans": st["scans"], "launches": st["launches"], "rows": int(len(rk)), "verified": bool(v),
                             "roofline": roofline_block(dict(cfg, mode="kth", kth_k=k, kth_bounded=bound is not None),
                                                        "scan kernels of one call", k_med, Q * N, alg_bytes_main, build_id)}
                ok = ok and bool(v)

    # ---- the block index (opt-in, smafa_db_build_index): the metric's launch answered by D + 1 probes per query
    indexed = None
    if side_legs and args.mode == "scan" and args.prefilter and not args.no_index and in_budget("indexed", 6):
        indexed = B.indexed_leg(store, qset, D, Q, kernel_ms_avg)
        ok = ok and indexed["verified"]
        if args.alphabet == "aa" and in_budget("indexed besthit", 4):
            indexed["besthit_planted"] = B.indexed_besthit(store, my_q)
            ok = ok and indexed["besthit_planted"]["verified"]

    # ---- stream mode: ONE query per pass — the HBM-bound form (north_star's literal "broadcast each query against
    #      all subjects"); three fractions of the 8 TB/s peak + the box's empirical read ceiling.
    def stream_leg(the_store, the_info, queries, n_rows_store, label, reps=200):
        K = min(reps, len(queries))
        qs_k = smafa_amd.QuerySet(the_store, queries[:K])
        cap_q = 256
        hits_k = torch.zeros(K * cap_q * 3, dtype=torch.int32, device=dev)
        counts_k = torch.zeros(K, dtype=torch.int64, device=dev)
        one = smafa_amd.QuerySet(the_store, queries[:1])

        def passes(graph):
            the_store.scan_each(qs_k, D, hits_k.data_ptr(), cap_q, counts_k.data_ptr(), use_graph=graph)  # warm / capture
            torch.cuda.synchronize()
            best = None
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                the_store.scan_each(qs_k, D, hits_k.data_ptr(), cap_q, counts_k.data_ptr(), use_graph=graph)
                e1.record(stream)
                torch.cuda.synchronize()
                w = e0.elapsed_time(e1) / K  # everything one query costs on the stream: launch, gaps, row bookkeeping
                best = w if best is None else min(best, w)
            return best

        def one_kernel():
            k_ms = []
            for _ in range(20):  # kernel-only time: HIP events recorded by the library right around the scan kernel
                the_store.scan_launch(one, D, None, B.d_hits.data_ptr(), B.cap, B.d_count.data_ptr())
                k_ms.append(the_store.last_scan_ms()[0])
            return float(np.median(k_ms)), the_store.last_scan_kernel(), the_store.last_scan_plan()

        # (a) the shipped form: on a sorted store a pass reads the zone words (16 B per 256 subjects) and fetches only the
        #     tiles the query survives; (b) zone level off: every pass streams the prefilter's whole bit-plane — the
        #     HBM-bound form the streamed fractions below are for
        wall_z = passes(True)
        k_z, kern_z, _ = one_kernel()
        rows_z = counts_k.cpu().numpy().copy()
        hz = hits_k.cpu().numpy().view(np.uint32).reshape(K, cap_q, 3)
        the_store.set_zone_level(0)
        wall_s_plain = passes(False)
        wall_s = passes(True)
        k_s1, kern_s, plan_s = one_kernel()
        rows_s = counts_k.cpu().numpy().copy()
        hs = hits_k.cpu().numpy().view(np.uint32).reshape(K, cap_q, 3)
        # (c) the same K passes as ONE launch: query blocks of one query (smafa_set_query_block 1) — the grid walks the query
        #     list workgroup after workgroup, every query still streams the whole plane, no kernel boundary between two passes
        big = torch.zeros(K * cap_q * 3, dtype=torch.int32, device=dev)
        total = torch.zeros(1, dtype=torch.int64, device=dev)
        the_store.set_query_block(1)
        wall_1 = None
        for rep in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            the_store.scan_launch(qs_k, D, None, big.data_ptr(), K * cap_q, total.data_ptr())
            e1.record(stream)
            torch.cuda.synchronize()
            if rep:
                w = e0.elapsed_time(e1) / K
                wall_1 = w if wall_1 is None else min(wall_1, w)
        plan_1, kern_1 = the_store.last_scan_plan(), the_store.last_scan_kernel()
        the_store.set_query_block(args.query_block or 0)
        n_1 = int(total.item())
        rows_1 = sorted_rows(big[: 3 * n_1].cpu().numpy().view(np.uint32).reshape(-1, 3))
        per_pass = [hs[i, : rows_s[i]].copy() for i in range(K)]
        for i, r in enumerate(per_pass):
            r[:, 0] = i  # scan_each numbers a pass's rows by the query's index in the set, as the one launch does
        same_1 = n_1 == int(rows_s.sum()) and rows_1.tobytes() == sorted_rows(np.concatenate(per_pass)).tobytes()
        # (d) prefilter off as well: every pass streams EVERY plane of every subject (north_star's literal "each query is
        #     broadcast against all subjects in a coalesced byte-wise mismatch-count scan") — scan_kernel, whole tiles
        the_store.set_prefilter(False)
        wall_f = passes(True)
        k_f, kern_f, _ = one_kernel()
        rows_f = counts_k.cpu().numpy().copy()
        hf = hits_k.cpu().numpy().view(np.uint32).reshape(K, cap_q, 3)
        the_store.set_prefilter(bool(args.prefilter))
        same_f = bool((rows_f == rows_s).all()) and all(
            sorted_rows(hf[i, : rows_f[i]]).tobytes() == sorted_rows(hs[i, : rows_s[i]]).tobytes() for i in range(K))
        the_store.set_zone_level(1)
        same = bool((rows_z == rows_s).all()) and all(
            sorted_rows(hz[i, : rows_z[i]]).tobytes() == sorted_rows(hs[i, : rows_s[i]]).tobytes() for i in range(K))
        sb = the_info.words_per_plane * 4 if plan_s["filter_plane_resident"] else the_info.bytes_per_subject
        streamed = the_info.hbm_bytes * sb / the_info.bytes_per_subject
        alg = n_rows_store * L * (8 if args.alphabet == "aa" else int(the_info.planes)) / 8
        # the same trivial sum over as many bytes as one pass streams (what a launch of that size can reach at all; a
        # buffer below 256 MB is also served by the Infinity Cache on repeats)
        ceiling_same = smafa_amd.hbm_read_probe(local_rank, int(streamed))
        one.close()
        qs_k.close()
        return {
            "store": label,
            "streamed_bytes_per_pass": int(streamed),
            "served_by": "HBM (plane larger than the 256 MB Infinity Cache)" if streamed > (256 << 20) else
                         "cache-resident: the plane fits the 256 MB Infinity Cache, NOT an HBM figure",
            "passes": K,
            "shipped": {"kernel": kern_z, "ms_per_query_wall": wall_z, "kernel_ms_median": k_z,
                        "algorithmic_x_of_peak": alg / k_z / 1e6 / HBM_PEAK_GBS, "rows": int(rows_z.sum())},
            "streaming": {"kernel": kern_s, "ms_per_query_wall": wall_s, "ms_per_query_wall_without_graph": wall_s_plain,
                          "kernel_ms_median": k_s1, "streamed_bytes_per_subject": int(sb),
                          "frac_kernel_streamed": streamed / k_s1 / 1e6 / HBM_PEAK_GBS,
                          "frac_wall_streamed": streamed / wall_s / 1e6 / HBM_PEAK_GBS,
                          "algorithmic_x_of_peak": alg / k_s1 / 1e6 / HBM_PEAK_GBS,
                          "kernel_streamed_GBs": streamed / k_s1 / 1e6, "wall_streamed_GBs": streamed / wall_s / 1e6,
                          "rows": int(rows_s.sum())},
            "streaming_one_launch": {"kernel": kern_1, "query_blocks": plan_1["query_blocks"], "ms_per_query_wall": wall_1,
                                     "frac_wall_streamed": streamed / wall_1 / 1e6 / HBM_PEAK_GBS,
                                     "wall_streamed_GBs": streamed / wall_1 / 1e6, "rows": n_1,
                                     "rows_identical_to_the_passes": same_1,
                                     "note": "the same passes as ONE launch with query blocks of one query "
                                             "(smafa_set_query_block 1): every query streams the whole plane, no kernel "
                                             "boundary between two passes; rows of all queries in one list"},
            "streaming_every_plane": {"kernel": kern_f, "streamed_bytes_per_pass": int(the_info.hbm_bytes),
                                      "served_by": "HBM (larger than the 256 MB Infinity Cache)" if the_info.hbm_bytes > (256 << 20)
                                                   else "cache-resident, NOT an HBM figure",
                                      "ms_per_query_wall": wall_f, "kernel_ms_median": k_f,
                                      "frac_wall_streamed": the_info.hbm_bytes / wall_f / 1e6 / HBM_PEAK_GBS,
                                      "frac_kernel_streamed": the_info.hbm_bytes / k_f / 1e6 / HBM_PEAK_GBS,
                                      "wall_streamed_GBs": the_info.hbm_bytes / wall_f / 1e6, "rows": int(rows_f.sum()),
                                      "rows_identical_to_the_passes": same_f,
                                      "note": "prefilter off (smafa_set_prefilter 0) and zone level off: every pass reads every "
                                              "plane of every subject and compares every pair in full"},
            "rows_identical": same and same_1 and same_f,
            "trivial_read_of_the_streamed_bytes_GBs": ceiling_same,
            "streaming_kernel_vs_trivial_read_of_the_same_bytes": (streamed / k_s1 / 1e6) / ceiling_same if ceiling_same else None,
        }

    stream_info = None
    if side_legs and args.mode == "scan" and in_budget("stream_small", 5):
        stream_info = {
            "note": "one query per store pass, %d passes enqueued back to back by ONE smafa_scan_each call (captured once as a "
                    "HIP graph, replayed; `ms_per_query_wall_without_graph`: the same launches enqueued one by one). "
                    "`shipped`: the default path (zone level: only the tiles the query survives are fetched, so the pass is "
                    "not a stream of the store and no streamed fraction is claimed for it). `streaming`: the same passes with "
                    "the zone level off (smafa_set_zone_level 0) — every pass streams the prefilter's bit-plane of every "
                    "subject: frac_kernel_streamed / frac_wall_streamed = those bytes over kernel time / wall time per "
                    "query as a fraction of 8 TB/s (roofline bound: hbm); algorithmic_x_of_peak = B_s per subject over "
                    "kernel time (a reuse figure, not an efficiency)" % 200,
            "metric_store": stream_leg(store, info, my_q, N, "%d x %d %s (the timed store)" % (N, L, args.alphabet)),
        }
        ok = ok and stream_info["metric_store"]["rows_identical"]

    # ---- host-buffer API (PCIe-inclusive): queries uploaded + packed, rows copied back and ordered on the host
    host_api = None
    if side_legs and args.mode == "scan":
        store.scan(my_q[:64], max_divergence=D)
        reps = 3
        tq = time.perf_counter()
        for _ in range(reps):
            rows_h = store.scan(my_q, max_divergence=D)
        dt = (time.perf_counter() - tq) / reps
        rh = rows3(rows_h)
        host_api = {"queries_per_s": Q / dt, "ms_per_batch": dt * 1e3, "rows": int(len(rows_h)),
                    "rows_identical_to_device_launch": rh.tobytes() == rows.tobytes(),
                    "note": "smafa_scan_hits: host code bytes in, ordered rows out (upload, pack, scan, copy back, sort)"}
        ok = ok and host_api["rows_identical_to_device_launch"]

    # ---- CPU baselines: the oracle's port of the reference's per-query loop, bounded samples
    cpu = None
    if rank == 0 and not multi and not args.no_cpu_baseline:
        b = cpu_baselines(N, L, D, args.alphabet, subj, my_q, args.cpu_seconds)
        cpu = {
            "value": b["b1"]["value"], "unit": "query seqs/s", "cores": 1, "kind": "port",
            "sample": "B1: first %d queries of a %d x %d nucleotide batch against a %d-row one-hot store (5-bit codes, "
                      "12 per u64 — src/lib.rs:29-52), oracle C port of src/lib.rs:238 (distances) + :298 (min) + :307 "
                      "(equality pass), gcc -O3 without POPCNT (cargo's release default), 1 thread — the reference is "
                      "single-threaded" % (b["b1"]["queries"], 512, L, N),
            "cpu_model": b["cpu_model"], "host_cores_available": b["nproc"],
            "b1": b["b1"], "b1n": b["b1n"], "b2": b["b2"], "kmode": b.get("kmode"),
        }
        if "aa_code_bytes" in b:
            cpu["aa_code_bytes"] = b["aa_code_bytes"]

    # the timed store is not needed any more: free its host rows and HBM before the other stores are built
    main_value = Q * world * args.steps / elapsed_s
    qset.close()
    store.close()
    del subj, all_q

    # ---- related store: 100 members per family at 10-25 % divergence from the family root — the regime smafa is used
    #      in (homologous windows), where a lower-bound prefilter has less to reject than on uniform letters
    related = None
    if side_legs and args.mode == "scan" and not args.no_related and args.store == "uniform" and in_budget("related", 25):
        tg = time.time()
        r_subj = synth.related_subjects(max(N // 100, 1), 100, L, alphabet, seed=7)
        r_q, r_row, r_subs = synth.queries(r_subj, Q, alphabet, seed=8, max_subs=max_subs)
        tg = time.time() - tg
        r_store = smafa_amd.SubjectStore(L, alphabet, local_rank)
        r_store.push(r_subj)
        r_store.set_stream(stream.cuda_stream)
        r_qset = smafa_amd.QuerySet(r_store, r_q)
        n_r, rows_r = B.launch_rows(r_store, r_qset, D)
        r_med, _ = B.kernel_ms(r_store, r_qset, D, 10)
        r_kernel = r_store.last_scan_kernel()
        r_store.set_prefilter(False)
        n_r_off, rows_r_off = B.launch_rows(r_store, r_qset, D)
        r_store.set_prefilter(True)
        r_ok = (n_r <= B.cap and n_r == n_r_off and rows_r.tobytes() == rows_r_off.tobytes()
                and verify_rows(np, r_subj, r_q, rows_r, D, r_row, r_subs))
        # the reference's default mode on this store (`smafa query` without --max-divergence, src/lib.rs:296-313) for queries
        # that are NEW members of the store's families: the family root re-diverged by 10-25 % (seed 11), i.e. sequences that
        # are not in the store but have relatives there, ~10-25 columns away — what a novel homologue looks like, as opposed to
        # the uniform-random queries of `besthit_unbounded`
        novel = synth.related_subjects(max(N // 100, 1), 1, L, alphabet, seed=7)  # one member per root, the same roots (seed 7)
        rng_n = np.random.default_rng(11)
        novel = novel[rng_n.integers(0, len(novel), size=Q)]
        r_store.scan(novel[:256], max_divergence=None, max_num_hits=1)
        walls, kms = [], []
        for _ in range(3):
            tq = time.perf_counter()
            nb = r_store.scan(novel, max_divergence=None, max_num_hits=1)
            walls.append((time.perf_counter() - tq) * 1e3)
            kms.append(r_store.last_call_stats()["kernel_ms"])
        nbr = rows3(nb)
        rec_n = (r_subj[nbr[:, 1]] != novel[nbr[:, 0]]).sum(axis=1)
        best_n = np.full(Q, 1 << 30, dtype=np.int64)
        np.minimum.at(best_n, nbr[:, 0], nbr[:, 2])
        import oracle

        ev_n = oracle.scan_codes(r_subj, novel[:2], L)
        nov_ok = bool((rec_n == nbr[:, 2]).all()) and len(np.unique(nbr[:, 0])) == Q and all(
            int(best_n[i]) == int(ev_n[ev_n["query"] == i]["dist"][0]) for i in range(2))
        besthit_related = {"queries": Q, "wall_ms": float(np.median(walls)), "kernel_ms": float(np.median(kms)),
                           "queries_per_s_wall": Q / (float(np.median(walls)) * 1e-3),
                           "best_hit_distance_quartiles": [int(x) for x in np.percentile(best_n, [25, 50, 75])],
                           "scans": r_store.last_call_stats()["scans"], "verified": nov_ok,
                           "note": "best hit without a bound for %d novel family members (family roots re-diverged by 10-25 %%, "
                                   "not in the store) against the related store: the near-hit ladder answers them" % Q}
        r_ok = r_ok and nov_ok
        if not args.no_index and args.alphabet == "aa" and in_budget("indexed besthit novel", 5):
            # ... and with a block index wide enough for most of them (bounds up to 14: 15 blocks of 4 columns)
            besthit_related["indexed"] = B.indexed_besthit(r_store, novel, wide=14)
            r_ok = r_ok and besthit_related["indexed"]["verified"]
        related = {"kernel": r_kernel, "kernel_ms": r_med, "queries_per_s": Q / (r_med * 1e-3), "rows": n_r,
                   "besthit_novel_members": besthit_related,
                   "slowdown_vs_uniform": r_med / kernel_ms_avg, "verified": bool(r_ok),
                   "roofline": roofline_block(dict(cfg, store="related", db_rows=len(r_subj)), r_kernel, r_med, Q * len(r_subj),
                                              Q * len(r_subj) * L * sym_bits // 8, build_id),
                   "workload": "%d families x 100 members, each member 10-25 %% of its columns substituted against the "
                               "family root (rows shuffled, seed 7); %d queries = store members with 0..%d substitutions "
                               "(seed 8); max-divergence %d" % (len(r_subj) // 100, Q, max_subs, D),
                   "generate_s": tg}
        ok = ok and bool(r_ok)
        r_qset.close()
        r_store.close()
        del r_subj

    # ---- BASELINE.json configs[1..4], each verified, each with kernel and wall time (N = 1: configs[3] = one rank's share)
    configs = None
    if side_legs and args.mode == "scan" and not args.no_configs and args.store == "uniform" and args.alphabet == "aa" \
            and ((N, Q, D) == (10_000_000, 10_000, 5) or args.configs_scale != 1.0):
        configs = {}
        cs = args.configs_scale

        def query_config(name, n, q, alpha, d, n_frac=0.0, stream_too=False, need_s=20):
            if not in_budget(name, need_s):
                return
            n, q = max(int(n * cs), 4096), max(int(q * cs), 256)
            a_name = "aa" if alpha else "nt"
            tg = time.time()
            s_c = synth.subjects(n, L, alpha, seed=1 if alpha else 2, n_frac=n_frac)
            q_c, p_row, p_subs = synth.queries(s_c, q, alpha, seed=3, max_subs=10 if alpha else 6)
            tg = time.time() - tg
            tp = time.time()
            st_c = smafa_amd.SubjectStore(L, alpha, local_rank)
            st_c.push(s_c)
            st_c.set_stream(stream.cuda_stream)
            qs_c = smafa_amd.QuerySet(st_c, q_c)
            tp = time.time() - tp
            i_c = st_c.info()
            B.ensure(max(4 * q, 1 << 16))
            n_c, rows_c = B.launch_rows(st_c, qs_c, d)
            k_ms, w_ms = B.kernel_ms(st_c, qs_c, d, 5 if n >= 50_000_000 else 10)
            kname = st_c.last_scan_kernel()
            import oracle

            pick = np.unique(np.concatenate([np.nonzero(p_subs <= d)[0][:4], np.random.default_rng(5).integers(0, q, size=4)]))[:8]
            v = (n_c <= B.cap and verify_rows(np, s_c, q_c, rows_c, d, p_row, p_subs) and oracle_sample(np, s_c, q_c, rows_c, d, pick))
            bits = 8 if alpha else int(i_c.planes)
            c_cfg = {"db_rows": n, "seq_len": L, "queries": q, "max_div": d, "alphabet": a_name, "store": "uniform",
                     "n_frac": n_frac, "prefilter": 1, "mode": "scan"}
            configs[name] = {
                "workload": "%d x %d %s store (uniform, seed %d%s), %d planted queries, max-divergence %d"
                            % (n, L, a_name, 1 if alpha else 2, ", N with probability %g per column" % n_frac if n_frac else "", q, d),
                "planes": int(i_c.planes), "kernel": kname, "kernel_ms": k_ms, "wall_ms_per_launch": w_ms,
                "queries_per_s": q / (k_ms * 1e-3), "queries_per_s_wall": q / (w_ms * 1e-3),
                "residues_per_s": q / (k_ms * 1e-3) * n * L, "rows": n_c, "verified": bool(v),
                "checks": "every distance recomputed, planted rows present, oracle scan of %d sampled queries identical" % len(pick),
                "roofline": roofline_block(c_cfg, kname, k_ms, q * n, q * n * L * bits // 8, build_id),
                "setup_s": {"generate": tg, "pack_upload": tp},
            }
            if not args.no_index and in_budget("indexed " + name, 4 + n // 5_000_000):
                configs[name]["indexed"] = B.indexed_leg(st_c, qs_c, d, q, k_ms, 5 if n >= 50_000_000 else 10)
                v = v and configs[name]["indexed"]["verified"]
                configs[name]["verified"] = bool(v)
            if n <= (256 << 20) // int(i_c.bytes_per_subject):
                configs[name]["cache_note"] = "the packed store (%d MB) fits the 256 MB Infinity Cache" % (i_c.hbm_bytes >> 20)
            extra = None
            if stream_too and in_budget("stream_" + name, 10):
                extra = stream_leg(st_c, i_c, q_c, n, "%d x %d %s" % (n, L, a_name))
            qs_c.close()
            st_c.close()
            return bool(v), extra

        r1 = query_config("configs[1] 1M aa", 1_000_000, 10_000, 1, 5, need_s=5)
        r2 = query_config("configs[2] 10M nt, N-free (2-bit store)", 10_000_000, 100_000, 0, 3, need_s=15)
        r3 = query_config("configs[2] 10M nt, 0.1 % N (3-plane store)", 10_000_000, 100_000, 0, 3, n_frac=0.001, need_s=15)
        r4 = query_config("configs[3] one rank's share: 50M aa x 125k of 1M queries", 50_000_000, 125_000, 1, 5,
                          stream_too=True, need_s=60)
        for r in (r1, r2, r3, r4):
            if r is not None:
                ok = ok and r[0]
        if r4 is not None and r4[1] is not None and stream_info is not None:
            stream_info["hbm_store"] = r4[1]
            ok = ok and r4[1]["rows_identical"]

        # configs[4]: `smafa cluster` (src/cluster.rs:13-94) on 5M x 60 aa records through the product CLI
        if in_budget("configs[4] cluster", 60):
            configs["configs[4] cluster 5M aa"] = cluster_config(np, synth, args, build_id, max(int(100_000 * cs), 200))
            ok = ok and configs["configs[4] cluster 5M aa"]["verified"]

    if stream_info is not None:
        # HBM bytes per streaming pass from the hardware counter (rocprofv3 --pmc FETCH_SIZE around tools/stream_probe.py on the
        # 50M store, tools/stream_pmc.py): the committed record, not measured in this run
        for path in STREAM_PMC_JSONS:
            if os.path.exists(path):
                rec = json.load(open(path))
                for kname, r in rec.items():
                    if isinstance(r, dict) and "scan_lazy_kernel" in kname:
                        stream_info["fetched_over_plane"] = r.get("fetched_over_plane")
                        stream_info["counters"] = {"source": os.path.relpath(path, ROOT), "kernel": kname, "build_id": rec.get("build_id"),
                                                   "hbm_bytes_per_pass": r.get("hbm_bytes_per_pass_avg"),
                                                   "kernel_us_median_under_profiler": r.get("kernel_us_median"),
                                                   "frac_of_peak_by_counter_bytes": r.get("frac_of_8TBs")}
                break
        # the box's empirical read ceiling: a trivial sum over 8 GiB
        ceiling = smafa_amd.hbm_read_probe(local_rank, 8 << 30)
        stream_info["empirical_read_ceiling_GBs"] = ceiling
        stream_info["empirical_read_ceiling_frac_of_peak"] = ceiling / HBM_PEAK_GBS
        best = stream_info.get("hbm_store") or stream_info["metric_store"]
        stream_info["roofline"] = {"bound": "hbm", "kernel": best["streaming"]["kernel"],
                                   "achieved": best["streaming"]["kernel_streamed_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": best["streaming"]["frac_kernel_streamed"],
                                   "frac_wall": best["streaming"]["frac_wall_streamed"],
                                   "frac_wall_one_launch": best["streaming_one_launch"]["frac_wall_streamed"],
                                   "frac_wall_every_plane_metric_store": stream_info["metric_store"]["streaming_every_plane"]["frac_wall_streamed"],
                                   "traffic": best["streamed_bytes_per_pass"], "store": best["store"], "served_by": best["served_by"]}

    if rank == 0:
        q_total = Q * world
        value = main_value
        pairs_per_launch = Q * N
        filt = os.environ.get("SMAFA_FILTER", "1") != "0" and bool(args.prefilter)
        roof = roofline_block(cfg, kernel_name if args.mode == "scan" else "scan kernels of one call", kernel_ms_avg,
                              pairs_per_launch, alg_bytes_main, build_id,
                              {"plan": plan, "prefilter": filt, "stored_bytes_per_subject": int(info.bytes_per_subject)})
        roof["note"] = ("integer compare/reduce: the launch is bound by VALU issue, not HBM. peak = 256 CU x 4 SIMD x "
                        "32 lanes x 2.4 GHz (one wave64 op per 2 cycles); measured issue rates on this chip "
                        "(profiles/r01_ubench_valu*.txt): all-VGPR xor/bitop3/add ~60e12, v_bcnt/v_cmp/v_readlane and any "
                        "op with an SGPR source ~37e12 lane-ops/s = 0.48 of peak, and the zone kernel's survivor loop is "
                        "made of those (frac / 0.48 = its share of the ceiling of its own instruction mix). "
                        "algorithmic_reuse_x = queries x subjects x %g B (SURVEY 8d) over kernel time over 8 TB/s: how many "
                        "times the naive one-query-per-pass traffic would exceed HBM peak — register reuse of a tile across "
                        "a query block plus exact early-outs, NOT an HBM efficiency; the HBM-bound form is in `stream` "
                        "(stream.roofline: bound hbm)." % (L * sym_bits / 8))
        out = {
            "metric": "query seqs/sec (DB residues/sec in `residues_per_s`) vs roofline, %dM x %d%s DB, d<=%d"
                      % (N // 1_000_000, L, args.alphabet, D),
            "value": value,
            "unit": "query seqs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed_s / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {
                "workload": ("%d x %d %s subject store (%s) replicated per GPU; %d planted queries per GPU per step "
                             "(0..%d substitutions, seed 3); %s"
                             % (N, L, args.alphabet,
                                "related families: 100 members at 10-25 %% divergence from their root, seed 7"
                                if args.store == "related" else
                                "uniform letters, 1%% duplicate rows, seed %d%s" % (
                                    1 if alphabet else 2,
                                    ", N with probability %g per column" % args.n_frac if args.n_frac else ""),
                                Q, max_subs,
                                "max-divergence %d" % D if args.mode == "scan" else
                                "max_num_hits %d (the K branch), %s" % (call_k, "max-divergence %d" % D if args.kth_bounded else "no bound")
                                if args.mode == "kth" else
                                "best hit without a bound (max_num_hits 1), %g of the queries uniform random" % args.far_frac)),
                "db_rows": N, "seq_len": L, "alphabet": args.alphabet, "queries_per_gpu": Q, "max_divergence": D,
                "store": args.store, "n_frac": args.n_frac, "prefilter": args.prefilter, "mode": args.mode,
                "far_frac": args.far_frac, "kth_k": args.kth_k, "kth_bounded": bool(args.kth_bounded),
                "parallelism": "query shards x%d, DB replicated, RCCL %s of row lists to rank 0" % (world, args.collective),
            },
            "residues_per_s": value * N * L,
            "pairs_per_s": value * N,
            "rows_per_step": n_rows,
            "verified": ok,
            "checks": checks,
            "build_id": build_id,
            "roofline": roof,
            "stream": stream_info,
            "unfiltered": unfiltered,
            "loose_bounds": loose,
            "besthit_unbounded": besthit,
            "kth": kth,
            "indexed": indexed,
            "related": related,
            "host_api": host_api,
            "configs": configs,
            "cpu_baseline": cpu,
            "gathered_bytes_per_rank_per_step": gathered_bytes,
            "calls_total": it[0],
            "wall_ms_per_step_besthit": float(np.mean(wall_ms)) if wall_ms else None,
            "setup_s": {"generate": t_gen, "pack_upload": t_up},
            "run_s": round(elapsed(), 1),
            "skipped_for_time": skipped,
        }
        emit(out, args.full_record)
    if multi:
        dist.destroy_process_group()
    return 0


def cluster_config(np, synth, args, build_id, n_roots=100_000):
    """BASELINE configs[4]: `smafa cluster -d 5 --alphabet aa` on 5M x 60 aa records (100k roots x 50 members, 0..4
    substitutions, shuffled, seed 4 — SURVEY 8d) through the product CLI; verified by (i) a 60k-record prefix run that must
    equal the oracle's sequential greedy loop byte for byte and (ii) the full-size properties of the reference's algorithm
    on the 5M output; CPU baseline = the oracle's loop on a bounded prefix, extrapolated by pairs compared."""
    import oracle
    from smafa_amd import _lib

    D = 5
    letters = np.array([ord("A") + i for i in range(26)] + [ord("*"), ord("-")], dtype=np.uint8)
    back = np.full(256, 255, dtype=np.uint8)
    back[letters] = np.arange(28, dtype=np.uint8)
    out = {"workload": "100 000 uniform roots x 50 members, each member its root with 0..4 substitutions, shuffled (seed 4); "
                       "max-divergence 5; `smafa cluster` through the product CLI (parse, duplicate skip, GPU scans, output)"}

    def write_fasta(path, codes):  # fixed-width headers: one vectorised write
        n, L = codes.shape
        ids = np.char.zfill(np.arange(n).astype("U8"), 8).astype("S8")
        rec = np.empty((n, 1 + 8 + 1 + L + 1), dtype=np.uint8)
        rec[:, 0] = ord(">")
        rec[:, 1:9] = np.frombuffer(ids.tobytes(), dtype=np.uint8).reshape(n, 8)
        rec[:, 9] = 10
        rec[:, 10:10 + L] = letters[codes]
        rec[:, 10 + L] = 10
        rec.tofile(path)

    def run_cli(path):
        # stdout to a file (as a shell redirection would): through a pipe the run would be timed at the speed this Python
        # process drains ~0.5 GB of lines, not at the CLI's
        out_path = path + ".out"
        t = time.perf_counter()
        with open(out_path, "wb") as fo:
            r = subprocess.run([_lib.CLI_PATH, "cluster", "-i", path, "-d", str(D), "--alphabet", "aa", "-v"], stdout=fo,
                               stderr=subprocess.PIPE)
        wall = time.perf_counter() - t
        r.stdout = open(out_path, "rb").read()
        os.unlink(out_path)
        return r, wall

    with tempfile.TemporaryDirectory() as tmp:
        # (i) prefix == oracle sequential
        small = synth.cluster_records(2_000, 30, 60, 1, seed=4, max_subs=4)
        f_small = os.path.join(tmp, "small.faa")
        write_fasta(f_small, small)
        r, _ = run_cli(f_small)
        assigned = oracle.cluster_codes(small, D, oracle.ALPHABET_AA)
        ascii_rows = letters[small]
        keep = assigned != 0xFFFFFFFF
        first_of = {}
        for i in np.nonzero(keep)[0]:
            first_of.setdefault(int(assigned[i]), i)
        cent_rows = ascii_rows[[first_of[int(a)] for a in assigned[keep]]]
        want = np.concatenate([ascii_rows[keep], np.full((keep.sum(), 1), 9, np.uint8), cent_rows,
                               np.full((keep.sum(), 1), 10, np.uint8)], axis=1).tobytes()
        prefix_ok = r.returncode == 0 and r.stdout == want
        # (ii) the full size
        tg = time.time()
        recs = synth.cluster_records(n_roots, 50, 60, 1, seed=4, max_subs=4)
        f_big = os.path.join(tmp, "big.faa")
        write_fasta(f_big, recs)
        tg = time.time() - tg
        r, wall = run_cli(f_big)
        log = r.stderr.decode(errors="replace")
        if r.returncode != 0:
            out.update({"verified": False, "error": log[-400:]})
            return out
        raw = np.frombuffer(r.stdout, dtype=np.uint8).reshape(-1, 122)
        member, centroid = back[raw[:, :60]], back[raw[:, 61:121]]
        d = (member != centroid).sum(axis=1)
        is_cent = d == 0
        cent, cent_pos = member[is_cent], np.nonzero(is_cent)[0]
        void = lambda a: np.ascontiguousarray(a).view(np.dtype((np.void, a.shape[1]))).ravel()
        _, first_idx = np.unique(void(recs), return_index=True)
        first_idx.sort()
        props = len(member) == len(first_idx) and bool((member == recs[first_idx]).all()) and int(d.max()) <= D \
            and len(np.unique(void(cent))) == len(cent) and len(np.unique(void(centroid))) == len(cent)
        rng = np.random.default_rng(5)
        for ci in rng.choice(len(cent), size=40, replace=False):  # a centroid is > D from every EARLIER centroid
            if ci:
                props = props and int((cent[:ci] != cent[ci]).sum(axis=1).min()) > D
        for ri in rng.choice(len(member), size=40, replace=False):  # a member's centroid = nearest earlier one, lowest index
            nb = int(np.searchsorted(cent_pos, ri, side="left"))
            if is_cent[ri]:
                continue
            dd = (cent[:nb] != member[ri]).sum(axis=1)
            best = int(np.argmin(dd))
            props = props and int(dd[best]) <= D and bool((cent[best] == centroid[ri]).all())
        # pairs the reference compares: record i is scanned against every centroid that exists when it arrives
        # (src/cluster.rs:51); duplicates are skipped before the scan (:46-48)
        pairs = int(np.searchsorted(cent_pos, np.arange(len(member)), side="left").sum())
        stages = {}
        for key, pat in (("parse_s", r"parsed \d+ records in ([\d.]+) s"), ("duplicates_s", r"distinct found in ([\d.]+) s"),
                         ("scans_vs_old_centroids_s", r"scans vs old centroids ([\d.]+) s"),
                         ("candidate_scans_s", r"candidate scans ([\d.]+) s"), ("sequential_pass_s", r"sequential pass ([\d.]+) s"),
                         ("centroid_appends_s", r"centroid appends ([\d.]+) s"), ("output_s", r"lines written in ([\d.]+) s"),
                         ("scan_kernels_ms", r"scan kernels ([\d.]+) ms")):
            m = re.search(pat, log)
            if m:
                stages[key] = float(m.group(1))
        out.update({"records": int(len(recs)), "lines": int(len(member)), "centroids": int(len(cent)), "wall_s": wall,
                    "records_per_s": len(recs) / wall, "stages": stages, "pairs_the_reference_compares": pairs,
                    "verified": bool(prefix_ok and props),
                    "checks": {"prefix_60k_equals_oracle_sequential": bool(prefix_ok), "full_size_properties": bool(props)},
                    "setup_s": {"generate_and_write": tg}})
        rec = None
        for r_ in pmc_records():
            if r_.get("config", {}).get("mode") == "cluster":
                rec = r_
        if rec and rec.get("per_launch", {}).get("SQ_INSTS_VALU") and stages.get("scan_kernels_ms"):
            lane_ops = float(rec["per_launch"]["SQ_INSTS_VALU"]) * 64.0 / (stages["scan_kernels_ms"] * 1e-3)
            out["roofline"] = {"bound": "valu", "kernel": "all scan kernels of the run", "achieved": lane_ops / 1e12,
                               "peak": VALU_PEAK_LANE_OPS / 1e12, "unit": "Tlane-op/s", "frac": lane_ops / VALU_PEAK_LANE_OPS,
                               "kernel_ms_total": stages["scan_kernels_ms"],
                               "insts_source": "%s (rocprofv3 --pmc on the same CLI command), build %s; this run is build %s"
                                               % (rec["_file"], rec.get("build_id"), build_id),
                               "insts_source_is_this_build": rec.get("build_id") == build_id}
        if not args.no_cpu_baseline:
            # the oracle's sequential loop on a prefix of the SAME records; its pair count from its own assignment
            n_pre = 40_000
            while True:
                t = time.perf_counter()
                a_pre = oracle.cluster_codes(recs[:n_pre], D, oracle.ALPHABET_AA)
                dt = time.perf_counter() - t
                if dt >= args.cpu_seconds * 0.5 or n_pre >= len(recs):
                    break
                n_pre = min(len(recs), int(n_pre * max(2.0, args.cpu_seconds / max(dt, 1e-3))))
            keep = a_pre != 0xFFFFFFFF
            is_new = np.zeros(len(a_pre), dtype=bool)
            seen_max = -1
            new_at = []
            for i in np.nonzero(keep)[0]:
                if int(a_pre[i]) > seen_max:
                    seen_max = int(a_pre[i])
                    new_at.append(i)
            cpos = np.array(new_at)
            pre_pairs = int(np.searchsorted(cpos, np.nonzero(keep)[0], side="left").sum())
            rate = pre_pairs / dt
            out["cpu_baseline"] = {"value": n_pre / dt, "unit": "records/s on the prefix", "cores": 1, "kind": "port",
                                   "sample": "oracle orc_cluster_codes (src/cluster.rs:35-85 restated: dedup, scan vs every "
                                             "centroid, first minimum) on the first %d records, 1 thread, gcc -O3" % n_pre,
                                   "prefix_seconds": dt, "prefix_pairs": pre_pairs, "pairs_per_s": rate,
                                   "extrapolated_full_run_s": pairs / rate if rate else None,
                                   "extrapolation": "the loop's cost is the pairs it compares (one scan of all current centroids "
                                                    "per record): full-run pairs / prefix pairs-per-second",
                                   "cpu_model": cpu_model()}
            out["speedup_vs_cpu_extrapolated"] = (pairs / rate) / wall if rate else None
    return out


if __name__ == "__main__":
    sys.exit(main())
