#!/usr/bin/env python3
"""Collect the rocprofv3 evidence bench.py's `roofline` block cites, on the GPU box:

    python3 tools/collect_pmc.py [--tag NAME] [-- extra bench.py flags]

  1. `rocprofv3 --kernel-trace --stats` of `python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-stream`
     -> per-kernel average durations (must agree with the HIP-event average of the bench line);
  2. four separate `--pmc` passes of the same command (counters never share a run with a trace other than the
     kernel trace, and FETCH_SIZE / WRITE_SIZE cannot share a pass — MI355X_MICROARCH.md, PMC slots):
     SQ instruction/cycle counters (two passes), FETCH_SIZE, WRITE_SIZE;
  3. per-launch averages over the dispatches of the bench line's dominant kernel, HBM bytes by the guide's rule
     (FETCH_SIZE is KiB and reads HALF of a wide streaming read on gfx950: bytes = FETCH_SIZE * 1024 * 2,
     plus WRITE_SIZE * 1024), derived issue figures.
Everything lands under gpurun_out/<tag>/ (scratch); the summary JSON is gpurun_out/<tag>/pmc_record.json — copy it
into profiles/r02_pmc.json ("records": [...]) to have bench.py use it.
The program after `--` is python3 itself: no env/bash hop between rocprofv3 and the process that opens the GPU.
"""
import argparse
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SQ_A = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES",
        "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY"]
SQ_B = ["GRBM_GUI_ACTIVE", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_LDS_BANK_CONFLICT",
        "SQ_WAIT_ANY", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD"]


def run(cmd, log):
    with open(log, "w") as f:
        r = subprocess.run(cmd, stdout=f, stderr=subprocess.STDOUT, cwd=ROOT)
    return r.returncode


def bench_line(log):
    """the FULL record of the bench run behind `log` (bench.py prints a compact line and names the file that holds the rest)"""
    for line in open(log, errors="replace"):
        line = line.strip()
        if line.startswith("{") and '"metric"' in line:
            short = json.loads(line)
            path = short.get("full_record")
            if path:
                path = path if os.path.isabs(path) else os.path.join(ROOT, path)
                if os.path.exists(path):
                    return json.load(open(path))
            return short
    return None


def counter_rows(out_dir):
    rows = []
    for path in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            rows.extend(csv.DictReader(f))
    return rows


def per_call(rows, calls):
    """counter totals over EVERY scan kernel dispatch of the run, divided by the number of identical calls the run made
    (mode besthit: one smafa_scan_hits call is a ladder of scans; cluster: calls = 1, the whole run)"""
    # (the k-th modes' bound kernel and row filter belong to the call as well)
    mine = [r for r in rows if any(k in r["Kernel_Name"] for k in ("smafa::scan_", "kth_seed_kernel", "kth_from_counts_kernel", "filter_rows_kernel"))]
    if not mine:
        return {}, 0, 0.0, {}
    disp, by_kernel = {}, {}
    for r in mine:
        d = disp.setdefault(r["Dispatch_Id"], {"_k": r["Kernel_Name"]})
        d[r["Counter_Name"]] = float(r["Counter_Value"])
        d["_ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    names = sorted({k for d in disp.values() for k in d if not k.startswith("_")})
    tot = {n: sum(d.get(n, 0.0) for d in disp.values()) / calls for n in names}
    ms = sum(d["_ns"] for d in disp.values()) / calls / 1e6
    for d in disp.values():
        k = by_kernel.setdefault(d["_k"].split("(")[0], {"dispatches": 0, "ms": 0.0})
        k["dispatches"] += 1.0 / calls
        k["ms"] += d["_ns"] / 1e6 / calls
        for n in names:
            k[n] = k.get(n, 0.0) + d.get(n, 0.0) / calls
    return tot, len(disp), ms, by_kernel


def per_launch(rows, kernel_sub):
    """average counter value per dispatch of the kernel, over its full-size dispatches"""
    mine = [r for r in rows if kernel_sub in r["Kernel_Name"]]
    if not mine:
        return {}, 0, 0.0
    grid = max(int(r["Grid_Size"]) for r in mine)
    mine = [r for r in mine if int(r["Grid_Size"]) == grid]
    disp = {}
    for r in mine:
        disp.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
        disp[r["Dispatch_Id"]]["_ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    names = sorted({k for d in disp.values() for k in d if not k.startswith("_")})
    avg = {n: sum(d.get(n, 0.0) for d in disp.values()) / len(disp) for n in names}
    ms = sum(d["_ns"] for d in disp.values()) / len(disp) / 1e6
    return avg, len(disp), ms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="r02_prof")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--skip-stats", action="store_true")
    ap.add_argument("--passes", default="sq_a,sq_b,fetch,write", help="counter passes to run (sq_a carries SQ_INSTS_VALU)")
    ap.add_argument("--cluster", action="store_true",
                    help="profile `smafa cluster -d 5 --alphabet aa` on BASELINE configs[4]'s 5M records instead of bench.py")
    ap.add_argument("bench_args", nargs="*", help="extra bench.py flags (after --)")
    a = ap.parse_args()
    out = os.path.join(ROOT, "gpurun_out", a.tag)
    os.makedirs(out, exist_ok=True)
    os.environ.setdefault("TMPDIR", "/tmp")
    bench = ["python3", os.path.join(ROOT, "bench.py"), "--steps", str(a.steps), "--warmup", "2", "--no-cpu-baseline",
             "--no-stream", *a.bench_args]
    prof = "/opt/rocm/bin/rocprofv3"
    if a.cluster:
        return cluster_main(a, out, prof)

    record = {"command": " ".join(bench[1:]).replace(ROOT + "/", "")}
    if not a.skip_stats:
        rc = run([prof, "--kernel-trace", "--stats", "-d", os.path.join(out, "stats"), "-o", "stats", "--output-format", "csv",
                  "--", *bench, "--full-record", os.path.join(out, "stats_full.json")], os.path.join(out, "stats.log"))
        print("stats pass rc", rc, flush=True)
        line = bench_line(os.path.join(out, "stats.log"))
        if line:
            record["bench_under_kernel_trace"] = {k: line[k] for k in ("value", "ms_per_step", "verified")}
            record["kernel_ms_avg_hip_events_under_trace"] = line["roofline"]["kernel_ms_avg"]
        for path in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
            with open(path, newline="") as f:
                record["kernel_stats"] = [r for r in csv.DictReader(f)][:12]
    passes = [p for p in (("sq_a", SQ_A), ("sq_b", SQ_B), ("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"]))
              if p[0] in a.passes.split(",")]
    merged, n_disp, kernel, cfg, ms_prof, by_kernel = {}, {}, None, None, {}, {}
    for name, counters in passes:
        d = os.path.join(out, name)
        rc = run([prof, "--pmc", *counters, "--kernel-trace", "-d", d, "-o", name, "--output-format", "csv", "--", *bench,
                  "--full-record", os.path.join(out, name + "_full.json")], os.path.join(out, name + ".log"))
        print(name, "pass rc", rc, flush=True)
        line = bench_line(os.path.join(out, name + ".log"))
        if not line:
            print("no bench line in", name, "— see", os.path.join(out, name + ".log"))
            continue
        kernel = line["roofline"]["kernel"].split(" (")[0]  # (a note like " (zone level on)" is not part of the symbol)
        cfg = line["config"]
        record["build_id"] = line.get("build_id")
        if cfg.get("mode") in ("besthit", "kth"):  # one call = several scans: totals over all scan kernels per call
            avg, nd, ms, bk = per_call(counter_rows(d), line["calls_total"])
            by_kernel[name] = bk
        else:
            avg, nd, ms = per_launch(counter_rows(d), kernel.replace("smafa::", ""))
        merged.update(avg)
        n_disp[name] = nd
        ms_prof[name] = ms
    if not merged:
        print("no counters collected")
        return 1
    if "FETCH_SIZE" in merged:
        merged["hbm_bytes"] = merged["FETCH_SIZE"] * 1024.0 * 2.0 + merged.get("WRITE_SIZE", 0.0) * 1024.0
    record.update({
        "kernel": kernel,
        "config": {"db_rows": cfg["db_rows"], "seq_len": cfg["seq_len"], "queries": cfg["queries_per_gpu"],
                   "max_div": cfg["max_divergence"], "alphabet": cfg["alphabet"], "store": cfg.get("store", "uniform"),
                   "n_frac": cfg.get("n_frac", 0.0), "prefilter": cfg.get("prefilter", 1), "mode": cfg.get("mode", "scan"),
                   "far_frac": cfg.get("far_frac", 0.5), "kth_k": cfg.get("kth_k", 0), "kth_bounded": cfg.get("kth_bounded", False)},
        "by_kernel_per_call": by_kernel.get("sq_a"),
        "per_launch": merged,
        "dispatches_averaged": n_disp,
        "kernel_ms_under_profiler": ms_prof,
        "unit_note": "FETCH_SIZE/WRITE_SIZE are KiB; gfx950 FETCH_SIZE counts half of a wide coalesced streaming read "
                     "(MI355X_MICROARCH.md): hbm_bytes = FETCH_SIZE*1024*2 + WRITE_SIZE*1024",
    })
    pairs = cfg["db_rows"] * cfg["queries_per_gpu"]
    ms = ms_prof.get("sq_a") or 0.0
    if "SQ_INSTS_VALU" in merged and ms:
        steps = pairs / 1024.0
        clock = merged.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 / (ms_prof.get("sq_b") or ms) / 1e6  # GHz
        derived = {
            "valu_wave_instructions_per_1024_pairs": merged["SQ_INSTS_VALU"] / steps,
            "salu_per_1024_pairs": merged.get("SQ_INSTS_SALU", 0.0) / steps,
            "lds_per_1024_pairs": merged.get("SQ_INSTS_LDS", 0.0) / steps,
            "lane_ops_per_s": merged["SQ_INSTS_VALU"] * 64.0 / (ms * 1e-3),
            "clock_GHz": clock,
        }
        cycles = merged.get("GRBM_GUI_ACTIVE", 0.0) / 8.0  # per XCD: rocprofv3 sums the 8 XCDs
        if cycles:
            # 1024 SIMDs; SQ_WAVE_CYCLES counts in units of 4 cycles (matches the round-1 derivation, r01_pmc_sq.json)
            derived["cycles_per_valu_instruction_per_simd"] = cycles * 1024.0 / merged["SQ_INSTS_VALU"]
            derived["salu_instructions_per_cycle_per_cu"] = merged.get("SQ_INSTS_SALU", 0.0) / 256.0 / cycles
            if merged.get("SQ_WAVE_CYCLES"):
                derived["resident_waves_per_simd"] = merged["SQ_WAVE_CYCLES"] * 4.0 / (cycles * 1024.0)
            if merged.get("SQ_WAIT_INST_ANY") and merged.get("SQ_WAVE_CYCLES") and merged.get("SQ_ACTIVE_INST_ANY"):
                wc = merged["SQ_WAVE_CYCLES"]
                derived["wave_time_split"] = {
                    "waiting (s_waitcnt/barrier)": merged.get("SQ_WAIT_ANY", 0.0) / wc,
                    "issuing": merged["SQ_ACTIVE_INST_ANY"] / wc,
                }
        record["derived"] = derived
    path = os.path.join(out, "pmc_record.json")
    json.dump(record, open(path, "w"), indent=1)
    print("wrote", path)
    print(json.dumps({k: record[k] for k in ("kernel", "config", "build_id")}))
    print(json.dumps(record.get("derived", {}), indent=1))
    return 0


def cluster_main(a, out, prof):
    """counter totals over every scan kernel of ONE `smafa cluster` run on BASELINE configs[4]'s records (the program after
    `--` is the CLI binary itself)"""
    sys.path.insert(0, ROOT)
    import numpy as np

    from smafa_amd import synth

    fasta = "/tmp/r04_cluster_5M.faa"
    if not os.path.exists(fasta):
        recs = synth.cluster_records(100_000, 50, 60, 1, seed=4, max_subs=4)
        letters = np.array([ord("A") + i for i in range(26)] + [ord("*"), ord("-")], dtype=np.uint8)
        n, L = recs.shape
        ids = np.char.zfill(np.arange(n).astype("U8"), 8).astype("S8")
        rec = np.empty((n, 1 + 8 + 1 + L + 1), dtype=np.uint8)
        rec[:, 0], rec[:, 9], rec[:, 10 + L] = ord(">"), 10, 10
        rec[:, 1:9] = np.frombuffer(ids.tobytes(), dtype=np.uint8).reshape(n, 8)
        rec[:, 10:10 + L] = letters[recs]
        rec.tofile(fasta)
    cli = [os.path.join(ROOT, "smafa_amd", "bin", "smafa"), "cluster", "-i", fasta, "-d", "5", "--alphabet", "aa", "-v"]
    bid = subprocess.run(["python3", "-c", "import smafa_amd; print(smafa_amd.build_id())"], cwd=ROOT, capture_output=True, text=True)
    record = {"command": "smafa cluster -i <5M x 60 aa, seed 4> -d 5 --alphabet aa", "build_id": bid.stdout.strip(),
              "kernel": "all scan kernels of the run", "config": {"mode": "cluster", "records": 5_000_000}}
    merged, ms_prof, bk_all = {}, {}, None
    for name, counters in (("sq_a", SQ_A), ("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"])):
        d = os.path.join(out, "cluster_" + name)
        with open(os.path.join(out, "cluster_" + name + ".log"), "w") as f:
            # (the CLI normally leaves through _exit, which would skip the profiler's own exit handler)
            rc = subprocess.run([prof, "--pmc", *counters, "--kernel-trace", "-d", d, "-o", name, "--output-format", "csv", "--", *cli],
                                stdout=subprocess.DEVNULL, stderr=f, cwd=ROOT, env=dict(os.environ, SMAFA_NO_FAST_EXIT="1")).returncode
        print("cluster", name, "pass rc", rc, flush=True)
        tot, nd, ms, bk = per_call(counter_rows(d), 1)
        merged.update(tot)
        ms_prof[name] = ms
        if name == "sq_a":
            bk_all = bk
            record["dispatches"] = nd
    if "FETCH_SIZE" in merged:
        merged["hbm_bytes"] = merged["FETCH_SIZE"] * 1024.0 * 2.0 + merged.get("WRITE_SIZE", 0.0) * 1024.0
    record.update({"per_launch": merged, "kernel_ms_under_profiler": ms_prof, "by_kernel_per_call": bk_all})
    if merged.get("SQ_INSTS_VALU") and ms_prof.get("sq_a"):
        record["derived"] = {"lane_ops_per_s": merged["SQ_INSTS_VALU"] * 64.0 / (ms_prof["sq_a"] * 1e-3)}
    path = os.path.join(out, "pmc_record.json")
    json.dump(record, open(path, "w"), indent=1)
    print("wrote", path)
    return 0


if __name__ == "__main__":
    sys.exit(main())
