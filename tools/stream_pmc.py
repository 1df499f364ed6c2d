#!/usr/bin/env python3
"""HBM traffic of the one-query-per-pass streaming form from the hardware counters (north_star: "evidenced by rocprof HBM
GB/s"): rocprofv3 --pmc FETCH_SIZE (its own pass, kernel trace only) around tools/stream_probe.py on the 50M x 60 aa store,
then per dispatch of the streaming kernel: FETCH_SIZE x 1024 x 2 (KiB; gfx950 counts half of a wide coalesced read —
MI355X_MICROARCH.md) over the dispatch's own duration.   python3 tools/stream_pmc.py  -> gpurun_out/r04_stream_pmc.json"""
import csv, glob, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "gpurun_out", "stream_pmc")
os.makedirs(out, exist_ok=True)
os.environ.setdefault("TMPDIR", "/tmp")
cmd = ["/opt/rocm/bin/rocprofv3", "--pmc", "FETCH_SIZE", "--kernel-trace", "-d", out, "-o", "fetch", "--output-format", "csv", "--",
       "python3", os.path.join(ROOT, "tools", "stream_probe.py"), "--db-rows", "50000000", "--passes", "60"]
with open(os.path.join(out, "run.log"), "w") as f:
    rc = subprocess.run(cmd, stdout=f, stderr=subprocess.STDOUT, cwd=ROOT).returncode
rows = []
for path in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    rows += list(csv.DictReader(open(path, newline="")))
sys.path.insert(0, ROOT)
import smafa_amd  # (loads the library for its build id; opens no device)
res = {"command": "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3 tools/stream_probe.py --db-rows 50000000 --passes 60", "rc": rc,
       "build_id": smafa_amd.build_id()}
for kernel in ("scan_lazy_kernel<5, 5, 2, 4, false, false>", "scan_zone_few_kernel<5, 5, 2>"):
    mine = [r for r in rows if kernel in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
    # one-query passes only: the probe also runs the same 200 queries as ONE launch (200 query blocks, 200 x the bytes)
    one_pass_grid = min((int(r["Grid_Size"]) for r in mine), default=0)
    mine = [r for r in mine if int(r["Grid_Size"]) == one_pass_grid]
    if not mine:
        continue
    fetch = [float(r["Counter_Value"]) * 1024.0 * 2.0 for r in mine]
    ns = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in mine]
    n = len(mine)
    med = sorted(range(n), key=lambda i: ns[i])[n // 2]
    res[kernel] = {"dispatches": n, "hbm_bytes_per_pass_avg": sum(fetch) / n, "kernel_us_median": ns[med] / 1e3,
                   "GBs_from_counters_median_dispatch": fetch[med] / ns[med], "frac_of_8TBs": fetch[med] / ns[med] / 8000.0,
                   "filter_plane_bytes": 400001024, "fetched_over_plane": sum(fetch) / n / 400001024.0}
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "r04_stream_pmc.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
