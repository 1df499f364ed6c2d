#!/usr/bin/env python3
"""Per-dispatch view of ONE k-th-mode call (rocprofv3 --kernel-trace around tools/kth_one.py): kernel, grid, duration.
    python3 tools/kth_dispatches.py K [SMAFA_KTH_SAMPLE]"""
import csv, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
k = sys.argv[1] if len(sys.argv) > 1 else "5"
env = dict(os.environ, TMPDIR="/tmp")
far = "far" in sys.argv[2:]
mixed = "mixed" in sys.argv[2:]
if len(sys.argv) > 2 and sys.argv[2] not in ("far", "mixed"):
    env["SMAFA_KTH_SAMPLE"] = sys.argv[2]
out = os.path.join(ROOT, "gpurun_out", "kth_trace_%s_%s%s" % (k, env.get("SMAFA_KTH_SAMPLE", "d"), "_far" if far else "_mixed" if mixed else ""))
subprocess.run(["/opt/rocm/bin/rocprofv3", "--kernel-trace", "-d", out, "-o", "t", "--output-format", "csv", "--",
                "python3", os.path.join(ROOT, "tools", "kth_one.py"), k] + (["far"] if far else ["mixed"] if mixed else []), cwd=ROOT, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
rows = []
for p in glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(p, newline="")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the LAST call: everything after the last pack_rows_kernel (the query upload of the timed call)
packs = [i for i, r in enumerate(rows) if "pack_rows_kernel" in r["Kernel_Name"]]
last_pack = packs[-3] if len(packs) >= 3 else packs[-1]  # a call packs its queries, the open ones after the first ladder step, and the planner's sample
t0 = int(rows[last_pack]["Start_Timestamp"])
tot = 0.0
for r in rows[last_pack:]:
    ms = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    tot += ms
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("smafa::", "")
    if "rocprim" in name or "rocclr" in name or ms < 0.02:
        continue
    print("%9.3f ms at %8.3f  grid %9s  %s" % (ms, (int(r["Start_Timestamp"]) - t0) / 1e6, r["Grid_Size_X"], name[:90]))
print("sum %.2f ms" % tot)
