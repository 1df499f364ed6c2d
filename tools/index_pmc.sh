#!/bin/bash
# On the GPU box: rocprofv3 records of the index probe kernel on the metric's launch (tools/index_one.py: 10M x 60 aa, 10 000 queries,
# bound 5, 20 launches) — kernel trace (true duration), then counters in their own passes (no trace domain beside --kernel-trace).
#   tools/index_pmc.sh [aa|nt] > gpurun_out/r04_index_pmc.txt
cd "$(dirname "$0")/.."
shape=${1:-aa}
O=gpurun_out/index_pmc
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python3 tools/index_one.py $shape > $O/trace.log 2>&1 || { tail -5 $O/trace.log; exit 1; }
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS --kernel-trace -d $O/sq -o p --output-format csv -- python3 tools/index_one.py $shape > $O/sq.log 2>&1 || { tail -5 $O/sq.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch -o p --output-format csv -- python3 tools/index_one.py $shape > $O/fetch.log 2>&1 || { tail -5 $O/fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write -o p --output-format csv -- python3 tools/index_one.py $shape > $O/write.log 2>&1 || { tail -5 $O/write.log; exit 1; }
tail -1 $O/trace.log
python3 - "$O" <<'PY'
import csv, glob, sys
O = sys.argv[1]
for f in glob.glob(O + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "index" in r["Name"] or "zone_kernel<" in r["Name"]:
            print("%-70s calls %4s avg %10.1f us  min %10.1f  max %10.1f" % (r["Name"].split("(")[0][-70:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
acc = {}
for p in ("sq", "fetch", "write"):
    for f in glob.glob(O + "/%s/**/*counter_collection.csv" % p, recursive=True):
        for r in csv.DictReader(open(f)):
            if "index_probe" in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
print("index_probe_kernel, per launch (average of %d):" % max(len(v) for v in acc.values()))
for k, v in sorted(acc.items()):
    print("  %-18s %14.1f" % (k, sum(v) / len(v)))
if "FETCH_SIZE" in acc:
    kb = sum(acc["FETCH_SIZE"]) / len(acc["FETCH_SIZE"])
    print("  fetched: %.2f MB per launch as counted (KiB units), %.2f MB with the guide's gfx950 x2 rule for wide streaming reads "
          "(an upper bound for these narrow gathers)" % (kb * 1024 / 1e6, kb * 2 * 1024 / 1e6))
PY
rm -rf $O/trace $O/sq $O/fetch $O/write
