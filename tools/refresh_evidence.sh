#!/bin/bash
# On the GPU box, after ANY change to kernels.hip.h / engine.hip (the build id changes): collect the counter records of
# the four workloads bench.py looks up, assemble profiles/r02_pmc.json, take the bench lines of record with it in place,
# and leave everything under gpurun_out/evidence/ in the names profiles/ uses (copy them over afterwards).
#   gpurun --timeout 1150 -- 'cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && tools/refresh_evidence.sh'
cd "$(dirname "$0")/.."
E=gpurun_out/evidence
mkdir -p $E
python3 tools/collect_pmc.py --tag ev_aa > $E/collect_aa.log 2>&1 || exit 1
python3 tools/collect_pmc.py --tag ev_nt -- --alphabet nt --queries 100000 --max-div 3 > $E/collect_nt.log 2>&1 || exit 1
python3 tools/collect_pmc.py --tag ev_rel -- --store related > $E/collect_rel.log 2>&1 || exit 1
python3 tools/collect_pmc.py --tag ev_50m -- --db-rows 50000000 --queries 125000 > $E/collect_50m.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, json, os, shutil
E = "gpurun_out/evidence"
recs = [json.load(open("gpurun_out/%s/pmc_record.json" % t)) for t in ("ev_aa", "ev_nt", "ev_50m", "ev_rel")]
json.dump({"records": recs}, open(E + "/r02_pmc.json", "w"), indent=1)
shutil.copy(E + "/r02_pmc.json", "profiles/r02_pmc.json")  # on the box: the bench lines below look it up
kernel = recs[0]["kernel"].replace("smafa::", "")
def only_scan(src_glob, dst):
    rows, head = [], None
    for path in glob.glob(src_glob, recursive=True):
        with open(path, newline="") as f:
            r = csv.reader(f); h = next(r)
            head = head or h
            k = h.index("Kernel_Name")
            rows += [x for x in r if kernel in x[k]]
    if head:
        with open(dst, "w", newline="") as f:
            w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC); w.writerow(head); w.writerows(rows)
for stats in glob.glob("gpurun_out/ev_aa/stats/**/*kernel_stats.csv", recursive=True): shutil.copy(stats, E + "/r02_kernel_stats.csv")
only_scan("gpurun_out/ev_aa/stats/**/*kernel_trace.csv", E + "/r02_kernel_trace_scan.csv")
for p in ("sq_a", "sq_b", "fetch", "write"):
    only_scan("gpurun_out/ev_aa/%s/**/*counter_collection.csv" % p, E + "/r02_pmc_%s_scan_rows.csv" % p)
PY
python3 bench.py --steps 20 > $E/r02_bench.json 2> $E/bench.err || exit 1
python3 bench.py --steps 20 --alphabet nt --queries 100000 --max-div 3 --no-related --no-cpu-baseline > $E/r02_bench_nt_10M_100k_d3.json 2>> $E/bench.err || exit 1
python3 bench.py --steps 20 --alphabet nt --queries 100000 --max-div 3 --n-frac 0.001 --no-related --no-cpu-baseline > $E/r02_bench_nt_10M_100k_d3_withN.json 2>> $E/bench.err || exit 1
python3 bench.py --steps 50 --db-rows 1000000 --no-related --no-cpu-baseline > $E/r02_bench_aa_1M_10k_d5.json 2>> $E/bench.err || exit 1
python3 bench.py --steps 5 --warmup 2 --db-rows 50000000 --queries 125000 --no-related --no-cpu-baseline > $E/r02_bench_aa_50M_125k_d5.json 2>> $E/bench.err || exit 1
python3 - <<'PY'
import json
for n in ("r02_bench", "r02_bench_nt_10M_100k_d3", "r02_bench_nt_10M_100k_d3_withN", "r02_bench_aa_1M_10k_d5", "r02_bench_aa_50M_125k_d5"):
    d = json.load(open("gpurun_out/evidence/%s.json" % n)); r = d["roofline"]
    print("%-34s %.3f ms/step  %.3f M q/s  frac %s  this build: %s  verified %s  %s" % (n, d["ms_per_step"], d["value"] / 1e6, r.get("frac"), r.get("insts_source_is_this_build"), d.get("verified"), r["kernel"]))
PY
