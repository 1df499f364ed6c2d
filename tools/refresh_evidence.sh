#!/bin/bash
# On the GPU box, after ANY change to kernels.hip.h / index.hip.h / engine.hip (the build id changes): collect the counter records of
# every workload bench.py looks up (tools/collect_pmc.py: rocprofv3 --pmc passes of the same bench command, program directly
# after `--`), assemble profiles/r04_pmc.json, take the bench line of record with it in place, and leave everything under
# gpurun_out/evidence/ in the names profiles/ uses (copy them over afterwards).  Three parts, in one call (about six minutes)
# or in several (then copy gpurun_out/evidence/part_<x>.json to profiles/r04_pmc_part_<x>.json in between):
#   gpurun --timeout 1150 -- 'cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && tools/refresh_evidence.sh a && tools/refresh_evidence.sh b && tools/refresh_evidence.sh c'
cd "$(dirname "$0")/.."
E=gpurun_out/evidence
mkdir -p $E
part=${1:-a}
light="--skip-stats --passes sq_a,fetch,write"
collect() {  # tag, collect_pmc flags..., -- bench flags
  local tag=$1; shift
  python3 tools/collect_pmc.py --tag $tag "$@" > $E/collect_$tag.log 2>&1 || { echo "collect $tag failed"; tail -5 $E/collect_$tag.log; return 1; }
  # the raw rocprofv3 output stays on the box (gpurun merges at most 64 MiB back): the record is what travels; ev_aa's CSVs are
  # cut down to the scan kernel's rows at the end of part a
  [ "$tag" = ev_aa ] || rm -rf gpurun_out/$tag/sq_a gpurun_out/$tag/sq_b gpurun_out/$tag/fetch gpurun_out/$tag/write gpurun_out/$tag/stats gpurun_out/$tag/*_full.json
  echo "collected $tag ($(date +%T))"
}
if [ "$part" = a ]; then
  collect ev_aa || exit 1
  collect ev_unf $light --steps 5 -- --prefilter 0 || exit 1
  collect ev_d8 $light -- --max-div 8 || exit 1
  collect ev_d14 $light -- --max-div 14 || exit 1
  collect ev_d24 $light -- --max-div 24 || exit 1
  collect ev_best $light --steps 4 -- --mode besthit || exit 1
  collect ev_k5 $light --steps 4 -- --mode kth --kth-k 5 || exit 1
  collect ev_k50 $light --steps 4 -- --mode kth --kth-k 50 || exit 1
  python3 - <<'PY'
import json
tags = ("ev_aa", "ev_unf", "ev_d8", "ev_d14", "ev_d24", "ev_best", "ev_k5", "ev_k50")
json.dump({"records": [json.load(open("gpurun_out/%s/pmc_record.json" % t)) for t in tags]}, open("gpurun_out/evidence/part_a.json", "w"), indent=1)
PY
elif [ "$part" = b ]; then
  collect ev_rel $light -- --store related || exit 1
  collect ev_1m $light -- --db-rows 1000000 || exit 1
  collect ev_nt $light -- --alphabet nt --queries 100000 --max-div 3 || exit 1
  collect ev_ntn $light -- --alphabet nt --queries 100000 --max-div 3 --n-frac 0.001 || exit 1
  collect ev_50m $light --steps 4 -- --db-rows 50000000 --queries 125000 || exit 1
  python3 - <<'PY'
import json
tags = ("ev_rel", "ev_1m", "ev_nt", "ev_ntn", "ev_50m")
json.dump({"records": [json.load(open("gpurun_out/%s/pmc_record.json" % t)) for t in tags]}, open("gpurun_out/evidence/part_b.json", "w"), indent=1)
PY
else  # part c: the cluster record, the assembled file, the bench line of record
  python3 tools/collect_pmc.py --tag ev_cluster --cluster > $E/collect_ev_cluster.log 2>&1 || { echo "collect cluster failed"; exit 1; }
  echo "collected cluster ($(date +%T))"
  python3 - <<'PY'
import csv, glob, json, os, shutil
E = "gpurun_out/evidence"
recs = []
for part in ("a", "b"):  # the parts travel as profiles/r04_pmc_part_<x>.json (copied there from gpurun_out/evidence/ between calls)
    for path in ("gpurun_out/evidence/part_%s.json" % part, "profiles/r04_pmc_part_%s.json" % part):
        if os.path.exists(path):
            recs += json.load(open(path))["records"]
            break
recs.append(json.load(open("gpurun_out/ev_cluster/pmc_record.json")))
json.dump({"records": recs}, open(E + "/r04_pmc.json", "w"), indent=1)
shutil.copy(E + "/r04_pmc.json", "profiles/r04_pmc.json")  # on the box: the bench line below looks it up
PY
  python3 tools/stream_pmc.py > $E/stream_pmc.log 2>&1 && cp gpurun_out/r04_stream_pmc.json $E/r04_stream_pmc.json && cp $E/r04_stream_pmc.json profiles/r04_stream_pmc.json
  python3 bench.py --steps 20 --warmup 5 --full-record $E/r04_bench_full.json > $E/r04_bench.json 2> $E/bench.err || exit 1
  wc -c $E/r04_bench.json
  python3 - <<'PY'
import json
d = json.load(open("gpurun_out/evidence/r04_bench_full.json"))
r = d["roofline"]
print("headline %.3f ms/step %.3f M q/s frac %s this build %s verified %s run %s s" % (d["ms_per_step"], d["value"] / 1e6, r.get("frac"), r.get("insts_source_is_this_build"), d["verified"], d["run_s"]))
for k in ("unfiltered", "besthit_unbounded", "related"):
    if d.get(k): print(k, d[k].get("kernel_ms"), d[k]["roofline"].get("frac"))
for x in d.get("loose_bounds") or []: print("bound", x["max_divergence"], x["kernel_ms"], x["roofline"].get("frac"))
for k, v in (d.get("configs") or {}).items(): print(k, v.get("kernel_ms"), (v.get("roofline") or {}).get("frac"), v.get("verified"))
for k, v in (d.get("kth") or {}).items():
    if isinstance(v, dict): print(k, v.get("wall_ms"), v.get("kernel_ms"), v["roofline"].get("frac"), v.get("verified"))
print("stream", d["stream"]["roofline"])
PY
fi
if [ "$part" = a ]; then
  python3 - <<'PY'
import csv, glob, json, os, shutil
E = "gpurun_out/evidence"
rec = json.load(open("gpurun_out/ev_aa/pmc_record.json"))
kernel = rec["kernel"].replace("smafa::", "")
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
for stats in glob.glob("gpurun_out/ev_aa/stats/**/*kernel_stats.csv", recursive=True): shutil.copy(stats, E + "/r04_kernel_stats.csv")
only_scan("gpurun_out/ev_aa/stats/**/*kernel_trace.csv", E + "/r04_kernel_trace_scan.csv")
for p in ("sq_a", "sq_b", "fetch", "write"):
    only_scan("gpurun_out/ev_aa/%s/**/*counter_collection.csv" % p, E + "/r04_pmc_%s_scan_rows.csv" % p)
for d in ("sq_a", "sq_b", "fetch", "write", "stats"): shutil.rmtree("gpurun_out/ev_aa/" + d, ignore_errors=True)
PY
fi
if [ "$part" = c ]; then rm -rf gpurun_out/ev_cluster/cluster_* gpurun_out/stream_pmc; fi
echo "part $part done"
