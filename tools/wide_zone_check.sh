#!/bin/bash
# More than 128 columns, 10M-row sorted store (run on the GPU box): scan_wide_kernel with its zone level (automatic) vs off.
cd "$(dirname "$0")/.."
for cfg in "aa 150 5 10000000" "aa 200 5 10000000" "nt 200 3 10000000" "nt 300 5 10000000" "aa 200 5 1000000"; do
  set -- $cfg
  for z in 1 0; do
    SMAFA_ZONE=$z python3 bench.py --alphabet $1 --seq-len $2 --max-div $3 --db-rows $4 --no-cpu-baseline --no-stream --no-related --steps 6 --warmup 2 > gpurun_out/wz.json 2> gpurun_out/wz.err
    python3 - "$1" "$2" "$3" "$4" "$z" <<'PY'
import json, sys
d = json.load(open("gpurun_out/wz.json"))
print("%s L=%-3s D=%-2s rows %-9s SMAFA_ZONE=%s  %8.3f ms/launch  %6.2f M q/s  rows %-8d verified=%s  %s" % (sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5], d["roofline"]["kernel_ms_avg"], d["value"] / 1e6, d["rows_per_step"], d["verified"], d["roofline"]["kernel"]))
PY
  done
done
