#!/bin/bash
# Bounds at which level 1 cannot prune (short sequences, loose bounds): zone kernel forced vs off, 10M-row store.
cd "$(dirname "$0")/.."
for cfg in "aa 20 5" "aa 60 8" "aa 60 10" "aa 60 14" "nt 60 8" "nt 60 12" "aa 30 8"; do
  set -- $cfg
  for z in 2 0; do
    SMAFA_ZONE=$z python3 bench.py --alphabet $1 --seq-len $2 --max-div $3 --no-cpu-baseline --no-stream --steps 6 --warmup 2 > gpurun_out/sc.json 2> gpurun_out/sc.err
    python3 - "$1" "$2" "$3" "$z" <<'PY'
import json, sys
d = json.load(open("gpurun_out/sc.json"))
print("%s L=%-3s D=%-2s  SMAFA_ZONE=%s  %8.3f ms/launch  %6.2f M q/s  rows %-8d verified=%s  %s" % (sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], d["roofline"]["kernel_ms_avg"], d["value"] / 1e6, d["rows_per_step"], d["verified"], d["roofline"]["kernel"]))
PY
  done
done
