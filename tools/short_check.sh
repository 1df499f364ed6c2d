#!/bin/bash
# Lengths other than 60 columns on a 10M-row store: zone kernel forced (SMAFA_ZONE=2) vs off (run on the GPU box).
cd "$(dirname "$0")/.."
for cfg in "aa 20 3" "aa 20 5" "aa 30 5" "nt 30 3" "aa 100 5" "nt 120 5"; do
  set -- $cfg
  for z in 2 0; do
    SMAFA_ZONE=$z python3 bench.py --alphabet $1 --seq-len $2 --max-div $3 --no-cpu-baseline --no-stream --steps 10 > gpurun_out/sc.json 2> gpurun_out/sc.err
    python3 - "$1" "$2" "$3" "$z" <<'PY'
import json, sys
d = json.load(open("gpurun_out/sc.json"))
print("%s L=%-3s D=%s  SMAFA_ZONE=%s  %8.3f ms/launch  %6.2f M q/s  verified=%s  %s" % (sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], d["roofline"]["kernel_ms_avg"], d["value"] / 1e6, d["verified"], d["roofline"]["kernel"]))
PY
  done
done
