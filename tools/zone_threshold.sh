#!/bin/bash
# Where does the zone level stop paying?  Same launch with the zone kernel forced (SMAFA_ZONE=2) and off (SMAFA_ZONE=0),
# over store sizes and bounds (run on the GPU box).
cd "$(dirname "$0")/.."
for cfg in "1000000 5" "1000000 3" "250000 5" "250000 3" "4000000 5" "10000000 6" "10000000 7" "10000000 8"; do
  set -- $cfg
  for z in 2 0; do
    SMAFA_ZONE=$z python3 bench.py --db-rows $1 --max-div $2 --no-cpu-baseline --no-stream --steps 20 > gpurun_out/zt.json 2> gpurun_out/zt.err
    python3 - "$1" "$2" "$z" <<'PY'
import json, sys
d = json.load(open("gpurun_out/zt.json"))
print("rows %9s  D %s  SMAFA_ZONE=%s  %8.3f ms/launch  verified=%s  %s" % (sys.argv[1], sys.argv[2], sys.argv[3], d["roofline"]["kernel_ms_avg"], d["verified"], d["roofline"]["kernel"]))
PY
  done
done
