#!/bin/bash
# On the GPU box: the bench line of the default library under two values of one environment switch, batched launch only.
#   tools/ab_env.sh SMAFA_ZONE_DIRECT "0 1" [bench flags]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
var=$1; vals=$2; shift 2
for v in $vals; do
  env $var=$v python3 bench.py --no-cpu-baseline --no-stream --steps 30 "$@" > gpurun_out/abenv.json 2> gpurun_out/abenv.err
  python3 - "$var=$v" <<'PY'
import json, sys
d = json.load(open("gpurun_out/abenv.json"))
print("%-24s %.3f ms/launch  %.3f M q/s  verified=%s  %s" % (sys.argv[1], d["roofline"]["kernel_ms_avg"], d["value"] / 1e6, d["verified"], d["roofline"]["kernel"]))
PY
done
