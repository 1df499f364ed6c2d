#!/bin/bash
# On the GPU box: the tightening-mode legs of the bench line (best hit without a bound: mixed / far / novel members; the K branch)
# under several values of one environment switch.
#   tools/ab_modes.sh SMAFA_ZONE_DIRECT "1 2" [bench flags]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
var=$1; vals=$2; shift 2
for v in $vals; do
  env $var=$v python3 bench.py --no-cpu-baseline --no-configs --steps 10 --warmup 3 --full-record gpurun_out/abmodes_full.json "$@" > gpurun_out/abmodes.json 2> gpurun_out/abmodes.err
  python3 - "$var=$v" <<'PY'
import json, sys
d = json.load(open("gpurun_out/abmodes.json"))
legs = d.get("legs") or {}
print(sys.argv[1], "headline %.3f ms" % d["ms_per_step"], " ".join("%s=%s" % (k, legs[k]) for k in sorted(legs) if k.startswith(("besthit", "kth", "host"))))
PY
done
