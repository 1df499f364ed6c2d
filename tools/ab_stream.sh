#!/bin/bash
# One-query-per-pass figures of the default library and of every smafa_amd/lib_v*/ build (run on the GPU box).
cd "$(dirname "$0")/.."
run() {
  SMAFA_AMD_LIB=$2 python3 bench.py --no-cpu-baseline --no-related --steps 5 "${@:3}" > gpurun_out/abs_$1.json 2> gpurun_out/abs_$1.err
  python3 - "$1" <<'PY'
import json, sys
d = json.load(open("gpurun_out/abs_%s.json" % sys.argv[1])); s = d["stream"]["shipped"]
print("%-8s one query: kernel %.1f us  wall %.1f us  %s  verified=%s" % (sys.argv[1], s["kernel_ms_median"] * 1e3, s["ms_per_query_wall"] * 1e3, s["kernel"], d["verified"]))
PY
}
run base "" "$@"
for lib in smafa_amd/lib_v*/libsmafa_amd.so; do
  [ -f "$lib" ] || continue
  run $(basename $(dirname $lib)) "$PWD/$lib" "$@"
done
