#!/bin/bash
# On the GPU box: the bench line of the default library and of every smafa_amd/lib_v*/ build, batched launch only.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() {  # name, lib ("" = default)
  SMAFA_AMD_LIB=$2 python3 bench.py --no-cpu-baseline --no-stream --steps 30 "${@:3}" > gpurun_out/ab_$1.json 2> gpurun_out/ab_$1.err
  python3 - "$1" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_%s.json" % sys.argv[1]))
print("%-8s %.3f ms/launch  %.3f M q/s  verified=%s  %s" % (sys.argv[1], d["roofline"]["kernel_ms_avg"], d["value"] / 1e6, d["verified"], d["roofline"]["kernel"]))
PY
}
run base "" "$@"
for lib in smafa_amd/lib_v*/libsmafa_amd.so; do
  [ -f "$lib" ] || continue
  v=$(basename $(dirname $lib))
  run $v "$PWD/$lib" "$@"
done
