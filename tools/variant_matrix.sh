#!/bin/bash
# Build the library with different kernel tuning macros and bench each (run on the GPU box).
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
CONFIGS="${CONFIGS:-1,4 1,6 2,4 2,6 4,4 4,5 4,6 8,4}"
for cfg in $CONFIGS; do
  g=${cfg%,*}; w=${cfg#*,}
  make -C smafa_amd/csrc -B -j8 EXTRA_HIPFLAGS="-DSMAFA_GROUP=$g -DSMAFA_MIN_WAVES=$w ${EXTRA:-}" > /dev/null 2>&1 || { echo "build failed $cfg"; continue; }
  timeout -k 10 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline ${BENCH_ARGS:-} > gpurun_out/var_${g}_${w}.json 2> gpurun_out/var_${g}_${w}.err
  python - "$g" "$w" <<'PY'
import json,sys
g,w=sys.argv[1],sys.argv[2]
try:
    d=json.load(open(f"gpurun_out/var_{g}_{w}.json"))
    print(f"group={g} minwaves={w}  q/s={d['value']:.0f}  kernel_ms={d['roofline']['kernel_ms_avg']:.3f} verified={d['verified']}")
except Exception as e:
    print(f"group={g} minwaves={w} FAILED {e}")
PY
done
make -C smafa_amd/csrc -B -j8 > /dev/null 2>&1
