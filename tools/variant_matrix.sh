#!/bin/bash
# A/B of scan-kernel variants on the GPU box.  Run-time knobs (no rebuild): SMAFA_TILES=1|2 (wave tiles per
# wave), SMAFA_FILTER=0|1, SMAFA_NT_PLANES=3.  Build-time knobs go through EXTRA (e.g. -DSMAFA_AND_PAIR=0).
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
if [ -n "${EXTRA:-}" ]; then make -C smafa_amd/csrc -B -j8 EXTRA_HIPFLAGS="$EXTRA" > /dev/null 2>&1 || { echo "build failed"; exit 1; }; fi
for alpha in ${ALPHABETS:-aa nt}; do
  for tiles in ${TILES:-1 2}; do
    for rep in 1 2; do
      if [ "$alpha" = nt ]; then args="--alphabet nt --max-div 3 --queries 100000"; else args=""; fi
      SMAFA_TILES=$tiles timeout -k 10 150 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-stream $args ${BENCH_ARGS:-} > gpurun_out/var.json 2> gpurun_out/var.err
      python - "$alpha" "$tiles" <<'PY'
import json,sys
try:
    d=json.load(open("gpurun_out/var.json"))
    print(f"{sys.argv[1]} tiles/wave={sys.argv[2]}  q/s={d['value']:.0f}  kernel_ms={d['roofline']['kernel_ms_avg']:.3f} verified={d['verified']} B/subject={d['roofline']['stored_bytes_per_subject']}")
except Exception as e:
    print(f"{sys.argv[1]} tiles={sys.argv[2]} FAILED {e}")
PY
    done
  done
done
if [ -n "${EXTRA:-}" ]; then make -C smafa_amd/csrc -B -j8 > /dev/null 2>&1; fi
