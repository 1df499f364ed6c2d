#!/bin/bash
# On the GPU box: stage times of `makedb --packed` (device and --no-gpu) for 10M x 60 nt records.
cd "$(dirname "$0")/.."
python3 - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
from smafa_amd import synth
synth.write_fasta("/dev/shm/mp_s.fna", synth.subjects(10_000_000, 60, 0, seed=2), 0)
PY
for flags in "" "--no-gpu"; do
  for rep in 1 2; do
    SMAFA_LOG=2 smafa_amd/bin/smafa makedb -i /dev/shm/mp_s.fna -d /dev/shm/mp.packed --packed $flags -v 2>&1 | grep "makedb --packed"
  done
done
rm -f /dev/shm/mp_s.fna /dev/shm/mp.packed
