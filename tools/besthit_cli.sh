#!/bin/bash
# On the GPU box: `smafa query` in the reference's default mode (best hit, no --max-divergence), 100 000 queries against the
# packed 10M x 60 aa store, with the ladder of bounded steps (default) and without (SMAFA_TWO_PHASE=0).
cd "$(dirname "$0")/.."
python3 - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
from smafa_amd import synth
import smafa_amd
subj = synth.subjects(10_000_000, 60, 1, seed=1)
synth.write_fasta("/dev/shm/bh_s.faa", subj, 1)
q, _, _ = synth.queries(subj, 100_000, 1, seed=3, max_subs=10)
synth.write_fasta("/dev/shm/bh_q.faa", q, 1)
smafa_amd.makedb_packed("/dev/shm/bh_s.faa", "/dev/shm/bh.packed", 1)
PY
for tp in 1 0; do
  for rep in 1 2; do
    t0=$(date +%s.%N)
    SMAFA_TWO_PHASE=$tp smafa_amd/bin/smafa query -d /dev/shm/bh.packed -q /dev/shm/bh_q.faa -v > /dev/shm/bh_$tp.out 2> /dev/shm/bh_$tp.err
    t1=$(date +%s.%N)
    echo "SMAFA_TWO_PHASE=$tp  wall $(python3 -c "print(round($t1 - $t0, 3))") s"; grep -E "near-hit|scans" /dev/shm/bh_$tp.err | tail -4
  done
done
cmp /dev/shm/bh_0.out /dev/shm/bh_1.out && echo "outputs identical ($(wc -l < /dev/shm/bh_1.out) lines)"
rm -f /dev/shm/bh_*
