cd "$(dirname "$0")/.."
for qb in 0 313 455 625 1250 2500 5000 10000; do
python3 bench.py --query-block $qb --no-cpu-baseline --no-stream --steps 30 > gpurun_out/qb.json 2> gpurun_out/qb.err
python3 - $qb <<'PY'
import json, sys
d = json.load(open("gpurun_out/qb.json"))
print("query block %6s  %.3f ms/launch  %.3f M q/s  blocks=%d" % (sys.argv[1], d["roofline"]["kernel_ms_avg"], d["value"]/1e6, d["roofline"]["plan"]["query_blocks"]))
PY
done
