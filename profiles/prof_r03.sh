#!/bin/bash
# rocprofv3 kernel statistics of the benchmark on ONE stream (kernel durations with nothing beside them).
set -o pipefail
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/${1:-r3p}
mkdir -p $OUT
STATS="python3 $R/bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extras --dense-pairs 0 --streams 1 ${BENCH_ARGS:-}"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats1 -o s -- $STATS > $OUT/bench_stats1.json 2> $OUT/bench_stats1.err || { tail -5 $OUT/bench_stats1.err; exit 1; }
cd $R
cp $OUT/stats1/s_kernel_stats.csv $OUT/kernel_stats_1stream.csv
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/kernel_stats_1stream.csv")))
tot = 0.0
for r in rows:
    name = r["Name"].split("(")[0][:70]
    calls, avg = int(r["Calls"]), float(r["AverageNs"]) / 1e3
    if calls >= 50:
        per_view = avg * calls / 105.0
        tot += per_view
        print(f"{name:72s} calls {calls:5d} avg {avg:8.2f} us  per view {per_view:8.2f} us")
print("sum per view (us)", round(tot, 1))
PY
