#!/bin/bash
# Round-4 A/B on one GPU box (run through gpurun from the repo root).
#   TESTS=1            run the GPU test suite first (TEST_ARGS narrows it)
#   VARIANTS="label|ENV=V,ENV2=V|streams ..."   benchmark variants (GSR_LIBRARY=ab/libgsraster_x.so selects another build)
set -o pipefail
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/${1:-r4a}
mkdir -p $OUT
if [ "${TESTS:-0}" = "1" ]; then
  timeout -k 10 1000 python -m pytest tests -m gpu -q ${TEST_ARGS:-} > $OUT/tests.log 2>&1
  echo "tests rc=$?" | tee -a $OUT/tests.log
  grep -E "^FAILED|^ERROR|passed|failed" $OUT/tests.log | tail -15
fi
B="python3 bench.py --steps ${STEPS:-200} --warmup 5 --no-cpu-baseline --no-extras --dense-pairs 0 ${BENCH_ARGS:-}"
for var in ${VARIANTS:-}; do
  label=$(echo $var | cut -d'|' -f1); envs=$(echo $var | cut -d'|' -f2); s=$(echo $var | cut -d'|' -f3)
  ( for kv in $(echo $envs | tr ',' ' '); do [ -n "$kv" ] && export "$kv"; done
    [ -n "$GSR_LIBRARY" ] && export GSR_LIBRARY=$R/$GSR_LIBRARY
    timeout -k 10 300 $B --streams $s > $OUT/bench_${label}_s$s.json 2> $OUT/bench_${label}_s$s.err || echo "bench $label s$s failed: $(tail -3 $OUT/bench_${label}_s$s.err)" )
  python3 - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_${label}_s$s.json").read().strip().splitlines()[-1])
    print("$label streams=$s", d["value"], "views/s", {k: v["avg_ms"] for k, v in d["stages"].items()}, "N", d["config"]["N_pairs"])
except Exception as e:
    print("$label streams=$s: no result", e)
PY
done
