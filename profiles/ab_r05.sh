#!/bin/bash
# Round-5 A/B on one GPU box (run through gpurun from the repo root).
#   TESTS=1            run GPU tests first (TEST_ARGS narrows them)
#   VARIANTS="label|ENV=V,ENV2=V|streams|extra bench args (':' for spaces) ..."   GSR_LIBRARY=abx/libgsr_x.so selects a build
set -o pipefail
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/${1:-r5a}
mkdir -p $OUT
if [ "${TESTS:-0}" = "1" ]; then
  timeout -k 10 ${TEST_TIMEOUT:-900} python -m pytest ${TEST_ARGS:-tests} -m gpu -q > $OUT/tests.log 2>&1
  echo "tests rc=$?" | tee -a $OUT/tests.log
  grep -E "^FAILED|^ERROR|passed|failed" $OUT/tests.log | tail -15
fi
[ -n "${PRE_CMD:-}" ] && bash -c "$PRE_CMD" > $OUT/pre.log 2>&1 && cat $OUT/pre.log
B="python3 bench.py --steps ${STEPS:-200} --warmup 5 --no-cpu-baseline --no-extras --dense-pairs 0 ${BENCH_ARGS:-}"
for var in ${VARIANTS:-}; do
  label=$(echo $var | cut -d'|' -f1); envs=$(echo $var | cut -d'|' -f2); s=$(echo $var | cut -d'|' -f3); extra=$(echo $var | cut -d'|' -f4 | tr ':' ' ')
  ( for kv in $(echo $envs | tr ',' ' '); do [ -n "$kv" ] && export "$kv"; done
    [ -n "$GSR_LIBRARY" ] && export GSR_LIBRARY=$R/$GSR_LIBRARY
    [ -n "$GSR_COMP_CUMASK" ] && export GSR_COMP_CUMASK=$(echo $GSR_COMP_CUMASK | tr '+' ',')
    timeout -k 10 300 $B --streams $s $extra > $OUT/bench_${label}_s$s.json 2> $OUT/bench_${label}_s$s.err || echo "bench $label s$s failed: $(tail -3 $OUT/bench_${label}_s$s.err)" )
  python3 - <<PY | tee -a $OUT/lines.txt
import json
try:
    d = json.loads(open("$OUT/bench_${label}_s$s.json").read().strip().splitlines()[-1])
    print("$label streams=$s $extra:", d["value"], "views/s", {k: v["avg_ms"] for k, v in d["stages"].items()}, "N", d["config"]["N_pairs"], "K7 in region", d["roofline"]["avg_launch_ms"])
except Exception as e:
    print("$label streams=$s: no result", e)
PY
done
[ -n "${POST_CMD:-}" ] && bash -c "$POST_CMD" > $OUT/post.log 2>&1; [ -f $OUT/post.log ] && tail -40 $OUT/post.log
true
