#!/bin/bash
# Collects the round-2 evidence on a GPU box (run through gpurun from the repo root):
#   1. rocprofv3 --kernel-trace --stats of the default bench command   -> profiles/r02_kernel_stats.csv
#   2. separate PMC passes FETCH_SIZE / WRITE_SIZE (MI355X_MICROARCH.md "HBM": never in one pass, never with traces
#      other than --kernel-trace)                                       -> profiles/r02_pmc_traffic.json
set -e
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r02
mkdir -p $OUT
BENCH="python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras"
# kernel statistics over a run whose launches are almost all inside the timed region (300 steps), so that the average
# duration of K7 is comparable with the bench line's live figure (r02_bench_under_rocprof.json is this run's line)
STATS="python3 bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- $STATS > $OUT/bench_stats.json 2> $OUT/bench_stats.err
# the same with ONE stream: kernel durations without other views' kernels beside them
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats1 -o s -- $STATS --streams 1 > $OUT/bench_stats1.json 2> $OUT/bench_stats1.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o f -- $BENCH > /dev/null 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o w -- $BENCH > /dev/null 2> $OUT/write.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $OUT/sq -o q -- $BENCH > /dev/null 2> $OUT/sq.err
python3 - <<'PY'
import csv, json, collections, os
out = os.environ.get("OUT", "gpurun_out/r02")
def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}
fetch = per_kernel(f"{out}/fetch/f_counter_collection.csv", "FETCH_SIZE")
write = per_kernel(f"{out}/write/w_counter_collection.csv", "WRITE_SIZE")
sq = {c: per_kernel(f"{out}/sq/q_counter_collection.csv", c) for c in
      ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_WAVES")}
res = {}
for k in sorted(set(fetch) | set(write)):
    if "gsr::" not in k:
        continue
    f_kb, w_kb = fetch.get(k, 0.0), write.get(k, 0.0)
    res[k] = {"FETCH_SIZE_KB_per_launch": f_kb, "WRITE_SIZE_KB_per_launch": w_kb,
              # gfx950: FETCH_SIZE tallies 128-B requests at 64 B => x2 for wide coalesced streams (MI355X_MICROARCH.md, HBM);
              # gathers of 48-byte records are NOT calibrated, so both figures are kept
              "hbm_bytes_raw": (f_kb + w_kb) * 1024, "hbm_bytes_fetch_x2": (2 * f_kb + w_kb) * 1024,
              # wave-level instruction counts per launch; SQ_ACTIVE_INST_VALU and SQ_WAVE_CYCLES count quad-cycles
              **{c: sq[c].get(k, 0.0) for c in sq}}
json.dump({"command": "python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras", "per_kernel": res}, open(f"{out}/pmc_traffic.json", "w"), indent=1)
print(json.dumps({k: v["hbm_bytes_fetch_x2"] / 1e6 for k, v in res.items()}, indent=1))
PY
cp $OUT/stats/s_kernel_stats.csv $R/gpurun_out/r02_kernel_stats.csv
cp $OUT/stats1/s_kernel_stats.csv $R/gpurun_out/r02_kernel_stats_1stream.csv
cp $OUT/pmc_traffic.json $R/gpurun_out/r02_pmc_traffic.json
cp $OUT/bench_stats.json $R/gpurun_out/r02_bench_under_rocprof.json
