#!/bin/bash
# Collects the round-6 evidence on a GPU box (run through gpurun from the repo root):
#   1. rocprofv3 --kernel-trace --stats of the benchmark: the headline regime (views five at a time through one launch
#      chain, --headline-only) and the per-view loop on one stream                -> profiles/r06_kernel_stats*.csv
#   2. separate PMC passes FETCH_SIZE / WRITE_SIZE / SQ counters (MI355X_MICROARCH.md "HBM": never in one pass, never
#      with traces other than --kernel-trace)                                    -> profiles/r06_pmc_traffic.json
#   3. bench lines: default, one stream only, config 2 and config 5 scenes       -> profiles/r06_bench*.json
#   4. the batch of views through one launch chain (kernel statistics, batch against loop) -> profiles/r06_*batch*
# Two gpurun calls (a call is limited to 20 minutes): PART=A (profiler passes + the main bench line), PART=B (the rest).
set -o pipefail
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r06
mkdir -p $OUT
PMCB="python3 $R/bench.py --steps 10 --warmup 2 --regions 1 --no-cpu-baseline --no-extras --headline-only"
STATS="python3 $R/bench.py --steps 100 --warmup 10 --regions 3 --no-cpu-baseline --no-extras --headline-only"
if [ "${PART:-A}" = "A" ]; then
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- $STATS > $OUT/bench_stats.json 2> $OUT/bench_stats.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats1 -o s -- python3 $R/bench.py --steps 100 --warmup 10 --regions 3 --no-cpu-baseline --no-extras --headline streams --streams 1 --batch 0 > $OUT/bench_stats1.json 2> $OUT/bench_stats1.err || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o f -- $PMCB > /dev/null 2> $OUT/fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o w -- $PMCB > /dev/null 2> $OUT/write.err || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $OUT/sq -o q -- $PMCB > /dev/null 2> $OUT/sq.err || exit 1
cd $R
OUT=$OUT python3 - <<'PY'
import csv, json, collections, os
out = os.environ["OUT"]
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
              **{c: sq[c].get(k, 0.0) for c in sq}}
# the digest of the kernel sources these counters belong to (bench.py quotes them only while it still matches)
import hashlib
def digest():
    h = hashlib.sha256()
    csrc = os.path.join(os.getcwd(), "3d-gaussian-splat-attack_amd", "csrc")
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h")))
    files.append(os.path.join(os.getcwd(), "include", "gsraster.h"))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()
json.dump({"command": "python3 bench.py --steps 10 --warmup 2 --regions 1 --no-cpu-baseline --no-extras --headline-only (every launch of the compositors covers the headline regime's 5 views: --steps 10, 20 and 100 all group views by 5)",
           "views_per_launch": 5,
           "sources_sha256": digest(), "per_kernel": res},
          open(f"{out}/pmc_traffic.json", "w"), indent=1)
print(json.dumps({k: round(v["hbm_bytes_fetch_x2"] / 1e6, 1) for k, v in res.items()}, indent=1))
PY
cp $OUT/stats/s_kernel_stats.csv $R/gpurun_out/r06_kernel_stats.csv
cp $OUT/stats1/s_kernel_stats.csv $R/gpurun_out/r06_kernel_stats_1stream.csv
cp $OUT/pmc_traffic.json $R/gpurun_out/r06_pmc_traffic.json
cp $OUT/pmc_traffic.json $R/profiles/r06_pmc_traffic.json     # (on the box: the bench lines below quote it; same sources => same digest)
cp $OUT/bench_stats.json $R/gpurun_out/r06_bench_under_rocprof.json
cp $OUT/bench_stats1.json $R/gpurun_out/r06_bench_1stream_under_rocprof.json
# bench lines without the profiler
python3 bench.py --steps 100 --warmup 10 > $R/gpurun_out/r06_bench.json 2> $OUT/bench.err || exit 1
# what shares the chip with what: under the profiler's kernel trace, and without a profiler (the library's own stage events)
python3 profiles/coresidency.py $OUT/stats/s_kernel_trace.csv "round 6, four streams, under rocprofv3 --kernel-trace" > $R/gpurun_out/r06_coresidency.txt
python3 profiles/coresidency.py $OUT/stats1/s_kernel_trace.csv "round 6, one stream, under rocprofv3 --kernel-trace" >> $R/gpurun_out/r06_coresidency.txt
python3 tests/diag_coresidency.py 4 64 2>/dev/null | grep -v amdgpu.ids >> $R/gpurun_out/r06_coresidency.txt
python3 tests/diag_coresidency.py 1 32 2>/dev/null | grep -v amdgpu.ids >> $R/gpurun_out/r06_coresidency.txt
fi
if [ "${PART:-A}" = "B" ]; then
python3 bench.py --steps 100 --warmup 10 --headline streams --streams 1 --batch 0 --no-cpu-baseline --no-extras > $R/gpurun_out/r06_bench_1stream.json 2> $OUT/bench1.err || exit 1
python3 bench.py --steps 100 --warmup 10 --scene hydrant-full --no-cpu-baseline --no-extras > $R/gpurun_out/r06_bench_hydrantfull.json 2> $OUT/bench_c2.err || exit 1
python3 bench.py --steps 50 --warmup 5 --scene airport-4K --no-cpu-baseline --no-extras > $R/gpurun_out/r06_bench_airport4k.json 2> $OUT/bench_c5.err || exit 1
python3 bench.py --steps 50 --warmup 5 --objects --no-cpu-baseline --no-extras > $R/gpurun_out/r06_bench_objects.json 2> $OUT/bench_obj.err || exit 1
# kernel statistics of config 2 (S-hydrant-full @800x800) and config 5 (S-airport-4K), one stream
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c2 -o s -- python3 $R/bench.py --steps 100 --warmup 10 --regions 2 --streams 1 --scene hydrant-full --no-cpu-baseline --no-extras > /dev/null 2> $OUT/stats_c2.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c5 -o s -- python3 $R/bench.py --steps 50 --warmup 5 --regions 2 --streams 1 --scene airport-4K --no-cpu-baseline --no-extras > /dev/null 2> $OUT/stats_c5.err || exit 1
cd $R
cp $OUT/stats_c2/s_kernel_stats.csv $R/gpurun_out/r06_kernel_stats_hydrantfull_1stream.csv
cp $OUT/stats_c5/s_kernel_stats.csv $R/gpurun_out/r06_kernel_stats_airport4k_1stream.csv
python3 tests/diag_dropin.py 30 2>/dev/null | grep -v amdgpu.ids > $R/gpurun_out/r06_dropin_breakdown.txt
# the batch of views through one launch chain: kernel statistics of 8-view batches (tests/diag_batch.py), and the bench line in
# the driver's shape (20-step regions)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_batch -o s -- python3 $R/tests/diag_batch.py nyc-1M 8 10 batched > $OUT/batch_under_rocprof.txt 2> $OUT/stats_batch.err || exit 1
cd $R
cp $OUT/stats_batch/s_kernel_stats.csv $R/gpurun_out/r06_kernel_stats_batch8.csv
python3 tests/diag_batch.py nyc-1M 8 10 2>/dev/null | grep -v amdgpu.ids > $R/gpurun_out/r06_batch_vs_loop.txt
python3 tests/diag_batch.py nyc-1M 4 10 batched 2>/dev/null | grep -v amdgpu.ids >> $R/gpurun_out/r06_batch_vs_loop.txt
python3 tests/diag_batch.py nyc-1M 16 6 batched 2>/dev/null | grep -v amdgpu.ids >> $R/gpurun_out/r06_batch_vs_loop.txt
python3 tests/diag_batch.py airport-4K 4 6 2>/dev/null | grep -v amdgpu.ids >> $R/gpurun_out/r06_batch_vs_loop.txt
python3 tests/diag_batch.py hydrant-full 8 10 2>/dev/null | grep -v amdgpu.ids >> $R/gpurun_out/r06_batch_vs_loop.txt
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $R/gpurun_out/r06_bench_driver_shape.json 2> $OUT/bench_ds.err || exit 1
fi
echo collected part ${PART:-A}
