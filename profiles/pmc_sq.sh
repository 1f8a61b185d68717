#!/bin/bash
# Extra SQ counter passes for the two compositing kernels (diagnostic; the judged summaries come from collect_r01.sh).
# usage: OUT=$PWD/gpurun_out/sq ./profiles/pmc_sq.sh    -> $OUT/sq_summary.json
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${OUT:-$ROOT/gpurun_out/sq}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --streams 1"
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INSTS_VALU_TRANS_F32" \
           "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p -- $BENCH > /dev/null 2> $OUT/p$i.err || { tail -5 $OUT/p$i.err; }
done
python3 - $OUT <<'PY'
import csv, json, sys, glob, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/p*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "render" in k or "k_pre_" in k or "preprocess" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
json.dump(res, open(f"{out}/sq_summary.json", "w"), indent=1)
for k, d in res.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} {v:16.0f}")
PY
