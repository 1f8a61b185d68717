"""Timeline of one launch chain of the headline regime from a rocprofv3 --kernel-trace CSV: every kernel between two
consecutive k_pack_views launches (one chain = the views of one step group), its start relative to the chain's first kernel,
its duration and the gap to the kernel in front of it on the timeline (negative: overlap with a side-stream kernel).

    python profiles/chain_timeline.py <kernel_trace.csv> [chain index from the end, default 3]
"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""))
             for r in rows), key=lambda t: t[0])
packs = [i for i, k in enumerate(ks) if "k_pack_views" in k[2]]
i0, i1 = packs[-back - 1], packs[-back]
chain = ks[i0:i1]
t0 = chain[0][0]
busy = 0
last_end = t0
print(f"chain of {len(chain)} kernels, span {(ks[i1][0] - t0) / 1e3:.1f} us")
gaps = 0.0
for s, e, n in chain:
    gap = (s - last_end) / 1e3
    print(f"{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:8.1f} us  gap {gap:7.1f}  {n[:70]}")
    if gap > 0:
        gaps += gap
    last_end = max(last_end, e)
print(f"sum of positive gaps {gaps:.1f} us; to the next chain's first kernel {(ks[i1][0] - last_end) / 1e3:.1f} us")
