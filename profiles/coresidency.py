#!/usr/bin/env python3
"""What shares the chip with what in the pipelined regime -- from a rocprofv3 --kernel-trace CSV of
`bench.py` (four streams).  Every dispatch is an interval [start, end]; kernels are put in classes

    compositor   k_render_fwd / k_render_bwd            (VALU-bound)
    hbm          k_pre_bwd / k_pre_color                 (HBM-bound)
    chain        everything else of the forward's front end (scans, sorts, emission, schedule: latency-bound)

and the timeline of the steady state (the middle 60 % of the trace's gsr:: dispatches) is swept once: for each class the
time during which at least one kernel of it is in flight, the pairwise overlaps, the time with nothing in flight, and the
mean number of kernels in flight.  "In flight" is dispatch-to-completion as rocprofv3 stamps it -- a kernel whose
workgroups are waiting for wave slots counts as in flight, which is exactly the queueing this analysis is after.

    python3 profiles/coresidency.py <s_kernel_trace.csv> [label]
"""
import csv
import sys


def klass(name):
    if "k_render_fwd" in name or "k_render_bwd" in name:
        return "compositor"
    if "k_pre_bwd" in name or "k_pre_color" in name or "k_preprocess_bwd" in name:
        return "hbm"
    return "chain"


def main():
    path = sys.argv[1]
    label = sys.argv[2] if len(sys.argv) > 2 else path
    ev = []
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if "gsr::" not in n:
            continue
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), klass(n), n.split("(")[0]))
    ev.sort()
    lo, hi = ev[len(ev) // 5][0], ev[len(ev) * 4 // 5][0]
    ev = [e for e in ev if e[0] >= lo and e[1] <= hi]
    pts = []
    for s, e, k, _ in ev:
        pts.append((s, 1, k))
        pts.append((e, -1, k))
    pts.sort()
    live = {"compositor": 0, "hbm": 0, "chain": 0}
    t_prev = pts[0][0]
    acc = {}
    inflight_time = 0
    for t, d, k in pts:
        dt = t - t_prev
        if dt > 0:
            key = tuple(sorted(c for c, v in live.items() if v > 0))
            acc[key] = acc.get(key, 0) + dt
            inflight_time += dt * sum(live.values())
        live[k] += d
        t_prev = t
    total = sum(acc.values())
    def share(pred):
        return sum(v for k, v in acc.items() if pred(k)) / total
    dur = {}
    for s, e, k, n in ev:
        d = dur.setdefault(n, [0, 0])
        d[0] += e - s
        d[1] += 1
    print(f"== {label}: steady-state window {total / 1e6:.2f} ms, {len(ev)} dispatches, mean kernels in flight {inflight_time / total:.2f}")
    print(f"  compositor in flight            {share(lambda k: 'compositor' in k):.3f}")
    print(f"  hbm-bound kernel in flight      {share(lambda k: 'hbm' in k):.3f}")
    print(f"  chain kernel in flight          {share(lambda k: 'chain' in k):.3f}")
    print(f"  compositor AND hbm              {share(lambda k: 'compositor' in k and 'hbm' in k):.3f}")
    print(f"  compositor AND chain            {share(lambda k: 'compositor' in k and 'chain' in k):.3f}")
    print(f"  hbm with NO compositor          {share(lambda k: 'hbm' in k and 'compositor' not in k):.3f}")
    print(f"  chain only                      {share(lambda k: k == ('chain',)):.3f}")
    print(f"  nothing in flight               {share(lambda k: k == ()):.3f}")
    comp = share(lambda k: 'compositor' in k)
    print(f"  of the compositors' in-flight time, an hbm-bound kernel is in flight beside them {share(lambda k: 'compositor' in k and 'hbm' in k) / max(comp, 1e-9):.3f}")
    print("  mean dispatch-to-completion per kernel (us):")
    for n, (d, c) in sorted(dur.items(), key=lambda kv: -kv[1][0])[:12]:
        print(f"    {n[:70]:70s} {d / c / 1e3:8.1f} x {c}")


if __name__ == "__main__":
    main()
