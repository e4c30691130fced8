"""Kernel-overlap summary of a `rocprofv3 --kernel-trace` run (CSV): per kernel name the call count and mean duration,
and for the whole trace the fraction of the busy time during which 1, 2, 3+ kernels were in flight (any queue).
usage: python tools/trace_overlap.py <dir with *_kernel_trace.csv> [t0_frac t1_frac]   (fractions of the trace to keep)"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    f0 = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
    f1 = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for fn in files:
        with open(fn) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
    if not rows:
        print("no kernel trace rows under", d)
        return
    rows.sort()
    t_lo, t_hi = rows[0][0], max(r[1] for r in rows)
    lo, hi = t_lo + (t_hi - t_lo) * f0, t_lo + (t_hi - t_lo) * f1
    rows = [r for r in rows if r[0] >= lo and r[1] <= hi]
    per = defaultdict(lambda: [0, 0])
    queues = defaultdict(int)
    for s, e, name, q in rows:
        key = name.split("(")[0][:70]
        per[key][0] += 1
        per[key][1] += e - s
        queues[q] += 1
    print(f"{len(rows)} dispatches on queues {dict(queues)}")
    for k, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"  {k:70s} calls {c:6d}  mean {t / c / 1e3:8.2f} us  total {t / 1e6:8.2f} ms")
    ev = []
    for s, e, _, _ in rows:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    depth, last = 0, ev[0][0]
    hist = defaultdict(int)
    for t, dlt in ev:
        hist[min(depth, 3)] += t - last
        last = t
        depth += dlt
    span = ev[-1][0] - ev[0][0]
    print(f"span {span / 1e6:.2f} ms: idle {hist[0] / span:.3f}, 1 kernel in flight {hist[1] / span:.3f}, "
          f"2 in flight {hist[2] / span:.3f}, 3+ {hist[3] / span:.3f}")


if __name__ == "__main__":
    main()
