"""Condense a rocprofv3 --kernel-trace --stats output dir into a small table."""
import csv
import glob
import os
import sys

d = sys.argv[1]
files = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
if not files:
    print("no kernel_stats.csv under", d)
    for f in glob.glob(os.path.join(d, "**", "*"), recursive=True)[:40]:
        print(" ", f)
    sys.exit(0)
rows = []
for f in files:
    with open(f) as fh:
        rows += list(csv.DictReader(fh))
print(f"# rocprofv3 --kernel-trace --stats summary ({len(files)} file(s))")
print(f"{'kernel':90s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
for r in sorted(rows, key=lambda r: -float(r.get("TotalDurationNs", 0))):
    name = r.get("Name", "?")[:90]
    print(f"{name:90s} {r.get('Calls','?'):>7s} {float(r.get('TotalDurationNs',0))/1e6:10.3f} "
          f"{float(r.get('AverageNs',0))/1e3:10.2f} {float(r.get('MinNs',0))/1e3:10.2f} {float(r.get('MaxNs',0))/1e3:10.2f} "
          f"{float(r.get('Percentage',0)):6.2f}")
