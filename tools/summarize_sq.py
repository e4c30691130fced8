"""Per-kernel SQ wave-cycle breakdown from one rocprofv3 --pmc pass (SQ_WAVE_CYCLES, SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY, ...):
every counter as a fraction of SQ_WAVE_CYCLES.  usage: python tools/summarize_sq.py <dir>"""
import csv, glob, os, sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
n = defaultdict(int)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = r.get("Kernel_Name", "?")
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_WAVE_CYCLES":
                n[k] += 1
names = sorted({c for v in acc.values() for c in v if c != "SQ_WAVE_CYCLES"})
print(f"{'kernel':60s} {'launches':>8s} " + " ".join(f"{c.replace('SQ_', '')[:16]:>16s}" for c in names))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:12]:
    w = v.get("SQ_WAVE_CYCLES", 0) or 1.0
    print(f"{k[:60]:60s} {n[k]:8d} " + " ".join(f"{v.get(c, 0) / w:16.3f}" for c in names))
