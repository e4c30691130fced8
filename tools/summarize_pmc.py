"""Per-kernel HBM traffic from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes.
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE under-reports wide coalesced
reads by exactly 2x -> doubled here; WRITE_SIZE is taken as reported. Units: KiB."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def load(d, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r.get("Counter_Name") != counter:
                    continue
                k = r.get("Kernel_Name", "?")
                acc[k][0] += float(r.get("Counter_Value", 0))
                acc[k][1] += 1
    return acc


fetch = load(sys.argv[1], "FETCH_SIZE")
write = load(sys.argv[2], "WRITE_SIZE")
out = {}
print(f"{'kernel':80s} {'launches':>8s} {'read_MB/launch(x2 corrected)':>30s} {'write_MB/launch':>16s}")
for k in sorted(set(fetch) | set(write)):
    fr = fetch.get(k, [0, 0]); wr = write.get(k, [0, 0])
    rd = 2.0 * fr[0] * 1024 / max(fr[1], 1)
    wt = wr[0] * 1024 / max(wr[1], 1)
    out[k] = {"launches": fr[1] or wr[1], "read_bytes_per_launch": rd, "write_bytes_per_launch": wt}
    print(f"{k[:80]:80s} {fr[1] or wr[1]:8d} {rd/1e6:30.1f} {wt/1e6:16.1f}")
json.dump(out, open(os.path.join(os.path.dirname(sys.argv[1].rstrip('/')), "pmc_traffic.json"), "w"), indent=1)
