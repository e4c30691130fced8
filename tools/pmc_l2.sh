#!/bin/bash
# L2 (TCC) hit / miss / request counters of the GEMM probe, per kernel (development aid).
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
cd /tmp
for C in "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum"; do
  tag=$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --kernel-trace -d $OUT/l2_$tag -o l2 --output-format csv -- python $OLDPWD/tools/probe_perf.py gemm > $OUT/l2_$tag.log 2>&1
done
cd $OLDPWD
python - <<'PY'
import csv, glob, os
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int)
for f in glob.glob("gpurun_out/l2_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"][:60], r.get("Grid_Size", ""))
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(acc.items()):
    if "gemm" not in k[0]: continue
    h, m = v.get("TCC_HIT_sum", 0), v.get("TCC_MISS_sum", 0)
    print(k, {n: f"{x:.3g}" for n, x in v.items()}, "hit_rate=%.3f" % (h / max(h + m, 1)))
PY
rm -rf gpurun_out/l2_TCC*
