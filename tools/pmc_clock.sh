#!/bin/bash
# effective shader clock of the FFN-out GEMM under ablation variants: GRBM_GUI_ACTIVE / duration
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/clk
cd /tmp
for v in 0 1 13; do
  SMI_GEMM_VAR=$v rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/clk/v$v -o c --output-format csv -- python $R/tools/probe_perf.py gemm > /dev/null 2>&1
done
cd $R
python - <<'PY'
import csv, glob
for v in (0, 1, 13):
    cc = {}
    for f in glob.glob(f"gpurun_out/clk/v{v}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm_tn256" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                cc.setdefault(r["Dispatch_Id"], [r["Kernel_Name"][:48], float(r["Counter_Value"]), None])
    for f in glob.glob(f"gpurun_out/clk/v{v}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            d = r.get("Dispatch_Id")
            if d in cc:
                cc[d][2] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    import collections
    agg = collections.defaultdict(list)
    for name, cyc, ns in cc.values():
        if ns: agg[(name, round(ns / 1e5))].append((cyc, ns))
    for k, lst in sorted(agg.items()):
        cyc = sum(c for c, _ in lst) / len(lst); ns = sum(n for _, n in lst) / len(lst)
        print(f"VAR {v} {k[0]} n={len(lst)} dur={ns/1e3:.0f}us GUI_ACTIVE={cyc:.3e} -> {cyc/ns:.2f} GHz(if per-XCC avg)")
PY
rm -rf gpurun_out/clk
