"""Per-kernel MFMA utilisation from a rocprofv3 --pmc pass
(SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE) joined with the
kernel trace of the same run.

  mfma_busy   = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * #SIMDs)     (the gfx94x MfmaUtil formula; the
                guide notes ROCm 7.2 has no gfx950 derived-counter section, so it is spelled out here)
  TFLOP/s     = SQ_INSTS_VALU_MFMA_MOPS_F16 * 512 flop / kernel duration   (hardware-counted fp16 MFMA flops)
  eff. clock  = GRBM_GUI_ACTIVE / kernel duration                          (DVFS: the power-limited clock)

rocprofv3 reports ONE GRBM_GUI_ACTIVE value per dispatch that is the SUM over the 8 XCCs of an MI355X (a bare
ratio to the wall time gives ~15 "GHz"); the script divides by the XCC count (or by the number of rows if a
future rocprofv3 reports one row per XCC).  Cross-check printed by the r02b run: busy% x eff_GHz / 2.4 GHz equals
the hardware-counted fraction of peak (62.6 % x 1.82 / 2.4 = 47.5 % vs 48.0 % counted) for the FFN-inner GEMM."""
import csv
import glob
import os
import sys
from collections import defaultdict

CUS, SIMDS, XCCS = 256, 1024, 8
agg = defaultdict(lambda: defaultdict(float))   # per kernel name, per-launch AVERAGES merged over the pass directories
for d in sys.argv[1:]:
    vals = defaultdict(lambda: defaultdict(float))
    rows = defaultdict(lambda: defaultdict(int))
    names = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = r["Dispatch_Id"]
                names[k] = r.get("Kernel_Name", "?")
                vals[k][r["Counter_Name"]] += float(r["Counter_Value"])
                rows[k][r["Counter_Name"]] += 1
    dur = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    per = defaultdict(lambda: defaultdict(float))
    for k, v in vals.items():
        if k not in dur or "GRBM_GUI_ACTIVE" not in v:
            continue
        a = per[names[k]]
        a["launches"] += 1
        a["ns"] += dur[k]
        nrow = rows[k]["GRBM_GUI_ACTIVE"]
        gui = v["GRBM_GUI_ACTIVE"] / (nrow if nrow > 1 else XCCS)
        a["gui"] += gui
        for c, key in (("SQ_VALU_MFMA_BUSY_CYCLES", "mfma_busy"), ("SQ_INSTS_VALU_MFMA_MOPS_F16", "mops")):
            if c in v:
                a[key] += v[c]
                a[key + "_gui"] += gui
                a[key + "_ns"] += dur[k]
    for n, a in per.items():
        g = agg[n]
        if not g["launches"]:
            g["launches"], g["ns"], g["gui"] = a["launches"], a["ns"], a["gui"]
        for key in ("mfma_busy", "mops"):
            if a[key + "_ns"]:
                g[key], g[key + "_gui"], g[key + "_ns"] = a[key], a[key + "_gui"], a[key + "_ns"]
print(f"{'kernel':72s} {'launches':>8s} {'avg_us':>9s} {'eff_GHz':>8s} {'mfma_busy%':>10s} {'hw TFLOP/s':>10s} {'%of 2.5PF':>9s}")
for n, a in sorted(agg.items(), key=lambda kv: -kv[1]["ns"]):
    if a["mfma_busy"] == 0 and a["mops"] == 0:
        continue
    ns = a["ns"]
    ghz = a["gui"] / ns
    busy = 100.0 * a["mfma_busy"] / (a["mfma_busy_gui"] * SIMDS) if a["mfma_busy_gui"] else 0.0
    tf = a["mops"] * 512 / a["mops_ns"] / 1e3 if a["mops_ns"] else 0.0
    print(f"{n[:72]:72s} {int(a['launches']):8d} {ns / a['launches'] / 1e3:9.1f} {ghz:8.2f} {busy:10.1f} {tf:10.0f} {100 * tf / 2500:9.1f}")
