#!/usr/bin/env python3
"""MFMA-busy fraction and effective clock per kernel from a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE`
pass plus the kernel_stats.csv of a separate --kernel-trace --stats pass (for the average duration).
usage: pmc_mfma.py COUNTER_CSV KERNEL_STATS_CSV
GRBM_GUI_ACTIVE is summed over the 8 XCDs (cycles = value / 8); SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs."""
import collections, csv, sys
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for r in csv.DictReader(open(sys.argv[1])):
    a = acc[r["Kernel_Name"]][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
dur = {r["Name"]: float(r["AverageNs"]) for r in csv.DictReader(open(sys.argv[2]))}
print("rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (own pass); eff_clock = GRBM_GUI_ACTIVE/8/duration; "
      "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES/(cycles*1024 SIMDs)")
for k, c in sorted(acc.items(), key=lambda kv: -dur.get(kv[0], 0) * kv[1]["GRBM_GUI_ACTIVE"][1]):
    if "GRBM_GUI_ACTIVE" not in c or k not in dur:
        continue
    cyc = c["GRBM_GUI_ACTIVE"][0] / c["GRBM_GUI_ACTIVE"][1] / 8
    busy = c["SQ_VALU_MFMA_BUSY_CYCLES"][0] / c["SQ_VALU_MFMA_BUSY_CYCLES"][1] / (cyc * 1024)
    print(f"{k[:80]:80s} eff_clock_GHz={cyc / dur[k]:.2f} mfma_busy={busy:.3f}")
