"""Aggregate gpurun_out/pmcb_MFMA into MFMA-pipe utilisation per kernel class.
SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles of all 1024 SIMDs (= 16 x the number of 16x16x32 MFMAs); GRBM_GUI_ACTIVE
sums the active cycles of the 8 XCDs.  utilisation = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024)."""
import csv, glob, json, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for f in glob.glob("gpurun_out/pmcb_MFMA/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        cls = "igemm" if ("igemm" in k or "conv3x3_c64" in k or "conv3x3_xres" in k) else "bneck_block" if "bneck_block" in k else "bneck_tail3" if "bneck_tail3" in k else "bneck_tail" if "bneck_tail" in k else \
              "conv1" if "stem" in k else "avgpool" if "avgpool" in k else None
        if cls is None: continue
        agg[cls][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE": cnt[cls] += 1
res = {}
for cls, c in agg.items():
    gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    res[cls] = {"launches_counted": cnt[cls], "mfma_busy_cycles_per_launch": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(1, cnt[cls]),
                "kernel_cycles_per_launch": gui / max(1, cnt[cls]), "mfma_insts_per_launch": c.get("SQ_INSTS_MFMA", 0.0) / max(1, cnt[cls]),
                "mfma_pipe_utilisation": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * 1024.0) if gui else None}
json.dump(res, sys.stdout, indent=1)
