"""Aggregate gpurun_out/pmcb_MFMA into MFMA-pipe utilisation per kernel class (full-batch launches only: see pmc_bench_summary.py).
SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles of all 1024 SIMDs (= 16 x the number of 16x16x32 MFMAs); GRBM_GUI_ACTIVE
sums the active cycles of the 8 XCDs.  utilisation = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024).
usage: python scripts/pmc_bench_mfma_summary.py [--dir gpurun_out] [--forwards N]"""
import argparse, collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_bench_summary import kernel_class

ap = argparse.ArgumentParser()
ap.add_argument("--dir", default="gpurun_out")
ap.add_argument("--forwards", type=int, default=0)
args = ap.parse_args()
rows = []
for f in glob.glob(f"{args.dir}/pmcb_MFMA/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kernel_class(r["Kernel_Name"]) is not None:
            rows.append(r)
if args.forwards > 0:        # drop (kernel, grid, workgroup) groups that do not occur a whole number of times per full-batch forward
    cnt = collections.Counter((r["Kernel_Name"], r["Grid_Size"], r["Workgroup_Size"]) for r in rows if r["Counter_Name"] == "GRBM_GUI_ACTIVE")
    bad = {k for k, n in cnt.items() if n % args.forwards}
    for k in sorted(bad):
        print(f"pmc_bench_mfma_summary: dropped {cnt[k]} launch(es) of {k[0][:70]} grid {k[1]}", file=sys.stderr)
    rows = [r for r in rows if (r["Kernel_Name"], r["Grid_Size"], r["Workgroup_Size"]) not in bad]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for r in rows:
    cls = kernel_class(r["Kernel_Name"])
    agg[cls][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        cnt[cls] += 1
res = {}
for cls, c in agg.items():
    gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    res[cls] = {"launches_counted": cnt[cls], "mfma_busy_cycles_per_launch": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(1, cnt[cls]),
                "kernel_cycles_per_launch": gui / max(1, cnt[cls]), "mfma_insts_per_launch": c.get("SQ_INSTS_MFMA", 0.0) / max(1, cnt[cls]),
                "mfma_pipe_utilisation": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * 1024.0) if gui else None}
json.dump(res, sys.stdout, indent=1)
