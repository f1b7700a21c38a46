"""Per-kernel summary of scripts/pmc_survey.sh: share of wave cycles waiting / issuing, instruction mix per MFMA, LDS conflicts."""
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for i in (1, 2, 3):
    fs = glob.glob(f"gpurun_out/pmcs_p{i}/**/*counter_collection.csv", recursive=True)
    if not fs: continue
    seen = set()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if i == 1 and r["Counter_Name"] == "SQ_WAVE_CYCLES": calls[k] += 1
print("%-64s %5s %9s %6s %6s %6s %6s | %6s %6s %6s %6s | %6s" % ("kernel", "calls", "wavecyc/c", "wait%", "stall%", "issue%", "mfma%", "valu/M", "lds/M", "vmem/M", "salu/M", "ldsconf"))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"]):
    wc = v["SQ_WAVE_CYCLES"]
    if wc < 1e7: continue
    n = max(1, calls[k]); m = max(1.0, v["SQ_INSTS_MFMA"])
    # SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD; wave cycles are quad-cycles summed over waves: report MFMA busy per kernel cycle via GRBM instead
    print("%-64s %5d %9.0f %6.1f %6.1f %6.1f %6s | %6.2f %6.2f %6.2f %6.2f | %6.3f" % (
        k[:64], n, wc / n, 100 * v["SQ_WAIT_ANY"] / wc, 100 * v["SQ_WAIT_INST_ANY"] / wc, 100 * v["SQ_ACTIVE_INST_ANY"] / wc, "",
        v["SQ_INSTS_VALU"] / m, v["SQ_INSTS_LDS"] / m, v["SQ_INSTS_VMEM"] / m, v["SQ_INSTS_SALU"] / m,
        v["SQ_LDS_BANK_CONFLICT"] / max(1.0, v["SQ_LDS_IDX_ACTIVE"])))
