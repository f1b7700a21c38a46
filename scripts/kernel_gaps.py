"""From a rocprofv3 --kernel-trace CSV of bench.py: per step, time inside kernels vs time between them.
usage: python scripts/kernel_gaps.py <..._kernel_trace.csv> [n_kernels_per_step]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda t: t[0])
# the timed region: take the last 60 % of the trace (steady state), find the per-step period by the stem kernel
stem = [i for i, k in enumerate(ks) if "stem_fused" in k[2]]
stem = stem[len(stem) // 3:]
tot_busy = tot_gap = 0; n = 0; gaps = []
for a, b in zip(stem[:-1], stem[1:]):
    seg = ks[a:b]
    busy = sum(e - s for s, e, _ in seg)
    wall = ks[b][0] - ks[a][0]
    tot_busy += busy; tot_gap += wall - busy; n += 1
    gaps += [seg[i + 1][0] - seg[i][1] for i in range(len(seg) - 1)]
print(f"steps analysed: {n}; kernels per step: {len(ks[stem[0]:stem[1]])}")
print(f"per step: wall {1e-3 * (tot_busy + tot_gap) / n:.1f} us, inside kernels {1e-3 * tot_busy / n:.1f} us, between kernels {1e-3 * tot_gap / n:.1f} us "
      f"({100 * tot_gap / (tot_busy + tot_gap):.1f} %)")
gaps.sort()
print(f"gap between consecutive kernels: median {gaps[len(gaps) // 2] * 1e-3:.2f} us, mean {sum(gaps) / len(gaps) * 1e-3:.2f} us, p90 {gaps[int(len(gaps) * 0.9)] * 1e-3:.2f} us")
