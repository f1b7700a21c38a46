"""HBM traffic of one bench step from the rocprofv3 PMC passes of scripts/pmc_bench.sh (gpurun_out/pmcb_{FETCH_SIZE,WRITE_SIZE}).

    python scripts/pmc_bench_summary.py [--dir gpurun_out] [--forwards N] [--layers gpurun_out/pmcb_layers.json] [--table out.txt]

Per kernel class: launches counted, FETCH_SIZE (raw and x2) and WRITE_SIZE per launch.  gfx950 correction
(/opt/skills/guides/MI355X_MICROARCH.md, HBM): FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads -> doubled; WRITE_SIZE is
exact for 16-B/lane stores.  Both counters are in KiB.

ONLY FULL-BATCH LAUNCHES COUNT.  bench.py's `checked` block runs one batch-2 forward outside the timed region; averaged in, it pulled
every class's bytes per launch down by 1/8 (round 2: 156 instead of 178 MB for the igemm class).  Two guards: scripts/pmc_bench.sh runs
bench.py with --no-check, and this script drops every (kernel, grid, workgroup) group whose launch count is not a multiple of
--forwards (the number of full-batch forward passes of the profiled run = warmup + 2 x steps: timed region + the event-bracket pass):
a batch-2 launch has another grid than the batch-256 launch of the same kernel and appears once.  Dropped groups go to stderr.
With --layers (bench.py --dump-layers: the launches of one forward in launch order with their algorithmic bytes) a per-launch table
is written: measured read x2 / written against algorithmic bytes, one row per launch of a step.
"""
from __future__ import annotations

import argparse
import collections
import csv
import glob
import json
import sys


def kernel_class(k: str):
    if "igemm" in k or "gemm8p" in k or "conv3x3_c64" in k or "conv3x3_xres" in k or "conv3x3_s2" in k:
        return "igemm"
    if "bneck_catchain" in k:
        return "bneck_catchain"
    if "bneck_block" in k:
        return "bneck_block"
    if "bneck_tail3" in k:
        return "bneck_tail3"
    if "bneck_tail" in k:
        return "bneck_tail"
    if "stem_conv" in k or "stem_fused" in k:
        return "conv1"
    if "stem_pack" in k:
        return "stem_pack"
    if "maxpool" in k:
        return "maxpool"
    if "avgpool" in k:
        return "avgpool"
    return None


def read_rows(files, counter):
    """Rows of our kernels for one counter, in dispatch order."""
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] != counter or kernel_class(r["Kernel_Name"]) is None:
                    continue
                rows.append({"id": int(r["Dispatch_Id"]), "kernel": r["Kernel_Name"], "grid": r["Grid_Size"], "wg": r["Workgroup_Size"],
                             "bytes": float(r["Counter_Value"]) * 1024.0})
    rows.sort(key=lambda r: r["id"])
    return rows


def full_batch_only(rows, forwards: int, log=sys.stderr):
    """Keep the launches that occur once (or k times) per full-batch forward: (kernel, grid, workgroup) groups whose count is a
    multiple of `forwards`.  forwards <= 0: keep everything (old behaviour)."""
    if forwards <= 0:
        return rows
    cnt = collections.Counter((r["kernel"], r["grid"], r["wg"]) for r in rows)
    bad = {k for k, n in cnt.items() if n % forwards}
    for k in sorted(bad):
        print(f"pmc_bench_summary: dropped {cnt[k]} launch(es) of {k[0][:70]} grid {k[1]} (not a multiple of {forwards} forwards: "
              f"not a full-batch launch)", file=log)
    return [r for r in rows if (r["kernel"], r["grid"], r["wg"]) not in bad]


def per_class(rows):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in rows:
        a = agg[kernel_class(r["kernel"])]
        a[0] += 1
        a[1] += r["bytes"]
    return {c: {"launches": n, "bytes_per_launch_raw": b / n} for c, (n, b) in agg.items()}


def per_launch(rows, forwards: int):
    """Average bytes per launch SLOT of a forward (launch order), or None if the rows do not split into `forwards` equal sequences."""
    if forwards <= 0 or len(rows) % forwards:
        return None
    L = len(rows) // forwards
    slots = []
    for i in range(L):
        names = {rows[f * L + i]["kernel"] for f in range(forwards)}
        if len(names) != 1:
            return None
        slots.append({"kernel": names.pop(), "bytes": sum(rows[f * L + i]["bytes"] for f in range(forwards)) / forwards})
    return slots


def summarize(fetch_files, write_files, forwards: int = 0, layers=None, log=sys.stderr):
    fr = full_batch_only(read_rows(fetch_files, "FETCH_SIZE"), forwards, log)
    wr = full_batch_only(read_rows(write_files, "WRITE_SIZE"), forwards, log)
    fc, wc = per_class(fr), per_class(wr)
    res = {}
    for cls in sorted(set(fc) | set(wc)):
        f = fc.get(cls, {}).get("bytes_per_launch_raw", 0.0)
        w = wc.get(cls, {}).get("bytes_per_launch_raw", 0.0)
        res[cls] = {"launches_counted": fc.get(cls, {}).get("launches", 0), "fetch_bytes_per_launch_raw": f,
                    "fetch_bytes_per_launch_x2": 2 * f, "write_bytes_per_launch": w, "hbm_bytes_per_launch": 2 * f + w}
    table = None
    fs, ws = per_launch(fr, forwards), per_launch(wr, forwards)
    if fs is not None and ws is not None and len(fs) == len(ws):
        table = []
        for i, (a, b) in enumerate(zip(fs, ws)):
            row = {"slot": i, "kernel": a["kernel"], "read_bytes_x2": 2 * a["bytes"], "written_bytes": b["bytes"]}
            if layers is not None and len(layers) == len(fs):
                row.update({"layer": layers[i]["name"], "algorithmic_bytes": layers[i]["bytes_per_launch"],
                            "ratio": (2 * a["bytes"] + b["bytes"]) / layers[i]["bytes_per_launch"] if layers[i]["bytes_per_launch"] else None})
            table.append(row)
        res["_step"] = {"launches": len(table), "hbm_bytes": sum(r["read_bytes_x2"] + r["written_bytes"] for r in table)}
    res["_meta"] = {"forwards": forwards, "full_batch_only": forwards > 0,
                    "note": "FETCH_SIZE x2 (gfx950) + WRITE_SIZE, KiB -> bytes; batch-2 check launches excluded"}
    return res, table


def format_table(table) -> str:
    lines = [f"{'#':>2s} {'layer':18s} {'read x2 MB':>11s} {'written MB':>11s} {'measured MB':>12s} {'algorithmic MB':>15s} {'ratio':>6s}  kernel"]
    for r in table:
        meas = (r["read_bytes_x2"] + r["written_bytes"]) / 1e6
        alg = r.get("algorithmic_bytes")
        lines.append(f"{r['slot']:2d} {r.get('layer', ''):18s} {r['read_bytes_x2'] / 1e6:11.1f} {r['written_bytes'] / 1e6:11.1f} {meas:12.1f} "
                     f"{(alg / 1e6 if alg else float('nan')):15.1f} {(r.get('ratio') or float('nan')):6.2f}  {r['kernel'][:60]}")
    return "\n".join(lines) + "\n"


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--dir", default="gpurun_out")
    ap.add_argument("--forwards", type=int, default=0, help="full-batch forward passes in the profiled run (bench.py: warmup + 2 x steps)")
    ap.add_argument("--layers", default=None, help="JSON written by bench.py --dump-layers")
    ap.add_argument("--table", default=None, help="write the per-launch table here")
    args = ap.parse_args()
    ff = glob.glob(f"{args.dir}/pmcb_FETCH_SIZE/**/*counter_collection.csv", recursive=True)
    wf = glob.glob(f"{args.dir}/pmcb_WRITE_SIZE/**/*counter_collection.csv", recursive=True)
    layers = json.load(open(args.layers)) if args.layers else None
    res, table = summarize(ff, wf, args.forwards, layers)
    if table is not None and args.table:
        with open(args.table, "w") as fh:
            fh.write(format_table(table))
    json.dump(res, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
