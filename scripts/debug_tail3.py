"""Diagnostic: where does the chained layer3 tail differ from the two igemm launches?  python scripts/debug_tail3.py N BP"""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from implementation_phd_lab_vision_amd import ops, _lib

_lib.build_library()
n, bp = int(sys.argv[1]), int(sys.argv[2])
g = torch.Generator().manual_seed(1)
d = "cuda:0"
def rb(shape, scale=1.0):
    return (torch.randn(shape, generator=g) * scale).to(torch.bfloat16)
m = n * 196
y2 = rb((n, 14, 14, 256)).to(d); idn = rb((n, 14, 14, 1024)).to(d)
w3 = rb((1024, 256), (2.0 / 256) ** 0.5).to(d); w1 = rb((256, 1024), (2.0 / 1024) ** 0.5).to(d)
b3 = (torch.randn(1024, generator=g) * 0.1).to(d); b1 = (torch.randn(256, generator=g) * 0.1).to(d)
if bp:
    os.environ["R50_TAIL3_BP"] = str(bp)
out, y1 = ops.bneck_tail_bf16(y2, w3, b3, idn, w1, b1)
out_u = ops.conv2d_bf16(y2, w3.view(1024, 1, 1, 256), b3, relu=True, residual=idn)
y1_u = ops.conv2d_bf16(out_u, w1.view(256, 1, 1, 1024), b1, relu=True)
torch.cuda.synchronize()
for name, a, b in (("out", out, out_u), ("y1n", y1, y1_u)):
    a, b = a.view(m, -1), b.view(m, -1)
    bad = (a != b)
    print(name, "mismatching elements:", int(bad.sum()), "of", bad.numel())
    if bad.any():
        rows = bad.any(1).nonzero().flatten()
        cols = bad.any(0).nonzero().flatten()
        print("  rows:", rows[:40].tolist(), "... n =", rows.numel(), " tiles:", sorted(set((rows // max(bp, 1)).tolist()))[:40] if bp else "")
        print("  cols:", cols[:40].tolist(), "... n =", cols.numel())
        r0 = int(rows[0]); c0 = int(bad[r0].nonzero()[0])
        print("  first:", r0, c0, float(a[r0, c0]), float(b[r0, c0]), "max abs diff", float((a.float() - b.float()).abs().max()))
