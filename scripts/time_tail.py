"""Time the fused bottleneck tail kernels (r50_op_bneck_tail) at the bench shapes.  usage: python scripts/time_tail.py [batch] [hw: only the shape at that resolution]"""
import sys, torch
sys.path.insert(0, '.')
from implementation_phd_lab_vision_amd import ops, _lib
_lib.load_library()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
d = torch.device('cuda:0'); g = torch.Generator().manual_seed(0)
SHAPES = ((56, 64, 64, False), (56, 64, 64, True), (56, 64, 128, False), (28, 128, 128, False), (14, 256, 256, False))
if len(sys.argv) > 2:
    SHAPES = tuple(sh for sh in SHAPES if sh[0] == int(sys.argv[2]))
for hw, cmid, c1, ds in SHAPES:
    y2 = torch.randn((B, hw, hw, cmid), generator=g).to(torch.bfloat16).to(d)
    idn = torch.randn((B, hw, hw, cmid if ds else 4 * cmid), generator=g).to(torch.bfloat16).to(d)
    wd = (torch.randn((4 * cmid, cmid), generator=g) * 0.12).to(torch.bfloat16).to(d) if ds else None
    bd = torch.randn(4 * cmid, generator=g).to(d) if ds else None
    w3 = (torch.randn((4 * cmid, cmid), generator=g) * 0.17).to(torch.bfloat16).to(d)
    w1 = (torch.randn((c1, 4 * cmid), generator=g) * 0.09).to(torch.bfloat16).to(d)
    b3 = torch.randn(4 * cmid, generator=g).to(d); b1 = torch.randn(c1, generator=g).to(d)
    for _ in range(3): ops.bneck_tail_bf16(y2, w3, b3, idn, w1, b1, wd=wd, bd=bd)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.bneck_tail_bf16(y2, w3, b3, idn, w1, b1, wd=wd, bd=bd)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    m = B * hw * hw
    byts = m * 2.0 * (cmid + (cmid if ds else 4 * cmid) + 4 * cmid + c1)
    print(f"tail {hw}x{hw} cmid={cmid} c1={c1} ds={int(ds)}: {us:.1f} us  {byts / us / 1e6:.2f} TB/s algorithmic", flush=True)
