"""A/B of the chained layer3 tail's kernel variants in one process (R50_TAIL3_VAR is read per call by r50_op_bneck_tail):
0 = bneck_tail3_kernel, 1 = bneck_tail3p_kernel (two-group pipeline, 112 rows), 2 = the same with 98-row slots.
Bits of both outputs are compared with variant 0's.  usage: python scripts/time_tail3_variants.py [batch] [rounds] [variants, e.g. 0,1,2]"""
import os, sys, torch
sys.path.insert(0, '.')
from implementation_phd_lab_vision_amd import ops, _lib
_lib.load_library()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
VARS = [int(v) for v in sys.argv[3].split(',')] if len(sys.argv) > 3 else [0, 1, 2]
d = torch.device('cuda:0'); g = torch.Generator().manual_seed(0)
hw, cmid, c1 = 14, 256, 256
y2 = torch.randn((B, hw, hw, cmid), generator=g).to(torch.bfloat16).to(d)
idn = torch.randn((B, hw, hw, 4 * cmid), generator=g).to(torch.bfloat16).to(d)
w3 = (torch.randn((4 * cmid, cmid), generator=g) * 0.17).to(torch.bfloat16).to(d)
w1 = (torch.randn((c1, 4 * cmid), generator=g) * 0.09).to(torch.bfloat16).to(d)
b3 = torch.randn(4 * cmid, generator=g).to(d); b1 = torch.randn(c1, generator=g).to(d)
m = B * hw * hw
flops = 2.0 * m * (cmid * 4 * cmid + 4 * cmid * c1)
byts = m * 2.0 * (cmid + 4 * cmid + 4 * cmid + c1)
ref = None
for v in VARS:
    os.environ["R50_TAIL3_VAR"] = str(v)
    out, y1n = ops.bneck_tail_bf16(y2, w3, b3, idn, w1, b1)
    torch.cuda.synchronize()
    if ref is None:
        ref = (out.clone(), y1n.clone())
    else:
        print(f"variant {v}: out equal {torch.equal(out, ref[0])} ({int((out != ref[0]).sum())} differ)  y1n equal {torch.equal(y1n, ref[1])} ({int((y1n != ref[1]).sum())} differ)", flush=True)
for rnd in range(ROUNDS):
    for v in VARS:
        os.environ["R50_TAIL3_VAR"] = str(v)
        for _ in range(3): ops.bneck_tail_bf16(y2, w3, b3, idn, w1, b1)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.bneck_tail_bf16(y2, w3, b3, idn, w1, b1)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 50        # (includes the per-call weight packing launch of the debug hook, ~3 us)
        print(f"round {rnd} variant {v}: {us:7.1f} us  {flops / us / 1e6:7.1f} TF/s  {byts / us / 1e6:5.2f} TB/s", flush=True)
