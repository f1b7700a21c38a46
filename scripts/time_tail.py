"""Time the fused layer1 tail kernel (r50_op_bneck_tail) at the bench shape.  usage: python scripts/time_tail.py [batch] [c1]"""
import sys, torch
sys.path.insert(0, '.')
from implementation_phd_lab_vision_amd import ops, _lib
_lib.load_library()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
d = torch.device('cuda:0'); g = torch.Generator().manual_seed(0)
for c1 in ([int(sys.argv[2])] if len(sys.argv) > 2 else [64, 128]):
    y2 = torch.randn((B, 56, 56, 64), generator=g).to(torch.bfloat16).to(d)
    idn = torch.randn((B, 56, 56, 256), generator=g).to(torch.bfloat16).to(d)
    w3 = (torch.randn((256, 64), generator=g) * 0.17).to(torch.bfloat16).to(d)
    w1 = (torch.randn((c1, 256), generator=g) * 0.09).to(torch.bfloat16).to(d)
    b3 = torch.randn(256, generator=g).to(d); b1 = torch.randn(c1, generator=g).to(d)
    for _ in range(3): ops.bneck_tail_bf16(y2, w3, b3, idn, w1, b1)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.bneck_tail_bf16(y2, w3, b3, idn, w1, b1)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    m = B * 56 * 56
    byts = m * 2.0 * (64 + 256 + 256 + c1)
    print(f"tail c1={c1}: {us:.1f} us  {byts / us / 1e6:.2f} TB/s algorithmic", flush=True)
