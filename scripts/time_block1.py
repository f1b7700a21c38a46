"""Layer1 bottleneck body at batch B: the one-launch form (r50_op_bneck_block1) against the launches it replaces
(resident-weights 3x3 + fused tail).  usage: python scripts/time_block1.py [batch] [rounds]"""
import sys, torch
sys.path.insert(0, '.')
from implementation_phd_lab_vision_amd import ops, _lib
_lib.load_library()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
d = torch.device('cuda:0'); g = torch.Generator().manual_seed(0)
rb = lambda shape, scale=1.0: (torch.randn(shape, generator=g) * scale).to(torch.bfloat16).to(d)
t1 = rb((B, 56, 56, 64)).clamp_(min=0); idn = rb((B, 56, 56, 256)).clamp_(min=0)
w2 = rb((64, 3, 3, 64), (2.0 / 576) ** 0.5); w3 = rb((256, 64), (2.0 / 64) ** 0.5)
b2 = (torch.randn(64, generator=g) * 0.1).to(d); b3 = (torch.randn(256, generator=g) * 0.1).to(d)
W1 = {c1: rb((c1, 256), (2.0 / 256) ** 0.5) for c1 in (64, 128)}
B1 = {c1: (torch.randn(c1, generator=g) * 0.1).to(d) for c1 in (64, 128)}


def t_us(fn, iters=20):
    for _ in range(3): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / iters


def unfused(c1):
    t2 = ops.conv2d_bf16(t1, w2, b2, stride=1, pad=1, relu=True, tile=ops.TILE_C64)
    return ops.bneck_tail_bf16(t2, w3, b3, idn, W1[c1], B1[c1])


cases = []
for c1 in (64, 128):
    cases.append((f"conv2 + tail, c1={c1} (2 launches)", lambda c1=c1: unfused(c1), c1))
    cases.append((f"block1, c1={c1}", lambda c1=c1: ops.bneck_block1_bf16(t1, w2, b2, w3, b3, idn, W1[c1], B1[c1]), c1))
res = {k: [] for k, _, _ in cases}
for r in range(ROUNDS):
    for k, fn, _ in cases:
        res[k].append(t_us(fn))
m = B * 3136
for k, _, c1 in cases:
    t = sorted(res[k]); med = t[len(t) // 2]
    byts = m * 2.0 * (64 + 256 + 256 + c1)
    print(f"{k:36s} median {med:7.1f} us  min {t[0]:7.1f} us   {byts / med / 1e6:5.2f} TB/s of the fused form's bytes", flush=True)
