"""Layer2 bottleneck body at batch B: the one-launch form (r50_op_bneck_block2) against the launches it replaces
(input-resident 3x3 + fused tail2 / + conv3 igemm).  usage: python scripts/time_block2.py [batch] [rounds]"""
import sys, torch
sys.path.insert(0, '.')
from implementation_phd_lab_vision_amd import ops, _lib
_lib.load_library()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
d = torch.device('cuda:0'); g = torch.Generator().manual_seed(0)
rb = lambda shape, scale=1.0: (torch.randn(shape, generator=g) * scale).to(torch.bfloat16).to(d)
t1 = rb((B, 28, 28, 128)).clamp_(min=0); idn = rb((B, 28, 28, 512)).clamp_(min=0)
w2 = rb((128, 3, 3, 128), (2.0 / 1152) ** 0.5); w3 = rb((512, 128), (2.0 / 128) ** 0.5); w1 = rb((128, 512), (2.0 / 512) ** 0.5)
b2 = (torch.randn(128, generator=g) * 0.1).to(d); b3 = (torch.randn(512, generator=g) * 0.1).to(d); b1 = (torch.randn(128, generator=g) * 0.1).to(d)
w3c = w3.view(512, 1, 1, 128)


def t_us(fn, iters=30):
    for _ in range(3): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / iters


def unfused_chain():
    t2 = ops.conv2d_bf16(t1, w2, b2, stride=1, pad=1, relu=True, tile=ops.TILE_XRES)
    return ops.bneck_tail_bf16(t2, w3, b3, idn, w1, b1)


def unfused_last():
    t2 = ops.conv2d_bf16(t1, w2, b2, stride=1, pad=1, relu=True, tile=ops.TILE_XRES)
    return ops.conv2d_bf16(t2, w3c, b3, stride=1, pad=0, relu=True, residual=idn)


cases = (("conv2 + tail2 (2 launches)", unfused_chain), ("block2, chained conv1", lambda: ops.bneck_block2_bf16(t1, w2, b2, w3, b3, idn, w1, b1)),
         ("conv2 + conv3 (2 launches)", unfused_last), ("block2, last block", lambda: ops.bneck_block2_bf16(t1, w2, b2, w3, b3, idn)))
res = {k: [] for k, _ in cases}
for r in range(ROUNDS):
    for k, fn in cases:
        res[k].append(t_us(fn))
m = B * 784
for k, _ in cases:
    t = sorted(res[k]); med = t[len(t) // 2]
    chain = "chained" in k or "tail2" in k
    fl = 2.0 * m * (128 * 1152 + 512 * 128 + (128 * 512 if chain else 0))
    print(f"{k:30s} median {med:7.1f} us  min {t[0]:7.1f} us   {fl / med / 1e6:7.1f} TF/s", flush=True)
