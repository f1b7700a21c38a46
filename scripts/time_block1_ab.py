"""Layer1 plain-identity bodies at batch B: bneck_block1_kernel (option block1_fat = 0) against bneck_block1f_kernel (= 1: 16-KB weight stages, identity
through the consumers' registers), interleaved rounds in ONE process, bits compared.  usage: python scripts/time_block1_ab.py [batch] [rounds] [iters]"""
import sys, torch
sys.path.insert(0, '.')
from implementation_phd_lab_vision_amd import ops, _lib
from implementation_phd_lab_vision_amd.backbone import ResNet50Backbone
_lib.load_library()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
IT = int(sys.argv[3]) if len(sys.argv) > 3 else 10
d = torch.device('cuda:0'); g = torch.Generator().manual_seed(0)
bb = ResNet50Backbone(seed=0, max_batch=2).to(d)          # only to reach the process-wide option
rb = lambda shape, scale=1.0: (torch.randn(shape, generator=g) * scale).to(torch.bfloat16).to(d)
t1 = rb((B, 56, 56, 64)).clamp_(min=0); idn = rb((B, 56, 56, 256)).clamp_(min=0)
w2 = rb((64, 3, 3, 64), (2.0 / 576) ** 0.5); w3 = rb((256, 64), (2.0 / 64) ** 0.5)
b2 = (torch.randn(64, generator=g) * 0.1).to(d); b3 = (torch.randn(256, generator=g) * 0.1).to(d)
W1 = {c1: rb((c1, 256), (2.0 / 256) ** 0.5) for c1 in (64, 128)}
B1 = {c1: (torch.randn(c1, generator=g) * 0.1).to(d) for c1 in (64, 128)}


def t_us(fn):
    for _ in range(2): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(IT): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / IT


m = B * 3136
for c1 in (64, 128):
    fn = lambda: ops.bneck_block1_bf16(t1, w2, b2, w3, b3, idn, W1[c1], B1[c1])
    outs = {}
    for v in (0, 1):
        bb.set_option("block1_fat", v)
        outs[v] = fn()
    torch.cuda.synchronize()
    same = all(bool(torch.equal(x, y)) for x, y in zip(outs[0], outs[1]))
    res = {0: [], 1: []}
    for _ in range(ROUNDS):
        for v in (0, 1):
            bb.set_option("block1_fat", v)
            res[v].append(t_us(fn))
    line = f"layer1 body, next conv1 -> {c1:3d}:"
    for v in (0, 1):
        t = sorted(res[v]); med = t[len(t) // 2]
        line += f"  block1_fat={v}: {med:7.1f} us (min {t[0]:.1f})  {m * 2.0 * (64 + 256 + 256 + c1) / med / 1e6:5.2f} TB/s"
    print(line + f"   bits={'same' if same else 'DIFFER'}", flush=True)
bb.set_option("block1_fat", 1)
