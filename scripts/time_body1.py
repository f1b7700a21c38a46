"""Layer1 bottleneck bodies at batch B: bneck_block1_kernel (weight-stage ring, option body1 = 0) against bneck_body1_kernel (round 4: weights and
identity through registers, six barriers per tile, option body1 = 7), interleaved rounds in ONE process, bits compared.
usage: python scripts/time_body1.py [batch] [rounds] [iters]"""
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
t1 = rb((B, 56, 56, 64)).clamp_(min=0); idn = rb((B, 56, 56, 256)).clamp_(min=0); x = rb((B, 56, 56, 64)).clamp_(min=0)
w2 = rb((64, 3, 3, 64), (2.0 / 576) ** 0.5); w3 = rb((256, 64), (2.0 / 64) ** 0.5); wd = rb((256, 64), (1.0 / 64) ** 0.5)
b2 = (torch.randn(64, generator=g) * 0.1).to(d); b3 = (torch.randn(256, generator=g) * 0.1).to(d); bd = (torch.randn(256, generator=g) * 0.1).to(d)
W1 = {c1: rb((c1, 256), (2.0 / 256) ** 0.5) for c1 in (64, 128)}
B1 = {c1: (torch.randn(c1, generator=g) * 0.1).to(d) for c1 in (64, 128)}


def t_us(fn):
    for _ in range(2): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(IT): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / IT


cases = {
    "layer1.0 (downsample identity, c1=64)": (lambda: ops.bneck_block1_ds_bf16(t1, w2, b2, w3, b3, x, wd, bd, W1[64], B1[64]), 64 + 64 + 256 + 64),
    "layer1.1 (c1=64)": (lambda: ops.bneck_block1_bf16(t1, w2, b2, w3, b3, idn, W1[64], B1[64]), 64 + 256 + 256 + 64),
    "layer1.2 (c1=128)": (lambda: ops.bneck_block1_bf16(t1, w2, b2, w3, b3, idn, W1[128], B1[128]), 64 + 256 + 256 + 128),
}
m = B * 3136
for name, (fn, ch) in cases.items():
    outs = {}
    for v in (0, 7):
        bb.set_option("body1", v)
        outs[v] = fn()
    torch.cuda.synchronize()
    same = all(bool(torch.equal(a, b)) for a, b in zip(outs[0], outs[7]))
    res = {0: [], 7: []}
    for _ in range(ROUNDS):
        for v in (0, 7):
            bb.set_option("body1", v)
            res[v].append(t_us(fn))
    line = f"{name:40s}"
    for v in (0, 7):
        t = sorted(res[v]); med = t[len(t) // 2]
        line += f"  body1={v}: {med:7.1f} us (min {t[0]:.1f})  {m * 2.0 * ch / med / 1e6:5.2f} TB/s"
    print(line + f"   bits={'same' if same else 'DIFFER'}", flush=True)
bb.set_option("body1", 0)
