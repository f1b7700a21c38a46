"""layer1.0's body in one launch (r50_op_bneck_block1_ds) against conv3x3_c64 + the fused downsample tail it replaces, interleaved rounds.
usage: python scripts/time_block1_ds.py [batch] [rounds]"""
import sys, torch
sys.path.insert(0, '.')
from implementation_phd_lab_vision_amd import ops, _lib
_lib.load_library()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 4
d = torch.device('cuda:0'); g = torch.Generator().manual_seed(0)
t1 = torch.randn((B, 56, 56, 64), generator=g).relu().to(torch.bfloat16).to(d)
x = torch.randn((B, 56, 56, 64), generator=g).relu().to(torch.bfloat16).to(d)
w2 = (torch.randn((64, 3, 3, 64), generator=g) * 0.06).to(torch.bfloat16).to(d)
w3 = (torch.randn((256, 64), generator=g) * 0.17).to(torch.bfloat16).to(d)
wd = (torch.randn((256, 64), generator=g) * 0.12).to(torch.bfloat16).to(d)
w1 = (torch.randn((64, 256), generator=g) * 0.09).to(torch.bfloat16).to(d)
b2 = torch.randn(64, generator=g).to(d); b3 = torch.randn(256, generator=g).to(d); bd = torch.randn(256, generator=g).to(d); b1 = torch.randn(64, generator=g).to(d)
m = B * 3136


def two():
    t2 = ops.conv2d_bf16(t1, w2, b2, stride=1, pad=1, relu=True, tile=ops.TILE_C64)
    return ops.bneck_tail_bf16(t2, w3, b3, x, w1, b1, wd=wd, bd=bd)


def one():
    return ops.bneck_block1_ds_bf16(t1, w2, b2, w3, b3, x, wd, bd, w1, b1)


a, b = two(); c, e = one(); torch.cuda.synchronize()
print("equal:", torch.equal(a, c), torch.equal(b, e), flush=True)
for rnd in range(ROUNDS):
    for name, fn in (("c64 + ds tail", two), ("one launch   ", one)):
        for _ in range(3): fn()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"round {rnd} {name}: {e0.elapsed_time(e1) * 50:7.1f} us", flush=True)
