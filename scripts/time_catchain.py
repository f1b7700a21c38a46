"""layer2.0's transition tail chained with layer2.1.conv1 (r50_op_bneck_cat_chain) against the two launches it replaces (two-source igemm +
1x1 igemm), interleaved rounds in one process.  usage: python scripts/time_catchain.py [batch] [rounds]"""
import sys, torch
sys.path.insert(0, '.')
from implementation_phd_lab_vision_amd import ops, _lib
_lib.load_library()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 4
d = torch.device('cuda:0'); g = torch.Generator().manual_seed(0)
t2 = torch.randn((B, 28, 28, 128), generator=g).relu().to(torch.bfloat16).to(d)
x = torch.randn((B, 56, 56, 256), generator=g).relu().to(torch.bfloat16).to(d)
wcat = (torch.randn((512, 384), generator=g) * 0.07).to(torch.bfloat16).to(d)
w1 = (torch.randn((128, 512), generator=g) * 0.06).to(torch.bfloat16).to(d)
bc = torch.randn(512, generator=g).to(d); b1 = torch.randn(128, generator=g).to(d)
m = B * 784
flops = 2.0 * m * (512 * 384 + 128 * 512)
byts = 2.0 * m * (128 + 256 + 512 + 128)


def two():
    o = ops.conv1x1_cat(t2, x, 2, wcat, bc, relu=True)
    return o, ops.conv2d_bf16(o, w1.view(128, 1, 1, 512), b1, relu=True)


def one():
    return ops.bneck_cat_chain_bf16(t2, x, wcat, bc, w1, b1)


a, b = two(); c, e = one(); torch.cuda.synchronize()
print("equal:", torch.equal(a, c), torch.equal(b, e), flush=True)
for rnd in range(ROUNDS):
    for name, fn in (("two launches", two), ("chained     ", one)):
        for _ in range(3): fn()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 50
        print(f"round {rnd} {name}: {us:7.1f} us  {flops / us / 1e6:7.1f} TF/s  {byts / us / 1e6:5.2f} TB/s algorithmic", flush=True)
