"""conv3 + downsample as one two-source 1x1 conv (r50_op_conv1x1_cat) at the three ResNet-50 shapes, every role-specialised tile,
beside the two launches it replaces (downsample conv, conv3 with residual).  usage: time_cat.py [batch] [iters]"""
import sys, torch
sys.path.insert(0, '.')
from implementation_phd_lab_vision_amd import _lib, ops
_lib.build_library()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
IT = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = "cuda:0"
def timeit(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(IT): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / IT * 1e3
for name, h, c1, h2, c2, cout in (("layer2.0", 28, 128, 56, 256, 512), ("layer3.0", 14, 256, 28, 512, 1024), ("layer4.0", 7, 512, 14, 1024, 2048)):
    g = torch.Generator().manual_seed(1)
    x1 = torch.randn(B, h, h, c1, generator=g).bfloat16().to(dev)
    x2 = torch.randn(B, h2, h2, c2, generator=g).bfloat16().to(dev)
    wc = (torch.randn(cout, c1 + c2, generator=g) * 0.05).bfloat16().to(dev)
    w3, wd = wc[:, :c1].contiguous().view(cout, 1, 1, c1), wc[:, c1:].contiguous().view(cout, 1, 1, c2)
    bias = torch.zeros(cout, device=dev)
    flops = 2.0 * B * h * h * cout * (c1 + c2)
    t_ds = timeit(lambda: ops.conv2d_bf16(x2, wd, bias, stride=2, pad=0, relu=False))
    idn = ops.conv2d_bf16(x2, wd, bias, stride=2, pad=0, relu=False)
    t_c3 = timeit(lambda: ops.conv2d_bf16(x1, w3, bias, stride=1, pad=0, relu=True, residual=idn))
    line = f"{name}: downsample {t_ds:.1f} us + conv3(res) {t_c3:.1f} us = {t_ds + t_c3:.1f} us | two-source:"
    for tile in (64 | 8, 64 | 4, 64 | 3, 64 | 1, 64 | 9):
        try:
            t = timeit(lambda: ops.conv1x1_cat(x1, x2, 2, wc, bias, True, tile))
            line += f" {tile}:{t:.1f}us/{flops / t / 1e6:.0f}TF"
        except Exception as e:
            line += f" {tile}:n/a"
    print(line, flush=True)
