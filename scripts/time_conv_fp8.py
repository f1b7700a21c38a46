"""fp8 conv path at kernel level (BASELINE configs[4]: batch 512, 224x224 frames): ResNet-50 conv shapes with Cin >= 128 through
r50_op_conv2d_fp8 (K = 128 scaled fp8 MFMA) beside the bf16 launch of the same shape.  usage: time_conv_fp8.py [batch] [iters]"""
import sys, torch
sys.path.insert(0, '.')
from implementation_phd_lab_vision_amd import _lib, ops
_lib.build_library()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
IT = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = "cuda:0"
def timeit(fn):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(IT): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / IT * 1e3
SHAPES = [  # name, h, cin, cout, k, stride, residual   (+ the stride-2 / downsample shapes of the first blocks)
    ("layer2.0 conv1 56x56 256->128 k1", 56, 256, 128, 1, 1, False), ("layer2.0 conv2 56x56 128->128 k3 s2", 56, 128, 128, 3, 2, False),
    ("layer2.0 ds 56x56 256->512 k1 s2", 56, 256, 512, 1, 2, False), ("layer3.0 conv1 28x28 512->256 k1", 28, 512, 256, 1, 1, False),
    ("layer3.0 conv2 28x28 256->256 k3 s2", 28, 256, 256, 3, 2, False), ("layer3.0 ds 28x28 512->1024 k1 s2", 28, 512, 1024, 1, 2, False),
    ("layer4.0 conv1 14x14 1024->512 k1", 14, 1024, 512, 1, 1, False), ("layer4.0 conv2 14x14 512->512 k3 s2", 14, 512, 512, 3, 2, False),
    ("layer4.0 ds 14x14 1024->2048 k1 s2", 14, 1024, 2048, 1, 2, False),
    ("layer2 conv2 28x28 128->128 k3", 28, 128, 128, 3, 1, False), ("layer2 conv1 28x28 512->128 k1", 28, 512, 128, 1, 1, False),
    ("layer2 conv3 28x28 128->512 k1 +res", 28, 128, 512, 1, 1, True), ("layer3 conv1 14x14 1024->256 k1", 14, 1024, 256, 1, 1, False),
    ("layer3 conv2 14x14 256->256 k3", 14, 256, 256, 3, 1, False), ("layer3 conv3 14x14 256->1024 k1 +res", 14, 256, 1024, 1, 1, True),
    ("layer4 conv1 7x7 2048->512 k1", 7, 2048, 512, 1, 1, False), ("layer4 conv2 7x7 512->512 k3", 7, 512, 512, 3, 1, False),
    ("layer4 conv3 7x7 512->2048 k1 +res", 7, 512, 2048, 1, 1, True)]
tot8 = tot16 = 0.0
for name, h, cin, cout, k, s, res in SHAPES:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, h, h, cin, generator=g)
    w = torch.randn(cout, k, k, cin, generator=g) * 0.05
    bias = torch.zeros(cout, device=dev)
    pad = k // 2
    ho = (h + 2 * (k // 2) - k) // s + 1
    r = torch.randn(B, ho, ho, cout, generator=g) if res else None
    x16, w16, r16 = x.bfloat16().to(dev), w.bfloat16().to(dev), (r.bfloat16().to(dev) if res else None)
    x8, w8, r8 = (x * 8).to(ops.FP8).to(dev), (w * 64).to(ops.FP8).to(dev), ((r * 8).to(ops.FP8).to(dev) if res else None)
    flops = 2.0 * B * h * h * cout * k * k * cin
    t16 = timeit(lambda: ops.conv2d_bf16(x16, w16, bias, stride=s, pad=pad, relu=True, residual=r16))
    best = None
    per = ""
    for tile in (0, 64 | 3, 64 | 8, 64 | 4, 64 | 1):
        try:
            t = timeit(lambda: ops.conv2d_fp8(x8, 0.125, w8, 1 / 64, bias, 0.125, stride=s, pad=pad, relu=True, residual=r8, sr=0.125, tile=tile))
        except Exception:
            continue
        per += f" {tile}:{t:.0f}"
        if tile and (best is None or t < best[0]): best = (t, tile)
    tot8 += best[0]; tot16 += t16
    print(f"{name:40s} bf16 {t16:7.1f} us {flops / t16 / 1e6:6.0f} TF/s | fp8 {best[0]:7.1f} us {flops / best[0] / 1e6:6.0f} TF/s (tile {best[1]}) x{t16 / best[0]:.2f} |{per}", flush=True)
print(f"sum over the shapes at batch {B}: bf16 {tot16:.0f} us, fp8 {tot8:.0f} us (x{tot16 / tot8:.2f})")
