"""Time the 3x3 STRIDE-2 conv shapes of layer2.0 / layer3.0 at batch B: polyphase input-resident kernel (ops.TILE_S2) against the tuned generic
tiles, interleaved in one process; max |diff| against an fp32 torch conv of the same 16-bit operands.  usage: python scripts/time_s2.py [batch] [rounds]"""
import sys, torch
sys.path.insert(0, '.')
from implementation_phd_lab_vision_amd import ops, _lib
_lib.load_library()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
d = torch.device('cuda:0'); g = torch.Generator().manual_seed(0)
for hw, c, tuned in ((56, 128, 65), (28, 256, 44), (14, 512, 68)):
    x = torch.randn((B, hw, hw, c), generator=g).to(torch.bfloat16).to(d)
    w = (torch.randn((c, 3, 3, c), generator=g) * (2.0 / (9 * c)) ** 0.5).to(torch.bfloat16).to(d)
    b = torch.randn(c, generator=g).to(d)
    nb = min(B, 4)
    ref = torch.nn.functional.conv2d(x[:nb].float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), b, stride=2, padding=1).relu().permute(0, 2, 3, 1)
    times = {"s2": [], "tuned": []}
    for r in range(ROUNDS):
        for name, tile in (("s2", ops.TILE_S2), ("tuned", tuned)):
            for _ in range(3): y = ops.conv2d_bf16(x, w, b, stride=2, pad=1, relu=True, tile=tile)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30): y = ops.conv2d_bf16(x, w, b, stride=2, pad=1, relu=True, tile=tile)
            e1.record(); torch.cuda.synchronize()
            times[name].append(e0.elapsed_time(e1) * 1000 / 30)
            if r == 0:
                diff = (y[:nb].float() - ref).abs()
                print(f"  {name}: max |diff vs fp32 conv| {float(diff.max()):.4g}  mean {float(diff.mean()):.3g}  (|ref| max {float(ref.abs().max()):.3g})", flush=True)
    fl = 2.0 * B * (hw // 2) ** 2 * c * c * 9
    for name in times:
        t = sorted(times[name]); med = t[len(t) // 2]
        print(f"{hw}->{hw // 2} c{c} {name:6s}: median {med:6.1f} us  min {t[0]:6.1f} us  {fl / med / 1e6:7.1f} TF/s", flush=True)
