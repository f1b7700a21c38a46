"""Time the 3x3 s1 conv shapes of layer2 / 3 / 4 (conv2 of the non-first blocks) at batch B: input-resident kernel vs the tuned generic tiles.
usage: python scripts/time_xres.py [batch]"""
import sys, torch
sys.path.insert(0, '.')
from implementation_phd_lab_vision_amd import ops, _lib
_lib.load_library()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
d = torch.device('cuda:0'); g = torch.Generator().manual_seed(0)
for hw, c, tuned in ((28, 128, 72), (14, 256, 68), (7, 512, 68)):
    x = torch.randn((B, hw, hw, c), generator=g).to(torch.bfloat16).to(d)
    w = (torch.randn((c, 3, 3, c), generator=g) * (2.0 / (9 * c)) ** 0.5).to(torch.bfloat16).to(d)
    b = torch.randn(c, generator=g).to(d)
    ref = None
    for name, tile in (("xres", ops.TILE_XRES), ("tuned", tuned), ("xres", ops.TILE_XRES), ("tuned", tuned)):
        for _ in range(3): y = ops.conv2d_bf16(x, w, b, stride=1, pad=1, relu=True, tile=tile)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): y = ops.conv2d_bf16(x, w, b, stride=1, pad=1, relu=True, tile=tile)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 50
        fl = 2.0 * B * hw * hw * c * c * 9
        if ref is None: ref = y.clone()
        diff = float((y.float() - ref.float()).abs().max())
        print(f"{hw}x{hw} c{c} {name:6s} tile {tile:3d}: {us:7.1f} us  {fl / us / 1e6:7.1f} TF/s   max|diff vs first| {diff:.3g}", flush=True)
