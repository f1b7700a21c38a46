"""The nine streaming 1x1 launches of layer3 / layer4 at batch 256: the tuned generic tile of each against the eight-phase GEMM
tiles (gemm8p_kernel, 256 and 224 pixels), interleaved rounds in ONE process; also checks that the eight-phase tiles give the generic tile's bits.
usage: python scripts/time_g8.py [batch] [rounds] [iters]"""
import sys, torch
sys.path.insert(0, '.')
from implementation_phd_lab_vision_amd import _lib, ops
_lib.load_library()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
IT = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dev = "cuda:0"
G8, G8N7 = ops.TILE_G8, ops.TILE_G8_224

def timeit(fn):
    for _ in range(2): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(IT): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / IT * 1e3

# name, h, cin, cout, residual, tuned tile (auto)
SINGLE = [("layer3.0.conv1", 28, 512, 256, False), ("layer3.1.conv1", 14, 1024, 256, False), ("layer4.0.conv1", 14, 1024, 512, False),
          ("layer4.1.conv1", 7, 2048, 512, False), ("layer4.1.conv3", 7, 512, 2048, True)]
CAT = [("layer3.0.conv3+ds", 14, 256, 28, 512, 1024), ("layer4.0.conv3+ds", 7, 512, 14, 1024, 2048)]
total = {}
for name, h, cin, cout, has_res in SINGLE:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, h, h, cin, generator=g).bfloat16().to(dev)
    w = (torch.randn(cout, 1, 1, cin, generator=g) * (2.0 / cin) ** 0.5).bfloat16().to(dev)
    bias = torch.randn(cout, generator=g).to(dev)
    r = torch.randn(B, h, h, cout, generator=g).bfloat16().to(dev) if has_res else None
    flops = 2.0 * B * h * h * cout * cin
    ref = ops.conv2d_bf16(x, w, bias, relu=True, residual=r, tile=ops.TILE_256x256)
    res = {}
    for tile in (0, G8, G8N7):
        y = ops.conv2d_bf16(x, w, bias, relu=True, residual=r, tile=tile)
        same = bool(torch.equal(y, ref))
        res[tile] = [same]
    for _ in range(ROUNDS):
        for tile in (0, G8, G8N7):
            res[tile].append(timeit(lambda: ops.conv2d_bf16(x, w, bias, relu=True, residual=r, tile=tile)))
    line = f"{name:18s} {flops / 1e9:6.1f} GF:"
    for tile in (0, G8, G8N7):
        ts = sorted(res[tile][1:])
        med = ts[len(ts) // 2]
        line += f"  tile {tile:2d}: {med:6.1f} us (min {ts[0]:.1f}) {flops / med / 1e6:5.0f} TF bits={'same' if res[tile][0] else 'DIFFER'}"
        total.setdefault(tile, 0.0)
        total[tile] += med * (2 if name.startswith("layer4.1") else 1)
    print(line, flush=True)
for name, h, c1, h2, c2, cout in CAT:
    g = torch.Generator().manual_seed(1)
    x1 = torch.randn(B, h, h, c1, generator=g).bfloat16().to(dev)
    x2 = torch.randn(B, h2, h2, c2, generator=g).bfloat16().to(dev)
    wc = (torch.randn(cout, c1 + c2, generator=g) * (2.0 / (c1 + c2)) ** 0.5).bfloat16().to(dev)
    bias = torch.randn(cout, generator=g).to(dev)
    flops = 2.0 * B * h * h * cout * (c1 + c2)
    ref = ops.conv1x1_cat(x1, x2, 2, wc, bias, True, 64 | 3)
    res = {}
    for tile in (0, G8, G8N7):
        res[tile] = [bool(torch.equal(ops.conv1x1_cat(x1, x2, 2, wc, bias, True, tile), ref))]
    for _ in range(ROUNDS):
        for tile in (0, G8, G8N7):
            res[tile].append(timeit(lambda: ops.conv1x1_cat(x1, x2, 2, wc, bias, True, tile)))
    line = f"{name:18s} {flops / 1e9:6.1f} GF:"
    for tile in (0, G8, G8N7):
        ts = sorted(res[tile][1:])
        med = ts[len(ts) // 2]
        line += f"  tile {tile:2d}: {med:6.1f} us (min {ts[0]:.1f}) {flops / med / 1e6:5.0f} TF bits={'same' if res[tile][0] else 'DIFFER'}"
        total[tile] += med
    print(line, flush=True)
print("nine launches (layer4.1 rows counted twice for layer4.2): " + "  ".join(f"tile {t}: {v:.0f} us" for t, v in total.items()), flush=True)
