"""Per-layer timing of every distinct conv shape of ResNet-50 at a given batch, for each igemm tile
variant (LDS-DMA and register staging), through the op-level C ABI.  Random bf16 data.
usage: python scripts/layer_bench.py [batch] [iters] [tiles csv]"""
import sys, json, torch
sys.path.insert(0, '.')
from implementation_phd_lab_vision_amd import ops, _lib
from implementation_phd_lab_vision_amd.weights import conv_specs
_lib.build_library(); _lib.load_library()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else 10
TILES = [int(t) for t in sys.argv[3].split(',')] if len(sys.argv) > 3 else [1, 2, 3, 5, 6, 7, 9, 10, 11, 12, 33, 34, 37, 38, 39, 40, 41, 42, 43, 44]
d = torch.device('cuda:0')
# distinct shapes in execution order with spatial size
shapes = {}
h = 56
for ck, bk, cin, cout, k, s, p in conv_specs()[1:]:
    blk = ck.split('.')
    stage = int(blk[0][5:])
    hin = {1: 56, 2: 28, 3: 14, 4: 7}[stage]
    if blk[1] == '0' and stage > 1 and (ck.endswith('conv1') or ck.endswith('conv2') or 'downsample' in ck):
        hin *= 2            # first block of a stage still sees the previous resolution until conv2
    if blk[1] == '0' and stage > 1 and ck.endswith('conv3'):
        pass
    key = (hin, cin, cout, k, s, ck.endswith('conv3'))
    shapes.setdefault(key, []).append(ck)
g = torch.Generator().manual_seed(0)
rows = []
tot = {}
for (hin, cin, cout, k, s, res), names in shapes.items():
    pad = 1 if k == 3 else 0
    ho = (hin + 2 * pad - k) // s + 1
    x = torch.randn((B, hin, hin, cin), generator=g).to(torch.bfloat16).to(d)
    w = (torch.randn((cout, k, k, cin), generator=g) * (2.0 / (cin * k * k)) ** 0.5).to(torch.bfloat16).to(d)
    bias = torch.randn(cout, generator=g).to(d)
    r = torch.randn((B, ho, ho, cout), generator=g).to(torch.bfloat16).to(d) if res else None
    flops = 2.0 * B * ho * ho * cout * cin * k * k
    byts = 2.0 * (B * hin * hin * cin + B * ho * ho * cout * (2 if res else 1) + cout * cin * k * k)
    best = None
    line = {}
    for tile in TILES:
        try:
            for _ in range(2):
                ops.conv2d_bf16(x, w, bias, stride=s, pad=pad, relu=True, residual=r, tile=tile)
        except RuntimeError:       # tile shape does not divide this layer's Cout
            continue
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(ITERS):
            ops.conv2d_bf16(x, w, bias, stride=s, pad=pad, relu=True, residual=r, tile=tile)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / ITERS
        line[tile] = us
        if best is None or us < best[1]: best = (tile, us)
    cnt = len(names)
    print(f"{hin:3d}x{hin:<3d} {cin:4d}->{cout:4d} k{k} s{s} res{int(res)} x{cnt}: best tile {best[0]:2d} {best[1]:8.1f} us "
          f"{flops/best[1]/1e6:7.1f} TF/s {byts/best[1]/1e3:7.1f} GB/s | " + ' '.join(f"{t}:{u:.0f}" for t, u in line.items()), flush=True)
    rows.append({"shape": [hin, cin, cout, k, s, int(res)], "count": cnt, "best": best[0], "us": line})
    for t, u in line.items(): tot[t] = tot.get(t, 0) + cnt * u
    tot['best'] = tot.get('best', 0) + cnt * best[1]
print("totals us (sum over 52 convs; missing variants excluded):", {k: round(v) for k, v in tot.items()})
json.dump(rows, open('gpurun_out/layer_bench.json', 'w'))
