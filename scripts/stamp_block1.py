"""In-kernel cycle stamps of the layer1 bottleneck body (bneck_block1_kernel): where a consumer wave's, a weight-loader wave's and an HBM-loader wave's
time goes.  Needs the diagnostic build:  scripts/build_variant.sh _stamp -DR50_STAMP=1
usage: R50_LIB=$PWD/implementation_phd_lab_vision_amd/libr50hip_stamp.so python scripts/stamp_block1.py [batch] [c1]"""
import ctypes, sys, torch
sys.path.insert(0, '.')
from implementation_phd_lab_vision_amd import ops, _lib
lib = _lib.load_library()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
C1 = int(sys.argv[2]) if len(sys.argv) > 2 else 64
d = torch.device('cuda:0'); g = torch.Generator().manual_seed(0)
rb = lambda shape, scale=1.0: (torch.randn(shape, generator=g) * scale).to(torch.bfloat16).to(d)
t1 = rb((B, 56, 56, 64)).clamp_(min=0); idn = rb((B, 56, 56, 256)).clamp_(min=0)
w2 = rb((64, 3, 3, 64), (2.0 / 576) ** 0.5); w3 = rb((256, 64), (2.0 / 64) ** 0.5); w1 = rb((C1, 256), (2.0 / 256) ** 0.5)
b2 = (torch.randn(64, generator=g) * 0.1).to(d); b3 = (torch.randn(256, generator=g) * 0.1).to(d); b1 = (torch.randn(C1, generator=g) * 0.1).to(d)
run = lambda: ops.bneck_block1_bf16(t1, w2, b2, w3, b3, idn, w1, b1)
dbg = torch.zeros((256, 12, 8), dtype=torch.int64, device=d)
lib.r50_debug_buffer.argtypes = [ctypes.c_void_p]
for _ in range(20):
    run()
lib.r50_debug_buffer(dbg.data_ptr())
run()
torch.cuda.synchronize()
lib.r50_debug_buffer(None)
t = dbg.double().cpu()
tiles = 14.0 * B / 256
for name, arr, labels in (("consumer waves 0-7", t[:, :8, :], ["reads + MFMAs", "stage barriers", "t2 / E(c)", "T2 / OUTC barriers", "next t1 stores"]),
                          ("weight loaders (waves 8, 9)", t[:, 8:10, :], ["DMA issue", "wait (L2)", "stage barriers", "extra barriers"]),
                          ("HBM loaders (waves 10, 11)", t[:, 10:12, :], ["DMA / copy-out issue", "waits (HBM)", "stage barriers", "extra barriers"])):
    m = arr.mean(dim=(0, 1))
    tot = m[:len(labels)].sum()
    print(f"{name}: total {tot:.0f} cycles per wave and launch ({tiles:.0f} tiles), {tot / tiles:.0f} per tile; held clock {100.0 * tot / m[7]:.0f} MHz ({m[7] / 100:.1f} us stamped)")
    for i, l in enumerate(labels):
        print(f"   {l:22s} {m[i]:10.0f}  {100 * m[i] / tot:5.1f}%   per tile {m[i] / tiles:7.0f}")
