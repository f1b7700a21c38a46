"""In-kernel cycle stamps of the pipelined layer3 tail (bneck_tail3p_kernel): where a group-A / group-B wave's time goes.
Needs the diagnostic build:  scripts/build_variant.sh _stamp -DR50_STAMP=1
usage: R50_LIB=$PWD/implementation_phd_lab_vision_amd/libr50hip_stamp.so python scripts/stamp_tail3p.py [batch]"""
import ctypes, os, sys, torch
sys.path.insert(0, '.')
from implementation_phd_lab_vision_amd import ops, _lib
lib = _lib.load_library()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
d = torch.device('cuda:0'); g = torch.Generator().manual_seed(0)
y2 = torch.randn((B, 14, 14, 256), generator=g).to(torch.bfloat16).to(d)
idn = torch.randn((B, 14, 14, 1024), generator=g).to(torch.bfloat16).to(d)
w3 = (torch.randn((1024, 256), generator=g) * 0.09).to(torch.bfloat16).to(d)
w1 = (torch.randn((256, 1024), generator=g) * 0.04).to(torch.bfloat16).to(d)
b3 = torch.randn(1024, generator=g).to(d); b1 = torch.randn(256, generator=g).to(d)
dbg = torch.zeros((256, 8, 8), dtype=torch.int64, device=d)
lib.r50_debug_buffer.argtypes = [ctypes.c_void_p]
for _ in range(3):
    ops.bneck_tail_bf16(y2, w3, b3, idn, w1, b1)
lib.r50_debug_buffer(dbg.data_ptr())
ops.bneck_tail_bf16(y2, w3, b3, idn, w1, b1)
torch.cuda.synchronize()
lib.r50_debug_buffer(None)
t = dbg.double().cpu()
for name, arr, labels in (("group A", t[:, :4, :], ["A (4 steps)", "E", "chunk barrier", "t2 issue + landing", "ninth barrier"]),
                          ("group B", t[:, 4:, :], ["first interval", "first barrier", "B (4 steps)", "copy-out", "chunk barrier"])):
    m = arr.mean(dim=(0, 1))
    tot = m[:len(labels)].sum()
    print(f"{name}: total {tot:.0f} cycles per wave (whole launch, 2 tiles = 16 chunks)")
    print(f"   held clock: {100.0 * tot / m[7]:.0f} MHz ({m[7] / 100:.1f} us stamped)")
    for i, l in enumerate(labels):
        print(f"   {l:20s} {m[i]:9.0f}  {100 * m[i] / tot:5.1f}%   per chunk {m[i] / 16:7.0f}")
