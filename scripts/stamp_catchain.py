"""In-kernel cycle stamps of bneck_catchain_kernel (layer2.0 transition tail + next conv1): where a group-A / group-B wave's time goes.
usage: R50_LIB=$PWD/implementation_phd_lab_vision_amd/libr50hip_stamp.so python scripts/stamp_catchain.py [batch]"""
import ctypes, sys, torch
sys.path.insert(0, '.')
from implementation_phd_lab_vision_amd import ops, _lib
lib = _lib.load_library()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
d = torch.device('cuda:0'); g = torch.Generator().manual_seed(0)
t2 = torch.randn((B, 28, 28, 128), generator=g).relu().to(torch.bfloat16).to(d)
x = torch.randn((B, 56, 56, 256), generator=g).relu().to(torch.bfloat16).to(d)
wcat = (torch.randn((512, 384), generator=g) * 0.07).to(torch.bfloat16).to(d)
w1 = (torch.randn((128, 512), generator=g) * 0.06).to(torch.bfloat16).to(d)
bc = torch.randn(512, generator=g).to(d); b1 = torch.randn(128, generator=g).to(d)
dbg = torch.zeros((256, 8, 8), dtype=torch.int64, device=d)
lib.r50_debug_buffer.argtypes = [ctypes.c_void_p]
for _ in range(3):
    ops.bneck_cat_chain_bf16(t2, x, wcat, bc, w1, b1)
lib.r50_debug_buffer(dbg.data_ptr())
ops.bneck_cat_chain_bf16(t2, x, wcat, bc, w1, b1)
torch.cuda.synchronize()
lib.r50_debug_buffer(None)
t = dbg.double().cpu()
nch = 4 * (B * 784 // 112 // 256 if B * 784 // 112 >= 256 else 1)
for name, arr, labels in (("group A", t[:, :4, :], ["A steps 4-5", "E", "chunk barrier", "-", "-", "A steps 0-3 + late-pair poll"]),
                          ("group B", t[:, 4:, :], ["y1n store + init", "-", "B (2 steps)", "copy-out", "chunk barrier", "release polls + operand issue", "landing waits"])):
    m = arr.mean(dim=(0, 1))
    tot = m[:len(labels)].sum()
    print(f"{name}: total {tot:.0f} cycles per wave (whole launch, {nch} chunks);  held clock {100.0 * tot / m[7]:.0f} MHz ({m[7] / 100:.1f} us stamped)")
    for i, l in enumerate(labels):
        print(f"   {l:24s} {m[i]:9.0f}  {100 * m[i] / tot:5.1f}%   per chunk {m[i] / nch:7.0f}")
