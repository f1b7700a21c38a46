"""In-kernel cycle stamps of the role-specialised igemm (loader / consumer waves).

Needs the diagnostic build:  hipcc ... -DR50_STAMP=1 -o implementation_phd_lab_vision_amd/libr50hip_stamp.so
usage: R50_LIB=.../libr50hip_stamp.so python scripts/stamp_conv.py "B,H,cin,cout,k,s,res" tile n_consumer_waves n_loader_waves
Prints cycles per K-step per wave, averaged over the workgroups, consumers and loaders apart.
"""
import ctypes, sys, torch
sys.path.insert(0, '.')
from implementation_phd_lab_vision_amd import ops, _lib
lib = _lib.load_library()
B, H, cin, cout, k, s, res = (int(v) for v in sys.argv[1].split(','))
tile, ncons, nload = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
d = torch.device('cuda:0'); g = torch.Generator().manual_seed(0)
pad = 1 if k == 3 else 0
ho = (H + 2 * pad - k) // s + 1
x = torch.randn((B, H, H, cin), generator=g).to(torch.bfloat16).to(d)
w = (torch.randn((cout, k, k, cin), generator=g) * (2.0 / (cin * k * k)) ** 0.5).to(torch.bfloat16).to(d)
bias = torch.randn(cout, generator=g).to(d)
r = torch.randn((B, ho, ho, cout), generator=g).to(torch.bfloat16).to(d) if res else None
nw = ncons + nload
dbg = torch.zeros((256, nw, 8), dtype=torch.int64, device=d)
lib.r50_debug_buffer.argtypes = [ctypes.c_void_p]
for _ in range(3):
    ops.conv2d_bf16(x, w, bias, stride=s, pad=pad, relu=True, residual=r, tile=tile)
lib.r50_debug_buffer(dbg.data_ptr())
ops.conv2d_bf16(x, w, bias, stride=s, pad=pad, relu=True, residual=r, tile=tile)
torch.cuda.synchronize()
lib.r50_debug_buffer(None)
t = dbg.double().cpu()
used = t[:, 0, :].sum(dim=1) > 0
t = t[used]
nk = k * k * (cin // 64)
# steps per workgroup are not uniform; normalise by each workgroup's own total
cons, load = t[:, :ncons, :], t[:, ncons:, :]
tot_c = cons[..., :4].sum(-1).mean()
steps = None
print(f"{H}x{H} {cin}->{cout} k{k} tile {tile}: {int(used.sum())} workgroups, mean cycles per workgroup {tot_c:.0f}")
for name, arr, labels in (("consumer", cons, ["tile begin", "reads+MFMA", "epilogue", "barrier"]),
                          ("loader", load, ["DMA issue", "wait landed", "barrier"])):
    m = arr.mean(dim=(0, 1))
    tot = m[:len(labels)].sum()
    print(f"  {name}: " + "  ".join(f"{l} {100 * m[i] / tot:.1f}%" for i, l in enumerate(labels)) + f"   (total {tot:.0f} cycles)")
