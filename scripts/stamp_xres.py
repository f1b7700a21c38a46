"""In-kernel cycle stamps of the input-resident 3x3 kernel (conv3x3_xres_kernel).  Needs scripts/build_variant.sh _stamp -DR50_STAMP=1
usage: R50_LIB=$PWD/implementation_phd_lab_vision_amd/libr50hip_stamp.so python scripts/stamp_xres.py [hw c]"""
import ctypes, sys, torch
sys.path.insert(0, '.')
from implementation_phd_lab_vision_amd import ops, _lib
lib = _lib.load_library()
hw, c = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (14, 256)
B = 256
d = torch.device('cuda:0'); g = torch.Generator().manual_seed(0)
x = torch.randn((B, hw, hw, c), generator=g).to(torch.bfloat16).to(d)
w = (torch.randn((c, 3, 3, c), generator=g) * (2.0 / (9 * c)) ** 0.5).to(torch.bfloat16).to(d)
b = torch.randn(c, generator=g).to(d)
dbg = torch.zeros((256, 12, 8), dtype=torch.int64, device=d)
lib.r50_debug_buffer.argtypes = [ctypes.c_void_p]
for _ in range(3): ops.conv2d_bf16(x, w, b, stride=1, pad=1, relu=True, tile=ops.TILE_XRES)
lib.r50_debug_buffer(dbg.data_ptr())
for _ in range(200): ops.conv2d_bf16(x, w, b, stride=1, pad=1, relu=True, tile=ops.TILE_XRES)   # steady clocks before the stamped launch
ops.conv2d_bf16(x, w, b, stride=1, pad=1, relu=True, tile=ops.TILE_XRES)
torch.cuda.synchronize()
lib.r50_debug_buffer(None)
t = dbg.double().cpu()
steps = 9 * (c // 64)
for name, arr, labels in (("consumer", t[:, :8, :], ["tile end", "reads+MFMA", "barrier", "epilogue"]), ("loader", t[:, 8:, :], ["DMA issue", "wait landed", "barrier"])):
    m = arr.mean(dim=(0, 1)); tot = m[:len(labels)].sum()
    print(f"{name}: total {tot:.0f} cycles per wave")
    for i, l in enumerate(labels):
        print(f"   {l:12s} {m[i]:9.0f} {100 * m[i] / tot:5.1f}%")
c = t[:, :8, :]
print(f"clock held during the kernel: {float((c[..., 6] / c[..., 7].clamp_min(1)).mean()) * 100:.0f} MHz  (loop: {float(c[..., 7].mean()) / 100:.1f} us)")
print(f"prologue (kernel entry -> first barrier released): {float(c[..., 5].mean()):.0f} cycles = {float(c[..., 5].mean()) / (float((c[..., 6] / c[..., 7].clamp_min(1)).mean()) * 100):.2f} us")
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): ops.conv2d_bf16(x, w, b, stride=1, pad=1, relu=True, tile=ops.TILE_XRES)
e1.record(); torch.cuda.synchronize()
print(f"back-to-back launches (stamp build): {e0.elapsed_time(e1) * 5:.1f} us each")
loop_us = c[..., 7].mean(dim=1) / 100          # per workgroup
print(f"loop time per workgroup: min {float(loop_us.min()):.1f}  median {float(loop_us.median()):.1f}  max {float(loop_us.max()):.1f} us;  prologue max {float(c[..., 5].max()):.0f} cycles")
