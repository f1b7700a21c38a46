"""In-kernel cycle stamps of the role-split fused stem (stem_fused3_kernel).  Needs scripts/build_variant.sh _stamp -DR50_STAMP=1
usage: R50_LIB=$PWD/implementation_phd_lab_vision_amd/libr50hip_stamp.so python scripts/stamp_stem.py [batch]"""
import ctypes, sys, torch
sys.path.insert(0, '.')
from implementation_phd_lab_vision_amd import _lib
from implementation_phd_lab_vision_amd.backbone import ResNet50Backbone
from implementation_phd_lab_vision_amd.weights import synthetic_frames
lib = _lib.load_library()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
d = torch.device('cuda:0')
x = synthetic_frames(8, seed=3).to(d).repeat((n + 7) // 8, 1, 1, 1)[:n].contiguous()
bb = ResNet50Backbone(seed=0, max_batch=n, precision="bf16").to(d).eval()
dbg = torch.zeros((256, 12, 8), dtype=torch.int64, device=d)
lib.r50_debug_buffer.argtypes = [ctypes.c_void_p]
for _ in range(100): bb.layer(x, "layer1.0.t1")          # steady clocks before the stamped launch
lib.r50_debug_buffer(dbg.data_ptr())
bb.layer(x, "layer1.0.t1")
torch.cuda.synchronize()
lib.r50_debug_buffer(None)
t = dbg.double().cpu()
pairs = 28 * max(1, n // 256)
for name, arr, labels in (("conv waves", t[:, :8, :], ["barrier wait", "K loop (reads + MFMA)", "epilogue (pool, bias, ReLU, stores)", "loop overhead"]),
                          ("service waves", t[:, 8:, :], ["barrier wait", "pack", "issue loads", "pool", "layer1.0.conv1"])):
    m = arr.mean(dim=(0, 1)); tot = m[:len(labels)].sum()
    print(f"{name}: stamped {tot:.0f} cycles per wave, {tot / pairs:.0f} per pair; kernel {m[6]:.0f} cycles")
    for i, l in enumerate(labels):
        print(f"   {l:38s} {m[i]:9.0f} {100 * m[i] / tot:5.1f}%   {m[i] / pairs:7.0f} / pair")
pw = t.mean(dim=0) / pairs
print("per wave and pair (slots 0-4):")
for w in range(12):
    print(f"   wave {w:2d}: " + "  ".join(f"{float(pw[w, i]):6.0f}" for i in range(5)))
c = t[:, :8, :]
print(f"clock held during the kernel: {float((c[..., 6] / c[..., 7].clamp_min(1)).mean()) * 100:.0f} MHz  (kernel body: {float(c[..., 7].mean()) / 100:.1f} us)")
