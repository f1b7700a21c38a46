"""Same-process A/B of the input-resident 3x3 kernel's two consumer schedules (option "xres_variant": 0 = barrier at the step's end,
5 = mid-step barrier + next-step operand prefetch), bit-for-bit comparison included.  usage: python scripts/time_xres_sched.py [batch] [rounds]"""
import sys, torch
sys.path.insert(0, '.')
from implementation_phd_lab_vision_amd import ops, _lib
from implementation_phd_lab_vision_amd.backbone import ResNet50Backbone
_lib.load_library()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 4
VARIANTS = [int(v) for v in sys.argv[3].split(',')] if len(sys.argv) > 3 else [0, 5]
d = torch.device('cuda:0'); g = torch.Generator().manual_seed(0)
bb = ResNet50Backbone(seed=0, max_batch=2).to(d)       # only to reach the process-wide option
for hw, c in ((28, 128), (14, 256), (7, 512)):
    x = torch.randn((B, hw, hw, c), generator=g).to(torch.bfloat16).to(d)
    w = (torch.randn((c, 3, 3, c), generator=g) * (2.0 / (9 * c)) ** 0.5).to(torch.bfloat16).to(d)
    b = torch.randn(c, generator=g).to(d)
    ref = None
    times = {v: [] for v in VARIANTS}
    for r in range(ROUNDS):
        for var in VARIANTS:
            bb.set_option('xres_variant', var)
            for _ in range(3): y = ops.conv2d_bf16(x, w, b, stride=1, pad=1, relu=True, tile=ops.TILE_XRES)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): y = ops.conv2d_bf16(x, w, b, stride=1, pad=1, relu=True, tile=ops.TILE_XRES)
            e1.record(); torch.cuda.synchronize()
            times[var].append(e0.elapsed_time(e1) * 20)
            if ref is None: ref = y.clone()
            elif not torch.equal(y, ref): print(f"  !! variant {var} differs from variant {VARIANTS[0]}: max |diff| {float((y.float() - ref.float()).abs().max()):.3g}", flush=True)
    fl = 2.0 * B * hw * hw * c * c * 9
    for var in VARIANTS:
        t = sorted(times[var]); med = t[len(t) // 2]
        print(f"{hw}x{hw} c{c} variant {var}: median {med:6.1f} us  min {t[0]:6.1f} us  {fl / med / 1e6:7.1f} TF/s", flush=True)
bb.set_option('xres_variant', 0)
bb.close()
