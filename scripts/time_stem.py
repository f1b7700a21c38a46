"""Fused stem launch (frames -> conv1+bn+relu -> maxpool -> layer1.0.conv1), the two kernels A/B in one process:
stem_variant 3 = stem_fused3_kernel (conv and pack/pool side by side), 2 = stem_fused2_kernel (in turn).  Same bits.
usage: python scripts/time_stem.py [batch] [rounds]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from implementation_phd_lab_vision_amd.backbone import ResNet50Backbone
from implementation_phd_lab_vision_amd.weights import synthetic_frames

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
x = synthetic_frames(8, seed=3).to("cuda:0").repeat((n + 7) // 8, 1, 1, 1)[:n].contiguous()
x8 = (torch.rand(n, 3, 224, 224, device="cuda:0") * 255).to(torch.uint8)
for prec in ["bf16", "fp16"]:
    bb = ResNet50Backbone(seed=0, max_batch=n, precision=prec).to("cuda:0").eval()
    ref = {}
    for rnd in range(rounds):
        for name, inp in (("fp32", x), ("u8", x8)):
            for var in (2, 3):
                bb.set_option("stem_variant", var)
                out = bb.layer(inp, "layer1.0.t1").clone()
                pool = bb.layer(inp, "pool").clone()
                key = (prec, name)
                if var == 2: ref[key] = (out, pool)
                same = torch.equal(out, ref[key][0]) and torch.equal(pool, ref[key][1])
                for _ in range(5): bb.layer(inp, "layer1.0.t1")
                torch.cuda.synchronize()
                ts = []
                for _ in range(20):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); bb.layer(inp, "layer1.0.t1"); e1.record(); e1.synchronize()
                    ts.append(e0.elapsed_time(e1) * 1e3)
                ts.sort()
                print(f"{prec} {name:5s} batch {n} variant {var}: median {ts[len(ts)//2]:7.1f} us  min {ts[0]:7.1f} us  bits={'same' if same else 'DIFFERENT'}", flush=True)
    bb.set_option("stem_variant", 3)
    bb.close()
