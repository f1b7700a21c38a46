"""Fused stem launch (frames -> conv1+bn+relu -> maxpool -> layer1.0.conv1; stem_fused3_kernel): time per launch for fp32 frames, bf16 and fp16, every strip length; the pooled rows and layer1.0.conv1's output must not depend on the strip length.
(profiles/r04_time_stem.txt is this script's round-4 output while the round-3 kernel was still in the tree beside it.)
usage: python scripts/time_stem.py [batch] [rounds]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from implementation_phd_lab_vision_amd.backbone import ResNet50Backbone
from implementation_phd_lab_vision_amd.weights import synthetic_frames

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
x = synthetic_frames(8, seed=3).to("cuda:0").repeat((n + 7) // 8, 1, 1, 1)[:n].contiguous()
for prec in ["bf16", "fp16"]:
    bb = ResNet50Backbone(seed=0, max_batch=n, precision=prec).to("cuda:0").eval()
    ref = {}
    for rnd in range(rounds):
        for name, inp in (("fp32", x),):       # (uint8 frames enter through features_u8; layer() would convert them to fp32 first)
            for var in (0, 28, 14, 7):
                bb.set_option("stem_strip", var)
                out = bb.layer(inp, "layer1.0.t1").clone()
                pool = bb.layer(inp, "pool").clone()
                key = (prec, name)
                if key not in ref: ref[key] = (out, pool)
                same = torch.equal(out, ref[key][0]) and torch.equal(pool, ref[key][1])
                for _ in range(5): bb.layer(inp, "layer1.0.t1")
                torch.cuda.synchronize()
                ts = []
                for _ in range(20):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); bb.layer(inp, "layer1.0.t1"); e1.record(); e1.synchronize()
                    ts.append(e0.elapsed_time(e1) * 1e3)
                ts.sort()
                print(f"{prec} {name:5s} batch {n} strip {var:2d}: median {ts[len(ts)//2]:7.1f} us  min {ts[0]:7.1f} us  bits={'same' if same else 'DIFFERENT'}", flush=True)
    bb.set_option("stem_strip", 0)
    bb.close()
