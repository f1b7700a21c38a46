import sys, time, torch
sys.path.insert(0, '.')
from implementation_phd_lab_vision_amd import _lib
from implementation_phd_lab_vision_amd.backbone import BackboneLanes
from implementation_phd_lab_vision_amd.weights import synthetic_frames, synthetic_state_dict
_lib.build_library()
dev = torch.device('cuda', 0)
sd = synthetic_state_dict(0)
x = synthetic_frames(256, seed=1234).to(dev)
outs = [torch.empty(256, 2048, device=dev) for _ in range(2)]
def rate(bl, lanes, steps=40):
    def run(n):
        for k in range(n):
            if lanes == 2: bl.submit(x, out=outs[k & 1])
            else: bl.lane0.features(x, outs[0])
    run(20); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(steps); torch.cuda.synchronize()
    return 256 * steps / (time.perf_counter() - t0)
for it in range(8):
    bl = BackboneLanes(lanes=2, state_dict=sd, max_batch=256).to(dev).eval()
    if len(sys.argv) > 1 and sys.argv[1] == "prio":
        bl._streams = [torch.cuda.Stream(dev, priority=0), torch.cuda.Stream(dev, priority=-1)]
    ids = [s.cuda_stream for s in bl._streams]
    print(f"instance {it}: retries {bl.stream_retries} streams {[hex(i) for i in ids]}  one lane {rate(bl,1):8.0f}  two lanes {rate(bl,2):8.0f}", flush=True)
    bl.close()
