"""Is the step limited by the clock the chip's power budget allows?  The same 256-frame forward on random frames and on all-zero frames
(identical launches, addresses and instruction streams; far fewer bits toggle): frames/s of both, one lane and two.
usage: python scripts/zero_input_probe.py [steps]"""
import sys, time, torch
sys.path.insert(0, '.')
from implementation_phd_lab_vision_amd import _lib
from implementation_phd_lab_vision_amd.backbone import BackboneLanes
from implementation_phd_lab_vision_amd.weights import synthetic_frames, synthetic_state_dict
_lib.build_library()
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device('cuda', 0)
bl = BackboneLanes(lanes=2, state_dict=synthetic_state_dict(0), max_batch=256).to(dev).eval()
outs = [torch.empty(256, 2048, device=dev) for _ in range(2)]


def rate(x, lanes):
    def run(n):
        for k in range(n):
            if lanes == 2:
                bl.submit(x, out=outs[k & 1])
            else:
                bl.lane0.features(x, outs[0])
    run(30); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(STEPS); torch.cuda.synchronize()
    return 256 * STEPS / (time.perf_counter() - t0)


xr = synthetic_frames(256, seed=1234).to(dev)
xz = torch.zeros_like(xr)
for rnd in range(2):
    for name, x in (("random frames", xr), ("zero frames  ", xz)):
        print(f"round {rnd}  {name}:  one lane {rate(x, 1):8.0f} frames/s   two lanes {rate(x, 2):8.0f} frames/s", flush=True)
