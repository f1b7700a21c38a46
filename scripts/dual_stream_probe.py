"""Do two half-batch pipelines on disjoint halves of the chip overlap the HBM-bound and the MFMA-bound launches?

Two handles, each on its own stream, each running 128-frame batches with every persistent grid capped at `cap` workgroups
(option "cu_cap"), free-running (no cross-stream dependency after the start).  Stream B starts `offset_us` after stream A so that
one pipeline is in stem / layer1 / layer2 (HBM-bound) while the other is in layer3 / layer4 (MFMA-bound).
usage: dual_stream_probe.py [steps] [cap,...] [offset_fraction,...]
"""
import sys, time, torch
sys.path.insert(0, '.')
from implementation_phd_lab_vision_amd import _lib
from implementation_phd_lab_vision_amd.backbone import ResNet50Backbone
from implementation_phd_lab_vision_amd.weights import synthetic_frames, synthetic_state_dict

_lib.build_library()
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 40
CAPS = [int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else [128]
OFFS = [float(v) for v in sys.argv[3].split(',')] if len(sys.argv) > 3 else [0.0, 0.5]
dev = torch.device('cuda', 0)
sd = synthetic_state_dict(0)
x = synthetic_frames(256, seed=1).to(dev)


def single(steps):
    bb = ResNet50Backbone(state_dict=sd, max_batch=256).to(dev).eval()
    bb.set_option('cu_cap', 0)
    out = torch.empty(256, 2048, device=dev)
    for _ in range(5): bb.features(x, out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): bb.features(x, out)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ref = out.clone()
    bb.close()
    return 256 * steps / dt, ref


def dual(steps, cap, off_frac, n_pipes=2, per=None):
    full = per is not None                    # full = every pipe runs whole 256-frame batches (two batches in flight)
    per = per or 256 // n_pipes
    bbs = [ResNet50Backbone(state_dict=sd, max_batch=per).to(dev).eval() for _ in range(n_pipes)]
    bbs[0].set_option('cu_cap', cap)          # process-wide
    streams = [torch.cuda.Stream(dev) for _ in range(n_pipes)]
    outs = [torch.empty(per, 2048, device=dev) for _ in range(n_pipes)]
    xs = [x if full else x[i * per:(i + 1) * per].contiguous() for i in range(n_pipes)]
    for i in range(n_pipes):
        with torch.cuda.stream(streams[i]):
            for _ in range(3): bbs[i].features(xs[i], outs[i])
    torch.cuda.synchronize()
    # offset: pipe i first runs a few layers' worth of delay = a partial pass (stem..layerK tap) so that it starts later
    t0 = time.perf_counter()
    for i in range(n_pipes):
        with torch.cuda.stream(streams[i]):
            if i and off_frac > 0:
                bbs[i].layer(xs[i], 'layer2.3' if off_frac >= 0.5 else 'layer1.2')      # a partial pass = the phase offset
    for s in range(steps):
        for i in range(n_pipes):
            with torch.cuda.stream(streams[i]):
                bbs[i].features(xs[i], outs[i])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res = outs[0] if full else torch.cat(outs)
    for b in bbs: b.close()
    return per * n_pipes * steps / dt, res


base, ref = single(STEPS)
print(f"single stream, 256 frames, all CUs: {base:9.0f} frames/s")
for cap in CAPS:
    for off in OFFS:
        v, res = dual(STEPS, cap, off)
        print(f"2 pipes x 128 frames, cu_cap {cap:3d}, offset {off:.2f}: {v:9.0f} frames/s  ({v / base - 1:+.1%})  equal={torch.equal(res, ref)}")
if len(sys.argv) > 4:                         # whole batches on two streams, every grid uncapped: launch tails of one fill with the other's heads
    for cap in CAPS:
        v, res = dual(STEPS, cap, 0.0, n_pipes=2, per=256)
        print(f"2 pipes x 256 frames, cu_cap {cap:3d}: {v:9.0f} frames/s  ({v / base - 1:+.1%})  equal={torch.equal(res, ref)}")
        v, res = dual(STEPS, cap, 0.5, n_pipes=2, per=256)
        print(f"2 pipes x 256 frames, cu_cap {cap:3d}, offset 0.5: {v:9.0f} frames/s  ({v / base - 1:+.1%})  equal={torch.equal(res, ref)}")
for cap in (() if len(sys.argv) > 4 else (64, 86)):
    v, res = dual(STEPS, cap, 0.5, n_pipes=4)
    print(f"4 pipes x  64 frames, cu_cap {cap:3d}, offset 0.50: {v:9.0f} frames/s  ({v / base - 1:+.1%})  equal={torch.equal(res, ref)}")
base2, _ = single(STEPS)
print(f"single stream again: {base2:9.0f} frames/s")
