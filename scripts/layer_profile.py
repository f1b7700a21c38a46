"""In-network per-layer timing (HIP events around every launch, one stream).  usage: layer_profile.py [batch] [micro_batch] [steps]"""
import sys, torch
sys.path.insert(0, '.')
from implementation_phd_lab_vision_amd import _lib
from implementation_phd_lab_vision_amd.backbone import ResNet50Backbone
from implementation_phd_lab_vision_amd.weights import synthetic_frames
_lib.build_library()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
MB = int(sys.argv[2]) if len(sys.argv) > 2 else 0
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 10
bb = ResNet50Backbone(seed=0, max_batch=B, micro_batch=MB).to('cuda:0').eval()
bb.set_option('streams', 1)
x = synthetic_frames(B, seed=1).to('cuda:0')
for _ in range(3): bb.features(x)
bb.set_option('profile', 1); bb.profile_reset()
for _ in range(STEPS): bb.features(x)
torch.cuda.synchronize()
prof = bb.profile_collect()
tot = 0.0
for name, p in prof.items():
    if p['launches'] == 0: continue
    ms = p['ms'] / STEPS
    if name not in ('igemm',): tot += ms if name in ('conv1', 'maxpool', 'avgpool', 'stem_pack') else 0
    tf = p['flops'] / (p['ms'] * 1e-3) / 1e12 if p['flops'] else 0
    gb = p['bytes'] / (p['ms'] * 1e-3) / 1e9
    print(f"{name:24s} {ms*1e3:9.1f} us/step  {tf:7.1f} TF/s  {gb:7.0f} GB/s  launches/step {p['launches']/STEPS:.0f}")
for st in (1, 2, 3, 4):
    t = sum(p['ms'] for n, p in prof.items() if n.startswith(f'layer{st}.')) / STEPS
    print(f"layer{st} total {t*1e3:8.1f} us/step")
