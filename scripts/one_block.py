"""One launch sequence of the bottleneck-body kernels (for rocprofv3 --pmc runs).  usage: one_block.py [batch] [reps]"""
import sys, torch
sys.path.insert(0, '.')
from implementation_phd_lab_vision_amd import ops, _lib
_lib.load_library()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
d = torch.device('cuda:0'); g = torch.Generator().manual_seed(0)
rb = lambda shape, scale=1.0: (torch.randn(shape, generator=g) * scale).to(torch.bfloat16).to(d)
fb = lambda n: (torch.randn(n, generator=g) * 0.1).to(d)
# layer2 body
t1 = rb((B, 28, 28, 128)).clamp_(min=0); idn = rb((B, 28, 28, 512)).clamp_(min=0)
w2 = rb((128, 3, 3, 128), (2.0 / 1152) ** 0.5); w3 = rb((512, 128), (2.0 / 128) ** 0.5); w1 = rb((128, 512), (2.0 / 512) ** 0.5)
b2, b3, b1 = fb(128), fb(512), fb(128)
# layer1 body
u1 = rb((B, 56, 56, 64)).clamp_(min=0); udn = rb((B, 56, 56, 256)).clamp_(min=0)
v2 = rb((64, 3, 3, 64), (2.0 / 576) ** 0.5); v3 = rb((256, 64), (2.0 / 64) ** 0.5); v1 = rb((64, 256), (2.0 / 256) ** 0.5)
c2, c3, c1 = fb(64), fb(256), fb(64)
for _ in range(REPS):
    ops.bneck_block2_bf16(t1, w2, b2, w3, b3, idn, w1, b1)
    ops.bneck_block2_bf16(t1, w2, b2, w3, b3, idn)
    ops.bneck_block1_bf16(u1, v2, c2, v3, c3, udn, v1, c1)
torch.cuda.synchronize()
print("done")
