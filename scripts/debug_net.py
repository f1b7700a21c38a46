import sys, torch
sys.path.insert(0, '.')
import torch.nn.functional as F
from implementation_phd_lab_vision_amd import _lib
from implementation_phd_lab_vision_amd.backbone import ResNet50Backbone
from implementation_phd_lab_vision_amd.weights import synthetic_frames, synthetic_state_dict
from oracle import resnet50_oracle as O
_lib.build_library()
sd = synthetic_state_dict(0)
x = synthetic_frames(2, seed=1234).to('cuda:0')
bb = ResNet50Backbone(state_dict=sd, max_batch=8).to('cuda:0').eval()
nchw = lambda t: t.float().cpu().permute(0, 3, 1, 2).contiguous()
x_in = nchw(bb.layer(x, 'layer1.2'))
ds = nchw(bb.layer(x, 'layer2.0.ds'))
w, b = O.folded(sd, 'layer2.0.downsample.0', 'layer2.0.downsample.1')
wq = O.bf16_round(w).double()
pre = F.conv2d(x_in.double(), wq, stride=2) + b.double().view(1, -1, 1, 1)
ref = pre.float().to(torch.bfloat16).float()
diff = (ds - ref).abs()
tol = ref.abs() * 2.0**-7 + 2.0**-16 * max(1.0, float(ref.abs().max()))
bad = diff > tol
idx = bad.nonzero()
print('nbad', len(idx), 'max|ref|', ref.abs().max().item())
import collections
print('channels', collections.Counter(idx[:, 1].tolist()).most_common(10))
print('rows', collections.Counter(idx[:, 2].tolist()).most_common(10))
print('cols', collections.Counter(idx[:, 3].tolist()).most_common(10))
for i in idx[:12]:
    i = tuple(i.tolist())
    print(i, 'got', ds[i].item(), 'ref', ref[i].item(), 'pre', pre[i].item())
wd_, bd_ = bb.packed_params('layer2.0.downsample.0')
wt_ = O.bf16_round(w).permute(0, 2, 3, 1).contiguous()
mm = (wd_.float() != wt_).nonzero()
print('packed weight mismatches vs torch fold:', len(mm), mm[:10].tolist())
print('bias mismatch:', (bd_ != b).nonzero().flatten().tolist())
for i in mm[:10]:
    i = tuple(i.tolist()); print(i, wd_[i].item(), wt_[i].item(), w[i[0], i[3], 0, 0].item())
