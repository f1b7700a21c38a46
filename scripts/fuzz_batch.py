"""Robustness sweep (run on the GPU box): a frame's features must be the same bits whatever the batch it is run in --
19 batch sizes from 2 to 300 (ragged tiles, the small-batch tile choice, chunking past max_batch), bf16 and fp16 --
and the uint8 boundary must not depend on how a batch is split."""
import sys, torch
sys.path.insert(0, '.')
from implementation_phd_lab_vision_amd.backbone import ResNet50Backbone
from implementation_phd_lab_vision_amd.weights import synthetic_frames
for prec in ("bf16", "fp16"):
    bb = ResNet50Backbone(seed=0, max_batch=256, precision=prec).to("cuda:0").eval()
    x = synthetic_frames(300, seed=3).to("cuda:0")
    base = torch.cat([bb.features(x[i:i + 1]).clone() for i in range(0, 300, 37)])   # single-frame results of a few frames
    idxs = list(range(0, 300, 37))
    bad = 0
    for n in (2, 3, 5, 13, 31, 47, 48, 63, 64, 65, 100, 127, 128, 129, 200, 255, 256, 257, 300):
        f = bb.features(x[:n])
        for j, i in enumerate(idxs):
            if i < n and not torch.equal(f[i], base[j]):
                bad += 1; print(prec, "MISMATCH n", n, "frame", i, float((f[i] - base[j]).abs().max()))
        assert torch.isfinite(f).all()
    print(prec, "batch-size sweep done, mismatches:", bad)
    u8 = torch.randint(0, 256, (70, 3, 224, 224), dtype=torch.uint8, device="cuda:0")
    a = bb.features_u8(u8); b = torch.cat([bb.features_u8(u8[:33]), bb.features_u8(u8[33:])])
    print(prec, "u8 split equal:", torch.equal(a, b))
