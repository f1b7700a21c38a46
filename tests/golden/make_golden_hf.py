"""Cross-check of the oracle against an INDEPENDENT implementation of the same architecture.

torchvision -- the module the reference delegates the arithmetic to (src/preprocess_resnet_features.py:11,207-208) -- is not installed in
this image, so feature values stay "parity unpinned".  What IS installed is `transformers`, whose `ResNetModel` (bottleneck, depths
[3,4,6,3], hidden sizes [256,512,1024,2048], `downsample_in_bottleneck=False` = ResNet v1.5: the stride on the 3x3 conv) is a separate
implementation of the network torchvision's `resnet50().children()[:-1]` computes: 7x7 s2 conv + BN + ReLU, MaxPool(3,2,1), the four
bottleneck stages, AdaptiveAvgPool2d((1,1)).  This script builds it in the build container, copies the seeded synthetic state dicts of
`weights.synthetic_state_dict` (torchvision key layout) onto it -- conv / BN keys map 1:1 -- and writes its `pooler_output` for 8 seeded
frames per weight family to `tests/golden/hf_resnet50_features.pt`.  `tests/test_oracle_cpu.py` asserts that
`oracle.resnet50_oracle.forward_reference` reproduces it: an independent implementation agreeing removes the risk that the oracle and the
kernels share one restatement error (it does NOT lift "parity unpinned": it is neither the reference nor torchvision).

Only the fixture travels (inputs are regenerated from their seed).  usage: python tests/golden/make_golden_hf.py
"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

from implementation_phd_lab_vision_amd.weights import synthetic_frames, synthetic_state_dict  # noqa: E402

N_FRAMES, FRAME_SEED = 8, 1234


def tv_to_hf_key(k: str):
    """torchvision ResNet state-dict key -> transformers.ResNetModel key (None: not part of the backbone)."""
    if k.startswith("fc."):
        return None
    parts = k.split(".")
    if parts[0] == "conv1":
        return "embedder.embedder.convolution." + parts[1]
    if parts[0] == "bn1":
        return "embedder.embedder.normalization." + parts[1]
    stage = int(parts[0][len("layer"):]) - 1
    blk = parts[1]
    base = f"encoder.stages.{stage}.layers.{blk}."
    if parts[2] == "downsample":                       # downsample.0 = conv, downsample.1 = bn
        return base + "shortcut." + ("convolution." if parts[3] == "0" else "normalization.") + parts[4]
    idx = int(parts[2][-1]) - 1                        # conv1 / bn1 -> layer.0, conv2 / bn2 -> layer.1, conv3 / bn3 -> layer.2
    kind = "convolution." if parts[2].startswith("conv") else "normalization."
    return base + f"layer.{idx}." + kind + parts[3]


def build_hf_model(sd_tv):
    from transformers import ResNetConfig, ResNetModel
    cfg = ResNetConfig(num_channels=3, embedding_size=64, hidden_sizes=[256, 512, 1024, 2048], depths=[3, 4, 6, 3], layer_type="bottleneck",
                       hidden_act="relu", downsample_in_first_stage=False, downsample_in_bottleneck=False)
    model = ResNetModel(cfg).eval()
    mapped = {}
    for k, v in sd_tv.items():
        hk = tv_to_hf_key(k)
        if hk is not None:
            mapped[hk] = v
    missing, unexpected = model.load_state_dict(mapped, strict=True), None
    assert not missing.missing_keys and not missing.unexpected_keys
    return model


def main() -> None:
    import transformers
    out = {"frames_seed": FRAME_SEED, "n_frames": N_FRAMES, "transformers_version": transformers.__version__, "torch_version": str(torch.__version__)}
    x = synthetic_frames(N_FRAMES, seed=FRAME_SEED)
    with torch.no_grad():
        for fam in ("uniform", "trained"):
            sd = synthetic_state_dict(0) if fam == "uniform" else synthetic_state_dict(0, family="trained")
            model = build_hf_model(sd)
            out[fam] = model(x).pooler_output.flatten(1).to(torch.float32).clone()
            print(fam, tuple(out[fam].shape), float(out[fam].abs().mean()))
    dst = Path(__file__).resolve().parent / "hf_resnet50_features.pt"
    torch.save(out, dst)
    print("wrote", dst, dst.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
