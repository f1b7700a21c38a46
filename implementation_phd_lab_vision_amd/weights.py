"""ResNet-50 backbone weights: torchvision state-dict key layout, seeded synthetic values,
and loading of a *local* checkpoint file.

The reference builds its backbone with
``models.resnet50(weights=ResNet50_Weights.IMAGENET1K_V2)`` and drops ``fc``
(/root/reference/src/preprocess_resnet_features.py:207-209).  That call downloads a checkpoint;
there is no network here, so this module accepts weights **only by local path** and otherwise
produces deterministic closed-form synthetic weights (same values on every box for a given seed).

Key layout (upstream torchvision ``models/resnet.py``; 318 tensors once ``fc.*`` is dropped):
    conv1.weight, bn1.{weight,bias,running_mean,running_var,num_batches_tracked},
    layer{1..4}.{b}.conv{1,2,3}.weight, layer{1..4}.{b}.bn{1,2,3}.*,
    layer{1..4}.0.downsample.0.weight, layer{1..4}.0.downsample.1.*
"""
from __future__ import annotations

import math
from typing import Dict, Iterator, List, Tuple

import torch

# (planes, blocks, stride of the first block) per stage; Bottleneck expansion 4 (ResNet v1.5:
# the stride sits on the 3x3 conv).
STAGES: Tuple[Tuple[int, int, int], ...] = ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2))
EXPANSION = 4
BN_EPS = 1e-5
FEATURE_DIM = 2048


def conv_specs() -> List[Tuple[str, str, int, int, int, int, int]]:
    """Every conv of ``resnet50.children()[:-1]`` in execution order.

    Returns tuples ``(conv_key, bn_key, cin, cout, ksize, stride, pad)`` where ``conv_key`` /
    ``bn_key`` are state-dict prefixes (``conv_key + '.weight'`` etc.).
    """
    specs = [("conv1", "bn1", 3, 64, 7, 2, 3)]
    inplanes = 64
    for si, (planes, blocks, stride) in enumerate(STAGES, start=1):
        for b in range(blocks):
            s = stride if b == 0 else 1
            p = f"layer{si}.{b}"
            specs.append((f"{p}.conv1", f"{p}.bn1", inplanes, planes, 1, 1, 0))
            specs.append((f"{p}.conv2", f"{p}.bn2", planes, planes, 3, s, 1))
            specs.append((f"{p}.conv3", f"{p}.bn3", planes, planes * EXPANSION, 1, 1, 0))
            if b == 0:
                specs.append((f"{p}.downsample.0", f"{p}.downsample.1", inplanes,
                              planes * EXPANSION, 1, s, 0))
            inplanes = planes * EXPANSION
    return specs


def synthetic_state_dict(seed: int = 0, family: str = "uniform") -> Dict[str, torch.Tensor]:
    """Closed-form seeded weights (no calibration pass, so every box regenerates identical bits).

    ``family="uniform"`` (the default, the benchmark's weights): conv Kaiming-normal, fan-out; BN gamma~U(0.5,1.5) (last BN of
    each residual branch x0.25 to bound residual growth), beta~U(-0.2,0.2), running_mean~U(-0.1,0.1), running_var~U(0.5,1.5).

    ``family="trained"``: the statistics a TRAINED checkpoint shows and the uniform family never does -- gamma of either sign,
    a tenth of the channels with |gamma| ~ 1e-3 (pruned channels), running_var log-uniform over [1e-3, 10], running_mean~U(-0.5,0.5),
    beta~U(-0.5,0.5).  Each conv output channel is scaled by sqrt(running_var) (a trained net's pre-BN activations HAVE the
    variance its BN recorded), so activations stay O(1) while the folded per-channel scales gamma / sqrt(var + eps) span
    four orders of magnitude -- the range the 16-bit and fp8 paths must survive."""
    if family not in ("uniform", "trained"):
        raise ValueError("family must be 'uniform' or 'trained'")
    g = torch.Generator(device="cpu").manual_seed(int(seed))
    sd: Dict[str, torch.Tensor] = {}

    def uni(n: int, lo: float, hi: float) -> torch.Tensor:
        return torch.rand(n, generator=g, dtype=torch.float32) * (hi - lo) + lo

    for conv_key, bn_key, cin, cout, k, _s, _p in conv_specs():
        std = math.sqrt(2.0 / (cout * k * k))
        w = torch.randn(cout, cin, k, k, generator=g, dtype=torch.float32) * std
        if family == "uniform":
            gamma = uni(cout, 0.5, 1.5)
            beta, mean, var = uni(cout, -0.2, 0.2), uni(cout, -0.1, 0.1), uni(cout, 0.5, 1.5)
        else:
            gamma = uni(cout, 0.3, 1.5) * torch.where(torch.rand(cout, generator=g) < 0.35, -1.0, 1.0)
            gamma = torch.where(torch.rand(cout, generator=g) < 0.1, gamma * 1e-3, gamma)
            beta, mean = uni(cout, -0.5, 0.5), uni(cout, -0.5, 0.5)
            var = torch.pow(10.0, uni(cout, -3.0, 1.0))
            w = w * var.sqrt().view(-1, 1, 1, 1)
        if bn_key.endswith("bn3"):
            gamma = gamma * 0.25
        sd[conv_key + ".weight"] = w
        sd[bn_key + ".weight"] = gamma
        sd[bn_key + ".bias"] = beta
        sd[bn_key + ".running_mean"] = mean
        sd[bn_key + ".running_var"] = var
        sd[bn_key + ".num_batches_tracked"] = torch.zeros((), dtype=torch.int64)
    return sd


def load_state_dict_from_path(path: str) -> Dict[str, torch.Tensor]:
    """Load a local torchvision-format ResNet-50 checkpoint (``resnet50-*.pth``); ``fc.*`` is dropped."""
    sd = torch.load(path, map_location="cpu", weights_only=True)
    if "state_dict" in sd and isinstance(sd["state_dict"], dict):
        sd = sd["state_dict"]
    out = {}
    for k, v in sd.items():
        k = k[7:] if k.startswith("module.") else k
        if k.startswith("fc."):
            continue
        out[k] = v
    validate_state_dict(out)
    return out


def validate_state_dict(sd: Dict[str, torch.Tensor]) -> None:
    for conv_key, bn_key, cin, cout, k, _s, _p in conv_specs():
        w = sd.get(conv_key + ".weight")
        if w is None or tuple(w.shape) != (cout, cin, k, k):
            raise ValueError(f"state dict: {conv_key}.weight missing or wrong shape "
                             f"(want {(cout, cin, k, k)}, got {None if w is None else tuple(w.shape)})")
        for suffix in ("weight", "bias", "running_mean", "running_var"):
            t = sd.get(f"{bn_key}.{suffix}")
            if t is None or tuple(t.shape) != (cout,):
                raise ValueError(f"state dict: {bn_key}.{suffix} missing or wrong shape")


def iter_named_tensors(sd: Dict[str, torch.Tensor]) -> Iterator[Tuple[str, torch.Tensor]]:
    """The float tensors the C ABI's ``r50_load_weights`` consumes, contiguous fp32 on the host."""
    for conv_key, bn_key, *_ in conv_specs():
        yield conv_key + ".weight", sd[conv_key + ".weight"].detach().to(torch.float32).contiguous()
        for suffix in ("weight", "bias", "running_mean", "running_var"):
            yield f"{bn_key}.{suffix}", sd[f"{bn_key}.{suffix}"].detach().to(torch.float32).contiguous()


def synthetic_frames(n: int, seed: int = 1234) -> torch.Tensor:
    """Seeded frames shaped like the reference's loader output: uint8 -> /255 -> ImageNet normalise
    (/root/reference/src/dataset.py:242-245,429).  fp32 NCHW (n,3,224,224); never zero-filled."""
    g = torch.Generator(device="cpu").manual_seed(int(seed))
    u8 = torch.randint(0, 256, (n, 3, 224, 224), generator=g, dtype=torch.uint8)
    mean = torch.tensor([0.485, 0.456, 0.406], dtype=torch.float32).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225], dtype=torch.float32).view(1, 3, 1, 1)
    return ((u8.to(torch.float32) / 255.0) - mean) / std
