"""CPU restatement of the ColorJitter augmentation variant.  TEST INFRASTRUCTURE ONLY: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Reference: ``_aug_color_jitter`` (src/dataset.py:188-197) = ``torchvision.transforms.v2.ColorJitter(brightness=0.3, contrast=0.3,
saturation=0.2, hue=0.05)`` applied to the float clip (T,3,H,W) in [0,1] (:416), then ``frame_tf`` = Normalize (:242-245,417).
The arithmetic lives in a third-party dependency -- torchvision, version not pinned by the reference, NOT installed in this image --
and the reference holds no fixtures for it, so this is a restatement of torchvision's published v2 float kernels
(``transforms/v2/functional/_color.py``: ``adjust_brightness_image``, ``adjust_contrast_image``, ``adjust_saturation_image``,
``adjust_hue_image``, ``_rgb_to_grayscale_image``, ``_blend``, ``_rgb_to_hsv``, ``_hsv_to_rgb``; ``ColorJitter.make_params`` /
``transform`` in ``transforms/v2/_color.py``) written with the same torch ops in the same order: PARITY UNPINNED.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

BRIGHTNESS, CONTRAST, SATURATION, HUE = (0.7, 1.3), (0.7, 1.3), (0.8, 1.2), (-0.05, 0.05)     # ColorJitter(0.3, 0.3, 0.2, 0.05)
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def sample_params(generator: Optional[torch.Generator] = None) -> Dict:
    """``ColorJitter.make_params``: the op order first, then the four factors, each one ``torch.empty(1).uniform_(lo, hi)``."""
    fn_idx = torch.randperm(4, generator=generator)
    val = lambda lo, hi: float(torch.empty(1).uniform_(lo, hi, generator=generator))
    return {"fn_idx": [int(v) for v in fn_idx], "brightness": val(*BRIGHTNESS), "contrast": val(*CONTRAST), "saturation": val(*SATURATION),
            "hue": val(*HUE)}


def _gray(img: torch.Tensor) -> torch.Tensor:
    r, g, b = img.unbind(dim=-3)
    return r.mul(0.2989).add_(g, alpha=0.587).add_(b, alpha=0.114).unsqueeze(dim=-3)


def _blend(a: torch.Tensor, b: torch.Tensor, ratio: float) -> torch.Tensor:
    return a.mul(ratio).add_(b, alpha=(1.0 - ratio)).clamp_(0, 1.0)


def adjust_brightness(img, f):
    return img.mul(f).clamp_(0, 1.0)


def adjust_contrast(img, f):
    mean = torch.mean(_gray(img), dim=(-3, -2, -1), keepdim=True)
    return _blend(img, mean, f)


def adjust_saturation(img, f):
    return _blend(img, _gray(img), f)


def _rgb_to_hsv(image):
    r, g, _ = image.unbind(dim=-3)
    minc, maxc = torch.aminmax(image, dim=-3)
    eqc = maxc == minc
    channels_range = maxc - minc
    ones = torch.ones_like(maxc)
    s = channels_range / torch.where(eqc, ones, maxc)
    divisor = torch.where(eqc, ones, channels_range).unsqueeze_(dim=-3)
    rc, gc, bc = ((maxc.unsqueeze(dim=-3) - image) / divisor).unbind(dim=-3)
    mask_maxc_neq_r = maxc != r
    mask_maxc_eq_g = maxc == g
    hg = rc.add(2.0).sub_(bc).mul_(mask_maxc_eq_g & mask_maxc_neq_r)
    hr = bc.sub_(gc).mul_(~mask_maxc_neq_r)
    hb = gc.add_(4.0).sub_(rc).mul_(mask_maxc_neq_r.logical_and_(mask_maxc_eq_g.logical_not_()))
    h = hr.add_(hg).add_(hb)
    h = h.mul_(1.0 / 6.0).add_(1.0).fmod_(1.0)
    return torch.stack((h, s, maxc), dim=-3)


def _hsv_to_rgb(img):
    h, s, v = img.unbind(dim=-3)
    h6 = h.mul(6)
    i = torch.floor(h6)
    f = h6.sub_(i)
    i = i.to(dtype=torch.int32)
    sxf = s * f
    one_minus_s = 1.0 - s
    q = (1.0 - sxf).mul_(v).clamp_(0.0, 1.0)
    t = sxf.add_(one_minus_s).mul_(v).clamp_(0.0, 1.0)
    p = one_minus_s.mul_(v).clamp_(0.0, 1.0)
    i.remainder_(6)
    vpqt = torch.stack((v, p, q, t), dim=-3)
    select = torch.tensor([[0, 2, 1, 1, 3, 0], [3, 0, 0, 2, 1, 1], [1, 1, 3, 0, 0, 2]], dtype=torch.long)
    select = select[:, i.long()]                                   # (3, ..., H, W): which of (v, p, q, t) each channel takes
    if select.ndim > 3:
        select = select.moveaxis(0, -3)
    return vpqt.gather(-3, select)


def adjust_hue(img, f):
    if f == 0:
        return img
    hsv = _rgb_to_hsv(img)
    h, s, v = hsv.unbind(dim=-3)
    h = h.add(f).remainder_(1.0)
    return _hsv_to_rgb(torch.stack((h, s, v), dim=-3))


def color_jitter(video01: torch.Tensor, params: Dict) -> torch.Tensor:
    """``ColorJitter.transform``: video01 (T,3,H,W) fp32 in [0,1]; ops in ``fn_idx`` order; one parameter set for the whole clip."""
    out = video01
    for fn_id in params["fn_idx"]:
        if fn_id == 0:
            out = adjust_brightness(out, params["brightness"])
        elif fn_id == 1:
            out = adjust_contrast(out, params["contrast"])
        elif fn_id == 2:
            out = adjust_saturation(out, params["saturation"])
        else:
            out = adjust_hue(out, params["hue"])
    return out


def normalize(video01: torch.Tensor) -> torch.Tensor:
    """``v2.Normalize(mean, std)`` (src/dataset.py:242-245)."""
    mean = torch.tensor(MEAN, dtype=video01.dtype).view(-1, 1, 1)
    std = torch.tensor(STD, dtype=video01.dtype).view(-1, 1, 1)
    return (video01 - mean) / std


def color_jitter_variant_u8(frames_u8: torch.Tensor, params: Dict) -> torch.Tensor:
    """uint8 resized crops (T,3,H,W) -> the normalized fp32 frames of the cjitter variant (:148-149,416-417)."""
    return normalize(color_jitter(frames_u8.to(torch.float32) / 255.0, params))
