"""Frame producer, CPU side: the oracle's restatements of the two resize routes against torch
itself, and frames.py's host functions against vectors produced by the reference's own functions."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import resize_oracle as ro

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "producer_golden.pt")
SIZES = [(231, 231), (500, 500), (100, 100), (224, 224), (225, 223), (1, 1), (2, 3), (448, 448), (449, 449), (1000, 1002),
         (333, 333), (223, 223)]


@pytest.mark.parametrize("hw", SIZES, ids=lambda v: f"{v[0]}x{v[1]}")
def test_fixed_point_restatement_equals_aten_uint8_kernel(hw):
    """ATen's native uint8 bilinear (the route torchvision v2 takes on an AVX2 CPU) is integer arithmetic: the numpy
    restatement must reproduce torch bit for bit, up- and down-scaling, identity, one-pixel sources."""
    if "AVX" not in torch.backends.cpu.get_cpu_capability():
        pytest.skip("torch has no native uint8 bilinear kernel on this CPU (it would convert to float itself)")
    h, w = hw
    g = torch.Generator().manual_seed(h * 1009 + w)
    x = torch.randint(0, 256, (2, 3, h, w), generator=g, dtype=torch.uint8)
    ref = F.interpolate(x, size=[224, 224], mode="bilinear", align_corners=False, antialias=False).numpy()
    assert np.array_equal(ro.resize_bilinear_u8(x.numpy(), 224, 224), ref)


@pytest.mark.parametrize("hw", SIZES, ids=lambda v: f"{v[0]}x{v[1]}")
def test_float_restatement_close_to_torch_float_route(hw):
    """The v1 API the reference imports: uint8 -> fp32 -> interpolate -> round.  The restatement (the HIP kernel's
    operation order) may differ from torch's CPU kernel only by 1 LSB at rounding ties (FMA contraction of the CPU
    build): at most 1e-4 of the bytes."""
    h, w = hw
    g = torch.Generator().manual_seed(h * 1013 + w)
    x = torch.randint(0, 256, (2, 3, h, w), generator=g, dtype=torch.uint8).numpy()
    a, b = ro.resize_bilinear_u8_float(x, 224, 224), ro.resize_bilinear_u8_float_restated(x, 224, 224)
    d = np.abs(a.astype(np.int32) - b.astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() <= 1e-4


def test_crop_and_resize_matches_reference_expression():
    """oracle.crop_and_resize_video_uint8 == the reference's statement sequence (src/dataset.py:141-149) written with
    torch: permute, slice, float, interpolate, round."""
    g = torch.Generator().manual_seed(5)
    fr = torch.randint(0, 256, (3, 120, 150, 3), generator=g, dtype=torch.uint8)
    top, left, side = 7, 20, 97
    x = fr.permute(0, 3, 1, 2)[:, :, top:top + side, left:left + side]
    ref = torch.round(F.interpolate(x.to(torch.float32), size=[224, 224], mode="bilinear", align_corners=False,
                                    antialias=False)).to(torch.uint8).numpy()
    got = ro.crop_and_resize_video_uint8(fr.numpy(), [top, left, side, side])
    d = np.abs(ref.astype(np.int32) - got.astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() <= 1e-4        # memory format picks a different CPU kernel: ties only


def test_host_functions_match_reference_goldens():
    from implementation_phd_lab_vision_amd import frames
    cases = torch.load(GOLDEN, weights_only=True)
    assert len(cases) >= 20
    for c in cases:
        box = frames.square_crop_from_2d(c["joints2d"], c["img_h"], c["img_w"])
        assert box.dtype == torch.int64 and torch.equal(box, c["box"])
        assert torch.equal(frames.adjust_joints2d_after_crop_and_resize(c["joints2d"], box), c["joints2d_adjusted"])
        k = frames.adjust_camera_after_crop_and_resize({"f": c["cam_f"].numpy(), "c": c["cam_c"].numpy()}, box)
        assert k.dtype == torch.float32 and torch.equal(k, c["K"])
        j3f, j2f, kf = frames.aug_hflip_annotations(c["joints3d"], c["joints2d_adjusted"], k)
        assert torch.equal(j3f, c["hflip_j3d"]) and torch.equal(j2f, c["hflip_j2d"]) and torch.equal(kf, c["hflip_K"])
        j3r, j2r = frames.aug_temporal_reverse_annotations(c["joints3d"], c["joints2d_adjusted"])
        assert torch.equal(j3r, c["trev_j3d"]) and torch.equal(j2r, c["trev_j2d"])


def test_device_op_refuses_cpu_tensors():
    from implementation_phd_lab_vision_amd import frames
    with pytest.raises(ValueError):
        frames.crop_and_resize_video_uint8(torch.zeros((1, 8, 8, 3), dtype=torch.uint8), [0, 0, 8, 8])
