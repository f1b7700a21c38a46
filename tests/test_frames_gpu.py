"""Frame producer on the MI355X: r50_op_crop_resize_u8 against torch's CPU kernels (the functions the
reference's torchvision call ends up in) and the oracle's restatements; decoded clip -> features end to end."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [  # frame h, w, box top, left, side
    (300, 280, 10, 20, 231), (1000, 1002, 317, 402, 520), (1002, 1000, 0, 0, 1000), (480, 640, 100, 300, 100),
    (224, 224, 0, 0, 224), (240, 250, 3, 5, 225), (64, 64, 10, 11, 1), (500, 500, 26, 31, 449), (1000, 1002, 776, 778, 224),
]


@pytest.fixture(scope="module")
def lib_built():
    from implementation_phd_lab_vision_amd import _lib
    _lib.build_library()
    return _lib.load_library()


def _frames(h, w, t, seed):
    g = torch.Generator().manual_seed(seed)
    fr = torch.randint(0, 256, (t, h, w, 3), generator=g, dtype=torch.uint8)
    fr[0, :, :, 0] = 255          # saturated and zero planes: clamping / rounding at the ends of the range
    fr[0, :, :, 1] = 0
    return fr


@pytest.mark.parametrize("case", CASES, ids=lambda c: "%dx%d_box%d_%d_%d" % c)
def test_fixed_point_mode_is_bit_exact(lib_built, case):
    from implementation_phd_lab_vision_amd import frames
    from oracle import resize_oracle as ro
    h, w, top, left, side = case
    fr = _frames(h, w, 3, h * 7 + side)
    got = frames.crop_and_resize_video_uint8(fr.to("cuda:0"), [top, left, side, side], mode=frames.RESIZE_FIXED).cpu()
    assert got.shape == (3, 3, 224, 224) and got.dtype == torch.uint8
    x = fr.permute(0, 3, 1, 2)[:, :, top:top + side, left:left + side]
    assert np.array_equal(got.numpy(), ro.crop_and_resize_video_uint8(fr.numpy(), [top, left, side, side], fixed_point=True))
    if "AVX" in torch.backends.cpu.get_cpu_capability():      # ATen's own uint8 kernel, when this host has it
        ref = F.interpolate(x, size=[224, 224], mode="bilinear", align_corners=False, antialias=False)
        assert torch.equal(got, ref)


@pytest.mark.parametrize("case", CASES, ids=lambda c: "%dx%d_box%d_%d_%d" % c)
def test_float_mode_matches_reference_route(lib_built, case):
    """Default mode = the reference's arithmetic (torchvision v1 resize on uint8: fp32, interpolate, round).  Bit-exact
    against the oracle's restatement of the same operation order; against torch's CPU kernel at most 1 LSB on at most
    1e-4 of the bytes (FMA contraction of the CPU build decides exact rounding ties)."""
    from implementation_phd_lab_vision_amd import frames
    from oracle import resize_oracle as ro
    h, w, top, left, side = case
    fr = _frames(h, w, 2, h * 11 + side)
    got = frames.crop_and_resize_video_uint8(fr.to("cuda:0"), [top, left, side, side]).cpu().numpy()
    x = np.transpose(fr.numpy(), (0, 3, 1, 2))[:, :, top:top + side, left:left + side]
    assert np.array_equal(got, ro.resize_bilinear_u8_float_restated(x, 224, 224))
    ref = ro.crop_and_resize_video_uint8(fr.numpy(), [top, left, side, side])            # torch itself
    d = np.abs(got.astype(np.int32) - ref.astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() <= 1e-4


@pytest.mark.parametrize("mode", [0, 1], ids=["float", "fixed"])
def test_hflip_and_trev_are_flips_of_the_plain_output(lib_built, mode):
    """The augmentation variants that are index permutations (src/dataset.py:158-166,199-207): the kernel's flags must
    give exactly torch.flip of its own plain output along W / T."""
    from implementation_phd_lab_vision_amd import frames
    fr = _frames(260, 300, 5, 99).to("cuda:0")
    box = [13, 40, 211, 211]
    plain = frames.crop_and_resize_video_uint8(fr, box, mode=mode)
    assert torch.equal(frames.crop_and_resize_video_uint8(fr, box, mode=mode, hflip=True), torch.flip(plain, dims=[-1]))
    assert torch.equal(frames.crop_and_resize_video_uint8(fr, box, mode=mode, trev=True), torch.flip(plain, dims=[0]))
    assert torch.equal(frames.crop_and_resize_video_uint8(fr, box, mode=mode, hflip=True, trev=True),
                       torch.flip(plain, dims=[0, -1]))


def test_box_outside_frame_is_refused(lib_built):
    from implementation_phd_lab_vision_amd import frames, _lib
    fr = torch.zeros((1, 32, 32, 3), dtype=torch.uint8, device="cuda:0")
    with pytest.raises(_lib.R50Error):
        frames.crop_and_resize_video_uint8(fr, [10, 10, 30, 30])
    with pytest.raises(ValueError):
        frames.crop_and_resize_video_uint8(fr.permute(0, 2, 1, 3), [0, 0, 8, 8])       # not contiguous


def test_decoded_clip_to_features(lib_built):
    """features_from_video(frames, box) == features_u8(crops resized by the same kernel), and within the bf16 path's
    usual distance of the features of crops resized by torch on the CPU (a 1-LSB input tie moves features by ~1e-4)."""
    from implementation_phd_lab_vision_amd import frames
    from implementation_phd_lab_vision_amd.backbone import ResNet50Backbone
    from oracle import resize_oracle as ro
    from oracle.resnet50_oracle import rel_l2
    g = torch.Generator().manual_seed(3)
    fr = torch.randint(0, 256, (4, 300, 320, 3), generator=g, dtype=torch.uint8)
    j2d = torch.rand((4, 17, 2), generator=g) * 120 + 90
    box = frames.square_crop_from_2d(j2d, 300, 320)
    bb = ResNet50Backbone(seed=0, max_batch=8).to("cuda:0").eval()
    a = bb.features_from_video(fr.to("cuda:0"), box).cpu()
    crops = frames.crop_and_resize_video_uint8(fr.to("cuda:0"), box)
    assert torch.equal(a, bb.features_u8(crops).cpu())
    cpu_crops = torch.from_numpy(ro.crop_and_resize_video_uint8(fr.numpy(), box.tolist()))
    b = bb.features_u8(cpu_crops.to("cuda:0")).cpu()
    assert torch.isfinite(a).all() and rel_l2(a, b) < 1e-3
