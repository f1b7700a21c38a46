"""The configurations bench.py measures, under test at their own sizes (BASELINE configs[1]: bf16, batch 256; configs[4]: fp8,
batch 512), a second weight family with the statistics of a trained checkpoint, and the BN fold the device holds.

At batch 256 / 512 the tuned large-batch tile table (`kTuned`), the persistent multi-tile streams, the chained layer3 tails
(512 tiles of 98 pixels) and the ragged last tiles all run as they do in the benchmark; the small-batch tests elsewhere never
reach them.  Properties checked: a frame's features are the same bits as when the frame runs alone (size-independent), and
sampled frames match the oracle's emulation tap by tap."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TAPS = ["stem", "pool", "layer1.0.t1", "layer1.0.t2", "layer1.0.ds", "layer1.0", "layer1.2", "layer2.0.t2",
        "layer2.0.ds", "layer2.0", "layer2.3", "layer3.0", "layer3.5", "layer4.0.t2", "layer4.0", "layer4.2"]


def test_bf16_batch256_is_the_same_bits_as_single_frames_and_matches_the_oracle(lib_built):
    from implementation_phd_lab_vision_amd.backbone import ResNet50Backbone
    from implementation_phd_lab_vision_amd.weights import synthetic_frames, synthetic_state_dict
    from oracle import resnet50_oracle as O
    sd = synthetic_state_dict(0)
    x = synthetic_frames(256, seed=1234)                   # the benchmark's frames
    bb = ResNet50Backbone(state_dict=sd, max_batch=256).to(DEV).eval()
    xd = x.to(DEV)
    big = bb.features(xd).clone()
    assert torch.isfinite(big).all()
    for i in (0, 97, 255):
        assert torch.equal(bb.features(xd[i:i + 1]), big[i:i + 1]), f"frame {i}: batch-256 features differ from the frame run alone"
    assert torch.equal(bb.features(xd[96:99]), big[96:99])
    # 16 named activations of 4 sampled frames, taken from the batch-256 run, against the emulation (tolerance: 3x the drift between
    # two CPU emulations that differ only in fp32-vs-fp64 accumulation, as in test_network_gpu.py)
    pick = [0, 97, 200, 255]
    xs = x[pick]
    taps, taps32 = {}, {}
    feats_emu = O.forward_bf16_emulated(sd, xs, taps=taps, fused_ds=True)
    O.forward_bf16_emulated(sd, xs, taps=taps32, acc_dtype=torch.float32, fused_ds=True)
    for name in TAPS:
        got = bb.layer(xd, name)[pick].float().cpu().permute(0, 3, 1, 2)
        ref = taps[name].float()
        assert got.shape == ref.shape, name
        r, drift = O.rel_l2(got, ref), O.rel_l2(taps32[name], taps[name])
        assert r < max(1e-4, 3.0 * drift), f"{name} at batch 256: rel-L2 {r} (emulation drift {drift})"
    assert float(O.per_row_rel_l2(big[pick].cpu(), feats_emu).max()) < 3e-3
    bb.close()


def test_fp8_batch512_is_the_same_bits_as_single_frames_and_matches_its_emulation(lib_built):
    from implementation_phd_lab_vision_amd.backbone import ResNet50Backbone
    from implementation_phd_lab_vision_amd.weights import synthetic_frames, synthetic_state_dict
    from oracle import resnet50_oracle as O
    sd = synthetic_state_dict(0)
    x = synthetic_frames(512, seed=1234)
    bb = ResNet50Backbone(state_dict=sd, max_batch=512, precision="fp8").to(DEV).eval()
    xd = x.to(DEV)
    big = bb.features(xd).clone()
    assert torch.isfinite(big).all()
    for i in (0, 97, 511):
        assert torch.equal(bb.features(xd[i:i + 1]), big[i:i + 1]), f"frame {i}: batch-512 fp8 features differ from the frame run alone"
    pick = [0, 97, 300, 511]
    emu = O.forward_fp8_emulated(sd, x[pick], bb.fp8_scales)
    ref = O.forward_reference(sd, x[pick]).flatten(1)
    got = big[pick].cpu()
    r_emu, r_ref, e_ref = O.per_row_rel_l2(got, emu), O.per_row_rel_l2(got, ref), O.per_row_rel_l2(emu, ref)
    assert float(r_emu.max()) < 2.5e-2, r_emu
    assert float(r_ref.max()) < 1.5 * float(e_ref.max()) + 1e-2, (r_ref, e_ref)
    bb.close()


@pytest.fixture(scope="module")
def trained_family(lib_built):
    from implementation_phd_lab_vision_amd.weights import synthetic_frames, synthetic_state_dict
    return synthetic_state_dict(0, family="trained"), synthetic_frames(2, seed=1234)


def test_trained_like_weights_every_conv_on_shared_inputs(trained_family):
    """Second weight family: gamma of either sign, a tenth of the channels nearly pruned (|gamma| ~ 1e-3), running_var over four
    decades -- what a trained checkpoint looks like and the benchmark's uniform family does not.  All 52 bottleneck convs, each on
    the device's own input activation, within one bf16 ulp of the oracle's fused-op emulation; stem and pool as well."""
    from implementation_phd_lab_vision_amd.backbone import ResNet50Backbone
    from oracle.resnet50_oracle import conv_bias_act_emulated, conv_cat_emulated, folded
    from tests.test_kernels_gpu import _check_bf16
    sd, x = trained_family
    g = torch.cat([sd[k].flatten() for k in sd if k.endswith("bn2.weight")])
    assert float((g < 0).float().mean()) > 0.2 and float((g.abs() < 5e-3).float().mean()) > 0.05       # it IS that family
    bb = ResNet50Backbone(state_dict=sd, max_batch=2).to(DEV).eval()
    xd = x.to(DEV)

    def dev(name):
        return bb.layer(xd, name)

    def nchw(t):
        return t.float().cpu().permute(0, 3, 1, 2).contiguous()

    w, b = folded(sd, "conv1", "bn1")
    from oracle.resnet50_oracle import elem_round
    _check_bf16(dev("stem"), conv_bias_act_emulated(elem_round(x), w, b, 2, 3, True), "stem")
    prev, n_checked = "pool", 0
    for si, (blocks, stride) in enumerate(((3, 1), (4, 2), (6, 2), (3, 2)), start=1):
        for bi in range(blocks):
            p = f"layer{si}.{bi}"
            s = stride if bi == 0 else 1
            x_in = nchw(dev(prev))
            t1, t2, out = dev(p + ".t1"), dev(p + ".t2"), dev(p)
            w, bias = folded(sd, p + ".conv1", p + ".bn1")
            _check_bf16(t1, conv_bias_act_emulated(x_in, w, bias, 1, 0, True), p + ".conv1")
            w, bias = folded(sd, p + ".conv2", p + ".bn2")
            _check_bf16(t2, conv_bias_act_emulated(nchw(t1), w, bias, s, 1, True), p + ".conv2")
            w, bias = folded(sd, p + ".conv3", p + ".bn3")
            if bi == 0:
                ds = dev(p + ".ds")
                wd, bd = folded(sd, p + ".downsample.0", p + ".downsample.1")
                _check_bf16(ds, conv_bias_act_emulated(x_in, wd, bd, s, 0, False), p + ".downsample")
                n_checked += 1
                if si >= 2:
                    _check_bf16(out, conv_cat_emulated(nchw(t2), w, bias, x_in, wd, bd, s), p + ".conv3+downsample")
                else:
                    _check_bf16(out, conv_bias_act_emulated(nchw(t2), w, bias, 1, 0, True, residual_bf=nchw(ds)), p + ".conv3")
            else:
                _check_bf16(out, conv_bias_act_emulated(nchw(t2), w, bias, 1, 0, True, residual_bf=x_in), p + ".conv3")
            n_checked += 3
            prev = p
    assert n_checked == 52
    bb.close()


def test_trained_like_weights_precisions_and_ranges(trained_family):
    """The same family through the 16-bit and fp8 modes end to end: bf16 / fp16 features against their emulations and the fp64
    view (fp16 must not saturate: the largest activation is checked against 65504), fp8 against its emulation with scales
    calibrated on this family (no tensor may clip: every calibrated scale covers the emulation's range)."""
    from implementation_phd_lab_vision_amd.backbone import ResNet50Backbone, fp8_tap_names
    from oracle import resnet50_oracle as O
    sd, x = trained_family
    xd = x.to(DEV)
    ref = O.forward_reference(sd, x, dtype=torch.float64).float().flatten(1)
    taps = {}
    emu = O.forward_bf16_emulated(sd, x, taps=taps, fused_ds=True)
    amax = max(float(t.abs().max()) for t in taps.values())
    assert amax < 65504 / 8, f"largest activation {amax}: too close to the fp16 range for a meaningful fp16 run"
    for prec, fmt, tol_emu, tol_ref in (("bf16", "bf16", 3e-3, 1e-2), ("fp16", "fp16", 4e-4, 1e-3)):
        bb = ResNet50Backbone(state_dict=sd, max_batch=2, precision=prec).to(DEV).eval()
        got = bb.features(xd).cpu()
        e = emu if fmt == "bf16" else O.forward_bf16_emulated(sd, x, fmt="fp16", fused_ds=True)
        assert torch.isfinite(got).all()
        r_emu, r_ref = O.per_row_rel_l2(got, e), O.per_row_rel_l2(got, ref)
        assert float(r_emu.max()) < tol_emu, f"{prec} vs its emulation: {r_emu.tolist()}"
        assert float(r_ref.max()) < tol_ref, f"{prec} vs the fp64 reference: {r_ref.tolist()}"
        bb.close()
    bb = ResNet50Backbone(state_dict=sd, max_batch=2, precision="fp8").to(DEV).eval()
    bb.calibrate_fp8(frames=xd)                               # scales from THIS family's activations (default: 8 noise frames)
    for name, s in zip(fp8_tap_names(), bb.fp8_scales):
        assert float(taps[name].abs().max()) <= 448.0 * s * 1.0001, f"{name}: calibrated scale {s} would clip"
    got = bb.features(xd).cpu()
    e8 = O.forward_fp8_emulated(sd, x, bb.fp8_scales)
    assert torch.isfinite(got).all()
    r_emu, r_ref, e_ref = O.per_row_rel_l2(got, e8), O.per_row_rel_l2(got, ref), O.per_row_rel_l2(e8, ref)
    assert float(r_emu.max()) < 3e-2, r_emu
    assert float(r_ref.max()) < 1.5 * float(e_ref.max()) + 1e-2, (r_ref, e_ref)
    bb.close()


@pytest.mark.parametrize("family", ["uniform", "trained"])
def test_bf16_bn_fold_on_the_device_equals_the_oracles(lib_built, family):
    """r50_load_weights folds BN in fp32 on the host (scale = gamma / sqrt(var + eps), w' = w * scale, b' = beta - mean * scale, each a
    single correctly rounded operation) and rounds to bf16; `r50_get_packed` hands the result back.  It must equal the oracle's fold
    (`oracle.fold_bn`, numpy float32 ops) BIT FOR BIT -- weights after bf16 rounding, biases in fp32 -- for 1x1, 3x3, strided and
    downsample convs, in both weight families (tiny variances and negative gammas included)."""
    from implementation_phd_lab_vision_amd.backbone import ResNet50Backbone
    from implementation_phd_lab_vision_amd.weights import synthetic_state_dict
    from oracle import resnet50_oracle as O
    sd = synthetic_state_dict(0, family=family)
    bb = ResNet50Backbone(state_dict=sd, max_batch=1).to(DEV).eval()
    for key, bn in (("layer1.0.conv1", "layer1.0.bn1"), ("layer1.1.conv2", "layer1.1.bn2"), ("layer2.0.conv2", "layer2.0.bn2"),
                    ("layer2.0.downsample.0", "layer2.0.downsample.1"), ("layer3.3.conv3", "layer3.3.bn3"), ("layer4.2.conv1", "layer4.2.bn1")):
        w, b = O.folded(sd, key, bn)                                     # (cout, cin, k, k) fp32, (cout) fp32
        got_w, got_b = bb.packed_params(key)                             # (cout, k, k, cin) bf16, (cout) fp32
        want_w = O.bf16_round(w).to(torch.bfloat16).permute(0, 2, 3, 1).contiguous()
        assert torch.equal(got_w.view(torch.int16), want_w.view(torch.int16)), f"{key}: folded bf16 weights differ"
        assert torch.equal(got_b, b.to(torch.float32)), f"{key}: folded bias differs"
    bb.close()


def test_max_batch_is_capped_per_precision(lib_built):
    """fp32x tensors carry [head | tail] pairs: twice the bytes per frame, so the 31-bit descriptor budget of the kernels ends at
    668 frames per chunk there (1337 in the 16-bit modes).  r50_create refuses what a forward pass could not run."""
    from implementation_phd_lab_vision_amd import _lib
    from implementation_phd_lab_vision_amd.backbone import ResNet50Backbone
    from implementation_phd_lab_vision_amd.weights import synthetic_frames
    with pytest.raises(_lib.R50Error, match="max_batch"):
        ResNet50Backbone(seed=0, max_batch=641, precision="fp32x").to(DEV)
    with pytest.raises(_lib.R50Error, match="max_batch"):
        ResNet50Backbone(seed=0, max_batch=1025, precision="bf16").to(DEV)
    bb = ResNet50Backbone(seed=0, max_batch=640, precision="fp32x").to(DEV).eval()
    x = synthetic_frames(3, seed=2).to(DEV)
    small = ResNet50Backbone(seed=0, max_batch=4, precision="fp32x").to(DEV).eval()
    assert torch.equal(bb.features(x), small.features(x))
    bb.close(); small.close()
