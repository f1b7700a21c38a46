"""Weight sources of the backbone (src/preprocess_resnet_features.py:207-209: ``resnet50(weights=IMAGENET1K_V2)``, ``children()[:-1]``):
local torchvision-layout checkpoints through ``weights.load_state_dict_from_path`` / the CLI's ``--weights``, and the CLI's refusal
to run without a named source."""
import pytest
import torch

from implementation_phd_lab_vision_amd import weights as W


def _same_sd(a, b):
    assert sorted(a) == sorted(b)
    for k in a:
        assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("layout", ["plain", "module_prefix", "nested_state_dict", "nested_module_prefix"])
def test_torchvision_layout_checkpoint_round_trips(tmp_path, layout):
    """A checkpoint file as torchvision writes it (320 keys: fc.weight / fc.bias and num_batches_tracked included), optionally saved
    from a DataParallel wrapper (``module.`` prefix) and / or nested under ``state_dict``: the loader returns the 318 backbone
    tensors (``fc.*`` dropped, as ``children()[:-1]`` drops the layer), bit for bit."""
    sd = W.synthetic_state_dict(3)
    assert len(sd) == 318
    full = dict(sd)
    full["fc.weight"], full["fc.bias"] = torch.randn(1000, 2048), torch.randn(1000)
    assert len(full) == 320
    if "module" in layout:
        full = {"module." + k: v for k, v in full.items()}
    obj = {"state_dict": full, "epoch": 3} if "nested" in layout else full
    path = tmp_path / "resnet50-test.pth"
    torch.save(obj, path)
    got = W.load_state_dict_from_path(str(path))
    assert not any(k.startswith("fc.") or k.startswith("module.") for k in got)
    _same_sd(got, sd)
    assert sum(1 for _ in W.iter_named_tensors(got)) == 53 * 5


def test_checkpoint_with_missing_or_misshaped_tensors_is_rejected(tmp_path):
    sd = W.synthetic_state_dict(0)
    bad = dict(sd); del bad["layer3.4.bn2.running_var"]
    torch.save(bad, tmp_path / "a.pth")
    with pytest.raises(ValueError, match="layer3.4.bn2.running_var"):
        W.load_state_dict_from_path(str(tmp_path / "a.pth"))
    bad = dict(sd); bad["layer2.0.downsample.0.weight"] = torch.zeros(512, 256, 3, 3)
    torch.save(bad, tmp_path / "b.pth")
    with pytest.raises(ValueError, match="layer2.0.downsample.0.weight"):
        W.load_state_dict_from_path(str(tmp_path / "b.pth"))


def test_cli_requires_an_explicit_weight_source(tmp_path):
    """Without --weights / --synthetic-weights the CLI refuses to run (it would silently write features of a random network);
    the check comes before any device work, so it needs no GPU."""
    from implementation_phd_lab_vision_amd.preprocess_resnet_features import _resolve_weights, build_parser, main, weights_digest
    base = ["--root", "unused", "--out", str(tmp_path / "o"), "--synthetic-clips", "2", "--num-workers", "0"]
    with pytest.raises(SystemExit, match="no backbone weights"):
        main(base)
    with pytest.raises(SystemExit, match="mutually exclusive"):
        main(base + ["--weights", "x.pth", "--synthetic-weights"])
    sd, source = _resolve_weights(build_parser().parse_args(base + ["--synthetic-weights", "--weights-seed", "5"]))
    _same_sd(sd, W.synthetic_state_dict(5))
    assert "seed 5" in source
    assert weights_digest(sd) == weights_digest(W.synthetic_state_dict(5)) != weights_digest(W.synthetic_state_dict(6))
    torch.save({"module." + k: v for k, v in sd.items()}, tmp_path / "ck.pth")
    sd2, source2 = _resolve_weights(build_parser().parse_args(base + ["--weights", str(tmp_path / "ck.pth")]))
    _same_sd(sd2, sd)
    assert "ck.pth" in source2
