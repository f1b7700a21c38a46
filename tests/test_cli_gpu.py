"""End-to-end on the MI355X: the CLI (14 reference flags) with synthetic clips -> shards + index, and the
features inside the shards equal what the HIP backbone returns for the same frames (bit-exact: same
kernels, same batch-split invariance)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_cli_end_to_end(tmp_path, lib_built):
    from implementation_phd_lab_vision_amd.backbone import ResNet50Backbone
    from implementation_phd_lab_vision_amd.preprocess_resnet_features import main
    from implementation_phd_lab_vision_amd.synthetic import SyntheticClips
    out = tmp_path / "cache"
    main(["--root", "unused", "--out", str(out), "--synthetic-clips", "7", "--seq-len", "4", "--batch-size", "3",
          "--num-workers", "0", "--shard-size", "4", "--shuffle-pool", "5", "--shuffle-seed", "9", "--device", "cuda",
          "--max-batch", "16", "--synthetic-weights"])
    idx = torch.load(out / "index.pt", weights_only=True)
    assert idx["n_clips"] == 7 and idx["n_shards"] == 2 and idx["feat_dtype"] == "float32" and idx["seq_len"] == 4
    ds = SyntheticClips(7, seq_len=4, subjects=(1, 5, 6, 7, 8, 9, 11), augment=False, stride=5)
    bb = ResNet50Backbone(seed=0, max_batch=16).to("cuda:0").eval()
    seen = set()
    for rec in idx["clips"]:
        shard = torch.load(out / f"shard_{rec['shard_id']:05d}.pt", weights_only=True)
        ci = next(i for i, c in enumerate(ds.index) if (c.subject, c.action, c.cam, c.start) ==
                  (rec["subject"], rec["action"], rec["cam"], rec["start"]) and i not in seen)
        seen.add(ci)
        video, j3d, j2d, k, box = ds[ci]
        feats = bb(video.to("cuda:0")).flatten(1).cpu()
        assert shard["feats"][rec["row"]].dtype == torch.float32
        assert torch.equal(shard["feats"][rec["row"]], feats), f"clip {ci}: shard features differ from the backbone's"
        assert torch.equal(shard["joints3d"][rec["row"]], j3d) and torch.equal(shard["K"][rec["row"]], k)
        assert torch.equal(shard["meta"][rec["row"]]["box"], box)
    assert len(seen) == 7


def test_cli_augment_trev_reuse_gives_identical_files(tmp_path, lib_built):
    """--augment on the MI355X: the shards written with the temporal-reverse shortcut (variant 3 = variant 0's features in
    reverse frame order, 3 forward passes) equal, tensor for tensor, the shards written with all 4 forward passes."""
    from implementation_phd_lab_vision_amd.preprocess_resnet_features import main
    common = ["--root", "unused", "--synthetic-clips", "5", "--seq-len", "3", "--batch-size", "2", "--num-workers", "0",
              "--shard-size", "3", "--shuffle-pool", "4", "--shuffle-seed", "3", "--device", "cuda", "--max-batch", "16", "--augment", "--synthetic-weights"]
    a, b = tmp_path / "reuse", tmp_path / "full"
    main(common + ["--out", str(a)])
    main(common + ["--out", str(b), "--no-trev-reuse"])
    ia, ib = torch.load(a / "index.pt", weights_only=True), torch.load(b / "index.pt", weights_only=True)
    assert ia["n_variants"] == 4 and ia["clips"] == ib["clips"] and ia["n_shards"] == ib["n_shards"]
    for sid in range(ia["n_shards"]):
        sa = torch.load(a / f"shard_{sid:05d}.pt", weights_only=True)
        sb = torch.load(b / f"shard_{sid:05d}.pt", weights_only=True)
        for key in ("feats", "joints3d", "joints2d", "K"):
            assert torch.equal(sa[key], sb[key]), (sid, key)
        assert [m["aug"] for m in sa["meta"]] == [m["aug"] for m in sb["meta"]]


def test_cli_with_a_checkpoint_file(tmp_path, lib_built):
    """--weights PATH: a torchvision-layout checkpoint (with fc.*, num_batches_tracked, a `module.` prefix) written to disk
    gives the same shards as the same weights handed over in memory, and its provenance is recorded beside the shards."""
    import json
    from implementation_phd_lab_vision_amd.preprocess_resnet_features import main, weights_digest
    from implementation_phd_lab_vision_amd.weights import synthetic_state_dict
    sd = synthetic_state_dict(7)
    ckpt = {"module." + k: v for k, v in sd.items()}
    ckpt["module.fc.weight"], ckpt["module.fc.bias"] = torch.zeros(1000, 2048), torch.zeros(1000)
    path = tmp_path / "resnet50-test.pth"
    torch.save({"state_dict": ckpt}, path)
    common = ["--root", "unused", "--synthetic-clips", "3", "--seq-len", "2", "--batch-size", "2", "--num-workers", "0",
              "--shard-size", "2", "--shuffle-pool", "2", "--device", "cuda", "--max-batch", "8"]
    a, b = tmp_path / "file", tmp_path / "seed"
    main(common + ["--out", str(a), "--weights", str(path)])
    main(common + ["--out", str(b), "--synthetic-weights", "--weights-seed", "7"])
    for sid in range(2):
        sa = torch.load(a / f"shard_{sid:05d}.pt", weights_only=True)
        sb = torch.load(b / f"shard_{sid:05d}.pt", weights_only=True)
        assert torch.equal(sa["feats"], sb["feats"])
    prov = json.loads((a / "backbone_weights.json").read_text())
    assert prov["sha256"] == weights_digest(sd) and prov["synthetic"] is False and str(path) in prov["source"]
