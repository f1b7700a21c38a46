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


def test_cli_augment_lanes_give_identical_files(tmp_path, lib_built):
    """--augment with two backbone lanes (the default: the variants of a batch run two at a time on their own streams) writes, tensor
    for tensor, the shards of the single-stream run."""
    from implementation_phd_lab_vision_amd.preprocess_resnet_features import main
    common = ["--root", "unused", "--synthetic-clips", "5", "--seq-len", "3", "--batch-size", "2", "--num-workers", "0",
              "--shard-size", "3", "--shuffle-pool", "4", "--shuffle-seed", "3", "--device", "cuda", "--max-batch", "16", "--augment", "--synthetic-weights",
              "--no-trev-reuse"]
    a, b = tmp_path / "lanes2", tmp_path / "lanes1"
    main(common + ["--out", str(a)])
    main(common + ["--out", str(b), "--lanes", "1"])
    ia, ib = torch.load(a / "index.pt", weights_only=True), torch.load(b / "index.pt", weights_only=True)
    assert ia["n_variants"] == 4 and ia["clips"] == ib["clips"] and ia["n_shards"] == ib["n_shards"]
    for sid in range(ia["n_shards"]):
        sa = torch.load(a / f"shard_{sid:05d}.pt", weights_only=True)
        sb = torch.load(b / f"shard_{sid:05d}.pt", weights_only=True)
        for key in ("feats", "joints3d", "joints2d", "K"):
            assert torch.equal(sa[key], sb[key]), (sid, key)


def test_cli_lanes_without_augment_give_identical_files(tmp_path, lib_built):
    """Without --augment a batch is ONE forward pass: run_extraction submits round q + 1 before it waits for round q, so consecutive batches
    share the two lanes.  Seven clips in batches of two (four rounds, a ragged last one): the shards equal the one-lane run's."""
    from implementation_phd_lab_vision_amd.preprocess_resnet_features import main
    common = ["--root", "unused", "--synthetic-clips", "7", "--seq-len", "3", "--batch-size", "2", "--num-workers", "0",
              "--shard-size", "3", "--shuffle-pool", "4", "--shuffle-seed", "5", "--device", "cuda", "--max-batch", "16", "--synthetic-weights"]
    a, b = tmp_path / "lanes2", tmp_path / "lanes1"
    main(common + ["--out", str(a)])
    main(common + ["--out", str(b), "--lanes", "1"])
    ia, ib = torch.load(a / "index.pt", weights_only=True), torch.load(b / "index.pt", weights_only=True)
    assert ia["n_variants"] == 1 and ia["clips"] == ib["clips"] and ia["n_shards"] == ib["n_shards"] and ia["n_clips"] == 7
    for sid in range(ia["n_shards"]):
        sa = torch.load(a / f"shard_{sid:05d}.pt", weights_only=True)
        sb = torch.load(b / f"shard_{sid:05d}.pt", weights_only=True)
        for key in ("feats", "joints3d", "joints2d", "K"):
            assert torch.equal(sa[key], sb[key]), (sid, key)


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


def test_cli_fp8_calibrates_on_the_first_real_batch(tmp_path, lib_built, capsys):
    """--precision fp8: the activation scales come from the first batch of the run's own clips (not from noise frames), the same on
    every rank; the run completes, its features are finite and close to the bf16 run's."""
    from implementation_phd_lab_vision_amd.preprocess_resnet_features import main
    common = ["--root", "unused", "--synthetic-clips", "3", "--seq-len", "2", "--batch-size", "2", "--num-workers", "0",
              "--shard-size", "4", "--shuffle-pool", "4", "--device", "cuda", "--max-batch", "8", "--synthetic-weights"]
    a, b = tmp_path / "fp8", tmp_path / "bf16"
    main(common + ["--out", str(a), "--precision", "fp8"])
    text = capsys.readouterr().out
    assert "calibrated on the first 2 clip(s) = 4 real frames" in text
    main(common + ["--out", str(b)])
    fa = torch.load(a / "shard_00000.pt", weights_only=True)["feats"]
    fb = torch.load(b / "shard_00000.pt", weights_only=True)["feats"]
    assert torch.isfinite(fa).all()
    rel = float((fa.double() - fb.double()).norm() / fb.double().norm())
    assert rel < 0.1, rel


def test_cli_fp8_with_the_device_producer_calibrates_on_producer_frames(tmp_path, lib_built, capsys):
    """--precision fp8 --device-producer --augment on decoded clips (round-2 ADVICE): the calibration frames come from the device
    producer's own crop + resize, not from ds[i] (the host producer, which under --synthetic-decoded --augment has no ColorJitter and
    raised before the extraction started)."""
    from implementation_phd_lab_vision_amd.preprocess_resnet_features import main
    out = tmp_path / "fp8dev"
    main(["--root", "unused", "--out", str(out), "--synthetic-clips", "3", "--synthetic-decoded", "--seq-len", "2", "--batch-size", "2",
          "--num-workers", "0", "--shard-size", "4", "--shuffle-pool", "4", "--device", "cuda", "--max-batch", "8", "--synthetic-weights",
          "--precision", "fp8", "--device-producer", "--resize-mode", "fixed", "--augment"])
    text = capsys.readouterr().out
    assert "calibrated on the first 2 clip(s) = 4 real frames made by the device producer" in text
    idx = torch.load(out / "index.pt", weights_only=True)
    feats = torch.load(out / "shard_00000.pt", weights_only=True)["feats"]
    assert idx["n_clips"] == 3 and idx["n_variants"] == 4 and torch.isfinite(feats).all()


def _run_host_and_device_producer(tmp_path, augment, cjitter_fn=None):
    """The same synthetic DECODED clips through (a) the host producer (the reference's __getitem__ restated: box, crop, ATen uint8
    resize, /255, variants, Normalize -> fp32 frames -> upload) and (b) the device producer (one uint8 upload per clip)."""
    from implementation_phd_lab_vision_amd import frames
    from implementation_phd_lab_vision_amd.backbone import ResNet50Backbone
    from implementation_phd_lab_vision_amd.preprocess_resnet_features import run_extraction
    from implementation_phd_lab_vision_amd.producer import DeviceProducer
    from implementation_phd_lab_vision_amd.synthetic import SyntheticDecodedClips
    from tests.helpers import cli_args
    dev = torch.device("cuda:0")
    bb = ResNet50Backbone(seed=0, max_batch=32).to(dev).eval()
    ds = SyntheticDecodedClips(5, seq_len=3, augment=augment, cjitter_fn=cjitter_fn)
    outs = {}
    for name in ("host", "device"):
        out = tmp_path / name
        args = cli_args(out, seq_len=3, batch_size=2, shard_size=3, shuffle_pool=4, shuffle_seed=5, augment=augment, save_fp16=False)
        producer = DeviceProducer(bb, dev, augment=augment, resize_mode=frames.RESIZE_FIXED) if name == "device" else None
        run_extraction(ds, args, bb, dev, log=lambda *_: None, producer=producer)
        outs[name] = out
    bb.close()
    return outs


def test_device_producer_equals_host_producer_byte_for_byte(tmp_path, lib_built):
    """--device-producer, no augmentation, fixed-point resize: the output directory equals the host-producer path's tensor for tensor
    (features, adjusted joints, intrinsics, boxes, index) -- src/dataset.py:141-152,395-405 moved onto the MI355X."""
    from tests.helpers import assert_same_feature_cache
    outs = _run_host_and_device_producer(tmp_path, augment=False)
    assert_same_feature_cache(outs["device"], outs["host"])


def test_device_producer_augment_variants(tmp_path, lib_built):
    """--device-producer --augment: four variants from ONE uint8 upload per clip.  orig / hflip / trev rows and every annotation are
    bit-identical to the host-producer path; the cjitter rows (device kernel vs the oracle's restatement of torchvision's
    ColorJitter, same per-clip draw) agree to the bf16 network's sensitivity to 5e-6 input differences."""
    from oracle.colorjitter_oracle import color_jitter as cj_oracle
    outs = _run_host_and_device_producer(tmp_path, augment=True, cjitter_fn=lambda video, p: cj_oracle(video, p))
    ih = torch.load(outs["host"] / "index.pt", weights_only=True)
    idv = torch.load(outs["device"] / "index.pt", weights_only=True)
    assert ih == idv and ih["n_variants"] == 4
    for sid in range(ih["n_shards"]):
        sh = torch.load(outs["host"] / f"shard_{sid:05d}.pt", weights_only=True)
        sd_ = torch.load(outs["device"] / f"shard_{sid:05d}.pt", weights_only=True)
        for key in ("joints3d", "joints2d", "K"):
            assert torch.equal(sh[key], sd_[key]), (sid, key)
        assert [m["aug"] for m in sh["meta"]] == [m["aug"] for m in sd_["meta"]]
        for row, meta in enumerate(sh["meta"]):
            a, b = sd_["feats"][row], sh["feats"][row]
            if meta["aug"] == "cjitter":
                rel = float((a.double() - b.double()).norm() / b.double().norm())
                assert rel < 2e-2, (sid, row, rel)
            else:
                assert torch.equal(a, b), (sid, row, meta["aug"])


def test_cli_device_producer_flag_end_to_end(tmp_path, lib_built, capsys):
    """The CLI itself with --device-producer (and --augment): same files as the host-producer CLI run for the variants whose frames are
    bit-defined (orig / hflip / trev), cjitter rows finite."""
    from implementation_phd_lab_vision_amd.preprocess_resnet_features import main
    common = ["--root", "unused", "--synthetic-clips", "4", "--synthetic-decoded", "--seq-len", "2", "--batch-size", "3", "--num-workers", "0",
              "--shard-size", "4", "--shuffle-pool", "4", "--device", "cuda", "--max-batch", "16", "--synthetic-weights"]
    a, b = tmp_path / "host", tmp_path / "dev"
    main(common + ["--out", str(a)])
    main(common + ["--out", str(b), "--device-producer", "--resize-mode", "fixed"])
    assert "Producer   : on the device" in capsys.readouterr().out
    from tests.helpers import assert_same_feature_cache
    for name in ("index.pt", "shard_00000.pt"):
        x, y = torch.load(a / name, weights_only=True), torch.load(b / name, weights_only=True)
        if name == "index.pt":
            assert x == y
        else:
            for key in ("feats", "joints3d", "joints2d", "K"):
                assert torch.equal(x[key], y[key]), key
            assert all(torch.equal(m1["box"], m2["box"]) for m1, m2 in zip(x["meta"], y["meta"]))
    c = tmp_path / "aug"
    main(common + ["--out", str(c), "--device-producer", "--resize-mode", "fixed", "--augment"])
    sh = torch.load(c / "shard_00000.pt", weights_only=True)
    assert sh["feats"].shape[0] == 16 and torch.isfinite(sh["feats"]).all() and [m["aug"] for m in sh["meta"][:4]] == ["orig", "cjitter", "hflip", "trev"]
    assert torch.equal(sh["feats"][3], sh["feats"][0].flip(0))          # trev = orig in reverse frame order


def _torchrun_one_rank(argv, tmp_path, timeout=600, nproc=1, extra_env=None):
    """A FRESH child process (started before any GPU call in that child): torchrun, `nproc` rank(s), rendezvous on 127.0.0.1."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4", PYTHONPATH=str(root) + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--nnodes=1", f"--nproc-per-node={nproc}", "--local-addr", "127.0.0.1"] + argv
    res = subprocess.run(cmd, env=env, cwd=str(root), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
    out = res.stdout.decode(errors="replace")
    assert res.returncode == 0, f"torchrun child failed ({res.returncode}):\n{out[-4000:]}"
    return out


def test_cli_under_torchrun_runs_the_rccl_path_and_writes_the_same_files(tmp_path, lib_built):
    """The multi-GPU code path on the hardware there is: the CLI as ONE rank under torchrun joins a process group with backend "nccl"
    (= RCCL) bound to its device, every round's feature block goes through ``dist.gather(async_op=True)`` on device tensors, the side
    stream is ordered behind the collective, the D2H copy lands in pinned memory and the packer thread writes the shards -- and the
    output directory equals the one the same command writes without a process group."""
    from implementation_phd_lab_vision_amd.preprocess_resnet_features import main
    from tests.helpers import assert_same_feature_cache
    common = ["--root", "unused", "--synthetic-clips", "9", "--seq-len", "4", "--batch-size", "2", "--num-workers", "0", "--shard-size", "4",
              "--shuffle-pool", "5", "--shuffle-seed", "11", "--device", "cuda", "--max-batch", "16", "--synthetic-weights", "--augment"]
    plain, grouped = tmp_path / "plain", tmp_path / "rccl"
    out = _torchrun_one_rank(["-m", "implementation_phd_lab_vision_amd.preprocess_resnet_features"] + common + ["--out", str(grouped)], tmp_path)
    assert "ranks: 1" in out and "process group: nccl" in out, out[-2000:]
    main(common + ["--out", str(plain)])
    assert_same_feature_cache(grouped, plain)


def test_bench_under_torchrun_runs_gather_barrier_and_fence_on_rccl(tmp_path, lib_built):
    """bench.py --gpus 1 as one torchrun rank: init_process_group("nccl", device_id=...), the per-step gather of the (256, 2048)
    feature block, the all-reduced pre-heat / elapsed times and the barrier + synchronize fences all execute on RCCL."""
    import json
    out = _torchrun_one_rank(["bench.py", "--gpus", "1", "--steps", "3", "--warmup", "1", "--preheat", "0.2", "--no-secondary",
                              "--no-cpu-baseline"], tmp_path)
    line = [ln for ln in out.splitlines() if ln.startswith("{") and '"metric"' in ln][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 1 and rec["value"] > 1000 and rec["checked"]["finite"] and rec["checked"]["equal_to_batch2_run"]
    assert "RCCL gather" in rec["config"]["workload"]
    # the default two lanes: step k's gather waits for lane k % 2's event, the lane's next step for the event behind that gather
    # (BackboneLanes.tune may fall back to one lane on a box where the lanes do not overlap: the line says which mode ran)
    assert rec["config"]["lanes_requested"] == 2 and rec["config"]["lanes"] in (1, 2) and rec["checked"]["lanes_equal"] is True
    assert rec["mode"] in ("2 batches in flight (BackboneLanes)", "one batch at a time") and rec["roofline"]["class"] in rec["kernels"]


def test_bench_two_ranks_rehearsal_runs_the_same_collective_sequence(tmp_path, lib_built):
    """TWO ranks through bench.py's lane + gather path on the one GPU there is.  R50_BENCH_REHEARSAL=1 puts every rank on
    cuda:0 with gloo collectives (RCCL refuses two ranks on one device); every rank must issue the same sequence -- all-reduced pre-heat count,
    per-step gather behind the lane's event, MAX-reduced failure flag, barriers -- or the run hangs / fails: rc 0 inside the timeout is the
    assertion.  Its numbers mean nothing (the line says REHEARSAL); no scaling figure is derived from it."""
    import json
    out = _torchrun_one_rank(["bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--preheat", "0.2", "--no-secondary", "--no-cpu-baseline"],
                             tmp_path, timeout=900, nproc=2, extra_env={"R50_BENCH_REHEARSAL": "1"})
    line = [ln for ln in out.splitlines() if ln.startswith("{") and '"metric"' in ln][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and "REHEARSAL" in rec["data"] and rec["steps"] == 3
    assert rec["checked"]["finite"] and rec["checked"]["equal_to_batch2_run"] and rec["config"]["lanes_requested"] == 2
