"""CLI: H36M clips -> per-frame ResNet-50 features -> shuffled ``shard_XXXXX.pt`` + ``index.pt``.

MI355X-native replacement of /root/reference/src/preprocess_resnet_features.py ``main()`` (:134-428):
the same 14 flags (:136-155) and the same output contract (shards.py), so the reference's
``src/dataset_features.py`` and ``src/train.py`` consume the result unchanged.  What changes is the
engine: the backbone is ``ResNet50Backbone`` (hand-written HIP kernels behind include/r50.h) instead of
torchvision + autocast + torch.compile, and multi-GPU is one process per GPU (``torchrun``) with an
RCCL gather of the feature blocks instead of ``nn.DataParallel`` (distributed.py).

    python -m implementation_phd_lab_vision_amd.preprocess_resnet_features --root R --out O [...]
    torchrun --standalone --local-addr 127.0.0.1 --nproc-per-node 8 -m \
        implementation_phd_lab_vision_amd.preprocess_resnet_features --root R --out O [...]

The clip dataset is the reference's own ``Human36MPreprocessedClips`` (src/dataset.py), imported from
``$H36M_REFERENCE_SRC`` or ``sys.path`` — the frame producer is upstream of this path.  Extra,
flags: ``--weights PATH`` (local torchvision checkpoint; nothing is downloaded) — REQUIRED unless
``--synthetic-weights`` (seeded random weights, benchmarks / tests only) is given; ``--synthetic-clips N``
(run without H36M data), ``--precision {bf16,fp16,bf16w2,fp32x,fp8}``, ``--micro-batch``, ``--max-batch``.
"""
from __future__ import annotations

import argparse
import os
import sys
import time
from pathlib import Path
from typing import Callable, List, Optional

import torch
from torch.utils.data import DataLoader, Subset

from . import distributed as D
from .prefetch import DevicePrefetcher, is_time_reverse_of
from .shards import AUG_NAMES, AsyncFileWriter, ShardPacker


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser("Precompute per-clip ResNet50 features for H36M (MI355X-native)")
    # ---- the reference's 14 flags, verbatim (:137-153) ----
    p.add_argument("--root", type=str, required=True, help="H36M preprocessed root")
    p.add_argument("--out", type=str, required=True, help="Output directory for cached features")
    p.add_argument("--seq-len", type=int, default=40)
    p.add_argument("--frame-skip", type=int, default=2)
    p.add_argument("--stride", type=int, default=5)
    p.add_argument("--batch-size", type=int, default=32)
    p.add_argument("--num-workers", type=int, default=8)
    p.add_argument("--subjects", type=int, nargs="+", default=[1, 5, 6, 7, 8, 9, 11])
    p.add_argument("--device", type=str, default="cuda")
    p.add_argument("--save-fp16", action="store_true", help="Store feats as float16")
    p.add_argument("--augment", action="store_true", help="Write the 4 augmentation variants per clip")
    p.add_argument("--shard-size", type=int, default=512, help="Number of clips per shard file")
    p.add_argument("--shuffle-pool", type=int, default=8192, help="Clips accumulated before a shuffle + flush")
    p.add_argument("--shuffle-seed", type=int, default=123, help="Seed for clip-level shuffling")
    # ---- additions (optional) ----
    p.add_argument("--weights", type=str, default=None,
                   help="Local torchvision-layout ResNet-50 checkpoint (e.g. resnet50-11ad3fa6.pth = IMAGENET1K_V2, what the reference "
                        "downloads at :207).  Required unless --synthetic-weights is given: nothing is downloaded here")
    p.add_argument("--synthetic-weights", action="store_true",
                   help="Run with seeded RANDOM backbone weights (--weights-seed): for benchmarks / tests only -- the shards then hold "
                        "features of an untrained network")
    p.add_argument("--weights-seed", type=int, default=0)
    p.add_argument("--synthetic-clips", type=int, default=0, help="Use N synthetic clips instead of reading --root")
    p.add_argument("--synthetic-decoded", action="store_true",
                   help="--synthetic-clips: synthetic DECODED clips (uint8 frames + raw joints + camera) run through the producer, "
                        "instead of ready-made normalised frames")
    p.add_argument("--device-producer", action="store_true",
                   help="Frame producer on the MI355X (producer.py): the loader hands over decoded uint8 frames, the person crop is uploaded "
                        "once per clip, crop / resize / flip / ColorJitter / normalisation run on the device (4 variants from one upload)")
    p.add_argument("--resize-mode", choices=["float", "fixed"], default="float",
                   help="--device-producer: bilinear resize arithmetic.  float = torchvision's v1 functional.resize on uint8 (what the "
                        "reference imports: fp32 + round); fixed = ATen's native uint8 kernel (the v2 API), bit-exact on the device")
    p.add_argument("--precision", choices=["bf16", "fp16", "bf16w2", "fp32x", "fp8"], default="bf16",
                   help="bf16 = the reference's CUDA autocast dtype (fast); fp32x = fp32-class accuracy (its CPU numerics), ~2.7x slower")
    p.add_argument("--fp8-calib", choices=["first-batch", "noise"], default="first-batch",
                   help="--precision fp8: where the 43 activation scales come from.  first-batch = the first batch of REAL clips (decoded "
                        "once more, like the reference's warm-up batch, :235-245); noise = 8 synthetic uniform-noise frames (scales that fit "
                        "noise, not H36M crops: anything above 448 x scale clips silently)")
    p.add_argument("--lanes", type=int, default=2,
                   help="Backbone copies on their own HIP streams (backbone.BackboneLanes): the variants of a batch and consecutive batches are "
                        "independent forward passes and two are kept in flight (same bits, +5 %% frames/s); 1 = one stream")
    p.add_argument("--micro-batch", type=int, default=0, help="Frames per pass through the layer stack (0 = auto)")
    p.add_argument("--max-batch", type=int, default=256, help="Frames per backbone call chunk (workspace size)")
    p.add_argument("--no-trev-reuse", action="store_true",
                   help="--augment: run the backbone on the temporal-reverse variant too instead of flipping variant 0's features")
    return p


def collate_variants(batch):
    """augment=True items are lists of variants: collate each variant across the batch -> list of
    (videos, joints3d, joints2d, K), one per variant (reference: augment_collate_fn, :59-69)."""
    stacked = []
    for v in range(len(batch[0])):
        cols = list(zip(*[sample[v][:4] for sample in batch]))
        stacked.append(tuple(torch.stack(col) for col in cols))
    return stacked


augment_collate_fn = collate_variants       # the reference's name for it


def _open_dataset(args):
    if args.synthetic_clips > 0 and args.synthetic_decoded:
        from .synthetic import SyntheticDecodedClips
        return SyntheticDecodedClips(args.synthetic_clips, seq_len=args.seq_len, subjects=tuple(args.subjects),
                                     augment=args.augment, stride=args.stride)
    if args.synthetic_clips > 0:
        from .synthetic import SyntheticClips
        return SyntheticClips(args.synthetic_clips, seq_len=args.seq_len, subjects=tuple(args.subjects),
                              augment=args.augment, stride=args.stride)
    src = os.environ.get("H36M_REFERENCE_SRC")
    if src and src not in sys.path:
        sys.path.insert(0, src)
    try:
        from dataset import Human36MPreprocessedClips      # the reference's src/dataset.py
    except Exception as exc:
        raise SystemExit(
            "cannot import the reference's clip dataset (src/dataset.py: Human36MPreprocessedClips, needs "
            f"torchvision.io): {exc!r}\nPoint H36M_REFERENCE_SRC at the reference's src/ directory, or use "
            "--synthetic-clips N to run the feature path without H36M data.")
    return Human36MPreprocessedClips(root=args.root, subjects=args.subjects, seq_len=args.seq_len,
                                     frame_skip=args.frame_skip, stride=args.stride, augment=args.augment,
                                     max_clips=None)


def _resolve_weights(args):
    """The reference's ``models.resnet50(weights=IMAGENET1K_V2)`` (:207) fetches a checkpoint; here the weights come from a local
    file (``--weights``) or, only on explicit request (``--synthetic-weights``), from the seeded synthetic generator.  Silently
    writing shards of a random network would produce files that ``dataset_features.py`` accepts and ``train.py`` cannot learn from."""
    from .weights import load_state_dict_from_path, synthetic_state_dict
    if args.weights and args.synthetic_weights:
        raise SystemExit("--weights and --synthetic-weights are mutually exclusive")
    if args.weights:
        return load_state_dict_from_path(args.weights), f"file {args.weights}"
    if args.synthetic_weights:
        return synthetic_state_dict(args.weights_seed), f"seeded synthetic (seed {args.weights_seed})"
    raise SystemExit("no backbone weights: pass --weights PATH (a local torchvision ResNet-50 checkpoint, e.g. the IMAGENET1K_V2 file "
                     "resnet50-11ad3fa6.pth the reference downloads; there is no network here) or, for benchmarks and tests only, "
                     "--synthetic-weights")


def weights_digest(state_dict) -> str:
    """sha256 over the 265 float tensors the backbone consumes (names and fp32 bytes, execution order)."""
    import hashlib
    from .weights import iter_named_tensors
    hsh = hashlib.sha256()
    for name, t in iter_named_tensors(state_dict):
        hsh.update(name.encode() + b"\0")
        hsh.update(t.numpy().tobytes())
    return hsh.hexdigest()


def _calibrate_fp8(backbone, ds, args, device, log) -> None:
    """fp8 mode: choose the activation scales from real frames.  Every rank calibrates on the SAME clips (global batch 0, loaded by
    each rank itself: the scales must agree across ranks or the shards would depend on the world size)."""
    if args.fp8_calib == "noise" or len(ds) == 0:
        log("fp8 scales : calibrated on 8 synthetic NOISE frames (--fp8-calib noise): real crops may exceed them and clip")
    else:
        n_clips = min(len(ds), args.batch_size)
        videos = []
        if args.device_producer:
            # the frames the run itself will feed the backbone: decoded clip -> box region -> crop + resize ON THE DEVICE (the producer's
            # kernel and resize mode), then the loader's /255 + Normalize (src/dataset.py:242-245,429).  ds[i] would go through the HOST
            # producer (other resize arithmetic; with --synthetic-decoded --augment it needs a host ColorJitter that is not there)
            from . import frames as F
            from .producer import DecodedClips
            dc = DecodedClips(ds, augment=False)
            mode = F.RESIZE_FIXED if args.resize_mode == "fixed" else F.RESIZE_FLOAT
            mean = torch.tensor([0.485, 0.456, 0.406], device=device).view(1, 3, 1, 1)
            std = torch.tensor([0.229, 0.224, 0.225], device=device).view(1, 3, 1, 1)
            for i in range(n_clips):
                reg = dc[i]["region"].to(device)
                u8 = F.crop_and_resize_video_uint8(reg, [0, 0, int(reg.shape[1]), int(reg.shape[2])], 224, mode)
                videos.append((u8.to(torch.float32) / 255.0 - mean) / std)
        else:
            for i in range(n_clips):
                item = ds[i]
                videos.append((item[0][0] if args.augment else item[0]))        # the orig variant's (T,3,224,224) frames
        frames = torch.cat(videos)[: args.max_batch].to(device=device, dtype=torch.float32)
        backbone.calibrate_fp8(frames=frames)
        log(f"fp8 scales : calibrated on the first {n_clips} clip(s) = {frames.shape[0]} real frames"
            + (" made by the device producer" if args.device_producer else "") + " (margin x1.25 over their largest activation per tensor)")
    sc = backbone.fp8_scales
    log(f"fp8 scales : 43 tensors, representable range 448 x scale from {448 * min(sc):.3g} to {448 * max(sc):.3g}")


def _resolve_device(name: str, ctx: D.RankContext) -> torch.device:
    if not name.startswith("cuda"):
        raise SystemExit(f"--device {name}: this build runs the backbone on an MI355X only (PyTorch-ROCm exposes it "
                         "as 'cuda'); there is no CPU path")
    if not torch.cuda.is_available():
        raise SystemExit("no GPU visible: the HIP feature path cannot run (the reference would fall back to CPU "
                         "torchvision here, :157-161; this build fails loudly instead)")
    if ctx.distributed:
        return torch.device("cuda", ctx.local_rank)
    dev = torch.device(name)
    return torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())


class PendingFeatures:
    """The forward passes of one batch, submitted but not waited for (``submit_features``); ``result()`` orders the current stream behind
    them and returns (B, V, T, 2048)."""

    def __init__(self, per_variant, pending):
        self._per_variant, self._pending = per_variant, pending

    def result(self) -> torch.Tensor:
        pv = self._per_variant
        for slot, ticket, _x, (b, t) in self._pending:
            pv[slot] = ticket.wait().view(b, t, -1)
        self._pending = []
        pv = [pv[0].flip(1) if isinstance(v, str) else v for v in pv]
        return torch.stack(pv, dim=1)


def extract_features(backbone: Callable[[torch.Tensor], torch.Tensor], variants_batch, device: torch.device,
                     reuse_trev: bool = True) -> torch.Tensor:
    """``submit_features(...).result()``: the reference's call site as one blocking call."""
    return submit_features(backbone, variants_batch, device, reuse_trev).result()


def submit_features(backbone: Callable[[torch.Tensor], torch.Tensor], variants_batch, device: torch.device,
                    reuse_trev: bool = True) -> PendingFeatures:
    """The hot call site (:287-297): every variant's (B,T,3,224,224) clip batch -> (B,T,2048) fp32.
    Returns (B, V, T, 2048) on ``device``.

    The backbone is a per-frame function, so the features of the temporal-reverse variant (``AUG_NAMES[3]``, built
    by ``_aug_temporal_reverse`` from the same clip as variant 0: src/dataset.py:199-207,424-426) are the
    features of variant 0 in reverse frame order, bit for bit: under ``--augment`` that forward pass (a quarter of
    the work) is replaced by a flip.  Only when the frames really are the reverse (``prefetch.is_time_reverse_of``: every frame
    takes part in the check, per batch).

    The variants are independent forward passes: a backbone with lanes (``backbone.BackboneLanes``: has ``submit``) gets all of
    them submitted before the first result is waited for, so two are in flight at a time (same bits, +5 % frames/s).  And nothing is
    waited for HERE: ``run_extraction`` submits round q + 1 before it asks for round q's ``result()``, so that without ``--augment`` (one
    forward pass per batch) consecutive batches share the two lanes the same way."""
    per_variant = []
    lanes = hasattr(backbone, "submit") and device.type == "cuda"
    pending = []                                       # (slot in per_variant, ticket, input kept alive, (b, t))
    ready = None
    for vi, (v_video, *_rest) in enumerate(variants_batch):
        if v_video is None:                            # the prefetcher found it to be the time reverse and did not upload it
            per_variant.append("flip0")
            continue
        if (reuse_trev and len(variants_batch) == len(AUG_NAMES) and vi == AUG_NAMES.index("trev")
                and v_video.device == variants_batch[0][0].device and is_time_reverse_of(v_video, variants_batch[0][0])):
            per_variant.append("flip0")
            continue
        v_video = v_video.to(device, non_blocking=True)
        b, t, c, h, w = v_video.shape
        x = v_video.view(b * t, c, h, w).contiguous()
        if lanes:
            if x.dtype != torch.float32:
                x = x.to(torch.float32)
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(device))       # x (and everything before it) is complete before the lane reads it
            pending.append((len(per_variant), backbone.submit(x, after=ready), x, (b, t)))
            per_variant.append(None)
        else:
            per_variant.append(backbone(x).flatten(1).view(b, t, -1).to(torch.float32))
    return PendingFeatures(per_variant, pending)


def run_extraction(ds, args, backbone, device: torch.device, ctx: Optional[D.RankContext] = None,
                   log: Callable[[str], None] = print, host_slots: int = 4, stats: Optional[dict] = None,
                   producer=None) -> Optional[Path]:
    """Whole job for one rank.  ``backbone(x)``: (N,3,224,224) fp32 on ``device`` -> (N,2048,1,1).
    Rank 0 packs and returns the index path; other ranks return None.

    Round q: rank r runs global batch ``q * world + r`` (distributed.py: batches round-robin over the ranks, NOT contiguous clip
    ranges -- every round's blocks, taken in rank order, continue the global clip order, which the shuffle pool's RNG sequence
    needs, :98,300,345).  The loop body is: forward passes of round q -> ``post`` (pack + async gather) -> ``collect`` round q-1
    (async D2H into a pinned slot, hand-over to the packing thread).  Nothing in it waits for the device, for a copy or for the
    packer, so round q+1's kernels are queued while round q's features travel and round q-1's clips are packed.
    ``stats`` (optional dict) receives ``compute_done_s`` / ``total_s`` (seconds since the loop started) and
    ``clips_packed_at_compute_done`` (clips the packing thread had consumed when the last round was collected).
    ``producer`` (``producer.DeviceProducer``): the loader then yields DECODED clips and crop / resize / variants / features all
    happen on the device (``--device-producer``)."""
    ctx = ctx or D.RankContext()
    n_vars = len(AUG_NAMES) if args.augment else 1
    n_clips = len(ds)
    bs = args.batch_size
    feat_dtype = torch.float16 if args.save_fp16 else torch.float32

    mine = D.rank_clip_indices(n_clips, bs, ctx.rank, ctx.world)
    loader = None
    if mine:
        kw = dict(batch_size=bs, shuffle=False, num_workers=args.num_workers, pin_memory=device.type == "cuda",
                  drop_last=False, collate_fn=collate_variants if args.augment else None)
        if args.num_workers > 0:
            kw["prefetch_factor"] = 2
        src = ds
        if producer is not None:
            from .producer import DecodedClips, collate_decoded
            src = DecodedClips(ds, augment=args.augment)
            kw["collate_fn"] = collate_decoded
        loader = DataLoader(Subset(src, mine) if ctx.distributed else src, **kw)
    it = iter(loader) if loader is not None else None
    if it is not None and producer is None:       # copy batch k+1 to the device on a side stream while batch k computes (no-op on a CPU device)
        it = DevicePrefetcher(it, device, augment=args.augment, skip_trev=not getattr(args, "no_trev_reuse", False),
                              trev_index=AUG_NAMES.index("trev"))

    packer = None
    if ctx.is_root:
        packer = ShardPacker(args.out, n_vars, args.shard_size, args.shuffle_pool, args.shuffle_seed, AsyncFileWriter())
        log(f"Processing {n_clips} clips × {n_vars} variant(s) = {n_clips * n_vars} entries …")
        log(f"Writing shards of {args.shard_size} clips each → {packer.out_root}")
        log("-" * 60)

    layout = D.BlockLayout(batch=bs, n_vars=n_vars, seq_len=args.seq_len, n_joints=int(getattr(ds, "joints_num", 17)))
    t_all = time.time()
    progress = {"done": 0, "t_last": t_all}

    def pack_round(q: int, host_blocks: torch.Tensor) -> None:
        """Packing thread, rank 0: the ``world`` blocks of round q, in rank order == global clip order (:299-341)."""
        for r in range(ctx.world):
            g = q * ctx.world + r
            clips = D.batch_clip_range(g, n_clips, bs)
            if len(clips) == 0:
                continue
            blk = layout.unpack(host_blocks[r], feat_dtype, has_box=not args.augment)
            if blk["count"] != len(clips):
                raise RuntimeError(f"rank {r} sent {blk['count']} clips for batch {g}, expected {len(clips)}")
            for b, ci in enumerate(clips):
                rec = ds.index[ci]
                group = []
                for v in range(n_vars):
                    group.append({
                        "feat": blk["feats"][b, v],
                        "joints3d": blk["joints3d"][b, v],
                        "joints2d": blk["joints2d"][b, v],
                        "K": blk["K"][b, v],
                        "meta": {"subject": rec.subject, "action": rec.action, "cam": rec.cam, "start": rec.start,
                                 "end": rec.end, "aug": AUG_NAMES[v] if args.augment else "orig",
                                 "box": blk["box"][b] if blk["box"] is not None else None},
                    })
                packer.add_group(group)
                progress["done"] += 1
            done = progress["done"]
            if done % 200 == 0 or done == n_clips:
                dt = time.time() - progress["t_last"]
                rate = 200 / dt if dt > 0 else 0.0
                progress["t_last"] = time.time()
                eta = (n_clips - done) / rate if rate > 0 else 0.0
                log(f"[{100 * done / n_clips:5.1f}%] {done:6d}/{n_clips} clips | {rate:6.1f} clips/s | ETA {eta:6.1f}s | "
                    f"shard {packer.shard_id} (pool: {len(packer.pool)} clips, carry: {len(packer.carry)} clips)")

    exchange = D.RoundExchange(ctx, layout, device, pack_round, host_slots=host_slots)
    my_batches = (D.n_batches(n_clips, bs) - ctx.rank + ctx.world - 1) // ctx.world if n_clips else 0
    n_rounds = D.n_rounds(n_clips, bs, ctx.world)
    # One round AHEAD: round q's forward passes are submitted (backbone lanes: queued on their own streams) before round q - 1's features are
    # waited for and posted, so that two batches are in flight on the device even without --augment.  The exchange sees the same sequence
    # as before -- post(0), post(1), collect(0), post(2), collect(1), ... -- only one submission later.
    held = None                                            # (features: tensor | PendingFeatures | None, variants_batch, box_batch) of round q - 1

    def finish_round(qp: int) -> None:
        feats, vb, bb_ = held
        if isinstance(feats, PendingFeatures):
            feats = feats.result()
        exchange.post(qp, feats, vb, bb_)
        if qp > 0:
            exchange.collect(qp - 1)

    for q in range(n_rounds):
        feats = variants_batch = box_batch = None
        if q < my_batches:
            batch = next(it)
            if producer is not None:                       # decoded clips in: crop / resize / variants / features on the device
                feats, variants_batch, box_batch = producer.compute(batch)
                t = feats.shape[2]
            else:
                if args.augment:
                    variants_batch, box_batch = batch, None
                else:
                    video, j3d, j2d, k, box = batch
                    variants_batch, box_batch = [(video, j3d, j2d, k)], box
                t = variants_batch[0][0].shape[1]
                feats = submit_features(backbone, variants_batch, device, reuse_trev=not getattr(args, "no_trev_reuse", False))
            if t != args.seq_len:
                raise RuntimeError(f"clips have {t} frames, --seq-len says {args.seq_len}")
        if q > 0:
            finish_round(q - 1)
        held = (feats, variants_batch, box_batch)
    if n_rounds > 0:
        finish_round(n_rounds - 1)
        exchange.collect(n_rounds - 1)
    if stats is not None:
        stats["compute_done_s"] = time.time() - t_all
        stats["clips_packed_at_compute_done"] = progress["done"]      # how far the packing thread was when the last round was collected
    exchange.finish()
    if stats is not None:
        stats["total_s"] = time.time() - t_all

    if not ctx.is_root:
        return None
    packer.finish()
    log("\nWaiting for all shards to be written to disk...")
    index_path = packer.write_index(seq_len=args.seq_len, frame_skip=args.frame_skip, save_fp16=args.save_fp16,
                                    augment=args.augment, n_clips=n_clips)
    log("✓ All shards written and index saved.")
    total = time.time() - t_all
    log("-" * 60)
    log(f"✓ Done!  {n_clips} clips × {n_vars} variant(s) packed into {packer.shard_id} shard(s)")
    log(f"✓ Total time        : {total:.1f}s")
    log(f"✓ Throughput        : {n_clips / total:.1f} clips/s  ({n_clips * n_vars / total:.1f} variant entries/s)")
    log(f"✓ Avg time per clip : {1000 * total / max(1, n_clips):.1f} ms")
    return index_path


@torch.no_grad()
def main(argv: Optional[List[str]] = None) -> None:
    args = build_parser().parse_args(argv)
    state_dict, source = _resolve_weights(args)          # fails here, before any device work, if no weight source was named
    ctx = D.init_from_env(use_gpu=True)
    device = _resolve_device(args.device, ctx)
    torch.cuda.set_device(device)
    log = print if ctx.is_root else (lambda *_a, **_k: None)

    log(f"Device     : {device}  ({torch.cuda.get_device_name(device)}), ranks: {ctx.world}"
        + (f", process group: {ctx.backend} (RCCL), one asynchronous gather per round" if ctx.distributed else ""))
    n_vars = len(AUG_NAMES) if args.augment else 1
    log(f"Augment    : {args.augment}  ({'4 variants/clip → ' + ', '.join(AUG_NAMES) if args.augment else 'none'})")
    log(f"Shard size : {args.shard_size} clips  ({args.shard_size * n_vars} variant entries/shard)")

    from . import _lib
    from .backbone import ResNet50Backbone
    if ctx.is_root:
        _lib.build_library()
    if ctx.distributed:
        torch.distributed.barrier()
    ds = _open_dataset(args)
    digest = weights_digest(state_dict)
    log(f"Weights    : {source}  (sha256 {digest[:16]}…)")
    if args.synthetic_weights:
        log("WARNING    : --synthetic-weights: the backbone is a RANDOM-initialised ResNet-50; the shards will NOT hold ImageNet features")
    n_lanes = max(1, int(getattr(args, "lanes", 1)))
    if n_lanes > 1 and not args.device_producer:
        from .backbone import BackboneLanes
        backbone = BackboneLanes(lanes=n_lanes, state_dict=state_dict, max_batch=args.max_batch, micro_batch=args.micro_batch,
                                 precision=args.precision).to(device).eval()
        log(f"Lanes      : {n_lanes} backbone copies on their own streams (forward passes of a batch's variants / of consecutive batches run {n_lanes} at a time)")
    else:
        n_lanes = 1
        backbone = ResNet50Backbone(state_dict=state_dict, max_batch=args.max_batch, micro_batch=args.micro_batch,
                                    precision=args.precision).to(device).eval()
    if ctx.is_root:       # provenance beside the shards (index.pt keeps the reference's exact key set)
        import json
        Path(args.out).mkdir(parents=True, exist_ok=True)
        (Path(args.out) / "backbone_weights.json").write_text(json.dumps(
            {"source": source, "sha256": digest, "precision": args.precision, "synthetic": bool(args.synthetic_weights)}) + "\n")

    if args.precision == "fp8":
        _calibrate_fp8(backbone, ds, args, device, log)

    log("Warming up the HIP kernels...")                               # reference warm-up: :235-245
    warm = torch.zeros((min(args.max_batch, args.batch_size * args.seq_len), 3, 224, 224), device=device)
    for _ in range(n_lanes):
        backbone(warm)
    torch.cuda.synchronize(device)
    if n_lanes > 1:
        backbone.tune(warm)               # informational ratio; too small a warm-up batch is not measured at all, a loss falls back to one lane
        log(f"Lanes      : {backbone.tune_mode} on the warm-up batch")
    del warm
    log("✓ Warmup complete\n")

    producer = None
    if args.device_producer:
        from . import frames
        from .producer import DeviceProducer
        producer = DeviceProducer(backbone, device, augment=args.augment,
                                  resize_mode=frames.RESIZE_FIXED if args.resize_mode == "fixed" else frames.RESIZE_FLOAT)
        log(f"Producer   : on the device ({args.resize_mode} resize): one uint8 upload per clip for {n_vars} variant(s)")
    try:
        run_extraction(ds, args, backbone, device, ctx, log, producer=producer)
    finally:
        backbone.close()
        D.shutdown(ctx)


if __name__ == "__main__":
    main()
