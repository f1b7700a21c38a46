"""Device frame producer for the CLI (``--device-producer``).

The reference's dataset does, per clip and on CPU workers (src/dataset.py:370-437): decode -> person box from the 2D joints ->
crop -> bilinear resize to 224 -> ``/255`` -> [4 augmentation variants] -> ``Normalize`` -> a (T,3,224,224) fp32 tensor PER VARIANT,
which the extraction loop then uploads (602 KB per frame and variant).  With the device producer the loader hands over the DECODED
uint8 frames and the raw annotations instead, and everything after the decode runs on the MI355X:

    host:    box = square_crop_from_2d(joints2d)   (frames.py, a few flops)            slice the box out of the frames (memcpy)
    H2D:     ONE uint8 upload of the cropped region per clip (side^2 x 3 bytes per frame -- for all 4 variants)
    device:  crop_and_resize_video_uint8            -> (T,3,224,224) uint8, variant "orig"  (r50_op_crop_resize_u8)
             the same kernel with its hflip flag    -> variant "hflip"
             aug_color_jitter_u8(orig crops)        -> variant "cjitter" as normalised fp32 (r50_op_color_jitter_u8)
             features_u8 / features                 -> (T,2048) per variant; "trev" = the orig features in reverse frame order
    host:    joints / intrinsics adjusted by the frames.py mirrors of the reference's functions

The features equal the host-producer path's bit for bit when both resize in fixed-point mode (ATen's native uint8 bilinear kernel,
what torchvision's v2 API dispatches to; ``frames.RESIZE_FIXED``); the v1 API the reference imports rounds through fp32
(``frames.RESIZE_FLOAT``, <= 1 LSB on ~1e-4 of the bytes apart).  The cjitter variant follows torchvision's published
kernels (the test-side oracle restates them; torchvision itself is absent here, so that variant is unpinned).
Video decode stays upstream: the adapter below calls the reference dataset's own reader.
"""
from __future__ import annotations

from typing import List, Optional

import torch

from . import frames as F
from .shards import AUG_NAMES


def decoded_item_from_reference_dataset(ds, idx: int) -> dict:
    """What ``Human36MPreprocessedClips.__getitem__`` (src/dataset.py:370-393) holds right after the decode: the clip's uint8 frames
    and its raw annotations.  Uses the reference dataset object's own reader / caches (duck-typed: ``index``, ``_gt_cache``,
    ``_read_video_uint8_clip_fast``, ``frame_skip``, ``crop_scale``)."""
    ci = ds.index[idx]
    frames_u8 = ds._read_video_uint8_clip_fast(ci.video_path, ci.start, ci.end)          # (T,H,W,3) uint8
    if frames_u8.dim() != 4 or frames_u8.shape[-1] != 3:                                  # the reference's `assert C == 3` (:375-376)
        raise AssertionError(f"decoded clip of {ci.video_path} is {tuple(frames_u8.shape)}, expected (T,H,W,3)")
    joints3d_all, joints2d_all = ds._gt_cache[ci.gt_path]
    orig_idx = torch.arange(ci.start, ci.end, dtype=torch.long) * ds.frame_skip
    if int(orig_idx[-1]) >= joints3d_all.shape[0]:
        raise RuntimeError(f"Joint index out of range for {ci.gt_path}: max orig_idx={int(orig_idx[-1])}, n_frames={joints3d_all.shape[0]}")
    joints3d, joints2d = joints3d_all[orig_idx], joints2d_all[orig_idx]
    # the reference's frame-count assert (:390-392): a reader that returned fewer frames than the clip has annotations for must not
    # silently pair frame t with the joints of another frame
    assert frames_u8.shape[0] == joints3d.shape[0], f"Mismatch T: video {frames_u8.shape[0]} vs joints {joints3d.shape[0]}"
    return {"frames": frames_u8, "joints3d": joints3d, "joints2d": joints2d, "cam": ci.cam_params,
            "crop_scale": float(getattr(ds, "crop_scale", 1.6))}


class DecodedClips(torch.utils.data.Dataset):
    """Loader-side view of a clip dataset for the device producer: item i = the decoded clip, its box (computed on the worker, as the
    reference does), the box region of the frames as one contiguous uint8 tensor, the adjusted annotations of all variants and --
    under ``augment`` -- the ColorJitter draw (``ColorJitter.make_params`` happens in the worker there too, src/dataset.py:188-197)."""

    def __init__(self, ds, augment: bool, out_size: int = 224):
        self.ds, self.augment, self.out_size = ds, augment, out_size
        self.index = ds.index
        self._decode = ds.decoded_item if hasattr(ds, "decoded_item") else (lambda i: decoded_item_from_reference_dataset(ds, i))

    def __len__(self) -> int:
        return len(self.ds)

    def __getitem__(self, i: int) -> dict:
        it = self._decode(i)
        frames_u8, j3d, j2d_raw = it["frames"], it["joints3d"], it["joints2d"]
        t, h, w, c = frames_u8.shape
        assert c == 3
        box = F.square_crop_from_2d(j2d_raw, h, w, scale=it.get("crop_scale", 1.6))
        top, left, hh, ww = box.tolist()
        region = frames_u8[:, top:top + hh, left:left + ww, :].contiguous()              # all the device needs: side^2 x 3 bytes per frame
        j2d = F.adjust_joints2d_after_crop_and_resize(j2d_raw, box, self.out_size)
        k = F.adjust_camera_after_crop_and_resize(it["cam"], box, self.out_size)
        annots = [(j3d, j2d, k)]
        cj = None
        if self.augment:
            cj = it.get("cj_params") or F.sample_color_jitter_params()
            annots.append((j3d, j2d, k))                                                  # cjitter: photometric only
            annots.append(F.aug_hflip_annotations(j3d, j2d, k, width=self.out_size))      # hflip
            tj3d, tj2d = F.aug_temporal_reverse_annotations(j3d, j2d)
            annots.append((tj3d, tj2d, k))                                                # trev
        return {"region": region, "box": box, "annots": annots, "cj": cj}


def collate_decoded(items: List[dict]) -> dict:
    """Regions differ in size from clip to clip (the box follows the person), so they stay a list; annotations are stacked per variant
    exactly as the reference's collate does (``augment_collate_fn``, :59-69)."""
    n_vars = len(items[0]["annots"])
    annots = []
    for v in range(n_vars):
        cols = list(zip(*[it["annots"][v] for it in items]))
        annots.append(tuple(torch.stack(col) for col in cols))
    return {"regions": [it["region"] for it in items], "box": torch.stack([it["box"] for it in items]), "annots": annots,
            "cj": [it["cj"] for it in items]}


class DeviceProducer:
    """``compute(batch)``: collated decoded batch -> ``(feats (B,V,T,2048) on the device, variants_annotations, box_batch)`` in the
    form ``run_extraction`` posts to the exchange.  One uint8 upload per clip; per batch ONE backbone call per uint8 variant (orig,
    hflip) over all B*T frames, plus one per clip for the cjitter variant (its fp32 frames are produced clip by clip)."""

    def __init__(self, backbone, device: torch.device, augment: bool, resize_mode: int = F.RESIZE_FIXED, out_size: int = 224):
        self.bb, self.device, self.augment, self.mode, self.out = backbone, device, augment, resize_mode, out_size
        self._buf = {}

    def _batch_buffer(self, key: str, n: int) -> torch.Tensor:
        cur = self._buf.get(key)
        if cur is None or cur.shape[0] < n:
            cur = torch.empty((n, 3, self.out, self.out), dtype=torch.uint8, device=self.device)
            self._buf[key] = cur
        return cur[:n]

    def compute(self, batch: dict):
        regions, cj = batch["regions"], batch["cj"]
        b, t = len(regions), regions[0].shape[0]
        orig = self._batch_buffer("orig", b * t)
        flip = self._batch_buffer("hflip", b * t) if self.augment else None
        for i, reg in enumerate(regions):
            dev = (reg.pin_memory() if (self.device.type == "cuda" and not reg.is_pinned()) else reg).to(self.device, non_blocking=True)
            hh, ww = int(reg.shape[1]), int(reg.shape[2])
            F.crop_and_resize_video_uint8(dev, [0, 0, hh, ww], self.out, self.mode, out=orig[i * t:(i + 1) * t])
            if self.augment:
                F.crop_and_resize_video_uint8(dev, [0, 0, hh, ww], self.out, self.mode, hflip=True, out=flip[i * t:(i + 1) * t])
        f_orig = self.bb.features_u8(orig).view(b, t, -1)
        if not self.augment:
            feats = f_orig.unsqueeze(1)
            return feats, [(None, *batch["annots"][0])], batch["box"]
        f_cj = torch.empty_like(f_orig)
        for i in range(b):
            x = F.aug_color_jitter_u8(orig[i * t:(i + 1) * t], cj[i])                     # (T,3,224,224) fp32, normalised
            f_cj[i] = self.bb.features(x)
        f_flip = self.bb.features_u8(flip).view(b, t, -1)
        per_variant = {"orig": f_orig, "cjitter": f_cj, "hflip": f_flip, "trev": f_orig.flip(1)}
        feats = torch.stack([per_variant[name] for name in AUG_NAMES], dim=1)
        return feats, [(None, *a) for a in batch["annots"]], None
