"""Synthetic stand-in for the reference's clip dataset (``Human36MPreprocessedClips``,
/root/reference/src/dataset.py:211-437) — same item contract, no video decode, no H36M files.

Used by the CLI's ``--synthetic-clips`` mode, the tests and the multi-GPU rehearsal: the real frame
producer (mp4 decode -> crop -> resize -> normalise) is upstream of the hot path and out of scope
(SURVEY.md §8f #1); what the hot path needs from it is only this contract:

* ``ds.index[i]`` has ``subject, action, cam, start, end`` (dataset.py:288-302);
* ``ds[i]`` -> ``(video (T,3,224,224) fp32 normalised, joints3d (T,17,3), joints2d (T,17,2),
  K (3,3), box (4,) int64)`` or, with ``augment=True``, a list of 4 such 4-tuples without box
  (orig, cjitter, hflip, trev; dataset.py:411-437).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List

import torch
from torch.utils.data import Dataset

_MEAN = torch.tensor([0.485, 0.456, 0.406], dtype=torch.float32).view(1, 3, 1, 1)
_STD = torch.tensor([0.229, 0.224, 0.225], dtype=torch.float32).view(1, 3, 1, 1)


@dataclass
class SyntheticClipIndex:
    subject: int
    action: str
    cam: str
    start: int
    end: int


class SyntheticClips(Dataset):
    """Deterministic clips: item ``i`` depends only on ``(seed, i)``."""

    def __init__(self, n_clips: int, seq_len: int = 40, subjects=(1, 5, 6, 7, 8, 9, 11), augment: bool = False,
                 seed: int = 0, stride: int = 5):
        self.seq_len = seq_len
        self.augment = augment
        self.seed = seed
        actions = ["Directions", "Discussion", "Eating", "Greeting", "Walking"]
        self.index: List[SyntheticClipIndex] = []
        for i in range(n_clips):
            start = (i // (len(subjects) * 2)) * stride
            self.index.append(SyntheticClipIndex(subject=int(subjects[i % len(subjects)]),
                                                 action=actions[(i // len(subjects)) % len(actions)],
                                                 cam=f"cam_{i % 4}", start=start, end=start + seq_len))

    def __len__(self) -> int:
        return len(self.index)

    def _base(self, i: int):
        g = torch.Generator().manual_seed(self.seed * 1_000_003 + i)
        t = self.seq_len
        u8 = torch.randint(0, 256, (t, 3, 224, 224), generator=g, dtype=torch.uint8)
        video = u8.to(torch.float32) / 255.0                       # [0,1], before normalisation
        joints3d = torch.randn((t, 17, 3), generator=g) * 500.0   # mm
        joints2d = torch.rand((t, 17, 2), generator=g) * 224.0
        k = torch.tensor([[1145.0, 0.0, 112.0], [0.0, 1144.0, 112.0], [0.0, 0.0, 1.0]]) + \
            torch.rand((3, 3), generator=g) * 1e-3
        top, left = int(torch.randint(0, 300, (1,), generator=g)), int(torch.randint(0, 300, (1,), generator=g))
        side = int(torch.randint(200, 600, (1,), generator=g))
        box = torch.tensor([top, left, side, side], dtype=torch.int64)
        return video, joints3d, joints2d, k, box

    @staticmethod
    def _norm(video01: torch.Tensor) -> torch.Tensor:
        return (video01 - _MEAN) / _STD

    def __getitem__(self, i: int):
        video, j3d, j2d, k, box = self._base(i)
        if not self.augment:
            return self._norm(video), j3d, j2d, k, box
        variants = [(self._norm(video), j3d, j2d, k)]
        jitter = (video * 0.9 + 0.05).clamp(0.0, 1.0)              # photometric stand-in for ColorJitter
        variants.append((self._norm(jitter), j3d, j2d, k))
        j2d_f = j2d.clone(); j2d_f[..., 0] = 223.0 - j2d_f[..., 0]
        j3d_f = j3d.clone(); j3d_f[..., 0] = -j3d_f[..., 0]
        k_f = k.clone(); k_f[0, 2] = 223.0 - k_f[0, 2]
        variants.append((self._norm(torch.flip(video, dims=[3])), j3d_f, j2d_f, k_f))
        variants.append((self._norm(torch.flip(video, dims=[0])), torch.flip(j3d, dims=[0]), torch.flip(j2d, dims=[0]), k))
        return variants


class SyntheticDecodedClips(Dataset):
    """Synthetic stand-in one step further upstream: what the reference's dataset holds right AFTER the video decode
    (src/dataset.py:370-393) -- uint8 (T,H,W,3) frames, raw joints and camera -- plus the reference's own host-side producer
    (``__getitem__``: box -> crop -> resize -> /255 -> variants -> Normalize, :395-437) restated with torch ops, so that the
    CLI's ``--device-producer`` path can be compared with the host-producer path on the same decoded clips.

    The host resize is ATen's native uint8 bilinear kernel (``F.interpolate`` on a uint8 tensor = what torchvision's v2 ``resize``
    dispatches to; ``frames.RESIZE_FIXED`` on the device).  ColorJitter lives in torchvision, which is absent here: the host path
    takes it as ``cjitter_fn(video01 (T,3,H,W), params) -> video01`` (the tests plug the oracle's restatement in); both paths use
    the SAME per-clip draw (``decoded_item(i)["cj_params"]``, seeded by the clip index)."""

    def __init__(self, n_clips: int, seq_len: int = 40, subjects=(1, 5, 6, 7, 8, 9, 11), augment: bool = False, seed: int = 0,
                 stride: int = 5, height: int = 260, width: int = 300, cjitter_fn=None):
        self.seq_len, self.augment, self.seed, self.h, self.w, self.cjitter_fn = seq_len, augment, seed, height, width, cjitter_fn
        self.crop_scale = 1.6
        self.index = SyntheticClips(n_clips, seq_len=seq_len, subjects=subjects, seed=seed, stride=stride).index

    def __len__(self) -> int:
        return len(self.index)

    def decoded_item(self, i: int) -> dict:
        from .frames import sample_color_jitter_params
        g = torch.Generator().manual_seed(self.seed * 1_000_003 + 7919 * i + 1)
        t = self.seq_len
        frames = torch.randint(0, 256, (t, self.h, self.w, 3), generator=g, dtype=torch.uint8)
        # a "person": joints scattered around a centre that drifts over the clip; some clips hug the image border (box clipping)
        cx = float(torch.rand(1, generator=g)) * self.w
        cy = float(torch.rand(1, generator=g)) * self.h
        spread = 20.0 + 60.0 * float(torch.rand(1, generator=g))
        joints2d = torch.stack([cx + spread * torch.randn((t, 17), generator=g) * 0.3, cy + spread * torch.randn((t, 17), generator=g) * 0.5], dim=-1)
        joints3d = torch.randn((t, 17, 3), generator=g) * 500.0
        cam = {"f": (1145.0 + torch.rand(2, generator=g)).numpy(), "c": (torch.tensor([self.w / 2.0, self.h / 2.0]) + torch.rand(2, generator=g)).numpy()}
        return {"frames": frames, "joints3d": joints3d, "joints2d": joints2d, "cam": cam, "crop_scale": self.crop_scale,
                "cj_params": sample_color_jitter_params(generator=g)}

    def __getitem__(self, i: int):
        import torch.nn.functional as TF
        from . import frames as F
        it = self.decoded_item(i)
        frames_u8, j3d, j2d_raw = it["frames"], it["joints3d"], it["joints2d"]
        box = F.square_crop_from_2d(j2d_raw, self.h, self.w, scale=self.crop_scale)
        top, left, hh, ww = box.tolist()
        crop = frames_u8.permute(0, 3, 1, 2)[:, :, top:top + hh, left:left + ww]
        video = TF.interpolate(crop, size=(224, 224), mode="bilinear", align_corners=False, antialias=False).to(torch.float32) / 255.0
        j2d = F.adjust_joints2d_after_crop_and_resize(j2d_raw, box, 224)
        k = F.adjust_camera_after_crop_and_resize(it["cam"], box, 224)
        norm = SyntheticClips._norm
        if not self.augment:
            return norm(video), j3d, j2d, k, box
        if self.cjitter_fn is None:
            raise RuntimeError("SyntheticDecodedClips(augment=True): the host-side ColorJitter needs `cjitter_fn` (torchvision is absent)")
        variants = [(norm(video), j3d, j2d, k)]
        variants.append((norm(self.cjitter_fn(video, it["cj_params"])), j3d, j2d, k))
        fj3d, fj2d, fk = F.aug_hflip_annotations(j3d, j2d, k, width=224)
        variants.append((norm(torch.flip(video, dims=[-1])), fj3d, fj2d, fk))
        tj3d, tj2d = F.aug_temporal_reverse_annotations(j3d, j2d)
        variants.append((norm(torch.flip(video, dims=[0])), tj3d, tj2d, k))
        return variants
