"""Host -> device prefetch of frame batches (the reference's call site copies each batch and then runs the backbone on the
same stream, src/preprocess_resnet_features.py:287-297: at 602 KB per fp32 frame the PCIe copy of a batch takes about as
long as the MI355X needs to process it, so the copy of batch k+1 runs on a side stream while batch k computes)."""
from __future__ import annotations

from typing import Iterator, Optional

import torch


def is_time_reverse_of(video_rev: torch.Tensor, video: torch.Tensor) -> bool:
    """True if ``video_rev`` (B,T,3,H,W) is ``video`` flipped along T.  EVERY frame takes part: all frames are compared on an
    8 x 8 pixel sub-lattice (1/64 of the data: ~12 MB per 1280-frame batch, a few ms on the host), the first and the last
    frame of every clip in full.  A variant that merely shares its end frames with the original, or differs in a few middle
    frames, is not mistaken for the time reverse (its features are then computed, not copied from variant 0)."""
    if video_rev.shape != video.shape or video_rev.dtype != video.dtype or video.dim() != 5:
        return False
    if not (torch.equal(video_rev[:, 0], video[:, -1]) and torch.equal(video_rev[:, -1], video[:, 0])):
        return False
    return torch.equal(video_rev[:, :, :, ::8, ::8], video[:, :, :, ::8, ::8].flip(1))


class DevicePrefetcher:
    """Wraps an iterator of loader batches; yields them with every ``video`` tensor already on ``device`` (copied one
    batch ahead on a side stream, pinned source -> asynchronous).  ``augment``: batches are lists of
    ``(video, joints3d, joints2d, K)`` variants, and with ``skip_trev`` the temporal-reverse variant (index
    ``trev_index``) is not uploaded at all when it is the time reverse of variant 0 -- its video slot becomes ``None``
    (``extract_features`` takes its features from variant 0).  On a CPU device the batches pass through untouched."""

    def __init__(self, it: Iterator, device: torch.device, augment: bool, skip_trev: bool = True, trev_index: int = 3):
        self._it = it
        self._device = device
        self._augment = augment
        self._skip_trev = skip_trev
        self._trev_index = trev_index
        self._cuda = device.type == "cuda"
        self._stream = torch.cuda.Stream(device) if self._cuda else None
        self._next = None
        self._preload()

    def _upload(self, video: torch.Tensor) -> torch.Tensor:
        return video.to(self._device, non_blocking=True)

    def _preload(self) -> None:
        try:
            batch = next(self._it)
        except StopIteration:
            self._next = None
            return
        if not self._cuda:
            self._next = (batch, None)
            return
        with torch.cuda.stream(self._stream):
            if self._augment:
                out = []
                for vi, (video, *rest) in enumerate(batch):
                    if (self._skip_trev and vi == self._trev_index and len(batch) > self._trev_index
                            and is_time_reverse_of(video, batch[0][0])):
                        out.append((None, *rest))
                    else:
                        out.append((self._upload(video), *rest))
                batch = out
            else:
                video, *rest = batch
                batch = (self._upload(video), *rest)
            ev = torch.cuda.Event()
            ev.record(self._stream)
        self._next = (batch, ev)

    def __iter__(self):
        return self

    def __next__(self):
        if self._next is None:
            raise StopIteration
        batch, ev = self._next
        if ev is not None:
            cur = torch.cuda.current_stream(self._device)
            cur.wait_event(ev)                       # compute stream: the copy of THIS batch is done
            vids = [v for v, *_ in batch] if self._augment else [batch[0]]
            for v in vids:
                if v is not None:
                    v.record_stream(cur)             # allocator: the block is in use on the compute stream too
        self._preload()                              # start the next copy now; it overlaps this batch's compute
        return batch
