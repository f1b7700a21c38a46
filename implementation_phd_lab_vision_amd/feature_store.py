"""HBM-resident feature store: the training-side reader of the shard files.

The reference's ``Human36MFeatureClips`` (src/dataset_features.py:29-126) keeps an LRU of two shards per DataLoader
worker and ``torch.load``s a 168-MB shard whenever a sampled clip falls outside it; its training loop measures that as
``data`` time (src/train.py:137,208).  On an MI355X the whole pre-extracted dataset is small against 288 GB of HBM (H36M:
~1.5 M frames x 2048 fp32 = 12 GB per variant), so this reader loads every needed shard ONCE, keeps
``feats / joints3d / joints2d / K`` as four device tensors, and serves a batch as four ``index_select``s on the device:
no worker processes, no host collate, no per-step H2D copy.  Same files, same filtering and item order, same values:
``store[i]`` equals the reference dataset's item ``i`` (tests/test_feature_store_cpu.py checks it against the reference
class itself on the golden shards), and ``get_batch(idx)`` equals the default collate of those items.
"""
from __future__ import annotations

from pathlib import Path
from typing import List, Optional, Sequence, Union

import torch


class DeviceFeatureStore:
    def __init__(self, root: str, subjects: Optional[List[int]] = None, max_clips: Optional[int] = None,
                 test_set: bool = False, augment: bool = False, device: Union[str, torch.device] = "cuda"):
        self.root = Path(root)
        self.test_set = test_set
        self.augment = augment
        self.device = torch.device(device)
        index_path = self.root / "index.pt"
        if not index_path.exists():
            raise RuntimeError(f"index.pt not found in {root}. Run preprocess_resnet_features.py first.")
        idx = torch.load(index_path, map_location="cpu", weights_only=True)
        self._n_vars = idx["n_variants"]
        self._aug_names = idx.get("aug_names", ["orig"])
        clips = idx["clips"]
        if subjects is not None:                                    # same filtering, same order (:58-66)
            keep = set(subjects)
            clips = [c for c in clips if c["subject"] in keep]
        if max_clips is not None:
            clips = clips[:max_clips]
        if len(clips) == 0:
            raise RuntimeError(f"No clips found in {root} for subjects={subjects}.")
        self._clips = clips
        variants = range(self._n_vars) if augment else (0,)          # item list as :76-81
        self._items = [(c, v) for c in clips for v in variants]

        # every shard that holds a kept clip, loaded once, concatenated row-wise on the device
        shard_ids = sorted({c["shard_id"] for c in clips})
        base, parts, metas, rows = {}, {"feats": [], "joints3d": [], "joints2d": [], "K": []}, [], 0
        for sid in shard_ids:
            shard = torch.load(self.root / f"shard_{sid:05d}.pt", map_location="cpu", weights_only=True)
            base[sid] = rows
            rows += shard["feats"].shape[0]
            for k in parts:
                t = shard[k]
                if k == "joints3d":
                    t = t / 1000.0          # mm -> m (:118) ON THE HOST, once: fp32 division on the device is not correctly rounded
                parts[k].append(t.to(self.device, non_blocking=True))
            if test_set:
                metas.extend(shard["meta"])
        self.feats = torch.cat(parts["feats"], dim=0)
        self.joints3d = torch.cat(parts["joints3d"], dim=0)
        self.joints2d = torch.cat(parts["joints2d"], dim=0)
        self.K = torch.cat(parts["K"], dim=0)
        self._metas = metas
        self._row = torch.tensor([base[c["shard_id"]] + c["row"] + v for c, v in self._items], dtype=torch.long,
                                 device=self.device)
        self._row_host = self._row.cpu().tolist()

    def __len__(self) -> int:
        return len(self._items)

    @property
    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in (self.feats, self.joints3d, self.joints2d, self.K))

    def __getitem__(self, idx: int):
        """The reference dataset's item (src/dataset_features.py:112-126), tensors on ``self.device``."""
        row = self._row_host[idx]
        out = (self.feats[row], self.joints3d[row], self.joints2d[row], self.K[row])       # joints3d already in metres
        return out + (self._metas[row],) if self.test_set else out

    def get_batch(self, idx: Union[Sequence[int], torch.Tensor]):
        """Items ``idx`` collated: ``(feats (B,T,2048), joints3d (B,T,17,3) in metres, joints2d (B,T,17,2), K (B,3,3))``
        on the device [+ list of meta dicts with ``test_set``]."""
        idx_t = torch.as_tensor(idx, dtype=torch.long, device=self.device)
        rows = self._row.index_select(0, idx_t)
        out = (self.feats.index_select(0, rows), self.joints3d.index_select(0, rows),
               self.joints2d.index_select(0, rows), self.K.index_select(0, rows))
        if self.test_set:
            return out + ([self._metas[r] for r in rows.cpu().tolist()],)
        return out

    def batches(self, batch_size: int, shuffle: bool = False, seed: int = 0, drop_last: bool = False):
        """One epoch of device-resident batches (``torch.randperm`` with a seeded generator when ``shuffle``)."""
        n = len(self)
        order = torch.randperm(n, generator=torch.Generator().manual_seed(seed)) if shuffle else torch.arange(n)
        for s in range(0, n, batch_size):
            sel = order[s:s + batch_size]
            if drop_last and sel.numel() < batch_size:
                break
            yield self.get_batch(sel)
