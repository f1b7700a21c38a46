"""Shard packer, async writer and index for the feature cache — the on-disk contract that
``src/dataset_features.py`` (Human36MFeatureClips, :44-54,89-125) and ``src/train.py`` read.

Behavioural restatement of /root/reference/src/preprocess_resnet_features.py:29-57 (AsyncFileWriter),
:71-131 (shard buffers and the clip-level shuffle pool), :343-396 (final flush) and :403-417
(index.pt).  The layout and the shuffle sequence are kept exactly, so that for the same features,
``--shuffle-seed``, ``--shuffle-pool`` and ``--shard-size`` the files are the same:

shard_XXXXX.pt (legacy, non-zip ``torch.save``)::
    {"feats": (rows,T,2048) fp32|fp16, "joints3d": (rows,T,17,3), "joints2d": (rows,T,17,2),
     "K": (rows,3,3), "meta": [dict]*rows, "n_vars": int}          rows = clips_in_shard * n_vars
index.pt (zip ``torch.save``)::
    {"clips": [{"shard_id","row","subject","action","cam","start","end"}], "n_shards", "n_clips",
     "n_variants", "aug_names", "seq_len", "frame_skip", "feat_dtype", "variants_grouped",
     "shuffle_seed", "shuffle_pool"}

A *group* is the list of a clip's ``n_vars`` variant entries (orig, cjitter, hflip, trev); shuffling
is by group and a group's rows stay contiguous (``row + var_offset`` addressing in the reader).
"""
from __future__ import annotations

import random
from pathlib import Path
from queue import Queue
from threading import Thread
from typing import Any, Dict, List, Optional, Sequence

import torch

AUG_NAMES = ["orig", "cjitter", "hflip", "trev"]
_TENSOR_FIELDS = ("feats", "joints3d", "joints2d", "K")
_ENTRY_KEY = {"feats": "feat", "joints3d": "joints3d", "joints2d": "joints2d", "K": "K"}
_INDEX_FIELDS = ("subject", "action", "cam", "start", "end")


class AsyncFileWriter:
    """Ordered background ``torch.save``: one daemon thread draining a bounded queue.

    ``save`` blocks when ``max_queue_size`` shards are pending (back-pressure on the GPU loop),
    ``wait`` returns once everything queued so far is on disk, ``stop`` ends the thread."""

    def __init__(self, max_queue_size: int = 100):
        self.queue: Queue = Queue(maxsize=max_queue_size)
        self.count = 0
        self.error: Optional[BaseException] = None
        self.thread = Thread(target=self._drain, name="shard-writer", daemon=True)
        self.thread.start()

    def _drain(self) -> None:
        while True:
            job = self.queue.get()
            try:
                if job is None:
                    return
                payload, path = job
                if self.error is None:
                    torch.save(payload, path, _use_new_zipfile_serialization=False)
            except BaseException as exc:      # surfaced by wait(); keep draining so save() never deadlocks
                self.error = exc
            finally:
                self.queue.task_done()

    def save(self, shard_dict: Dict[str, Any], save_path) -> None:
        self.queue.put((shard_dict, save_path))
        self.count += 1

    def wait(self) -> None:
        self.queue.join()
        if self.error is not None:
            raise RuntimeError(f"shard writer failed: {self.error!r}") from self.error

    def stop(self) -> None:
        self.queue.put(None)
        self.thread.join()


def empty_shard_buffer() -> Dict[str, list]:
    return {name: [] for name in (*_TENSOR_FIELDS, "meta")}


def flush_shard(buf: Dict[str, list], sid: int, n_vars: int, out_root: Path, writer: AsyncFileWriter) -> Dict[str, list]:
    """Stack a filled buffer into one shard dict, queue it as shard_{sid:05d}.pt, return a fresh buffer."""
    shard = {name: torch.stack(buf[name]) for name in _TENSOR_FIELDS}
    shard["meta"] = buf["meta"]
    shard["n_vars"] = n_vars
    writer.save(shard, Path(out_root) / f"shard_{sid:05d}.pt")
    return empty_shard_buffer()


def _write_groups(groups: Sequence[List[dict]], shard_id: int, n_vars: int, out_root: Path,
                  writer: AsyncFileWriter, clip_index: List[dict]) -> None:
    """One shard from ``groups``: one index record per clip (pointing at its first variant row), the
    variants' rows appended contiguously."""
    buf = empty_shard_buffer()
    for pos, group in enumerate(groups):
        first_meta = group[0]["meta"]
        record = {"shard_id": shard_id, "row": pos * n_vars}
        record.update({k: first_meta[k] for k in _INDEX_FIELDS})
        clip_index.append(record)
        for entry in group:
            for name in _TENSOR_FIELDS:
                buf[name].append(entry[_ENTRY_KEY[name]])
            buf["meta"].append(entry["meta"])
    flush_shard(buf, shard_id, n_vars, out_root, writer)


def flush_pool_groups_to_shards(pool_groups, carry_over_groups, shard_id, n_vars, out_root, writer, shard_size,
                                clip_index, rng):
    """Shuffle ``carry_over_groups + pool_groups`` (one ``rng.shuffle`` call), write every full shard,
    return ``(next_shard_id, leftover_groups)``.  Same signature and effects as the reference's
    function of this name (:94-131)."""
    mixed = list(carry_over_groups) + list(pool_groups)
    rng.shuffle(mixed)
    full = len(mixed) // shard_size
    for s in range(full):
        _write_groups(mixed[s * shard_size:(s + 1) * shard_size], shard_id + s, n_vars, out_root, writer, clip_index)
    return shard_id + full, mixed[full * shard_size:]


class ShardPacker:
    """Clip-level shuffle pool -> shards -> index.  Feed groups in GLOBAL CLIP ORDER (the shuffle
    sequence of ``random.Random(shuffle_seed)`` depends on arrival order, :98,345)."""

    def __init__(self, out_root, n_vars: int, shard_size: int, shuffle_pool: int, shuffle_seed: int,
                 writer: Optional[AsyncFileWriter] = None):
        if shard_size < 1 or shuffle_pool < 1:
            raise ValueError("shard_size and shuffle_pool must be >= 1")
        self.out_root = Path(out_root)
        self.out_root.mkdir(parents=True, exist_ok=True)
        self.n_vars = n_vars
        self.shard_size = shard_size
        self.shuffle_pool = shuffle_pool
        self.shuffle_seed = shuffle_seed
        self.rng = random.Random(shuffle_seed)
        self.writer = writer if writer is not None else AsyncFileWriter()
        self.pool: List[List[dict]] = []
        self.carry: List[List[dict]] = []
        self.shard_id = 0
        self.clip_index: List[dict] = []
        self.n_clips = 0

    def add_group(self, group: List[dict]) -> None:
        if len(group) != self.n_vars:
            raise ValueError(f"group has {len(group)} variants, expected {self.n_vars}")
        self.pool.append(group)
        self.n_clips += 1
        if len(self.pool) >= self.shuffle_pool:            # checked after every clip, as :325
            self.shard_id, self.carry = flush_pool_groups_to_shards(
                self.pool, self.carry, self.shard_id, self.n_vars, self.out_root, self.writer,
                self.shard_size, self.clip_index, self.rng)
            self.pool = []

    def finish(self) -> None:
        """Final flush (:343-396): shuffle carry + pool once more, full shards, then one partial shard."""
        self.shard_id, rest = flush_pool_groups_to_shards(
            self.pool, self.carry, self.shard_id, self.n_vars, self.out_root, self.writer,
            self.shard_size, self.clip_index, self.rng)
        self.pool, self.carry = [], []
        if rest:
            _write_groups(rest, self.shard_id, self.n_vars, self.out_root, self.writer, self.clip_index)
            self.shard_id += 1

    def write_index(self, *, seq_len: int, frame_skip: int, save_fp16: bool, augment: bool,
                    n_clips: Optional[int] = None) -> Path:
        """Wait for the shards, stop the writer, save index.pt (:399-417)."""
        self.writer.wait()
        self.writer.stop()
        index = {
            "clips": self.clip_index,
            "n_shards": self.shard_id,
            "n_clips": self.n_clips if n_clips is None else n_clips,
            "n_variants": self.n_vars,
            "aug_names": AUG_NAMES if augment else ["orig"],
            "seq_len": seq_len,
            "frame_skip": frame_skip,
            "feat_dtype": "float16" if save_fp16 else "float32",
            "variants_grouped": True,
            "shuffle_seed": self.shuffle_seed,
            "shuffle_pool": self.shuffle_pool,
        }
        path = self.out_root / "index.pt"
        torch.save(index, path)
        return path
