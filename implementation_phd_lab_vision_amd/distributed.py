"""Multi-GPU sharding of the feature path: one process per GPU, frames sharded by batch, features
returned to rank 0 with one gather per round (RCCL over xGMI on MI355X; gloo in the CPU tests).

Replaces the reference's ``nn.DataParallel`` wrap (/root/reference/src/preprocess_resnet_features.py:
214-217), whose implicit per-call scatter (771 MB of fp32 frames out of GPU 0), module replicate
and gather disappear: every rank decodes/loads its own batches and holds its own packed weights;
the only exchange is the (n, V, T, 2048) fp32 feature block per rank per round (8 KB per frame).

Sharding: global batch ``g`` (``--batch-size`` consecutive clips) belongs to rank ``g % world``.
In round ``q`` rank ``r`` processes batch ``q * world + r``; rank 0 appends the gathered blocks in
rank order, i.e. in GLOBAL CLIP ORDER, which is what the shuffle pool's RNG sequence requires
(:98,300,345) — so the shards equal the single-GPU run's for the same ``--shuffle-seed``.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Any, List, Optional, Sequence

import torch
import torch.distributed as dist


@dataclass
class RankContext:
    rank: int = 0
    world: int = 1
    local_rank: int = 0
    backend: Optional[str] = None
    side_group: Any = None        # gloo group for small host-side objects (joints, K, box)

    @property
    def is_root(self) -> bool:
        return self.rank == 0

    @property
    def distributed(self) -> bool:
        return self.world > 1


def init_from_env(use_gpu: bool) -> RankContext:
    """Join the process group torchrun described (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*).
    backend "nccl" IS RCCL on ROCm; CPU rehearsals use gloo."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return RankContext()
    rank = int(os.environ["RANK"])
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    backend = "nccl" if use_gpu else "gloo"
    if not dist.is_initialized():
        if use_gpu:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    side = dist.new_group(backend="gloo") if backend != "gloo" else dist.group.WORLD
    return RankContext(rank=rank, world=world, local_rank=local_rank, backend=backend, side_group=side)


def shutdown(ctx: RankContext) -> None:
    if ctx.distributed and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def n_batches(n_clips: int, batch_size: int) -> int:
    return (n_clips + batch_size - 1) // batch_size


def n_rounds(n_clips: int, batch_size: int, world: int) -> int:
    return (n_batches(n_clips, batch_size) + world - 1) // world


def rank_clip_indices(n_clips: int, batch_size: int, rank: int, world: int) -> List[int]:
    """Clip indices rank ``rank`` loads, in processing order: its batches g = rank, rank+world, ...
    Feeding this list to a DataLoader(batch_size=batch_size, shuffle=False) reproduces the batch
    boundaries of the global loader (only the globally last batch can be short)."""
    out: List[int] = []
    for g in range(rank, n_batches(n_clips, batch_size), world):
        out.extend(range(g * batch_size, min((g + 1) * batch_size, n_clips)))
    return out


def batch_clip_range(g: int, n_clips: int, batch_size: int) -> range:
    return range(g * batch_size, min((g + 1) * batch_size, n_clips))


def gather_features(ctx: RankContext, feats: Optional[torch.Tensor], block_shape: Sequence[int],
                    device: torch.device) -> Optional[List[torch.Tensor]]:
    """One exchange step.  ``feats``: this rank's (n, V, T, 2048) fp32 block on ``device`` (``None`` or
    n = 0 when the rank has no batch this round); ``block_shape`` = (B, V, T, 2048) of a full batch.
    Every rank sends a fixed-size block plus its valid count; rank 0 receives the list of per-rank
    blocks trimmed to their counts (host tensors), other ranks get ``None``."""
    if not ctx.distributed:
        return [feats.cpu()] if feats is not None else [torch.empty((0, *block_shape[1:]))]
    send = torch.zeros(tuple(block_shape), dtype=torch.float32, device=device)
    n = 0
    if feats is not None and feats.shape[0] > 0:
        n = feats.shape[0]
        send[:n].copy_(feats)
    count = torch.tensor([n], dtype=torch.int64, device=device)
    if ctx.is_root:
        blocks = [torch.empty_like(send) for _ in range(ctx.world)]
        counts = [torch.empty_like(count) for _ in range(ctx.world)]
        dist.gather(send, blocks, dst=0)
        dist.gather(count, counts, dst=0)
        return [b[: int(c.item())].cpu() for b, c in zip(blocks, counts)]
    dist.gather(send, None, dst=0)
    dist.gather(count, None, dst=0)
    return None


def gather_objects(ctx: RankContext, obj: Any) -> Optional[List[Any]]:
    """Small host-side payloads (joints3d/2d, K, box of a batch) to rank 0 over the gloo side group."""
    if not ctx.distributed:
        return [obj]
    if ctx.is_root:
        out: List[Any] = [None] * ctx.world
        dist.gather_object(obj, out, dst=0, group=ctx.side_group)
        return out
    dist.gather_object(obj, None, dst=0, group=ctx.side_group)
    return None
