"""Multi-GPU sharding of the feature path: one process per GPU, frames sharded by batch, features
returned to rank 0 with one gather per round (RCCL over xGMI on MI355X; gloo in the CPU tests).

Replaces the reference's ``nn.DataParallel`` wrap (/root/reference/src/preprocess_resnet_features.py:
214-217), whose implicit per-call scatter (771 MB of fp32 frames out of GPU 0), module replicate
and gather disappear: every rank decodes/loads its own batches and holds its own packed weights;
the only exchange is the (n, V, T, 2048) fp32 feature block per rank per round (8 KB per frame).

Sharding: global batch ``g`` (``--batch-size`` consecutive clips) belongs to rank ``g % world`` -- ROUND-ROBIN BATCHES, not the
contiguous clip ranges SURVEY.md section 8(e) first proposed: with contiguous ranges rank 0 would receive clip 0.. from itself and
clip C/G.. from rank 1 at the same time, and would have to buffer whole ranges to restore the global order; with round-robin
batches every round's ``world`` blocks, taken in rank order, ARE the next ``world`` batches of the global clip order, which is what
the shuffle pool's RNG sequence requires (:98,300,345) -- so the shards equal the single-GPU run's for the same ``--shuffle-seed``.
In round ``q`` rank ``r`` processes batch ``q * world + r``.  The exchange itself (``RoundExchange``) is one asynchronous gather
of one flat block per rank per round (features + annotations + count), overlapped with the next round's forward passes; rank 0's
D2H copies go to pinned memory on a side stream and the packing runs in a worker thread.
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
        """A process group exists: more than one rank, or ONE rank started by torchrun (RANK and WORLD_SIZE = 1 in the environment).
        The one-rank group runs exactly the multi-rank code -- ``init_process_group("nccl", device_id=...)``, the asynchronous gather on
        device tensors, the side-stream ordering behind it -- so the RCCL path executes on a one-GPU box (tests/test_cli_gpu.py)."""
        return self.backend is not None


def single_rank_group_requested(env=None) -> bool:
    """ONE rank joins a process group only when a launcher described a complete rendezvous -- RANK, WORLD_SIZE and MASTER_PORT all present, as
    torchrun sets them -- and ``R50_SINGLE_RANK_GROUP`` is not "0".  A scheduler or container that merely exports RANK=0 WORLD_SIZE=1 (no
    MASTER_PORT) gets a plain single process: no env:// rendezvous to fail in, no RCCL init, no gloo side group."""
    env = os.environ if env is None else env
    return all(k in env for k in ("RANK", "WORLD_SIZE", "MASTER_PORT")) and env.get("R50_SINGLE_RANK_GROUP", "1") != "0"


def init_from_env(use_gpu: bool) -> RankContext:
    """Join the process group torchrun described (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*).
    backend "nccl" IS RCCL on ROCm; CPU rehearsals use gloo."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 and not single_rank_group_requested():
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


@dataclass
class BlockLayout:
    """One rank's contribution to a round, as ONE flat fp32 buffer (a single collective moves everything):

        [count, 0, 0, 0 | feats (B,V,T,2048) | joints3d (B,V,T,J,3) | joints2d (B,V,T,J,2) | K (B,V,3,3) | box (B,4)]

    ``count`` = valid clips of the block (the globally last batch can be short; a rank without a batch this round sends 0).
    The annotations are fp32 in the reference's dataset (src/dataset.py:64-65,135); the crop box is int64 with values far
    below 2^24, so it travels exactly as fp32 and is restored to int64."""
    batch: int
    n_vars: int
    seq_len: int
    n_joints: int = 17
    feat_dim: int = 2048

    def __post_init__(self):
        b, v, t, j = self.batch, self.n_vars, self.seq_len, self.n_joints
        sizes = [("hdr", 4), ("feats", b * v * t * self.feat_dim), ("joints3d", b * v * t * j * 3),
                 ("joints2d", b * v * t * j * 2), ("K", b * v * 9), ("box", b * 4)]
        self.offsets = {}
        pos = 0
        for name, n in sizes:
            self.offsets[name] = (pos, pos + n)
            pos += n
        self.total = pos
        self.shapes = {"feats": (b, v, t, self.feat_dim), "joints3d": (b, v, t, j, 3), "joints2d": (b, v, t, j, 2),
                       "K": (b, v, 3, 3), "box": (b, 4)}

    def view(self, flat: torch.Tensor, name: str) -> torch.Tensor:
        lo, hi = self.offsets[name]
        return flat[lo:hi] if name == "hdr" else flat[lo:hi].view(self.shapes[name])

    def pack(self, flat: torch.Tensor, feats: Optional[torch.Tensor], variants_batch, box_batch) -> None:
        """Fill ``flat`` (on the compute device) from this rank's batch: ``feats`` (n,V,T,2048) on the device, the loader's
        host-side annotations copied in asynchronously.  ``feats`` None = no batch this round."""
        n = 0 if feats is None else int(feats.shape[0])
        self.view(flat, "hdr")[:1].fill_(float(n))
        if n == 0:
            return
        if tuple(feats.shape[1:]) != self.shapes["feats"][1:] or n > self.batch:
            raise RuntimeError(f"feature block {tuple(feats.shape)} does not fit the exchange layout {self.shapes['feats']}")
        self.view(flat, "feats")[:n].copy_(feats)
        for v, (_video, j3d, j2d, k) in enumerate(variants_batch):
            for name, src in (("joints3d", j3d), ("joints2d", j2d), ("K", k)):
                if src.dtype != torch.float32:
                    raise RuntimeError(f"{name} is {src.dtype}; the exchange block (and the reference's dataset) carries fp32 annotations")
                self.view(flat, name)[:n, v].copy_(src, non_blocking=True)
        if box_batch is not None:
            if box_batch.dtype != torch.int64 or int(box_batch.abs().max()) >= (1 << 24):
                raise RuntimeError("crop boxes must be int64 with values below 2^24")
            self.view(flat, "box")[:n].copy_(box_batch.to(torch.float32), non_blocking=True)

    def unpack(self, flat_host: torch.Tensor, feat_dtype, has_box: bool) -> dict:
        """Host side (rank 0): fresh tensors (the buffer behind ``flat_host`` is recycled) trimmed to the block's count."""
        n = int(self.view(flat_host, "hdr")[0].item())
        out = {"count": n}
        out["feats"] = self.view(flat_host, "feats")[:n].to(feat_dtype, copy=True)
        for name in ("joints3d", "joints2d", "K"):
            out[name] = self.view(flat_host, name)[:n].clone()
        out["box"] = self.view(flat_host, "box")[:n].to(torch.int64) if has_box else None
        return out


class RoundExchange:
    """The one exchange step per round, taken off the critical path.

    * buffers are allocated ONCE: two device slots (send block; on rank 0 also the ``world`` receive blocks) used alternately,
      and ``host_slots`` pinned host blocks;
    * ``post(q, ...)`` packs round q's block and starts its gather with ``async_op=True`` (RCCL's own stream on GPUs): it runs
      while round q+1 computes;
    * ``collect(q)`` (called one round later) orders a side stream behind that gather, copies the received blocks to a pinned host
      slot asynchronously and hands (slot, event) to the consumer thread -- rank 0's compute stream never waits for a D2H copy, and
      the Python-side unpacking / group building / packing runs in that thread, in global clip order;
    * the host only ever blocks when every host slot is still owned by the consumer (back-pressure, bounded memory).

    The reference's ``feats.cpu()`` (:297) and the per-clip packing loop (:299-330) sit between two forward passes; here neither does."""

    def __init__(self, ctx: RankContext, layout: BlockLayout, device: torch.device, consume, host_slots: int = 4):
        import queue
        import threading
        self.ctx, self.layout, self.device = ctx, layout, device
        self.cuda = device.type == "cuda"
        self.send = [torch.zeros(layout.total, dtype=torch.float32, device=device) for _ in range(2)]
        self.recv = self.host = None
        self.pending = {}                       # round -> (slot, work)
        self.copied = [None, None]              # per device slot: event of the last D2H copy that read it
        self.packed = [None, None]              # per device slot: event behind the packing kernels of the round that filled it
        self.error: Optional[BaseException] = None
        if ctx.is_root:
            self.recv = [torch.zeros((ctx.world, layout.total), dtype=torch.float32, device=device) for _ in range(2)] \
                if ctx.distributed else [s.view(1, -1) for s in self.send]
            self.host = [torch.zeros((ctx.world, layout.total), dtype=torch.float32, pin_memory=self.cuda) for _ in range(host_slots)]
            self.free = queue.Queue()
            for i in range(host_slots):
                self.free.put(i)
            self.ready = queue.Queue()
            self.side = torch.cuda.Stream(device) if self.cuda else None
            self.consume = consume
            self.thread = threading.Thread(target=self._drain, name="round-packer", daemon=True)
            self.thread.start()

    def _drain(self) -> None:
        while True:
            item = self.ready.get()
            if item is None:
                return
            q, hs, ev = item
            try:
                if self.error is None:
                    if ev is not None:
                        ev.synchronize()
                    self.consume(q, self.host[hs])
            except BaseException as exc:          # surfaced by the main thread at its next call; keep draining so it never deadlocks
                self.error = exc
            finally:
                self.free.put(hs)

    def _check(self) -> None:
        if self.error is not None:
            raise RuntimeError(f"packing thread failed: {self.error!r}") from self.error

    def post(self, q: int, feats, variants_batch, box_batch) -> None:
        self._check()
        s = q & 1
        if q - 2 in self.pending:
            raise RuntimeError("collect(q - 2) must come before post(q): its device slot is being reused")
        if self.cuda and self.copied[s] is not None:
            torch.cuda.current_stream(self.device).wait_event(self.copied[s])     # slot s: round q-2's D2H copy has read it
        self.layout.pack(self.send[s], feats, variants_batch, box_batch)
        if self.cuda:                            # round q's block is complete HERE on the compute stream (not one round of kernels later)
            self.packed[s] = torch.cuda.Event()
            self.packed[s].record(torch.cuda.current_stream(self.device))
        work = None
        if self.ctx.distributed:                 # issued behind the packing kernels (the collective orders itself after the current stream)
            work = dist.gather(self.send[s], list(self.recv[s].unbind(0)) if self.ctx.is_root else None, dst=0, async_op=True)
        self.pending[q] = (s, work)

    def collect(self, q: int) -> None:
        """Stream ordering (round-2 ADVICE): the D2H copy of round q waits for round q's GATHER only -- the side stream is ordered
        directly behind the collective (``work.wait()`` with the side stream current) or, without a process group, behind the event
        recorded after the packing kernels -- never behind round q+1's forward pass, which the compute stream has queued meanwhile.
        The compute stream meets these buffers again in ``post(q + 2)``, which waits for ``copied[s]`` (the copy, hence the gather,
        of round q is done: send[s] and recv[s] are free).  A non-root rank only has send[s] to protect: it orders its compute
        stream behind its (8 KB-per-frame) gather here."""
        self._check()
        s, work = self.pending.pop(q)
        if not self.ctx.is_root:
            if work is not None:
                work.wait()                      # GPU: orders the current stream behind the collective (no host block); gloo: blocks
            return
        hs = self.free.get()                     # blocks only when the consumer is `host_slots` rounds behind
        ev = None
        if self.cuda:
            with torch.cuda.stream(self.side):
                if work is not None:
                    work.wait()                  # the SIDE stream waits for the collective; the compute stream does not
                else:
                    self.side.wait_event(self.packed[s])
                self.host[hs].copy_(self.recv[s], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self.side)
            self.copied[s] = ev
        else:
            if work is not None:
                work.wait()
            self.host[hs].copy_(self.recv[s])
        self.ready.put((q, hs, ev))

    def finish(self) -> None:
        """All posted rounds collected: wait for the consumer thread and surface its error, if any."""
        if self.pending:
            raise RuntimeError(f"rounds {sorted(self.pending)} were posted but never collected")
        if self.ctx.is_root:
            self.ready.put(None)
            self.thread.join()
        self._check()
