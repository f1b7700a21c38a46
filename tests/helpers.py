"""Shared test doubles / comparison helpers (CPU side)."""
from pathlib import Path
from types import SimpleNamespace

import torch

GOLDEN = Path(__file__).resolve().parent / "golden"


class GatherBackbone:
    """Same exact-gather stand-in for the backbone that tests/golden/make_golden.py plugged into the
    REFERENCE's main(): (N,3,224,224) -> (N,2048,1,1) by indexing only."""

    def __call__(self, x):
        j = torch.arange(2048)
        return x[:, j % 3, (j // 3) % 224, (j * 7) % 224].reshape(x.shape[0], 2048, 1, 1)


def cli_args(out, *, seq_len, batch_size, shard_size, shuffle_pool, shuffle_seed, augment, save_fp16, num_workers=0):
    return SimpleNamespace(root="unused", out=str(out), seq_len=seq_len, frame_skip=2, stride=5, batch_size=batch_size,
                           num_workers=num_workers, subjects=[1, 5, 6, 7, 8, 9, 11], device="cpu", save_fp16=save_fp16,
                           augment=augment, shard_size=shard_size, shuffle_pool=shuffle_pool, shuffle_seed=shuffle_seed)


def _same(a, b, path="") -> None:
    assert type(a) is type(b), f"{path}: type {type(a)} vs {type(b)}"
    if isinstance(a, torch.Tensor):
        assert a.dtype == b.dtype and a.shape == b.shape, f"{path}: {a.dtype}{tuple(a.shape)} vs {b.dtype}{tuple(b.shape)}"
        assert torch.equal(a, b), f"{path}: tensor values differ"
    elif isinstance(a, dict):
        assert list(a.keys()) == list(b.keys()), f"{path}: keys {list(a.keys())} vs {list(b.keys())}"
        for k in a:
            _same(a[k], b[k], f"{path}.{k}")
    elif isinstance(a, (list, tuple)):
        assert len(a) == len(b), f"{path}: len {len(a)} vs {len(b)}"
        for i, (x, y) in enumerate(zip(a, b)):
            _same(x, y, f"{path}[{i}]")
    else:
        assert a == b, f"{path}: {a!r} vs {b!r}"


def assert_same_feature_cache(got_dir, want_dir) -> None:
    """index.pt and every shard equal: same keys in the same order, dtypes, shapes, values, meta."""
    got_dir, want_dir = Path(got_dir), Path(want_dir)
    names = sorted(p.name for p in want_dir.iterdir())
    assert sorted(p.name for p in got_dir.iterdir()) == names
    for name in names:
        if not name.endswith(".pt"):             # provenance files written beside the shards (backbone_weights.json): byte for byte
            assert (got_dir / name).read_bytes() == (want_dir / name).read_bytes(), name
            continue
        a = torch.load(got_dir / name, map_location="cpu", weights_only=True)
        b = torch.load(want_dir / name, map_location="cpu", weights_only=True)
        _same(a, b, name)
