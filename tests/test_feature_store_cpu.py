"""HBM-resident feature store against the reference's own reader class, imported from /root/reference when
it is there (this container) and against the invariants of the golden shards otherwise."""
import os
import sys

import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden")
REF_SRC = os.environ.get("H36M_REFERENCE_SRC", "/root/reference/src")


def _reference_class():
    path = os.path.join(REF_SRC, "dataset_features.py")
    if not os.path.exists(path):
        return None
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_dataset_features", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)                     # needs torch only
    return mod.Human36MFeatureClips


def _same(a, b):
    if isinstance(a, torch.Tensor):
        return torch.equal(a.cpu(), b.cpu())
    if isinstance(a, dict):
        return a.keys() == b.keys() and all(_same(a[k], b[k]) for k in a)
    return a == b


@pytest.mark.parametrize("name,kw", [("ref_plain", {}), ("ref_plain", {"test_set": True}), ("ref_plain", {"subjects": [1, 5]}),
                                     ("ref_plain", {"max_clips": 3}), ("ref_aug", {"augment": True}),
                                     ("ref_aug", {"augment": False, "test_set": True}), ("ref_aug", {"augment": True, "subjects": [9, 11]})],
                         ids=lambda v: str(v).replace(" ", ""))
def test_store_items_equal_reference_reader(name, kw):
    from implementation_phd_lab_vision_amd.feature_store import DeviceFeatureStore
    ref_cls = _reference_class()
    if ref_cls is None:
        pytest.skip("reference sources not present on this machine")
    root = os.path.join(GOLD, name)
    try:
        ref = ref_cls(root, **kw)
    except RuntimeError:
        with pytest.raises(RuntimeError):
            DeviceFeatureStore(root, device="cpu", **kw)
        return
    store = DeviceFeatureStore(root, device="cpu", **kw)
    assert len(store) == len(ref) and len(store) > 0
    for i in range(len(ref)):
        a, b = store[i], ref[i]
        assert len(a) == len(b) and all(_same(x, y) for x, y in zip(a, b)), f"item {i}"
    # a batch == the default collate of the reference's items
    idx = list(range(len(ref)))[::-1][: max(1, len(ref) // 2 + 1)]
    batch = store.get_batch(idx)
    items = [ref[i] for i in idx]
    for j in range(4):
        assert torch.equal(batch[j], torch.stack([it[j] for it in items])), f"batch field {j}"
    if kw.get("test_set"):
        assert all(_same(m, it[4]) for m, it in zip(batch[4], items))


def test_store_items_equal_committed_reference_items():
    """The same comparison against the committed fixture (written by the reference's reader, make_golden_reader.py), so it
    also runs where /root/reference does not exist."""
    from implementation_phd_lab_vision_amd.feature_store import DeviceFeatureStore
    cases = torch.load(os.path.join(GOLD, "reader_golden.pt"), weights_only=True)
    assert len(cases) >= 4
    for case in cases:
        root = os.path.join(GOLD, case["dir"])
        if case["items"] is None:
            with pytest.raises(RuntimeError):
                DeviceFeatureStore(root, device="cpu", **case["kwargs"])
            continue
        store = DeviceFeatureStore(root, device="cpu", **case["kwargs"])
        assert len(store) == len(case["items"])
        for i, want in enumerate(case["items"]):
            got = store[i]
            assert len(got) == len(want) and all(_same(x, y) for x, y in zip(got, want)), (case["dir"], case["kwargs"], i)


def test_store_batches_cover_every_item_once():
    from implementation_phd_lab_vision_amd.feature_store import DeviceFeatureStore
    store = DeviceFeatureStore(os.path.join(GOLD, "ref_aug"), augment=True, device="cpu")
    seen = 0
    feats_sum = torch.zeros(2048, dtype=torch.float64)
    for feats, j3d, j2d, k in store.batches(3, shuffle=True, seed=5):
        assert feats.shape[0] == j3d.shape[0] == j2d.shape[0] == k.shape[0] <= 3
        seen += feats.shape[0]
        feats_sum += feats.double().sum(dim=(0, 1))
    assert seen == len(store)
    assert torch.allclose(feats_sum, store.feats[store._row].double().sum(dim=(0, 1)))
    assert store.nbytes > 0


def test_missing_index_is_an_error(tmp_path):
    from implementation_phd_lab_vision_amd.feature_store import DeviceFeatureStore
    with pytest.raises(RuntimeError):
        DeviceFeatureStore(str(tmp_path), device="cpu")
