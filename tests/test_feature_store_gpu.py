"""HBM-resident feature store on the MI355X: same items as the host-resident store, batches served from device memory."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_device_store_equals_host_store():
    from implementation_phd_lab_vision_amd.feature_store import DeviceFeatureStore
    root = os.path.join(GOLD, "ref_aug")
    host = DeviceFeatureStore(root, augment=True, test_set=True, device="cpu")
    dev = DeviceFeatureStore(root, augment=True, test_set=True, device="cuda:0")
    assert len(dev) == len(host) and dev.feats.is_cuda
    for i in range(len(host)):
        a, b = dev[i], host[i]
        assert all(torch.equal(x.cpu(), y) for x, y in zip(a[:4], b[:4])), f"item {i}"
        assert a[4].keys() == b[4].keys()
    idx = torch.tensor([5, 0, 11, 3, 3])
    gb, hb = dev.get_batch(idx), host.get_batch(idx)
    for j in range(4):
        assert gb[j].is_cuda and torch.equal(gb[j].cpu(), hb[j])
    n = sum(b[0].shape[0] for b in dev.batches(4, shuffle=True, seed=1))
    assert n == len(dev)
