"""Golden vectors for the frame producer's host functions (implementation_phd_lab_vision_amd/frames.py), produced by
the REFERENCE's own functions: ``_compute_square_crop_from_2d``, ``_adjust_joints2d_after_crop_and_resize``,
``_adjust_camera_after_crop_and_resize``, ``_aug_hflip``, ``_aug_temporal_reverse`` of src/dataset.py, imported from /root/reference with ``torchvision`` (absent
from this image, unused by these three functions) replaced by empty stub modules.  Inputs are seeded; the file holds
inputs and expected outputs only.

    python tests/golden/make_golden_producer.py      # run in the build container (needs /root/reference)
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = os.environ.get("H36M_REFERENCE_SRC", "/root/reference/src")


def _import_reference_dataset():
    def stub(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m
    tv = stub("torchvision")
    tv.io = stub("torchvision.io")
    tv.io.VideoReader = object
    tv.transforms = stub("torchvision.transforms")
    tv.transforms.v2 = stub("torchvision.transforms.v2")
    tv.transforms.functional = stub("torchvision.transforms.functional")
    sys.path.insert(0, REF_SRC)
    import dataset as ref_dataset
    return ref_dataset


def main():
    ref = _import_reference_dataset()
    g = torch.Generator().manual_seed(20260104)
    cases = []
    for i in range(24):
        img_h, img_w = [(1000, 1002), (1002, 1000), (480, 640), (224, 224)][i % 4]
        t = 1 + i % 5
        centre = torch.rand(2, generator=g) * torch.tensor([img_w, img_h], dtype=torch.float32)
        spread = 5.0 + 400.0 * torch.rand(1, generator=g).item()
        joints2d = centre + (torch.rand(t, 17, 2, generator=g) - 0.5) * spread
        if i % 6 == 5:
            joints2d = joints2d * 0 + centre                     # degenerate: all joints on one point
        box = ref._compute_square_crop_from_2d(joints2d, img_h, img_w)
        j2 = ref._adjust_joints2d_after_crop_and_resize(joints2d, box, out_size=224)
        cam = {"f": (torch.rand(2, generator=g) * 500 + 900).numpy().astype(np.float64),
               "c": (torch.rand(2, generator=g) * 100 + 450).numpy().astype(np.float64)}
        k = ref._adjust_camera_after_crop_and_resize(cam, box, out_size=224)
        joints3d = torch.randn(t, 17, 3, generator=g) * 500.0
        dummy_video = torch.zeros(t, 3, 2, 224)
        _v, j3_f, j2_f, k_f = ref._aug_hflip(dummy_video, joints3d, j2, k)
        _v, j3_r, j2_r = ref._aug_temporal_reverse(dummy_video, joints3d, j2)
        cases.append({"img_h": img_h, "img_w": img_w, "joints2d": joints2d, "box": box, "joints2d_adjusted": j2,
                      "cam_f": torch.from_numpy(cam["f"]), "cam_c": torch.from_numpy(cam["c"]), "K": k,
                      "joints3d": joints3d, "hflip_j3d": j3_f, "hflip_j2d": j2_f, "hflip_K": k_f,
                      "trev_j3d": j3_r, "trev_j2d": j2_r})
    torch.save(cases, os.path.join(HERE, "producer_golden.pt"))
    print("wrote", len(cases), "cases; boxes:", [c["box"].tolist() for c in cases[:6]])


if __name__ == "__main__":
    main()
