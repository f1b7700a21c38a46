"""Golden items of the training-side reader: what the REFERENCE's ``Human36MFeatureClips`` (src/dataset_features.py,
imported from /root/reference; it needs torch only) returns for the golden shard directories ``ref_plain`` / ``ref_aug``
under several constructor configurations.  The file holds the configurations and the returned items only.

    python tests/golden/make_golden_reader.py        # run in the build container (needs /root/reference)
"""
import importlib.util
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = os.environ.get("H36M_REFERENCE_SRC", "/root/reference/src")

CONFIGS = [("ref_plain", {"test_set": True}), ("ref_plain", {"max_clips": 3}), ("ref_aug", {"augment": True}),
           ("ref_aug", {"augment": True, "subjects": [9, 11]})]


def main():
    spec = importlib.util.spec_from_file_location("ref_dataset_features", os.path.join(REF_SRC, "dataset_features.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = []
    for name, kw in CONFIGS:
        try:
            ds = mod.Human36MFeatureClips(os.path.join(HERE, name), **kw)
            items = [tuple(ds[i]) for i in range(len(ds))]
        except RuntimeError as e:
            items = None
        out.append({"dir": name, "kwargs": kw, "items": items})
        print(name, kw, "->", "error" if items is None else len(items), "items")
    torch.save(out, os.path.join(HERE, "reader_golden.pt"))


if __name__ == "__main__":
    main()
