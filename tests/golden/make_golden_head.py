"""Golden outputs of the lifting head: the REFERENCE module ``PHDFor3DJoints`` (src/model.py, imported from /root/reference
with ``torchvision`` -- unused by the head -- replaced by an empty stub) in eval mode, loaded with the seeded weights of
``oracle.lifting_oracle.synthetic_head_state_dict`` and run on seeded features.  The fixture holds the configurations,
the input features and the module's outputs (the weights are regenerated from their seed by the test).

    python tests/golden/make_golden_head.py         # run in the build container (needs /root/reference)
"""
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_SRC = os.environ.get("H36M_REFERENCE_SRC", "/root/reference/src")
sys.path.insert(0, ROOT)


def main():
    tv = types.ModuleType("torchvision"); tv.models = types.ModuleType("torchvision.models")
    sys.modules["torchvision"] = tv; sys.modules["torchvision.models"] = tv.models
    sys.path.insert(0, REF_SRC)
    import model as ref_model
    from oracle.lifting_oracle import synthetic_head_state_dict
    cases = []
    for latent, blocks, b, t, seed in ((64, 2, 2, 5, 1), (128, 2, 3, 4, 2), (64, 3, 1, 7, 3)):
        m = ref_model.PHDFor3DJoints(latent_dim=latent, joints_num=17, number_blocks=blocks).eval()
        sd = synthetic_head_state_dict(latent, blocks, seed)
        missing = m.load_state_dict(sd, strict=True)
        g = torch.Generator().manual_seed(100 + seed)
        feats = torch.randn(b, t, 2048, generator=g).abs()            # ResNet features are post-ReLU means: non-negative
        with torch.no_grad():
            phi, phi_hat, joints_phi, joints_hat = m(feats, predict_future=True)
        cases.append({"latent_dim": latent, "number_blocks": blocks, "seed": seed, "feats": feats, "phi": phi, "phi_hat": phi_hat,
                      "joints_phi": joints_phi, "joints_hat": joints_hat})
        print(latent, blocks, b, t, "joints", tuple(joints_phi.shape), float(joints_phi.abs().max()))
    torch.save(cases, os.path.join(HERE, "head_golden.pt"))


if __name__ == "__main__":
    main()
