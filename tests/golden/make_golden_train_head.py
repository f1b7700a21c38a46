"""Golden training steps of the lifting head: the REFERENCE module ``PHDFor3DJoints`` (src/model.py, imported from /root/reference
with ``torchvision`` -- unused by the head -- replaced by an empty stub), set up as src/train.py:370-389 does (f_AR frozen,
``torch.optim.AdamW(trainable, lr, weight_decay=1e-2)``), run for two steps of ``loss = (joints_pred - joints3d).pow(2).mean()``
(:161) on the CPU.  Dropout is random and not seed-pinned upstream, so the module is stepped in eval mode (dropout = identity;
gradients flow the same way); the dropout sites are covered by the oracle with explicit masks.  The fixture keeps, per parameter,
the gradient's norm and its first 64 entries after step 1, and the norm / first 64 entries of the parameter after step 2.

    python tests/golden/make_golden_train_head.py         # run in the build container (needs /root/reference)
"""
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_SRC = os.environ.get("H36M_REFERENCE_SRC", "/root/reference/src")
sys.path.insert(0, ROOT)

LR = 1e-4      # train.py's default (src/config.py LR)


def batches_for(case_seed, b, t):
    g = torch.Generator().manual_seed(500 + case_seed)
    out = []
    for _ in range(2):
        feats = torch.randn(b, t, 2048, generator=g).abs()
        gt = torch.randn(b, t, 17, 3, generator=g) * 0.5
        out.append((feats, gt))
    return out


def main():
    tv = types.ModuleType("torchvision"); tv.models = types.ModuleType("torchvision.models")
    sys.modules["torchvision"] = tv; sys.modules["torchvision.models"] = tv.models
    sys.path.insert(0, REF_SRC)
    import model as ref_model
    from oracle.lifting_oracle import synthetic_head_state_dict
    cases = []
    for latent, blocks, b, t, seed in ((64, 2, 3, 5, 11), (128, 2, 2, 40, 12)):
        m = ref_model.PHDFor3DJoints(latent_dim=latent, joints_num=17, number_blocks=blocks).eval()
        m.load_state_dict(synthetic_head_state_dict(latent, blocks, seed), strict=True)
        for p in m.f_AR.parameters():
            p.requires_grad = False
        trainable = [p for p in m.parameters() if p.requires_grad]
        optim = torch.optim.AdamW(trainable, lr=LR, weight_decay=1e-2)
        case = {"latent_dim": latent, "number_blocks": blocks, "seed": seed, "b": b, "t": t, "lr": LR, "losses": [], "grads": {}, "params": {}}
        for s, (feats, gt) in enumerate(batches_for(seed, b, t)):
            optim.zero_grad(set_to_none=True)
            _phi, _phi_hat, joints_pred, _ = m.forward(feats, predict_future=False)
            loss = (joints_pred - gt).pow(2).mean()
            loss.backward()
            if s == 0:
                for k, p in m.named_parameters():
                    if p.requires_grad:
                        case["grads"][k] = {"norm": float(p.grad.norm()), "head": p.grad.reshape(-1)[:64].clone()}
            optim.step()
            case["losses"].append(float(loss.detach()))
        for k, p in m.named_parameters():
            if p.requires_grad:
                case["params"][k] = {"norm": float(p.detach().norm()), "head": p.detach().reshape(-1)[:64].clone()}
        print(latent, blocks, b, t, "losses", case["losses"], "params", len(case["params"]))
        cases.append(case)
    torch.save(cases, os.path.join(HERE, "train_head_golden.pt"))


if __name__ == "__main__":
    main()
