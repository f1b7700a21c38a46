"""CPU restatement of the lifting head's forward (first step).  TEST INFRASTRUCTURE ONLY: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Reference: ``PHDFor3DJoints.forward`` (src/model.py:146-178) in eval mode (dropout = identity), built by
``PHD(latent_dim=1024, joints_num=17, number_blocks=2)`` (src/train.py:370): ``input_proj`` Linear 2048 -> D,
``f_movie`` = ``number_blocks`` ResidualBlocks (GroupNorm(32) -> ReLU -> causal conv1d k3 with replicate left padding,
twice, plus skip; :18-57), ``f_AR`` = 3 more such blocks on phi, shifted by one frame (:166-168), ``f_3D`` = iterative
regressor (3 iterations of Linear(D+51,1024)-ReLU-Linear(1024,1024)-ReLU-Linear(1024,51) on [phi | y], y += dy; :86-126).
Written with torch.nn.functional on plain tensors from the reference's state-dict keys; pinned by
tests/golden/head_golden.pt (outputs of the reference module itself, tests/golden/make_golden_head.py).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F


def _causal_conv1d(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:      # x (B,C,T); :20-35
    return F.conv1d(F.pad(x, (w.shape[-1] - 1, 0), mode="replicate"), w, b)


def _residual_block(x: torch.Tensor, sd: Dict[str, torch.Tensor], p: str, groups: int = 32) -> torch.Tensor:   # :37-57
    r = x
    x = F.relu(F.group_norm(x, groups, sd[p + ".gn1.weight"], sd[p + ".gn1.bias"], eps=1e-5))
    x = _causal_conv1d(x, sd[p + ".conv1.conv.weight"], sd[p + ".conv1.conv.bias"])
    x = F.relu(F.group_norm(x, groups, sd[p + ".gn2.weight"], sd[p + ".gn2.bias"], eps=1e-5))
    x = _causal_conv1d(x, sd[p + ".conv2.conv.weight"], sd[p + ".conv2.conv.bias"])
    return x + r


def _temporal_net(x_btd: torch.Tensor, sd, prefix: str) -> torch.Tensor:                     # :69-78
    x = x_btd.permute(0, 2, 1)
    i = 0
    while f"{prefix}.blocks.{i}.gn1.weight" in sd:
        x = _residual_block(x, sd, f"{prefix}.blocks.{i}")
        i += 1
    return x.permute(0, 2, 1)


def _regressor(phi: torch.Tensor, sd, iters: int = 3) -> torch.Tensor:                       # :86-126
    b, t, _ = phi.shape
    y = sd["f_3D.y0"].to(phi.dtype).view(1, 1, -1).expand(b, t, -1).contiguous()
    for _ in range(iters):
        h = torch.cat([phi, y], dim=-1)
        h = F.relu(F.linear(h, sd["f_3D.mlp.0.weight"], sd["f_3D.mlp.0.bias"]))
        h = F.relu(F.linear(h, sd["f_3D.mlp.3.weight"], sd["f_3D.mlp.3.bias"]))
        y = y + F.linear(h, sd["f_3D.mlp.5.weight"], sd["f_3D.mlp.5.bias"])
    return y.view(b, t, -1, 3)


@torch.no_grad()
def forward_reference(sd: Dict[str, torch.Tensor], feats: torch.Tensor, predict_future: bool = False,
                      dtype=torch.float32) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, Optional[torch.Tensor]]:
    """(phi, phi_hat, joints_phi, joints_hat) exactly as ``PHDFor3DJoints.forward`` returns them (eval mode)."""
    sd = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
    x = F.linear(feats.to(dtype), sd["input_proj.weight"], sd["input_proj.bias"])
    phi = _temporal_net(x, sd, "f_movie")
    ar = _temporal_net(phi, sd, "f_AR")
    phi_hat = torch.zeros_like(ar)
    phi_hat[:, 1:, :] = ar[:, :-1, :]
    joints_phi = _regressor(phi, sd)
    joints_hat = _regressor(phi_hat, sd) if predict_future else None
    return phi, phi_hat, joints_phi, joints_hat


def synthetic_head_state_dict(latent_dim: int = 1024, number_blocks: int = 2, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded weights with the reference module's keys and shapes (values are NOT the reference's initialisation: the
    fixture pins the arithmetic, not the init).  Scales keep activations O(1)."""
    g = torch.Generator().manual_seed(seed)
    d = latent_dim
    sd: Dict[str, torch.Tensor] = {}

    def rnd(*shape, scale=1.0):
        return torch.randn(*shape, generator=g) * scale

    sd["input_proj.weight"] = rnd(d, 2048, scale=(1.0 / 2048) ** 0.5)
    sd["input_proj.bias"] = rnd(d, scale=0.1)
    for net, nb in (("f_movie", number_blocks), ("f_AR", 3)):
        for i in range(nb):
            p = f"{net}.blocks.{i}"
            for gn in ("gn1", "gn2"):
                sd[f"{p}.{gn}.weight"] = 1.0 + rnd(d, scale=0.1)
                sd[f"{p}.{gn}.bias"] = rnd(d, scale=0.1)
            for cv in ("conv1", "conv2"):
                sd[f"{p}.{cv}.conv.weight"] = rnd(d, d, 3, scale=(1.0 / (3 * d)) ** 0.5)
                sd[f"{p}.{cv}.conv.bias"] = rnd(d, scale=0.05)
    sd["f_3D.y0"] = torch.zeros(51)
    sd["f_3D.mlp.0.weight"] = rnd(1024, d + 51, scale=(1.0 / (d + 51)) ** 0.5)
    sd["f_3D.mlp.0.bias"] = rnd(1024, scale=0.05)
    sd["f_3D.mlp.3.weight"] = rnd(1024, 1024, scale=(1.0 / 1024) ** 0.5)
    sd["f_3D.mlp.3.bias"] = rnd(1024, scale=0.05)
    sd["f_3D.mlp.5.weight"] = rnd(51, 1024, scale=(1.0 / 1024) ** 0.5)
    sd["f_3D.mlp.5.bias"] = rnd(51, scale=0.05)
    return sd


# ---- training step (src/train.py:137-176, 370-393): CPU restatement with torch autograd, fp32 / fp64 ----------------------
TRAINABLE_PREFIXES = ("input_proj.", "f_movie.", "f_3D.mlp.")        # f_AR frozen (:375-376); y0 is a buffer


def _forward_train(p: Dict[str, torch.Tensor], feats: torch.Tensor, masks: Optional[Dict[str, torch.Tensor]], keep: float = 0.5):
    """joints_phi of ``PHDFor3DJoints.forward`` in train mode with the dropout keep-masks given explicitly (None: identity).
    masks: "f_movie.blocks.i" (B*T, D) after conv1 (src/model.py:52), "f_3D.i" (B*T, 1024) after the first ReLU (:98)."""
    b, t, _ = feats.shape
    x = F.linear(feats, p["input_proj.weight"], p["input_proj.bias"]).permute(0, 2, 1)      # (B,D,T)
    i = 0
    while f"f_movie.blocks.{i}.gn1.weight" in p:
        q = f"f_movie.blocks.{i}"
        r = x
        h = F.relu(F.group_norm(x, 32, p[q + ".gn1.weight"], p[q + ".gn1.bias"], eps=1e-5))
        h = _causal_conv1d(h, p[q + ".conv1.conv.weight"], p[q + ".conv1.conv.bias"])
        if masks is not None:
            m = masks[q].view(b, t, -1).permute(0, 2, 1).to(h.dtype)
            h = h * m / keep
        h = F.relu(F.group_norm(h, 32, p[q + ".gn2.weight"], p[q + ".gn2.bias"], eps=1e-5))
        x = _causal_conv1d(h, p[q + ".conv2.conv.weight"], p[q + ".conv2.conv.bias"]) + r
        i += 1
    phi = x.permute(0, 2, 1)
    y = p["f_3D.y0"].to(phi.dtype).view(1, 1, -1).expand(b, t, -1).contiguous()
    for it in range(3):
        h = F.relu(F.linear(torch.cat([phi, y], dim=-1), p["f_3D.mlp.0.weight"], p["f_3D.mlp.0.bias"]))
        if masks is not None:
            h = h * masks[f"f_3D.{it}"].view(b, t, -1).to(h.dtype) / keep
        h = F.relu(F.linear(h, p["f_3D.mlp.3.weight"], p["f_3D.mlp.3.bias"]))
        y = y + F.linear(h, p["f_3D.mlp.5.weight"], p["f_3D.mlp.5.bias"])
    return y.view(b, t, -1, 3)


def train_steps_reference(sd: Dict[str, torch.Tensor], batches, masks_per_step=None, lr: float = 1e-4, weight_decay: float = 1e-2,
                          dtype=torch.float32):
    """Run len(batches) phase-1 steps (l3d loss :161, AdamW :389; no loss scaling on the CPU, as the reference's own CPU path).
    batches: [(feats (B,T,2048), joints3d (B,T,17,3))].  Returns (losses, mpjpes, grads of the FIRST step, final state dict)."""
    p = {k: v.detach().clone().to(dtype) for k, v in sd.items()}
    trainable = [k for k in p if k.startswith(TRAINABLE_PREFIXES)]
    for k in trainable:
        p[k].requires_grad_(True)
    opt = torch.optim.AdamW([p[k] for k in trainable], lr=lr, weight_decay=weight_decay)
    losses, mpjpes, first_grads = [], [], None
    for s, (feats, gt) in enumerate(batches):
        opt.zero_grad(set_to_none=True)
        pred = _forward_train(p, feats.to(dtype), masks_per_step[s] if masks_per_step is not None else None)
        loss = (pred - gt.to(dtype)).pow(2).mean()
        loss.backward()
        if first_grads is None:
            first_grads = {k: p[k].grad.detach().clone() for k in trainable}
        opt.step()
        losses.append(float(loss.detach()))
        mpjpes.append(float(torch.norm(pred.detach() - gt.to(dtype), dim=-1).mean()))
    return losses, mpjpes, first_grads, {k: v.detach().clone() for k, v in p.items()}
