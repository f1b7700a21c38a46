"""Host-side mirror of the reference's phase-1 training step for the lifting head on one MI355X.

The reference (src/train.py:114-176,370-393)::

    model = PHD(latent_dim=1024, joints_num=17, number_blocks=2); f_AR frozen (:375-376)
    optim = torch.optim.AdamW(trainable, lr=args.lr, weight_decay=1e-2); scaler = torch.amp.GradScaler('cuda')
    with torch.autocast(dtype=torch.float16):
        _phi, _phi_hat, joints_pred, _ = model.forward(feats, predict_future=False)
        loss = l3d = (joints_pred - joints3d).pow(2).mean()
    scaler.scale(loss).backward(); scaler.step(optim); scaler.update()

``TrainableHead`` keeps the model surface of ``model.PHDFor3DJoints`` (constructor, ``load_state_dict`` / ``state_dict`` with the
reference's keys) and adds ``train_step(feats, joints3d, optim, scaler)``.  Arithmetic on the device, through the C ABI:

* forward as in model.py plus the two ``nn.Dropout(0.5)`` sites (src/model.py:44,52 and :98), applied with byte masks;
* backward: every dX = dY W and dW = dY^T X is an igemm MFMA launch (``r50_op_conv2d(_f16)``) on operands made K-contiguous by
  ``r50_op_transpose16``; GroupNorm + ReLU + causal-row backward, ReLU / dropout backward, bias column sums, the MSE gradient are
  the kernels of include/r50.h "Lifting head, backward + optimizer".  16-bit activations and gradients, fp32 accumulation in the
  GEMMs, fp32 master weights and fp32 flat gradient buffer (the shared regressor weights accumulate their three uses in fp32);
* ``AdamW`` / ``GradScaler``: one flat fp32 parameter / moment / gradient buffer, one ``r50_op_adamw`` launch per step, skipped on
  the device when ``r50_op_check_finite`` raised the flag; the scale follows torch.amp.GradScaler's rule (x0.5 on overflow, x2
  after 2000 clean steps).
* f_AR is frozen and its output does not enter the loss (:158-161), so the training step does not run it.
* multi-GPU: one process per GPU, ``all_reduce_gradients`` averages the flat gradient buffer with ONE RCCL all-reduce per step
  (35.8 M parameters x 4 B = 67.6 MB at train.py's configuration; 16.9 M trainable) instead of nn.DataParallel's scatter /
  replicate / gather (:381-383).

PyTorch is used for device memory, the stream, the dropout masks' random bits and torch.distributed.  No CPU fallback.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch

from . import _lib
from .model import _GN_EPS, _GROUPS, _REG_HIDDEN, _REG_ITERS, PHDFor3DJoints, _round_up

DROPOUT_P = 0.5        # ResidualBlock(dropout=0.5), JointRegressor(dropout=0.5): src/model.py:39,87


class GradScaler:
    """torch.amp.GradScaler('cuda') defaults (src/train.py:392): init 65536, x2 after 2000 clean steps, x0.5 on inf/nan."""

    def __init__(self, init_scale: float = 65536.0, growth_factor: float = 2.0, backoff_factor: float = 0.5,
                 growth_interval: int = 2000, enabled: bool = True):
        self.enabled = enabled
        self._scale = float(init_scale) if enabled else 1.0
        self.growth_factor, self.backoff_factor, self.growth_interval = growth_factor, backoff_factor, growth_interval
        self._growth_tracker = 0

    def get_scale(self) -> float:
        return self._scale

    def update(self, found_inf: bool) -> None:
        if not self.enabled:
            return
        if found_inf:
            self._scale *= self.backoff_factor
            self._growth_tracker = 0
        else:
            self._growth_tracker += 1
            if self._growth_tracker == self.growth_interval:
                self._scale *= self.growth_factor
                self._growth_tracker = 0


class AdamW:
    """torch.optim.AdamW(trainable, lr, weight_decay=1e-2) (src/train.py:389) over the head's flat fp32 buffers."""

    def __init__(self, head: "TrainableHead", lr: float = 1e-4, betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2):
        self.head, self.lr, self.betas, self.eps, self.weight_decay = head, lr, betas, eps, weight_decay
        self.step_count = 0
        self.exp_avg = torch.zeros_like(head.flat_master)
        self.exp_avg_sq = torch.zeros_like(head.flat_master)

    def step(self, found_inf_flag: Optional[torch.Tensor]) -> None:
        """One update from ``head.flat_grad`` (already unscaled).  The step counter advances only when the update is applied."""
        h = self.head
        self.step_count += 1
        rc = _lib.load_library().r50_op_adamw(h.flat_master.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                                              h.flat_grad.data_ptr(), h.flat_w16.data_ptr(), h.flat_master.numel(), self.lr,
                                              self.betas[0], self.betas[1], self.eps, self.weight_decay, self.step_count,
                                              found_inf_flag.data_ptr() if found_inf_flag is not None else None, h._et, h._stream())
        _lib.check(rc, None, "r50_op_adamw")


def all_reduce_gradients(flat_grad: torch.Tensor, group=None) -> None:
    """Average the flat gradient buffer over the ranks: one all-reduce per step (RCCL over xGMI on GPUs; gloo in the CPU tests)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return
    world = dist.get_world_size(group)
    if world == 1:
        return
    dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
    flat_grad.mul_(1.0 / world)


def sync_overflow_flag(found: torch.Tensor, group=None) -> None:
    """Make the skip decision of a data-parallel step GLOBAL: MAX-reduce the found-overflow flag over the ranks, so every
    replica skips (and backs its loss scale off) or steps together.  Needed because an fp16 overflow on ONE rank is a
    saturated 65504 -- finite -- so after the gradient all-reduce the averaged buffer is finite everywhere and only the
    overflowing rank's own arena check fires.  The reference's ``nn.DataParallel`` has one scaler and one optimizer
    (src/train.py:382-393) and cannot disagree with itself; this is the one-process-per-GPU equivalent."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    dist.all_reduce(found, op=dist.ReduceOp.MAX, group=group)


class _Arena:
    """Bump allocator for the backward pass's GEMM outputs: one 16-bit buffer, so ONE overflow check covers every gradient the matrix
    cores produced in a step (and nothing is allocated per step once the first step has sized it)."""

    def __init__(self, device, dtype):
        self.device, self.dtype = device, dtype
        self.chunks: List[torch.Tensor] = []
        self.used: List[int] = []

    def reset(self) -> None:
        total = sum(self.used)
        if len(self.chunks) != 1 or self.chunks[0].numel() < total:
            self.chunks = [torch.empty(max(total, 1 << 20), dtype=self.dtype, device=self.device)]
        self.used = [0]

    def take(self, rows: int, cols: int) -> torch.Tensor:
        n = _round_up(rows * cols, 64)
        if self.used[-1] + n > self.chunks[-1].numel():
            self.chunks.append(torch.empty(max(n, 1 << 22), dtype=self.dtype, device=self.device))
            self.used.append(0)
        o = self.used[-1]
        self.used[-1] = o + n
        return self.chunks[-1][o: o + rows * cols].view(rows, cols)


class TrainableHead(PHDFor3DJoints):
    """``PHDFor3DJoints`` with the phase-1 trainable parameters (input_proj, f_movie, f_3D) in flat fp32 / 16-bit buffers."""

    def __init__(self, latent_dim: int = 2048, joints_num: int = 17, number_blocks: int = 3, precision: str = "fp16"):
        super().__init__(latent_dim, joints_num, number_blocks, precision)
        self.flat_master: Optional[torch.Tensor] = None
        self._layout: List[Tuple[str, int, Tuple[int, ...]]] = []
        self._use_graphs = False
        self._graphs: Dict[tuple, tuple] = {}

    def enable_graphs(self, on: bool = True) -> "TrainableHead":
        """Replay forward + loss + backward (~250 short launches) as one captured HIP graph per (B, T, loss scale, mode): the step
        is launch-bound otherwise.  The optimizer part stays outside (its bias corrections are per-step host scalars)."""
        self._use_graphs = bool(on)
        return self

    def train(self, mode: bool = True):
        self.training = bool(mode)
        return self

    # ---- flat parameter buffers (GEMM layout) -------------------------------------------------
    def _upload(self) -> None:
        super()._upload()                      # f_AR and y0 (frozen) + everything eval() needs; trainable entries are re-pointed below
        sd, dev = self._sd, self._device
        d, o = self.latent_dim, self.out_dim
        items: List[Tuple[str, torch.Tensor]] = [("input_proj.w", sd["input_proj.weight"]), ("input_proj.b", sd["input_proj.bias"])]
        for i in range(self.number_blocks):
            p = f"f_movie.blocks.{i}"
            for gn in ("gn1", "gn2"):
                items += [(f"{p}.{gn}.g", sd[f"{p}.{gn}.weight"]), (f"{p}.{gn}.b", sd[f"{p}.{gn}.bias"])]
            for cv in ("conv1", "conv2"):
                items += [(f"{p}.{cv}.w", sd[f"{p}.{cv}.conv.weight"].permute(0, 2, 1).reshape(d, 3 * d)),
                          (f"{p}.{cv}.b", sd[f"{p}.{cv}.conv.bias"])]
        w0 = torch.zeros(_REG_HIDDEN, self._dp); w0[:, : d + o] = sd["f_3D.mlp.0.weight"]
        w5 = torch.zeros(self._op, _REG_HIDDEN); w5[:o] = sd["f_3D.mlp.5.weight"]
        b5 = torch.zeros(self._op); b5[:o] = sd["f_3D.mlp.5.bias"]
        items += [("mlp0.w", w0), ("mlp0.b", sd["f_3D.mlp.0.bias"]), ("mlp3.w", sd["f_3D.mlp.3.weight"]),
                  ("mlp3.b", sd["f_3D.mlp.3.bias"]), ("mlp5.w", w5), ("mlp5.b", b5)]
        self._layout, off = [], 0
        for name, t in items:
            assert t.numel() % 64 == 0
            self._layout.append((name, off, tuple(t.shape)))
            off += t.numel()
        self.flat_master = torch.cat([t.reshape(-1).to(torch.float32) for _, t in items]).to(dev)
        self.flat_w16 = self.flat_master.to(self._dtype)
        self.flat_grad = torch.zeros_like(self.flat_master)
        self._off = {name: (o_, shape) for name, o_, shape in self._layout}
        for name, o_, shape in self._layout:       # weights: the 16-bit copy; biases and GroupNorm parameters: the fp32 master itself
            n = int(torch.Size(shape).numel())
            src = self.flat_w16 if name.endswith(".w") else self.flat_master
            self._dev[name] = src[o_: o_ + n].view(shape)
        self._wt: Dict[str, torch.Tensor] = {}     # transposed 16-bit weights for the dX products
        self._refresh_transposes()
        self._zero_bias = torch.zeros(max(3 * d, 2048, self._dp, _REG_HIDDEN), dtype=torch.float32, device=dev)
        self._found = torch.zeros(1, dtype=torch.int32, device=dev)
        self._arena = _Arena(dev, self._dtype)

    def _refresh_transposes(self) -> None:
        lib = _lib.load_library()
        for name, _, shape in self._layout:
            if not name.endswith(".w") or name == "input_proj.w":
                continue
            n, k = shape
            if name not in self._wt:
                self._wt[name] = torch.zeros((k, n), dtype=self._dtype, device=self._device)
            _lib.check(lib.r50_op_transpose16(self._dev[name].data_ptr(), n, k, self._wt[name].data_ptr(), n, self._stream()), None,
                       "r50_op_transpose16")

    def grad_view(self, name: str) -> torch.Tensor:
        o_, shape = self._off[name]
        return self.flat_grad[o_: o_ + int(torch.Size(shape).numel())].view(shape)

    def state_dict(self) -> Dict[str, torch.Tensor]:
        """The reference's keys and layouts (fp32, CPU) from the flat master buffer; frozen entries as loaded."""
        d, o = self.latent_dim, self.out_dim
        out = {k: v.clone() for k, v in self._sd.items()}
        get = lambda name: self.flat_master[self._off[name][0]: self._off[name][0] + int(torch.Size(self._off[name][1]).numel())] \
            .view(self._off[name][1]).cpu()
        out["input_proj.weight"], out["input_proj.bias"] = get("input_proj.w"), get("input_proj.b")
        for i in range(self.number_blocks):
            p = f"f_movie.blocks.{i}"
            for gn in ("gn1", "gn2"):
                out[f"{p}.{gn}.weight"], out[f"{p}.{gn}.bias"] = get(f"{p}.{gn}.g"), get(f"{p}.{gn}.b")
            for cv in ("conv1", "conv2"):
                out[f"{p}.{cv}.conv.weight"] = get(f"{p}.{cv}.w").view(d, 3, d).permute(0, 2, 1).contiguous()
                out[f"{p}.{cv}.conv.bias"] = get(f"{p}.{cv}.b")
        out["f_3D.mlp.0.weight"], out["f_3D.mlp.0.bias"] = get("mlp0.w")[:, : d + o].contiguous(), get("mlp0.b")
        out["f_3D.mlp.3.weight"], out["f_3D.mlp.3.bias"] = get("mlp3.w"), get("mlp3.b")
        out["f_3D.mlp.5.weight"], out["f_3D.mlp.5.bias"] = get("mlp5.w")[:o].contiguous(), get("mlp5.b")[:o].contiguous()
        return out

    def named_gradients(self) -> Dict[str, torch.Tensor]:
        """flat_grad under the reference's parameter names and layouts (fp32, CPU): what ``p.grad`` holds after ``backward()``."""
        d, o = self.latent_dim, self.out_dim
        g = lambda name: self.grad_view(name).cpu()
        out = {"input_proj.weight": g("input_proj.w"), "input_proj.bias": g("input_proj.b")}
        for i in range(self.number_blocks):
            p = f"f_movie.blocks.{i}"
            for gn in ("gn1", "gn2"):
                out[f"{p}.{gn}.weight"], out[f"{p}.{gn}.bias"] = g(f"{p}.{gn}.g"), g(f"{p}.{gn}.b")
            for cv in ("conv1", "conv2"):
                out[f"{p}.{cv}.conv.weight"] = g(f"{p}.{cv}.w").view(d, 3, d).permute(0, 2, 1).contiguous()
                out[f"{p}.{cv}.conv.bias"] = g(f"{p}.{cv}.b")
        out["f_3D.mlp.0.weight"], out["f_3D.mlp.0.bias"] = g("mlp0.w")[:, : d + o].contiguous(), g("mlp0.b")
        out["f_3D.mlp.3.weight"], out["f_3D.mlp.3.bias"] = g("mlp3.w"), g("mlp3.b")
        out["f_3D.mlp.5.weight"], out["f_3D.mlp.5.bias"] = g("mlp5.w")[:o].contiguous(), g("mlp5.b")[:o].contiguous()
        return out

    # ---- launch helpers -----------------------------------------------------------------------
    def _mm(self, x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
        """x (R, K) @ w (N, K)^T -> (R, N), 16-bit out, fp32 accumulation, no bias: one igemm launch."""
        rows, k = x.shape
        n = w.shape[0]
        assert w.shape[1] == k and x.is_contiguous() and w.is_contiguous() and k % 64 == 0 and n % 64 == 0
        y = self._arena.take(rows, n)
        lib = _lib.load_library()
        fn = lib.r50_op_conv2d_f16 if self._et else lib.r50_op_conv2d
        _lib.check(fn(x.data_ptr(), rows, 1, 1, k, w.data_ptr(), self._zero_bias.data_ptr(), None, y.data_ptr(), n, 1, 1, 0, 0, 0,
                      self._stream()), None, "r50_op_conv2d (lifting head backward)")
        return y

    def _t(self, x: torch.Tensor) -> torch.Tensor:
        """(R, C) -> (C, Rp) transposed, Rp = R rounded up to 64 with zero padding (the K of a dW product)."""
        rows, cols = x.shape
        rp = _round_up(rows, 64)
        out = torch.zeros((cols, rp), dtype=self._dtype, device=self._device) if rp != rows else \
            torch.empty((cols, rp), dtype=self._dtype, device=self._device)
        _lib.check(_lib.load_library().r50_op_transpose16(x.data_ptr(), rows, cols, out.data_ptr(), rp, self._stream()), None,
                   "r50_op_transpose16")
        return out

    def _wgrad(self, name: str, dy: torch.Tensor, x: torch.Tensor, inv_scale: float, accumulate: bool, bias: Optional[str] = None) -> None:
        """flat_grad[name] (N, K) [+]= inv_scale * dy (R, N)^T x (R, K); flat_grad[bias] (N) [+]= inv_scale * column sums of dy."""
        lib = _lib.load_library()
        dw = self._mm(self._t(dy), self._t(x))                    # (N, Rp) @ (K, Rp)^T -> (N, K)
        gv = self.grad_view(name)
        assert tuple(dw.shape) == tuple(gv.shape)
        _lib.check(lib.r50_op_grad_accum(dw.data_ptr(), inv_scale, gv.data_ptr(), dw.numel(), int(accumulate), self._et, self._stream()),
                   None, "r50_op_grad_accum")
        if bias is not None:
            gb = self.grad_view(bias)
            _lib.check(lib.r50_op_colsum(dy.data_ptr(), dy.shape[0], dy.shape[1], dy.shape[1], inv_scale, gb.data_ptr(), int(accumulate),
                                         self._et, self._stream()), None, "r50_op_colsum")

    def _mask_scale(self, x: torch.Tensor, mask: torch.Tensor, scale: float) -> None:
        assert mask.dtype == torch.uint8 and mask.numel() == x.numel() and mask.is_contiguous()
        _lib.check(_lib.load_library().r50_op_mask_scale(x.data_ptr(), mask.data_ptr(), scale, x.numel(), self._et, self._stream()), None,
                   "r50_op_mask_scale")

    def _relu_bwd(self, dy: torch.Tensor, act: torch.Tensor, scale: float) -> None:
        _lib.check(_lib.load_library().r50_op_relu_bwd(dy.data_ptr(), act.data_ptr(), scale, dy.numel(), self._et, self._stream()), None,
                   "r50_op_relu_bwd")

    def _gn_bwd(self, dr: torch.Tensor, x: torch.Tensor, b: int, t: int, prefix: str, add: Optional[torch.Tensor], inv_scale: float) -> torch.Tensor:
        d = self.latent_dim
        lib = _lib.load_library()
        dx = torch.empty((b * t, d), dtype=self._dtype, device=self._device)
        part = torch.empty((2, b, d), dtype=torch.float32, device=self._device)
        _lib.check(lib.r50_op_gn_relu_causal3_bwd(dr.data_ptr(), x.data_ptr(), b, t, d, _GROUPS, self._dev[prefix + ".g"].data_ptr(),
                                                  self._dev[prefix + ".b"].data_ptr(), _GN_EPS, add.data_ptr() if add is not None else None,
                                                  dx.data_ptr(), part[0].data_ptr(), part[1].data_ptr(), self._et, self._stream()), None,
                   "r50_op_gn_relu_causal3_bwd")
        for j, suffix in ((0, ".g"), (1, ".b")):
            _lib.check(lib.r50_op_colsum_f32(part[j].data_ptr(), b, d, inv_scale, self.grad_view(prefix + suffix).data_ptr(), 0,
                                             self._stream()), None, "r50_op_colsum_f32")
        return dx

    def make_dropout_masks(self, b: int, t: int, generator: Optional[torch.Generator] = None) -> Dict[str, torch.Tensor]:
        """Byte keep-masks (1 = keep, probability 1 - p) for the dropout sites of one step: one per f_movie block (src/model.py:52)
        and one per regressor iteration (:98)."""
        def bern(*shape):
            return (torch.rand(*shape, device=self._device, generator=generator) >= DROPOUT_P).to(torch.uint8)
        masks = {f"f_movie.blocks.{i}": bern(b * t, self.latent_dim) for i in range(self.number_blocks)}
        masks.update({f"f_3D.{i}": bern(b * t, _REG_HIDDEN) for i in range(_REG_ITERS)})
        return masks

    # ---- one training step ------------------------------------------------------------------------
    def forward_backward(self, feats: torch.Tensor, joints3d: torch.Tensor, loss_scale: float = 1.0,
                         masks: Optional[Dict[str, torch.Tensor]] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """Forward (train mode when ``masks`` is given or ``self.training``; else dropout is identity), l3d loss (src/train.py:161),
        backward into ``flat_grad`` (UNSCALED: the 16-bit backward runs on loss_scale * loss, the fp32 buffer receives grad / loss_scale).
        ``self._found`` is raised when a 16-bit gradient overflowed (fp16 saturates at 65504 here instead of producing inf).
        Returns (joints_pred (B,T,J,3) fp32, loss2 = [l3d, mpjpe] fp32 device tensor)."""
        if self.flat_master is None:
            raise _lib.R50Error("call .load_state_dict(...) and .to('cuda:N') first")
        if feats.dim() != 3 or feats.shape[-1] != 2048 or feats.device != self._device:
            raise ValueError("feats: expected (B,T,2048) on the head's device")
        b, t, _ = feats.shape
        if tuple(joints3d.shape) != (b, t, self.joints_num, 3) or joints3d.device != self._device:
            raise ValueError("joints3d: expected (B,T,J,3) on the head's device")
        if masks is None and self.training:
            masks = self.make_dropout_masks(b, t)
        keep_scale = 1.0 / (1.0 - DROPOUT_P)
        lib = _lib.load_library()
        rows, d, o = b * t, self.latent_dim, self.out_dim
        inv = 1.0 / loss_scale
        self._arena.reset()
        self._found.zero_()
        with torch.cuda.device(self._device):
            # ---------------- forward, keeping what the backward needs ----------------
            f = feats.to(torch.float32).contiguous()
            x0 = torch.empty((rows, 2048), dtype=self._dtype, device=self._device)
            _lib.check(lib.r50_op_cast_rows(f.data_ptr(), rows, 2048, x0.data_ptr(), 2048, self._et, self._stream()), None, "r50_op_cast_rows")
            x = self._gemm(x0, "input_proj", relu=False)
            saved = []
            for i in range(self.number_blocks):
                p = f"f_movie.blocks.{i}"
                r1 = self._gn_relu_rows(x, b, t, p + ".gn1")
                h = self._gemm(r1, p + ".conv1", relu=False)
                m = masks[p] if masks is not None else None
                if m is not None:
                    self._mask_scale(h, m, keep_scale)
                r2 = self._gn_relu_rows(h, b, t, p + ".gn2")
                xo = self._gemm(r2, p + ".conv2", relu=False, residual=x)
                saved.append((x, r1, h, r2, m))
                x = xo
            phi = x
            y = self._dev["y0"].view(1, o).expand(rows, o).contiguous()
            reg = []
            for i in range(_REG_ITERS):
                inp = torch.empty((rows, self._dp), dtype=self._dtype, device=self._device)
                _lib.check(lib.r50_op_concat_pad(phi.data_ptr(), d, y.data_ptr(), o, rows, inp.data_ptr(), self._dp, self._et, self._stream()),
                           None, "r50_op_concat_pad")
                h1 = self._gemm(inp, "mlp0", relu=True)
                if masks is not None:
                    self._mask_scale(h1, masks[f"f_3D.{i}"], keep_scale)
                h2 = self._gemm(h1, "mlp3", relu=True)
                dy = self._gemm(h2, "mlp5", relu=False)
                _lib.check(lib.r50_op_add_rows(y.data_ptr(), o, dy.data_ptr(), self._op, rows, self._et, self._stream()), None, "r50_op_add_rows")
                reg.append((inp, h1, h2))
            # ---------------- loss and its gradient ----------------
            gt = joints3d.to(torch.float32).contiguous()
            dyacc = torch.empty((rows, o), dtype=torch.float32, device=self._device)
            loss2 = torch.empty(2, dtype=torch.float32, device=self._device)
            _lib.check(lib.r50_op_mse_loss_grad(y.data_ptr(), gt.data_ptr(), rows * o, loss_scale, dyacc.data_ptr(), loss2.data_ptr(),
                                                self._stream()), None, "r50_op_mse_loss_grad")
            # ---------------- backward: regressor, last iteration first ----------------
            dphi = torch.zeros((rows, d), dtype=torch.float32, device=self._device)
            g5 = torch.empty((rows, self._op), dtype=self._dtype, device=self._device)
            for i in reversed(range(_REG_ITERS)):
                inp, h1, h2 = reg[i]
                first = i == _REG_ITERS - 1
                _lib.check(lib.r50_op_cast_rows(dyacc.data_ptr(), rows, o, g5.data_ptr(), self._op, self._et, self._stream()), None, "r50_op_cast_rows")
                self._wgrad("mlp5.w", g5, h2, inv, not first, bias="mlp5.b")
                dh2 = self._mm(g5, self._wt["mlp5.w"])                         # (rows, H)
                self._relu_bwd(dh2, h2, 1.0)
                self._wgrad("mlp3.w", dh2, h1, inv, not first, bias="mlp3.b")
                dh1 = self._mm(dh2, self._wt["mlp3.w"])
                self._relu_bwd(dh1, h1, keep_scale if masks is not None else 1.0)
                self._wgrad("mlp0.w", dh1, inp, inv, not first, bias="mlp0.b")
                dinp = self._mm(dh1, self._wt["mlp0.w"])                       # (rows, Dp) = [dphi | dy | 0]
                _lib.check(lib.r50_op_add_rows(dphi.data_ptr(), d, dinp.data_ptr(), self._dp, rows, self._et, self._stream()), None, "r50_op_add_rows")
                _lib.check(lib.r50_op_add_rows(dyacc.data_ptr(), o, dinp.data_ptr() + 2 * d, self._dp, rows, self._et, self._stream()), None,
                           "r50_op_add_rows")
            dx = torch.empty((rows, d), dtype=self._dtype, device=self._device)
            _lib.check(lib.r50_op_cast_rows(dphi.data_ptr(), rows, d, dx.data_ptr(), d, self._et, self._stream()), None, "r50_op_cast_rows")
            # ---------------- backward: f_movie blocks, last first ----------------
            for i in reversed(range(self.number_blocks)):
                p = f"f_movie.blocks.{i}"
                xin, r1, h, r2, m = saved[i]
                self._wgrad(p + ".conv2.w", dx, r2, inv, False, bias=p + ".conv2.b")
                dr2 = self._mm(dx, self._wt[p + ".conv2.w"])                   # (rows, 3D)
                dh = self._gn_bwd(dr2, h, b, t, p + ".gn2", None, inv)
                if m is not None:
                    self._mask_scale(dh, m, keep_scale)
                self._wgrad(p + ".conv1.w", dh, r1, inv, False, bias=p + ".conv1.b")
                dr1 = self._mm(dh, self._wt[p + ".conv1.w"])
                dx = self._gn_bwd(dr1, xin, b, t, p + ".gn1", dx, inv)         # + the skip connection's gradient
            self._wgrad("input_proj.w", dx, x0, inv, False, bias="input_proj.b")
            for chunk, used in zip(self._arena.chunks, self._arena.used):     # every 16-bit gradient the GEMMs wrote this step
                if used:
                    _lib.check(lib.r50_op_check_overflow16(chunk.data_ptr(), used, self._found.data_ptr(), self._et, self._stream()), None,
                               "r50_op_check_overflow16")
        return y.view(b, t, self.joints_num, 3), loss2

    def _forward_backward_graphed(self, feats: torch.Tensor, joints3d: torch.Tensor, loss_scale: float):
        b, t, _ = feats.shape
        key = (b, t, float(loss_scale), self.training)
        if key not in self._graphs:
            if len(self._graphs) >= 8:                      # loss scales come and go; keep the cache bounded
                self._graphs.pop(next(iter(self._graphs)))
            s_feats = torch.empty((b, t, 2048), dtype=torch.float32, device=self._device)
            s_gt = torch.empty((b, t, self.joints_num, 3), dtype=torch.float32, device=self._device)
            s_feats.copy_(feats); s_gt.copy_(joints3d)
            eager_arena, self._arena = self._arena, _Arena(self._device, self._dtype)
            try:
                side = torch.cuda.Stream(self._device)
                side.wait_stream(torch.cuda.current_stream(self._device))
                with torch.cuda.stream(side):              # warm-up off the default stream: sizes the arena, loads every kernel
                    for _ in range(2):
                        self.forward_backward(s_feats, s_gt, loss_scale)
                torch.cuda.current_stream(self._device).wait_stream(side)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    pred, loss2 = self.forward_backward(s_feats, s_gt, loss_scale)
                self._graphs[key] = (graph, s_feats, s_gt, pred, loss2, self._arena)
            finally:
                self._arena = eager_arena
        graph, s_feats, s_gt, pred, loss2, _ = self._graphs[key]
        s_feats.copy_(feats); s_gt.copy_(joints3d)
        graph.replay()
        return pred, loss2

    def train_step(self, feats: torch.Tensor, joints3d: torch.Tensor, optim: AdamW, scaler: Optional[GradScaler] = None,
                   masks: Optional[Dict[str, torch.Tensor]] = None, group=None) -> Tuple[float, float, bool]:
        """src/train.py:137-176 for one batch: forward + l3d, scaled backward, inf check, AdamW, scale update.
        Returns (loss, mpjpe, skipped)."""
        scale = scaler.get_scale() if scaler is not None else 1.0
        if self._use_graphs and masks is None:
            _, loss2 = self._forward_backward_graphed(feats, joints3d, scale)
        else:
            _, loss2 = self.forward_backward(feats, joints3d, scale, masks)
        lib = _lib.load_library()
        with torch.cuda.device(self._device):
            all_reduce_gradients(self.flat_grad, group)
            _lib.check(lib.r50_op_check_finite(self.flat_grad.data_ptr(), self.flat_grad.numel(), self._found.data_ptr(), self._stream()), None,
                       "r50_op_check_finite")
            sync_overflow_flag(self._found, group)        # any rank overflowed -> every rank skips this step
            found = bool(self._found.item())              # the reference's scaler.step() synchronises on the same flag
            if not found:
                optim.step(self._found)
                self._refresh_transposes()
            if scaler is not None:
                scaler.update(found)
            l = loss2.cpu()
        return float(l[0]), float(l[1]), found


def mpjpe_m(pred: torch.Tensor, gt: torch.Tensor) -> float:
    """src/train.py:42-45 (reported by train_step from the device; this is the host form for evaluation code)."""
    return float(torch.norm(pred - gt, dim=-1).mean().item())


def train(model: TrainableHead, loader, optim: AdamW, scaler: Optional[GradScaler], device, log_every: int = 500):
    """The epoch loop of the reference's ``train()`` (src/train.py:114-215): same batch tuple, same returned (mean loss, mean mpjpe)."""
    model.train()
    running_loss = running_mpjpe = 0.0
    n_batches = 0
    for it, batch in enumerate(loader):
        feats, joints3d = batch[0].to(device, non_blocking=True), batch[1].to(device, non_blocking=True)
        loss, mpjpe, _ = model.train_step(feats, joints3d, optim, scaler)
        running_loss += loss
        running_mpjpe += mpjpe
        n_batches += 1
        if log_every > 0 and (it + 1) % log_every == 0:
            print(f"[3D]  iter {it + 1:05d} | loss {running_loss / n_batches:.6f} | mpjpe {running_mpjpe / n_batches:.3f}")
    return running_loss / max(n_batches, 1), running_mpjpe / max(n_batches, 1)
