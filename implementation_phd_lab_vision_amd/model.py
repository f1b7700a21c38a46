"""Host-side mirror of the reference's lifting head for the MI355X path, forward / inference only.

The reference builds and calls it as (src/train.py:370, src/model.py:127-178)::

    model = PHD(latent_dim=1024, joints_num=17, number_blocks=2)
    phi, phi_hat, joints_phi, joints_hat = model(feats, predict_future=True)        # feats (B,T,2048)

``PHDFor3DJoints`` keeps that surface (constructor arguments, ``load_state_dict`` with the reference's keys, ``.to(device)``,
``.eval()``, ``__call__`` returning the same 4-tuple of fp32 tensors) and routes the arithmetic to libr50hip.so:

* every ``nn.Linear`` (input_proj :143, the regressor's MLP :95-102) and every ``CausalConv1d`` (:20-35) is one launch of the
  implicit-GEMM MFMA kernel (``r50_op_conv2d_f16`` / ``r50_op_conv2d``) as a 1x1 convolution over the B*T rows.  A causal
  conv1d with kernel 3 and replicate left padding is a GEMM with K = 3*C against the row [x(t-2) | x(t-1) | x(t)], indices
  clamped at 0; its weight (C_out, C_in, 3) is repacked once to (C_out, 3*C_in) in that order.  Bias, ReLU and the residual
  add (:57) are the kernel's fused epilogue;
* GroupNorm(32) + ReLU (:47-55) and the construction of those rows are one kernel (``r50_op_gn_relu_causal3``);
* ``torch.cat([phi, y])`` (:113) and ``y = y + dy`` (:114-115) are ``r50_op_concat_pad`` / ``r50_op_add_rows`` (y stays fp32).

Element type: IEEE half by default -- the reference runs the head under ``torch.autocast(dtype=torch.float16)`` on the GPU
(src/train.py:154); ``precision="bf16"`` selects bf16.  Accumulation is fp32.  Dropout is identity (eval mode); there is no
backward pass here, so ``.train()`` raises.  PyTorch is used for device memory, the stream and the phi_hat shift (a copy).
No fallback: without the shared library or a gfx950 GPU the calls raise.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

from . import _lib

_GROUPS = 32           # ResidualBlock(groups=32), src/model.py:39
_GN_EPS = 1e-5         # nn.GroupNorm default
_AR_BLOCKS = 3         # CausalTemporalNet(latent_dim) default num_blocks, src/model.py:69,141
_REG_ITERS = 3         # JointRegressor(iters=3), src/model.py:87
_REG_HIDDEN = 1024


def _round_up(v: int, m: int) -> int:
    return (v + m - 1) // m * m


def expected_keys(latent_dim: int, joints_num: int, number_blocks: int) -> Dict[str, Tuple[int, ...]]:
    """state-dict keys and shapes of the reference module (src/model.py:127-143)."""
    d, o = latent_dim, joints_num * 3
    keys: Dict[str, Tuple[int, ...]] = {"input_proj.weight": (d, 2048), "input_proj.bias": (d,)}
    for net, nb in (("f_movie", number_blocks), ("f_AR", _AR_BLOCKS)):
        for i in range(nb):
            p = f"{net}.blocks.{i}"
            for gn in ("gn1", "gn2"):
                keys[f"{p}.{gn}.weight"] = (d,)
                keys[f"{p}.{gn}.bias"] = (d,)
            for cv in ("conv1", "conv2"):
                keys[f"{p}.{cv}.conv.weight"] = (d, d, 3)
                keys[f"{p}.{cv}.conv.bias"] = (d,)
    keys["f_3D.y0"] = (o,)
    keys["f_3D.mlp.0.weight"] = (_REG_HIDDEN, d + o)
    keys["f_3D.mlp.0.bias"] = (_REG_HIDDEN,)
    keys["f_3D.mlp.3.weight"] = (_REG_HIDDEN, _REG_HIDDEN)
    keys["f_3D.mlp.3.bias"] = (_REG_HIDDEN,)
    keys["f_3D.mlp.5.weight"] = (o, _REG_HIDDEN)
    keys["f_3D.mlp.5.bias"] = (o,)
    return keys


class PHDFor3DJoints:
    """``PHDFor3DJoints(latent_dim, joints_num, number_blocks)`` of src/model.py in eval mode on one MI355X."""

    def __init__(self, latent_dim: int = 2048, joints_num: int = 17, number_blocks: int = 3, precision: str = "fp16"):
        if latent_dim % 64 or latent_dim % _GROUPS:
            raise ValueError("latent_dim must be a multiple of 64 (GEMM granularity) and of 32 (GroupNorm groups)")
        if precision not in ("fp16", "bf16"):
            raise ValueError("precision must be 'fp16' or 'bf16'")
        self.latent_dim = int(latent_dim)
        self.joints_num = int(joints_num)
        self.number_blocks = int(number_blocks)
        self.out_dim = self.joints_num * 3
        self._et = 1 if precision == "fp16" else 0
        self._dtype = torch.float16 if precision == "fp16" else torch.bfloat16
        self._sd: Optional[Dict[str, torch.Tensor]] = None
        self._dev: Dict[str, torch.Tensor] = {}
        self._device: Optional[torch.device] = None
        self.training = False

    # ---- nn.Module-like surface -------------------------------------------------------------
    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = True):
        want = expected_keys(self.latent_dim, self.joints_num, self.number_blocks)
        missing = [k for k in want if k not in state_dict]
        unexpected = [k for k in state_dict if k not in want]
        if missing or (strict and unexpected):
            raise KeyError(f"load_state_dict: missing {missing[:4]}{'...' if len(missing) > 4 else ''}, "
                           f"unexpected {unexpected[:4]}{'...' if len(unexpected) > 4 else ''}")
        for k, shape in want.items():
            if tuple(state_dict[k].shape) != shape:
                raise ValueError(f"load_state_dict: {k} has shape {tuple(state_dict[k].shape)}, expected {shape}")
        self._sd = {k: state_dict[k].detach().to("cpu", torch.float32).contiguous() for k in want}
        if self._device is not None:
            self._upload()
        return self

    def to(self, device) -> "PHDFor3DJoints":
        device = torch.device(device)
        if device.type != "cuda":
            raise _lib.R50Error(f"PHDFor3DJoints runs on an MI355X only (got device '{device}'); "
                                "there is no CPU fallback in the product path")
        _lib.load_library()
        index = device.index if device.index is not None else torch.cuda.current_device()
        self._device = torch.device("cuda", index)
        if self._sd is not None:
            self._upload()
        return self

    def eval(self) -> "PHDFor3DJoints":
        self.training = False
        return self

    def train(self, mode: bool = True):
        if mode:
            raise _lib.R50Error("PHDFor3DJoints here is the forward pass only (eval mode); training is not implemented")
        return self

    # ---- weights: repacked once, GEMM-ready --------------------------------------------------
    def _upload(self) -> None:
        sd, dev, dt = self._sd, self._device, self._dtype
        d, o = self.latent_dim, self.out_dim
        w: Dict[str, torch.Tensor] = {}

        def elem(t):
            return t.to(dt).contiguous().to(dev)

        def f32(t):
            return t.to(torch.float32).contiguous().to(dev)

        w["input_proj.w"] = elem(sd["input_proj.weight"])                        # (D, 2048) = (cout, 1, 1, cin)
        w["input_proj.b"] = f32(sd["input_proj.bias"])
        for net, nb in (("f_movie", self.number_blocks), ("f_AR", _AR_BLOCKS)):
            for i in range(nb):
                p = f"{net}.blocks.{i}"
                for gn in ("gn1", "gn2"):
                    w[f"{p}.{gn}.g"] = f32(sd[f"{p}.{gn}.weight"])
                    w[f"{p}.{gn}.b"] = f32(sd[f"{p}.{gn}.bias"])
                for cv in ("conv1", "conv2"):
                    # (cout, cin, k) -> (cout, k, cin): column k*D + c multiplies x(t-2+k)[c]
                    w[f"{p}.{cv}.w"] = elem(sd[f"{p}.{cv}.conv.weight"].permute(0, 2, 1).reshape(d, 3 * d))
                    w[f"{p}.{cv}.b"] = f32(sd[f"{p}.{cv}.conv.bias"])
        self._dp = _round_up(d + o, 64)                                          # K of the regressor's first Linear
        self._op = _round_up(o, 64)                                              # its last Linear's cout
        w0 = torch.zeros(_REG_HIDDEN, self._dp)
        w0[:, : d + o] = sd["f_3D.mlp.0.weight"]
        w5 = torch.zeros(self._op, _REG_HIDDEN)
        w5[:o] = sd["f_3D.mlp.5.weight"]
        b5 = torch.zeros(self._op)
        b5[:o] = sd["f_3D.mlp.5.bias"]
        w["mlp0.w"], w["mlp0.b"] = elem(w0), f32(sd["f_3D.mlp.0.bias"])
        w["mlp3.w"], w["mlp3.b"] = elem(sd["f_3D.mlp.3.weight"]), f32(sd["f_3D.mlp.3.bias"])
        w["mlp5.w"], w["mlp5.b"] = elem(w5), f32(b5)
        w["y0"] = f32(sd["f_3D.y0"])
        self._dev = w

    # ---- launches ---------------------------------------------------------------------------
    def _stream(self) -> int:
        return torch.cuda.current_stream(self._device).cuda_stream

    def _gemm(self, x: torch.Tensor, wname: str, relu: bool, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        """rows (R, K) element @ W (cout, K)^T + bias [+ residual] [ReLU] -> (R, cout) element: a 1x1 convolution over R pixels."""
        wt, b = self._dev[wname + ".w"], self._dev[wname + ".b"]
        rows, k = x.shape
        cout = wt.shape[0]
        assert wt.shape[1] == k and x.is_contiguous()
        y = torch.empty((rows, cout), dtype=self._dtype, device=self._device)
        lib = _lib.load_library()
        fn = lib.r50_op_conv2d_f16 if self._et else lib.r50_op_conv2d
        rc = fn(x.data_ptr(), rows, 1, 1, k, wt.data_ptr(), b.data_ptr(), residual.data_ptr() if residual is not None else None,
                y.data_ptr(), cout, 1, 1, 0, int(relu), 0, self._stream())
        _lib.check(rc, None, "r50_op_conv2d (lifting head)")
        return y

    def _gn_relu_rows(self, x: torch.Tensor, b: int, t: int, prefix: str) -> torch.Tensor:
        d = self.latent_dim
        out = torch.empty((b * t, 3 * d), dtype=self._dtype, device=self._device)
        rc = _lib.load_library().r50_op_gn_relu_causal3(x.data_ptr(), b, t, d, _GROUPS, self._dev[prefix + ".g"].data_ptr(),
                                                        self._dev[prefix + ".b"].data_ptr(), _GN_EPS, out.data_ptr(), self._et,
                                                        self._stream())
        _lib.check(rc, None, "r50_op_gn_relu_causal3")
        return out

    def _temporal_net(self, x: torch.Tensor, b: int, t: int, net: str, nb: int) -> torch.Tensor:   # CausalTemporalNet, :69-78
        for i in range(nb):
            p = f"{net}.blocks.{i}"
            h = self._gemm(self._gn_relu_rows(x, b, t, p + ".gn1"), p + ".conv1", relu=False)
            x = self._gemm(self._gn_relu_rows(h, b, t, p + ".gn2"), p + ".conv2", relu=False, residual=x)
        return x

    def _regressor(self, phi: torch.Tensor, b: int, t: int) -> torch.Tensor:                       # JointRegressor.forward, :104-126
        rows, d, o = b * t, self.latent_dim, self.out_dim
        lib = _lib.load_library()
        y = self._dev["y0"].view(1, o).expand(rows, o).contiguous()
        inp = torch.empty((rows, self._dp), dtype=self._dtype, device=self._device)
        for _ in range(_REG_ITERS):
            _lib.check(lib.r50_op_concat_pad(phi.data_ptr(), d, y.data_ptr(), o, rows, inp.data_ptr(), self._dp, self._et,
                                             self._stream()), None, "r50_op_concat_pad")
            h = self._gemm(inp, "mlp0", relu=True)
            h = self._gemm(h, "mlp3", relu=True)
            dy = self._gemm(h, "mlp5", relu=False)
            _lib.check(lib.r50_op_add_rows(y.data_ptr(), o, dy.data_ptr(), self._op, rows, self._et, self._stream()), None,
                       "r50_op_add_rows")
        return y.view(b, t, self.joints_num, 3)

    def __call__(self, feats: torch.Tensor, predict_future: bool = False):
        if self._device is None or not self._dev:
            raise _lib.R50Error("call .load_state_dict(...) and .to('cuda:N') before running the head")
        if feats.dim() != 3 or feats.shape[-1] != 2048:
            raise ValueError(f"expected (B,T,2048) features, got {tuple(feats.shape)}")
        if feats.device != self._device:
            raise ValueError(f"features are on {feats.device}, head on {self._device}")
        b, t, _ = feats.shape
        if b * t == 0:
            raise ValueError("empty batch")
        d = self.latent_dim
        lib = _lib.load_library()
        with torch.cuda.device(self._device):
            f = feats.to(torch.float32).contiguous()
            x0 = torch.empty((b * t, 2048), dtype=self._dtype, device=self._device)
            _lib.check(lib.r50_op_cast_rows(f.data_ptr(), b * t, 2048, x0.data_ptr(), 2048, self._et, self._stream()), None,
                       "r50_op_cast_rows")
            x = self._gemm(x0, "input_proj", relu=False)                                     # :155
            phi = self._temporal_net(x, b, t, "f_movie", self.number_blocks)                 # :156
            ar = self._temporal_net(phi, b, t, "f_AR", _AR_BLOCKS)                           # :158
            phi_hat = torch.zeros_like(ar).view(b, t, d)                                     # :159-160 (a shifted copy)
            phi_hat[:, 1:, :] = ar.view(b, t, d)[:, :-1, :]
            joints_phi = self._regressor(phi, b, t)                                          # :162
            joints_hat = self._regressor(phi_hat.view(b * t, d), b, t) if predict_future else None   # :164-166
        return phi.view(b, t, d).float(), phi_hat.float(), joints_phi, joints_hat

    forward = __call__


PHD = PHDFor3DJoints       # the name src/train.py imports it under
