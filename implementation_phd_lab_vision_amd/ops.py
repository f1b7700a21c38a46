"""Thin wrappers over the op-level C-ABI entry points (``r50_op_*`` in include/r50.h).

Each mirrors one module of the reference's ``nn.Sequential`` (upstream torchvision
``models/resnet.py``): Conv2d(+folded BatchNorm2d)(+residual)(+ReLU), MaxPool2d(3,2,1),
AdaptiveAvgPool2d((1,1)).  Tensors are bf16 NHWC on an MI355X; used by the per-kernel parity tests.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib

# tile-config ids of the implicit-GEMM kernel (BC couts x BP pixels; DESIGN.md "Kernels")
TILE_AUTO = 0
TILE_128x128 = 1         # 4 waves, 2 LDS stages
TILE_64x128 = 2
TILE_64x256 = 3
TILE_128x64 = 5
TILE_256x128_P3 = 6      # 8 waves, 3 LDS stages, counted vmcnt
TILE_128x256_P3 = 7
TILE_128x128_P3 = 8
TILE_256x256 = 9         # 8 waves, 2 LDS stages, wave tile 64c x 128p
TILE_256x256_B = 10      # 8 waves, 2 LDS stages, wave tile 128c x 64p
TILE_256x208 = 11        # 8 waves, 2 LDS stages, wave tile 32c x 208p (208 = 13 x 16 fits M = 2^10 * 49)
TILE_256x224 = 12        # 8 waves, 2 LDS stages, wave tile 64c x 112p
WS = 64                  # role-specialised kernel (4 loader waves + 4 or 8 consumer waves, one workgroup per CU):
                         # WS|1 = 128x128 (4), WS|3 = 256x128 (8), WS|4 = 128x224 (4), WS|8 = 128x224 (8), WS|9 = 64x224 (4), WS|10 = 128x208 (4)
TILE_XRES = 81           # 3x3 s1 p1 with Cin = Cout on 28x28 (128), 14x14 (256), 7x7 (512): input resident in LDS, only the weights stream (the automatic choice for these shapes)
TILE_S2 = 82             # 3x3 STRIDE 2 p1 with Cin = Cout, 56 -> 28 (128) and 28 -> 14 (256): input resident by polyphase planes (the automatic choice for these shapes)
TILE_G8 = 83              # 1x1, pad 0, Cout % 256 == 0: 256 couts x 256 pixels on the eight-phase schedule (gemm8p_kernel); also for conv1x1_cat
TILE_G8_224 = 84          # the same with 224 pixels per tile (14 x 16 divides the pixel counts 2^k * 49 of the network)
TILE_C64 = 80            # 3x3 s1 p1 64 -> 64 on 56x56 only (layer1 conv2): filter bank resident in LDS, input staged once per tile
PERSISTENT = 32          # + PERSISTENT: chip-sized grid, tiles streamed through the LDS ring


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def _need(t: torch.Tensor, dtype, name: str) -> None:
    if not t.is_cuda or t.dtype != dtype or not t.is_contiguous():
        raise ValueError(f"{name}: need a contiguous {dtype} tensor on the GPU")


def conv2d_bf16(x: torch.Tensor, w_ohwi: torch.Tensor, bias: torch.Tensor, stride: int = 1, pad: int = 0,
                relu: bool = True, residual: Optional[torch.Tensor] = None, tile: int = TILE_AUTO,
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x (N,H,W,Cin) bf16, w_ohwi (Cout,k,k,Cin) bf16, bias (Cout) fp32 -> (N,Ho,Wo,Cout) bf16.
    ``out``: optional flat bf16 buffer with at least N*Ho*Wo*Cout elements (tests use it to put a
    guard band behind the result)."""
    et = x.dtype                      # bf16, or fp16 (the R50_PREC_FP16 element type): every tensor in the same format
    if et not in (torch.bfloat16, torch.float16):
        raise ValueError("conv2d_bf16: x must be bf16 or fp16")
    _need(x, et, "x"); _need(w_ohwi, et, "w"); _need(bias, torch.float32, "bias")
    n, h, w, cin = x.shape
    cout, k, k2, cin2 = w_ohwi.shape
    if k != k2 or cin2 != cin or bias.numel() != cout:
        raise ValueError("conv2d_bf16: inconsistent shapes")
    ho = (h + 2 * pad - k) // stride + 1
    wo = (w + 2 * pad - k) // stride + 1
    if out is not None:
        _need(out, et, "out")
        if out.numel() < n * ho * wo * cout:
            raise ValueError("conv2d_bf16: out buffer too small")
        y = out.view(-1)[: n * ho * wo * cout].view(n, ho, wo, cout)
    else:
        y = torch.empty((n, ho, wo, cout), dtype=et, device=x.device)
    if residual is not None:
        _need(residual, et, "residual")
        if residual.shape != y.shape:
            raise ValueError("conv2d_bf16: residual shape mismatch")
    lib = _lib.load_library()
    with torch.cuda.device(x.device):
        fn = lib.r50_op_conv2d if et == torch.bfloat16 else lib.r50_op_conv2d_f16
        rc = fn(x.data_ptr(), n, h, w, cin, w_ohwi.data_ptr(), bias.data_ptr(),
                               residual.data_ptr() if residual is not None else None, y.data_ptr(),
                               cout, k, stride, pad, int(relu), int(tile), _stream(x))
    _lib.check(rc, None, "r50_op_conv2d")
    return y


FP8 = torch.float8_e4m3fn        # OCP e4m3: what gfx950's fp8 MFMA and conversions use
FP8_MAX = 448.0


def conv2d_fp8(x: torch.Tensor, sx: float, w_ohwi: torch.Tensor, sw: float, bias: torch.Tensor, sy: float, stride: int = 1, pad: int = 0,
               relu: bool = True, residual: Optional[torch.Tensor] = None, sr: float = 1.0, tile: int = TILE_AUTO) -> torch.Tensor:
    """fp8 conv (BASELINE configs[4]), kernel level: x (N,H,W,Cin) / w (Cout,k,k,Cin) / residual / result in ``float8_e4m3fn`` with
    per-tensor scales (real = stored * scale), bias fp32 in real units -> y (N,Ho,Wo,Cout) fp8 with scale ``sy``:
    ``y = fp8(act(sx*sw * sum(x*w) + bias + sr*residual) / sy)``, fp32 accumulation on the K = 128 scaled fp8 MFMA."""
    for t, nm in ((x, "x"), (w_ohwi, "w")):
        _need(t, FP8, nm)
    _need(bias, torch.float32, "bias")
    n, h, w, cin = x.shape
    cout, k, k2, cin2 = w_ohwi.shape
    if k != k2 or cin2 != cin or bias.numel() != cout:
        raise ValueError("conv2d_fp8: inconsistent shapes")
    ho = (h + 2 * pad - k) // stride + 1
    wo = (w + 2 * pad - k) // stride + 1
    y = torch.empty((n, ho, wo, cout), dtype=FP8, device=x.device)
    if residual is not None:
        _need(residual, FP8, "residual")
        if residual.shape != y.shape:
            raise ValueError("conv2d_fp8: residual shape mismatch")
    bias_scaled = (bias / (sx * sw)).contiguous()
    with torch.cuda.device(x.device):
        rc = _lib.load_library().r50_op_conv2d_fp8(x.data_ptr(), n, h, w, cin, w_ohwi.data_ptr(), bias_scaled.data_ptr(),
                                                   residual.data_ptr() if residual is not None else None, y.data_ptr(), cout, k, stride,
                                                   pad, int(relu), float(sx * sw / sy), float(sr / sy), int(tile), _stream(x))
    _lib.check(rc, None, "r50_op_conv2d_fp8")
    return y


def conv1x1_cat(x1: torch.Tensor, x2: torch.Tensor, stride2: int, wcat: torch.Tensor, bias: torch.Tensor, relu: bool = True,
                tile: int = TILE_AUTO) -> torch.Tensor:
    """``act([W1 | W2] . [x1 ; x2 at stride2] + bias)``: conv3 + downsample + add + ReLU of a stage's first bottleneck as one 1x1
    conv.  x1 (N,H,W,C1), x2 (N,H2,W2,C2) with (H2-1)//stride2+1 == H, wcat (Cout, C1+C2), all bf16 or all fp16; bias fp32."""
    et = x1.dtype
    if et not in (torch.bfloat16, torch.float16):
        raise ValueError("conv1x1_cat: x1 must be bf16 or fp16")
    _need(x1, et, "x1"); _need(x2, et, "x2"); _need(wcat, et, "wcat"); _need(bias, torch.float32, "bias")
    n, h, w, c1 = x1.shape
    n2, h2, w2, c2 = x2.shape
    cout = wcat.shape[0]
    if n2 != n or tuple(wcat.shape) != (cout, c1 + c2) or bias.numel() != cout:
        raise ValueError("conv1x1_cat: inconsistent shapes")
    y = torch.empty((n, h, w, cout), dtype=et, device=x1.device)
    with torch.cuda.device(x1.device):
        rc = _lib.load_library().r50_op_conv1x1_cat(x1.data_ptr(), n, h, w, c1, x2.data_ptr(), h2, w2, c2, int(stride2), wcat.data_ptr(),
                                                    bias.data_ptr(), y.data_ptr(), cout, int(relu), int(tile),
                                                    1 if et == torch.float16 else 0, _stream(x1))
    _lib.check(rc, None, "r50_op_conv1x1_cat")
    return y


def bneck_tail_bf16(y2: torch.Tensor, w3: torch.Tensor, b3: torch.Tensor, identity: torch.Tensor, w1: torch.Tensor,
                    b1: torch.Tensor, wd: Optional[torch.Tensor] = None, bd: Optional[torch.Tensor] = None,
                    out: Optional[torch.Tensor] = None, y1n: Optional[torch.Tensor] = None):
    """Fused tail of a bottleneck: ``out = relu(conv3(y2) + b3 + identity)`` and the next block's
    ``y1n = relu(conv1(out) + b1)``.  y2 (...,cmid), w3 (4*cmid,cmid) / w1 (c1,4*cmid) folded bf16, biases fp32 ->
    (out (...,4*cmid), y1n (...,c1)) bf16; cmid = 64 (layer1), 128 (layer2, c1 = 128) or 256 (layer3, c1 = 256: the chained
    kernel, weights streamed through LDS).  ``out`` / ``y1n``: optional preallocated outputs.  ``identity`` is the
    (...,4*cmid) identity tensor, or -- cmid = 64 with the folded downsample weights ``wd`` (256,64) / ``bd`` --
    the (...,64) block input."""
    for t, n in ((y2, "y2"), (w3, "w3"), (identity, "identity"), (w1, "w1")):
        _need(t, torch.bfloat16, n)
    _need(b3, torch.float32, "b3"); _need(b1, torch.float32, "b1")
    cmid = y2.shape[-1]
    cout = 4 * cmid
    m = y2.numel() // cmid
    c1 = w1.shape[0]
    cid = cout if wd is None else cmid
    if (wd is None) != (bd is None):
        raise ValueError("bneck_tail_bf16: wd and bd go together")
    if wd is not None:
        _need(wd, torch.bfloat16, "wd"); _need(bd, torch.float32, "bd")
        if tuple(wd.shape) != (cout, cmid) or bd.numel() != cout:
            raise ValueError("bneck_tail_bf16: inconsistent downsample shapes")
    if identity.shape[-1] != cid or identity.numel() != m * cid or tuple(w3.shape) != (cout, cmid) \
            or tuple(w1.shape) != (c1, cout) or b3.numel() != cout or b1.numel() != c1:
        raise ValueError("bneck_tail_bf16: inconsistent shapes")
    if out is None:
        out = torch.empty(tuple(y2.shape[:-1]) + (cout,), dtype=torch.bfloat16, device=y2.device)
    if y1n is None:
        y1n = torch.empty(tuple(y2.shape[:-1]) + (c1,), dtype=torch.bfloat16, device=y2.device)
    _need(out, torch.bfloat16, "out"); _need(y1n, torch.bfloat16, "y1n")
    if out.numel() != m * cout or y1n.numel() != m * c1:
        raise ValueError("bneck_tail_bf16: output tensors have the wrong size")
    with torch.cuda.device(y2.device):
        rc = _lib.load_library().r50_op_bneck_tail(y2.data_ptr(), m, cmid, w3.data_ptr(), b3.data_ptr(), identity.data_ptr(),
                                                   wd.data_ptr() if wd is not None else None,
                                                   bd.data_ptr() if bd is not None else None,
                                                   out.data_ptr(), w1.data_ptr(), c1, b1.data_ptr(), y1n.data_ptr(), _stream(y2))
    _lib.check(rc, None, "r50_op_bneck_tail")
    return out, y1n


def conv3_identity_tail3_bf16(y2: torch.Tensor, w3: torch.Tensor, b3: torch.Tensor, identity: torch.Tensor) -> torch.Tensor:
    """``relu(conv3(y2) + b3 + identity)`` of a layer3 block (y2 (...,256), w3 (1024,256), identity (...,1024)) through the pipelined tail kernel
    WITHOUT a next conv1 (``r50_op_bneck_tail`` with c1 = 0): the form the stage's last block, layer3.5, runs in."""
    for t, n in ((y2, "y2"), (w3, "w3"), (identity, "identity")):
        _need(t, torch.bfloat16, n)
    _need(b3, torch.float32, "b3")
    m = y2.numel() // 256
    if y2.shape[-1] != 256 or tuple(w3.shape) != (1024, 256) or identity.numel() != m * 1024 or b3.numel() != 1024:
        raise ValueError("conv3_identity_tail3_bf16: inconsistent shapes")
    out = torch.empty(tuple(y2.shape[:-1]) + (1024,), dtype=torch.bfloat16, device=y2.device)
    with torch.cuda.device(y2.device):
        rc = _lib.load_library().r50_op_bneck_tail(y2.data_ptr(), m, 256, w3.data_ptr(), b3.data_ptr(), identity.data_ptr(), None, None,
                                                   out.data_ptr(), None, 0, None, None, _stream(y2))
    _lib.check(rc, None, "r50_op_bneck_tail")
    return out


def bneck_block2_bf16(t1: torch.Tensor, w2: torch.Tensor, b2: torch.Tensor, w3: torch.Tensor, b3: torch.Tensor, identity: torch.Tensor,
                      w1: Optional[torch.Tensor] = None, b1: Optional[torch.Tensor] = None):
    """Layer2 bottleneck body in one launch (``r50_op_bneck_block2``): t1 (N,28,28,128), identity (N,28,28,512) bf16 NHWC; w2 (128,3,3,128),
    w3 (512,128), w1 (128,512) bf16, K contiguous; biases fp32.  Returns (block output (N,28,28,512), next t1 (N,28,28,128) or None)."""
    for t, name in ((t1, "t1"), (w2, "w2"), (w3, "w3"), (identity, "identity")):
        _need(t, torch.bfloat16, name)
    _need(b2, torch.float32, "b2"); _need(b3, torch.float32, "b3")
    n = t1.shape[0]
    if tuple(t1.shape) != (n, 28, 28, 128) or tuple(identity.shape) != (n, 28, 28, 512) or tuple(w2.shape) != (128, 3, 3, 128) \
            or tuple(w3.shape) != (512, 128) or b2.numel() != 128 or b3.numel() != 512:
        raise ValueError("bneck_block2_bf16: inconsistent shapes")
    if (w1 is None) != (b1 is None):
        raise ValueError("bneck_block2_bf16: w1 and b1 go together")
    out = torch.empty((n, 28, 28, 512), dtype=torch.bfloat16, device=t1.device)
    y1n = None
    if w1 is not None:
        _need(w1, torch.bfloat16, "w1"); _need(b1, torch.float32, "b1")
        if tuple(w1.shape) != (128, 512) or b1.numel() != 128:
            raise ValueError("bneck_block2_bf16: inconsistent next-conv1 shapes")
        y1n = torch.empty((n, 28, 28, 128), dtype=torch.bfloat16, device=t1.device)
    with torch.cuda.device(t1.device):
        rc = _lib.load_library().r50_op_bneck_block2(t1.data_ptr(), n, w2.data_ptr(), b2.data_ptr(), w3.data_ptr(), b3.data_ptr(),
                                                     identity.data_ptr(), out.data_ptr(),
                                                     w1.data_ptr() if w1 is not None else None, b1.data_ptr() if b1 is not None else None,
                                                     y1n.data_ptr() if y1n is not None else None, _stream(t1))
    _lib.check(rc, None, "r50_op_bneck_block2")
    return out, y1n


def bneck_block1_ds_bf16(t1: torch.Tensor, w2: torch.Tensor, b2: torch.Tensor, w3: torch.Tensor, b3: torch.Tensor, x: torch.Tensor,
                         wd: torch.Tensor, bd: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor):
    """layer1.0's bottleneck body in one launch (``r50_op_bneck_block1_ds``): t1 (N,56,56,64), block input x (N,56,56,64) bf16 NHWC; w2 (64,3,3,64),
    w3 (256,64), wd (256,64), w1 (64,256) bf16; biases fp32.  Returns (block output (N,56,56,256), next t1 (N,56,56,64))."""
    for t, name in ((t1, "t1"), (w2, "w2"), (w3, "w3"), (x, "x"), (wd, "wd"), (w1, "w1")):
        _need(t, torch.bfloat16, name)
    for t, name in ((b2, "b2"), (b3, "b3"), (bd, "bd"), (b1, "b1")):
        _need(t, torch.float32, name)
    n = t1.shape[0]
    if tuple(t1.shape) != (n, 56, 56, 64) or tuple(x.shape) != (n, 56, 56, 64) or tuple(w2.shape) != (64, 3, 3, 64) or tuple(w3.shape) != (256, 64) \
            or tuple(wd.shape) != (256, 64) or tuple(w1.shape) != (64, 256) or b2.numel() != 64 or b3.numel() != 256 or bd.numel() != 256 or b1.numel() != 64:
        raise ValueError("bneck_block1_ds_bf16: inconsistent shapes")
    out = torch.empty((n, 56, 56, 256), dtype=torch.bfloat16, device=t1.device)
    y1n = torch.empty((n, 56, 56, 64), dtype=torch.bfloat16, device=t1.device)
    with torch.cuda.device(t1.device):
        rc = _lib.load_library().r50_op_bneck_block1_ds(t1.data_ptr(), n, w2.data_ptr(), b2.data_ptr(), w3.data_ptr(), b3.data_ptr(), x.data_ptr(),
                                                        wd.data_ptr(), bd.data_ptr(), out.data_ptr(), w1.data_ptr(), b1.data_ptr(), y1n.data_ptr(),
                                                        _stream(t1))
    _lib.check(rc, None, "r50_op_bneck_block1_ds")
    return out, y1n


def bneck_cat_chain_bf16(t2: torch.Tensor, x: torch.Tensor, wcat: torch.Tensor, bcat: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor):
    """layer2.0's transition tail chained with layer2.1.conv1 in one launch (``r50_op_bneck_cat_chain``): t2 (N,28,28,128), block input x
    (N,56,56,256) bf16 NHWC; wcat (512,384) = [W3 | Wd], w1 (128,512) bf16, K contiguous; bcat = b3 + bd, b1 fp32.
    Returns (block output (N,28,28,512), next t1 (N,28,28,128))."""
    for t, name in ((t2, "t2"), (x, "x"), (wcat, "wcat"), (w1, "w1")):
        _need(t, torch.bfloat16, name)
    _need(bcat, torch.float32, "bcat"); _need(b1, torch.float32, "b1")
    n = t2.shape[0]
    if tuple(t2.shape) != (n, 28, 28, 128) or tuple(x.shape) != (n, 56, 56, 256) or tuple(wcat.shape) != (512, 384) or tuple(w1.shape) != (128, 512) \
            or bcat.numel() != 512 or b1.numel() != 128:
        raise ValueError("bneck_cat_chain_bf16: inconsistent shapes")
    out = torch.empty((n, 28, 28, 512), dtype=torch.bfloat16, device=t2.device)
    y1n = torch.empty((n, 28, 28, 128), dtype=torch.bfloat16, device=t2.device)
    with torch.cuda.device(t2.device):
        rc = _lib.load_library().r50_op_bneck_cat_chain(t2.data_ptr(), x.data_ptr(), n, 28, wcat.data_ptr(), bcat.data_ptr(), out.data_ptr(),
                                                        w1.data_ptr(), b1.data_ptr(), y1n.data_ptr(), _stream(t2))
    _lib.check(rc, None, "r50_op_bneck_cat_chain")
    return out, y1n


def bneck_block1_bf16(t1: torch.Tensor, w2: torch.Tensor, b2: torch.Tensor, w3: torch.Tensor, b3: torch.Tensor, identity: torch.Tensor,
                      w1: torch.Tensor, b1: torch.Tensor):
    """Layer1 bottleneck body in one launch (``r50_op_bneck_block1``): t1 (N,56,56,64), identity (N,56,56,256) bf16 NHWC; w2 (64,3,3,64),
    w3 (256,64), w1 (c1,256) bf16 with c1 in {64, 128}; biases fp32.  Returns (block output (N,56,56,256), next t1 (N,56,56,c1))."""
    for t, name in ((t1, "t1"), (w2, "w2"), (w3, "w3"), (identity, "identity"), (w1, "w1")):
        _need(t, torch.bfloat16, name)
    _need(b2, torch.float32, "b2"); _need(b3, torch.float32, "b3"); _need(b1, torch.float32, "b1")
    n = t1.shape[0]
    c1 = w1.shape[0]
    if tuple(t1.shape) != (n, 56, 56, 64) or tuple(identity.shape) != (n, 56, 56, 256) or tuple(w2.shape) != (64, 3, 3, 64) \
            or tuple(w3.shape) != (256, 64) or tuple(w1.shape) != (c1, 256) or c1 not in (64, 128) or b2.numel() != 64 or b3.numel() != 256 \
            or b1.numel() != c1:
        raise ValueError("bneck_block1_bf16: inconsistent shapes")
    out = torch.empty((n, 56, 56, 256), dtype=torch.bfloat16, device=t1.device)
    y1n = torch.empty((n, 56, 56, c1), dtype=torch.bfloat16, device=t1.device)
    with torch.cuda.device(t1.device):
        rc = _lib.load_library().r50_op_bneck_block1(t1.data_ptr(), n, w2.data_ptr(), b2.data_ptr(), w3.data_ptr(), b3.data_ptr(),
                                                     identity.data_ptr(), out.data_ptr(), w1.data_ptr(), c1, b1.data_ptr(), y1n.data_ptr(),
                                                     _stream(t1))
    _lib.check(rc, None, "r50_op_bneck_block1")
    return out, y1n


def stem_bf16(x_nchw: torch.Tensor, w_folded_oihw: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """conv1 7x7 s2 p3 + folded bn1 + ReLU: (N,3,224,224) fp32 NCHW -> (N,112,112,64) bf16 NHWC.
    ``w_folded_oihw``: (64,3,7,7) fp32 on the HOST (already BN-folded); ``bias``: (64) fp32 on the GPU."""
    _need(x_nchw, torch.float32, "x"); _need(bias, torch.float32, "bias")
    if tuple(x_nchw.shape[1:]) != (3, 224, 224) or tuple(w_folded_oihw.shape) != (64, 3, 7, 7):
        raise ValueError("stem_bf16: bad shapes")
    w_host = w_folded_oihw.detach().to("cpu", torch.float32).contiguous()
    n = x_nchw.shape[0]
    lib = _lib.load_library()
    scratch = torch.empty(lib.r50_stem_scratch_bytes(n), dtype=torch.uint8, device=x_nchw.device)
    y = torch.empty((n, 112, 112, 64), dtype=torch.bfloat16, device=x_nchw.device)
    with torch.cuda.device(x_nchw.device):
        rc = lib.r50_op_stem(x_nchw.data_ptr(), n, w_host.data_ptr(), bias.data_ptr(), scratch.data_ptr(),
                             y.data_ptr(), _stream(x_nchw))
    _lib.check(rc, None, "r50_op_stem")
    return y


def maxpool_bf16(x: torch.Tensor) -> torch.Tensor:
    """MaxPool2d(3, stride 2, pad 1) on (N,H,W,C) bf16."""
    _need(x, torch.bfloat16, "x")
    n, h, w, c = x.shape
    y = torch.empty((n, (h - 1) // 2 + 1, (w - 1) // 2 + 1, c), dtype=torch.bfloat16, device=x.device)
    with torch.cuda.device(x.device):
        rc = _lib.load_library().r50_op_maxpool(x.data_ptr(), n, h, w, c, y.data_ptr(), _stream(x))
    _lib.check(rc, None, "r50_op_maxpool")
    return y


def avgpool_bf16(x: torch.Tensor) -> torch.Tensor:
    """AdaptiveAvgPool2d((1,1)) + flatten(1) on (N,H,W,C) bf16 -> (N,C) fp32."""
    _need(x, torch.bfloat16, "x")
    n, h, w, c = x.shape
    y = torch.empty((n, c), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = _lib.load_library().r50_op_avgpool(x.data_ptr(), n, h * w, c, y.data_ptr(), _stream(x))
    _lib.check(rc, None, "r50_op_avgpool")
    return y
