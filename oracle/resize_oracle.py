"""CPU restatement of the frame producer's crop + resize.  TEST INFRASTRUCTURE ONLY:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Reference: ``_crop_and_resize_video_uint8`` (src/dataset.py:141-152): ``frames[:, :, top:top+hh, left:left+ww]`` of
the decoded ``(T,H,W,3)`` uint8 clip, then ``torchvision.transforms.v2.functional.resize(frames, [224, 224],
antialias=False)`` (bilinear).  torchvision (not installed here, not pinned by the reference) passes a uint8 CPU tensor
straight to ``torch.nn.functional.interpolate(mode="bilinear", align_corners=False, antialias=False)`` when the CPU has
AVX2 (``_do_native_uint8_resize_on_cpu``), i.e. to ATen's native uint8 kernel
(aten/src/ATen/native/cpu/UpSampleKernelAVXAntialias.h, ``upsample_avx_bilinear_bicubic_uint8``): a separable
two-pass (horizontal, then vertical) fixed-point resample with int16 weights and a uint8 intermediate -- the
Pillow-SIMD scheme.  That is what ``resize_bilinear_u8`` restates; it is pinned against ``F.interpolate`` itself by
tests/test_resize_oracle_cpu.py.  The reference file imports the v1 API (``import torchvision.transforms.functional as
F``, :13), whose tensor path converts uint8 to float32, calls the same ``interpolate`` and rounds
(``torchvision/transforms/_functional_tensor.py: resize -> _cast_squeeze_in / _cast_squeeze_out``):
``resize_bilinear_u8_float`` (torch's own kernel) and ``resize_bilinear_u8_float_restated`` (numpy, the HIP kernel's
operation order).  The two routes differ by at most 1 LSB on ~20 % of the pixels.
"""
from __future__ import annotations

import numpy as np


def _linear_filter(x: float) -> float:
    x = abs(x)
    return 1.0 - x if x < 1.0 else 0.0


def index_weights_int16(in_size: int, out_size: int):
    """Per output index: (first source index, number of taps (1 or 2), int16 weights), and the weight precision.
    ATen: ``_compute_indices_min_size_weights`` (UpSampleKernel.cpp) for the non-antialiased linear filter, then
    ``_compute_index_ranges_int16_weights``: precision = the largest p with round(max_weight * 2^(p+1)) < 2^15."""
    scale = in_size / out_size                       # area_pixel_compute_scale, align_corners=False, no scale_factor
    xmin = np.zeros(out_size, dtype=np.int64)
    xsize = np.zeros(out_size, dtype=np.int64)
    w = np.zeros((out_size, 2), dtype=np.float64)
    for i in range(out_size):
        real = scale * (i + 0.5) - 0.5               # area_pixel_compute_source_index (linear: clamped at 0)
        if real < 0.0:
            real = 0.0
        idx = min(int(np.floor(real)), in_size - 1)  # guard_index_and_lambda
        lam = min(max(real - idx, 0.0), 1.0)
        support = 1
        umin = idx - support + 1
        umax = idx + support + 1
        lo = max(umin, 0)
        size = min(umax, in_size) - lo
        w_index = 0
        for j in range(2):
            wj = _linear_filter(float(j + 1 - support) - lam)
            if umin + j <= 0:
                w_index = 0
            elif umin + j >= in_size - 1:
                w_index = size - 1
            w[i, w_index] += wj
            w_index += 1
        xmin[i], xsize[i] = lo, size
    wt_max = float(w.max())
    precision = 0
    while precision < 22:
        if int(0.5 + wt_max * (1 << (precision + 1))) >= (1 << 15):
            break
        precision += 1
    wi = np.where(w < 0, (-0.5 + w * (1 << precision)).astype(np.int64), (0.5 + w * (1 << precision)).astype(np.int64))
    return xmin, xsize, wi.astype(np.int64), precision


def _clip8(v: np.ndarray) -> np.ndarray:
    return np.clip(v, 0, 255).astype(np.uint8)


def resize_bilinear_u8(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """img (..., H, W) uint8 -> (..., out_h, out_w) uint8, ATen's native uint8 bilinear (antialias=False)."""
    img = np.asarray(img, dtype=np.uint8)
    h, w = img.shape[-2:]
    cur = img.astype(np.int64)
    if w != out_w:                                   # horizontal pass first, uint8 intermediate
        xmin, xsize, wi, p = index_weights_int16(w, out_w)
        x1 = np.minimum(xmin + 1, w - 1)
        acc = cur[..., :, xmin] * wi[:, 0] + np.where(xsize > 1, cur[..., :, x1] * wi[:, 1], 0)
        cur = _clip8((acc + (1 << (p - 1))) >> p).astype(np.int64)
    if h != out_h:
        ymin, ysize, wi, p = index_weights_int16(h, out_h)
        y1 = np.minimum(ymin + 1, h - 1)
        acc = cur[..., ymin, :] * wi[:, 0][:, None] + np.where((ysize > 1)[:, None], cur[..., y1, :] * wi[:, 1][:, None], 0)
        cur = _clip8((acc + (1 << (p - 1))) >> p).astype(np.int64)
    return cur.astype(np.uint8)


def resize_bilinear_u8_float(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """The non-AVX2 route of torchvision: float32 bilinear (align_corners=False), round half to even, uint8."""
    import torch
    import torch.nn.functional as F
    t = torch.from_numpy(np.ascontiguousarray(img)).to(torch.float32)
    lead = t.shape[:-2]
    t = t.reshape((-1, 1) + tuple(t.shape[-2:]))
    y = F.interpolate(t, size=[out_h, out_w], mode="bilinear", align_corners=False, antialias=False)
    return y.round_().to(torch.uint8).reshape(tuple(lead) + (out_h, out_w)).numpy()


def resize_bilinear_u8_float_restated(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """numpy restatement of the fp32 route with the operation order the HIP kernel uses (source index as one fma,
    w_ij = h_i * w_j, two fma-paired sums, round half to even).  Equals ``resize_bilinear_u8_float`` (= torch) on all but
    a few bytes per million: the CPU kernel's FMA contraction is a property of the torch build."""
    f32 = np.float32
    img = np.asarray(img, dtype=np.uint8).astype(np.float32)
    h, w = img.shape[-2:]

    def fma(a, b, c):
        return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(np.float32)

    def table(n_in, n_out):
        if n_in == n_out:
            i0 = np.arange(n_out)
            return i0, i0, np.ones(n_out, np.float32), np.zeros(n_out, np.float32)
        scale = f32(n_in) / f32(n_out)
        i = np.arange(n_out, dtype=np.float32)
        real = np.maximum(fma(np.full(n_out, scale, np.float32), (i + f32(0.5)).astype(np.float32), np.full(n_out, -0.5, np.float32)), f32(0))
        i0 = np.minimum(np.floor(real).astype(np.int64), n_in - 1)
        i1 = np.minimum(i0 + 1, n_in - 1)
        l1 = np.clip((real - i0.astype(np.float32)).astype(np.float32), 0, 1).astype(np.float32)
        return i0, i1, (f32(1) - l1).astype(np.float32), l1

    y0, y1, h0, h1 = table(h, out_h)
    x0, x1, w0, w1 = table(w, out_w)
    p00, p01 = img[..., y0[:, None], x0[None, :]], img[..., y0[:, None], x1[None, :]]
    p10, p11 = img[..., y1[:, None], x0[None, :]], img[..., y1[:, None], x1[None, :]]
    w00 = (h0[:, None] * w0[None, :]).astype(np.float32); w01 = (h0[:, None] * w1[None, :]).astype(np.float32)
    w10 = (h1[:, None] * w0[None, :]).astype(np.float32); w11 = (h1[:, None] * w1[None, :]).astype(np.float32)
    b = lambda a: np.broadcast_to(a, p00.shape)
    s0 = fma(p01, b(w01), (p00 * w00).astype(np.float32))
    s1 = fma(p11, b(w11), (p10 * w10).astype(np.float32))
    return np.clip(np.rint((s0 + s1).astype(np.float32)), 0, 255).astype(np.uint8)


def crop_and_resize_video_uint8(frames_thwc: np.ndarray, box, out_size: int = 224, fixed_point: bool = False) -> np.ndarray:
    """src/dataset.py:141-149 up to (not including) the ``/255``: (T,H,W,3) uint8 + box [top,left,hh,ww] ->
    (T,3,out,out) uint8.  Default: the fp32 route of the v1 API the reference imports (computed by torch itself)."""
    top, left, hh, ww = (int(v) for v in box)
    x = np.transpose(np.asarray(frames_thwc), (0, 3, 1, 2))[:, :, top:top + hh, left:left + ww]
    return resize_bilinear_u8(x, out_size, out_size) if fixed_point else resize_bilinear_u8_float(x, out_size, out_size)
