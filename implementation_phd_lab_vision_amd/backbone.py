"""Host-side mirror of the reference's ``backbone`` object for the MI355X path.

The reference builds and calls it as (src/preprocess_resnet_features.py:207-209,242,296)::

    resnet   = models.resnet50(weights=models.ResNet50_Weights.IMAGENET1K_V2)
    backbone = nn.Sequential(*list(resnet.children())[:-1])
    backbone = backbone.to(device).eval()
    feats    = backbone(x).flatten(1).view(Bv, T, -1)      # x: (Bv*T, 3, 224, 224) fp32 NCHW

``ResNet50Backbone`` keeps that surface (``.to(device)``, ``.eval()``, ``__call__`` returning
``(N, 2048, 1, 1)`` on the same device) and routes the arithmetic to libr50hip.so through the C ABI
of include/r50.h.  PyTorch is used for device memory and the stream only.  No fallback: without the
shared library or without a gfx950 GPU the calls raise.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch

from . import _lib
from .weights import FEATURE_DIM, iter_named_tensors, load_state_dict_from_path, synthetic_state_dict, validate_state_dict

PREC_BF16 = 1     # bf16 operands, fp32 accumulation: the fast path (the reference's CUDA autocast dtype)
PREC_FP32X = 2    # fp32-class accuracy on the bf16 matrix cores (bf16 head/tail pairs, 3 products per conv)
PREC_BF16W2 = 3   # bf16 activations, weights as bf16 head/tail pairs (2 products per conv): within 1e-3 of the fp32 reference
PREC_FP16 = 4     # IEEE half operands / activations, same kernels and speed as bf16; 3e-4 from the fp32 reference
PREC_FP8 = 5      # BASELINE configs[4]: layer2-4 on e4m3 tensors and the K = 128 fp8 MFMA (stem + layer1 stay bf16); a throughput mode
_PRECISIONS = {"bf16": PREC_BF16, "fp32x": PREC_FP32X, "bf16w2": PREC_BF16W2, "fp16": PREC_FP16, "fp8": PREC_FP8, PREC_BF16: PREC_BF16,
               PREC_FP32X: PREC_FP32X, PREC_BF16W2: PREC_BF16W2, PREC_FP16: PREC_FP16, PREC_FP8: PREC_FP8}
FP8_NUM_SCALES = 43
FP8_MARGIN = 1.25     # scale = margin * absmax / 448: head-room for frames whose activations exceed the calibration batch's


def fp8_tap_names():
    """The fp8 part's tensors in execution order (r50_set_fp8_scales): layer1's output, then per bottleneck of layer2-4
    conv1 output, conv2 output, [downsample output], block output."""
    names = ["layer1.2"]
    for si, blocks in ((2, 4), (3, 6), (4, 3)):
        for b in range(blocks):
            p = f"layer{si}.{b}"
            names += [p + ".t1", p + ".t2"] + ([p + ".ds"] if b == 0 else []) + [p]
    return names


class ResNet50Backbone:
    """``resnet50.children()[:-1]`` in eval mode on one MI355X.

    weights: ``state_dict`` (torchvision keys) or ``weights_path`` (local ``resnet50-*.pth``);
    with neither, seeded synthetic weights (there is no network to fetch IMAGENET1K_V2).
    """

    def __init__(self, state_dict: Optional[Dict[str, torch.Tensor]] = None, weights_path: Optional[str] = None,
                 seed: int = 0, max_batch: int = 256, micro_batch: int = 0, precision="bf16"):
        if state_dict is None:
            state_dict = load_state_dict_from_path(weights_path) if weights_path else synthetic_state_dict(seed)
        validate_state_dict(state_dict)
        self._sd = state_dict
        self._max_batch = int(max_batch)
        self._micro_batch = int(micro_batch)
        if precision not in _PRECISIONS:
            raise ValueError(f"precision must be 'bf16', 'fp16', 'bf16w2', 'fp32x' or 'fp8', got {precision!r}")
        self._precision = _PRECISIONS[precision]
        self._handle: Optional[int] = None
        self._device: Optional[torch.device] = None
        self.training = False

    # ---- nn.Module-like surface -------------------------------------------------------------
    def to(self, device, share_from: Optional["ResNet50Backbone"] = None) -> "ResNet50Backbone":
        """``share_from``: another backbone of the same precision already on ``device`` -- this one then reads ITS folded / packed weight buffers
        (``r50_share_weights``) instead of loading a copy of its own (bf16 / fp16; ``BackboneLanes`` uses it for the lanes behind the first)."""
        device = torch.device(device)
        if device.type != "cuda":
            raise _lib.R50Error(f"ResNet50Backbone runs on an MI355X only (got device '{device}'); "
                                "there is no CPU fallback in the product path")
        index = device.index if device.index is not None else torch.cuda.current_device()
        device = torch.device("cuda", index)
        if self._handle is not None and self._device == device:
            return self
        self._release()
        lib = _lib.load_library()
        h = C.c_void_p()
        _lib.check(lib.r50_create(C.byref(h), index, self._precision, self._max_batch), None, "r50_create")
        self._handle = h.value
        self._device = device
        if share_from is not None:
            if share_from._handle is None or share_from._device != device or share_from._precision != self._precision:
                raise ValueError("share_from must be a backbone of the same precision on the same device")
            _lib.check(lib.r50_share_weights(self._handle, share_from._handle), self._handle, "r50_share_weights")
            if self._micro_batch:
                self.set_option("micro_batch", self._micro_batch)
            return self
        self._load_weights()
        if self._micro_batch:
            self.set_option("micro_batch", self._micro_batch)
        if self._precision == PREC_FP8:
            self.calibrate_fp8()
        return self

    def _load_weights(self) -> None:
        """``r50_load_weights`` of this backbone's state dict into its handle (refused with R50_ERR_STATE once the handle shares weight buffers)."""
        named = list(iter_named_tensors(self._sd))
        descs = (_lib.TensorDesc * len(named))()
        keep = []
        for i, (name, t) in enumerate(named):
            keep.append(t)
            descs[i].name = name.encode()
            descs[i].data = C.cast(t.data_ptr(), C.POINTER(C.c_float))
            descs[i].numel = t.numel()
        _lib.check(_lib.load_library().r50_load_weights(self._handle, descs, len(named)), self._handle, "r50_load_weights")

    # ---- fp8 mode: activation scales ----------------------------------------------------------
    def set_fp8_scales(self, scales) -> None:
        """Per-tensor activation scales of the fp8 part in ``fp8_tap_names()`` order (real value = stored e4m3 value x scale)."""
        vals = [float(v) for v in scales]
        if len(vals) != FP8_NUM_SCALES:
            raise ValueError(f"expected {FP8_NUM_SCALES} scales, got {len(vals)}")
        arr = (C.c_float * FP8_NUM_SCALES)(*vals)
        _lib.check(_lib.load_library().r50_set_fp8_scales(self._handle, arr, FP8_NUM_SCALES), self._handle, "r50_set_fp8_scales")
        self.fp8_scales = vals

    def calibrate_fp8(self, frames: Optional[torch.Tensor] = None, margin: float = FP8_MARGIN):
        """Choose the activation scales from the largest magnitude each tensor takes in the bf16 network (same weights) on a
        calibration batch: ``frames`` (N,3,224,224) fp32 on the device, or 8 seeded synthetic frames.  Returns the scales."""
        from .weights import synthetic_frames
        if frames is None:
            frames = synthetic_frames(8, seed=4321).to(self._device)
        ref = ResNet50Backbone(state_dict=self._sd, max_batch=int(frames.shape[0]), precision="bf16").to(self._device).eval()
        try:
            scales = []
            for name in fp8_tap_names():
                amax = float(ref.layer(frames, name).float().abs().max())
                scales.append(margin * max(amax, 1e-6) / 448.0)
        finally:
            ref.close()
        self.set_fp8_scales(scales)
        return scales

    def cuda(self, device=None) -> "ResNet50Backbone":
        return self.to(torch.device("cuda", device if device is not None else torch.cuda.current_device()))

    def eval(self) -> "ResNet50Backbone":
        self.training = False      # BN always uses running statistics (folded at load time)
        return self

    def train(self, mode: bool = True):
        if mode:
            raise _lib.R50Error("ResNet50Backbone is inference-only (the reference calls .eval(), :209)")
        return self

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        return self.features(x).view(x.shape[0], FEATURE_DIM, 1, 1)

    forward = __call__

    # ---- feature path -----------------------------------------------------------------------
    def _check_input(self, x: torch.Tensor) -> torch.Tensor:
        if self._handle is None:
            raise _lib.R50Error("call .to('cuda:N') before running the backbone")
        if x.dim() != 4 or tuple(x.shape[1:]) != (3, 224, 224):
            raise ValueError(f"expected (N,3,224,224) frames, got {tuple(x.shape)}")
        if x.device != self._device:
            raise ValueError(f"frames are on {x.device}, backbone on {self._device}")
        if x.dtype != torch.float32:
            x = x.to(torch.float32)
        return x.contiguous()

    def features(self, x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """(N,3,224,224) fp32 NCHW on the backbone's device -> (N,2048) fp32 (= backbone(x).flatten(1))."""
        x = self._check_input(x)
        n = x.shape[0]
        if out is None:
            out = torch.empty((n, FEATURE_DIM), dtype=torch.float32, device=self._device)
        elif out.shape != (n, FEATURE_DIM) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != self._device:
            raise ValueError("out must be a contiguous fp32 (N,2048) tensor on the backbone's device")
        if n == 0:
            return out
        lib = _lib.load_library()
        stream = torch.cuda.current_stream(self._device).cuda_stream
        _lib.check(lib.r50_forward(self._handle, x.data_ptr(), n, out.data_ptr(), stream), self._handle, "r50_forward")
        return out

    def features_u8(self, x_u8: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """(N,3,224,224) uint8 NCHW resized crops (what the reference holds before ``/255`` and ``Normalize``,
        src/dataset.py:141-150,242-245) -> (N,2048) fp32.  Normalisation happens inside the stem kernel with the
        reference's fp32 operations, so the result equals ``features(((x/255) - mean) / std)`` bit for bit."""
        if self._handle is None:
            raise _lib.R50Error("call .to('cuda:N') before running the backbone")
        if x_u8.dim() != 4 or tuple(x_u8.shape[1:]) != (3, 224, 224) or x_u8.dtype != torch.uint8:
            raise ValueError(f"expected uint8 (N,3,224,224) frames, got {x_u8.dtype} {tuple(x_u8.shape)}")
        if x_u8.device != self._device:
            raise ValueError(f"frames are on {x_u8.device}, backbone on {self._device}")
        x_u8 = x_u8.contiguous()
        n = x_u8.shape[0]
        if out is None:
            out = torch.empty((n, FEATURE_DIM), dtype=torch.float32, device=self._device)
        if n == 0:
            return out
        lib = _lib.load_library()
        stream = torch.cuda.current_stream(self._device).cuda_stream
        _lib.check(lib.r50_forward_u8(self._handle, x_u8.data_ptr(), n, out.data_ptr(), stream), self._handle, "r50_forward_u8")
        return out

    def features_from_video(self, frames_thwc_u8: torch.Tensor, box, mode: int = 0) -> torch.Tensor:
        """Decoded clip in, features out: ``(T,H,W,3)`` uint8 frames on the device + crop box ``[top,left,hh,ww]``
        (``frames.square_crop_from_2d``) -> crop + bilinear resize on the device (``frames.crop_and_resize_video_uint8``,
        src/dataset.py:141-149) -> ``/255``, ``Normalize`` and the backbone (``features_u8``) -> (T,2048) fp32."""
        from .frames import crop_and_resize_video_uint8
        return self.features_u8(crop_and_resize_video_uint8(frames_thwc_u8, box, 224, mode))

    def layer(self, x: torch.Tensor, name: str) -> torch.Tensor:
        """Debug hook: named intermediate activation, NHWC (per-layer parity tests).  bf16 tensor in the bf16
        modes, fp16 tensor in fp16 mode; in fp32x mode the (head, tail) pair is recombined into an fp32 tensor."""
        x = self._check_input(x)
        n = x.shape[0]
        lib = _lib.load_library()
        cap = n * 112 * 112 * 64 * (2 if self._precision == PREC_FP32X else 1)
        buf = torch.empty(cap, dtype=torch.bfloat16, device=self._device)
        dims = (C.c_int64 * 4)()
        stream = torch.cuda.current_stream(self._device).cuda_stream
        _lib.check(lib.r50_forward_layer(self._handle, x.data_ptr(), n, name.encode(), buf.data_ptr(), cap * 2, dims, stream),
                   self._handle, "r50_forward_layer")
        d = [int(v) for v in dims]
        t = buf[: d[0] * d[1] * d[2] * d[3]].view(*d)
        if self._precision == PREC_FP32X:
            c = d[3] // 2
            return t[..., :c].float() + t[..., c:].float()
        if self._precision == PREC_FP16:          # same bytes, the other 16-bit format
            return t.view(torch.float16)
        return t

    def packed_params(self, conv_key: str):
        """Debug hook: (folded bf16 weights in (cout,k,k,cin) order, folded fp32 bias) as the device holds them."""
        from .weights import conv_specs
        spec = {c[0]: c for c in conv_specs()}[conv_key]
        _ck, _bk, cin, cout, k, _s, _p = spec
        lib = _lib.load_library()
        n = C.c_int64()
        wbytes = 7 * 64 * 64 if conv_key == "conv1" else cout * k * k * cin * 2
        w = torch.empty(wbytes // 2, dtype=torch.bfloat16)
        _lib.check(lib.r50_get_packed(self._handle, conv_key.encode(), 0, w.data_ptr(), wbytes, C.byref(n)),
                   self._handle, "r50_get_packed")
        b = torch.empty(cout, dtype=torch.float32)
        _lib.check(lib.r50_get_packed(self._handle, conv_key.encode(), 1, b.data_ptr(), cout * 4, C.byref(n)),
                   self._handle, "r50_get_packed")
        if conv_key != "conv1":
            w = w.view(cout, k, k, cin)
        return w, b

    # ---- options / profiling ----------------------------------------------------------------
    def set_option(self, key: str, value: int) -> None:
        _lib.check(_lib.load_library().r50_set_option(self._handle, key.encode(), int(value)), self._handle, "r50_set_option")

    def get_option(self, key: str) -> int:
        v = C.c_int64()
        _lib.check(_lib.load_library().r50_get_option(self._handle, key.encode(), C.byref(v)), self._handle, "r50_get_option")
        return int(v.value)

    def profile_reset(self) -> None:
        _lib.check(_lib.load_library().r50_profile_reset(self._handle), self._handle, "r50_profile_reset")

    def profile_collect(self) -> Dict[str, Dict[str, float]]:
        lib = _lib.load_library()
        _lib.check(lib.r50_profile_collect(self._handle), self._handle, "r50_profile_collect")
        out = {}
        for i in range(lib.r50_profile_count(self._handle)):
            name = C.c_char_p(); launches = C.c_int64(); ms = C.c_double(); fl = C.c_double(); by = C.c_double()
            _lib.check(lib.r50_profile_entry(self._handle, i, C.byref(name), C.byref(launches), C.byref(ms),
                                             C.byref(fl), C.byref(by)), self._handle, "r50_profile_entry")
            out[name.value.decode()] = {"launches": int(launches.value), "ms": float(ms.value),
                                        "flops": float(fl.value), "bytes": float(by.value)}
        return out

    # ---- lifetime ---------------------------------------------------------------------------
    def _release(self) -> None:
        if self._handle is not None:
            _lib.load_library().r50_destroy(self._handle)
            self._handle = None

    def close(self) -> None:
        self._release()

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass


class LaneTicket:
    """One batch in flight on a lane: ``out`` is complete once ``event`` has fired (``wait()`` orders the current stream behind it)."""
    __slots__ = ("out", "event", "lane")

    def __init__(self, out: torch.Tensor, event: "torch.cuda.Event", lane: int):
        self.out, self.event, self.lane = out, event, lane

    def wait(self, stream: Optional["torch.cuda.Stream"] = None) -> torch.Tensor:
        (stream or torch.cuda.current_stream(self.out.device)).wait_event(self.event)
        return self.out


class BackboneLanes:
    """Several independent batches in flight on one MI355X: ``lanes`` copies of the backbone, each on its own HIP stream.

    Why: a forward pass is 35 launches of persistent kernels, one workgroup per CU; every launch has a head (cold LDS, the
    first tiles' operands) and a tail (the last workgroups finishing alone), and the next launch of the SAME batch cannot start
    before the tail has drained.  Batches are independent (the reference's call site runs them one after the other,
    src/preprocess_resnet_features.py:287-297), so with two batches in flight the workgroups of one batch's next launch fill the
    CUs the other batch's tail leaves idle: +5 % frames/s at batch 256 with the same bits (scripts/dual_stream_probe.py).
    Each lane owns its activation buffers; the folded / packed weights exist once (``r50_share_weights``; the fp8 and the multi-term modes keep
    a copy per lane).

    ``submit`` enqueues a batch on the next lane and returns at once; the caller orders later work behind ``ticket.event``.
    ``features`` / ``__call__`` keep the one-batch surface of ``ResNet50Backbone`` (current stream waits for the result)."""

    def __init__(self, lanes: int = 2, **backbone_kwargs):
        if lanes < 1:
            raise ValueError("lanes must be >= 1")
        sd = backbone_kwargs.pop("state_dict", None)
        if sd is None:
            wp = backbone_kwargs.pop("weights_path", None)
            sd = load_state_dict_from_path(wp) if wp else synthetic_state_dict(backbone_kwargs.pop("seed", 0))
        self._bbs = [ResNet50Backbone(state_dict=sd, **backbone_kwargs) for _ in range(lanes)]
        self._streams = []
        self._next = 0
        self._active = lanes              # lanes `submit` deals batches over (tune() may fall back to 1)
        self.tune_log = []
        self.tune_mode = f"{lanes} lanes (not tuned)" if lanes > 1 else "one lane"
        self._device: Optional[torch.device] = None
        self.training = False

    # ---- nn.Module-like surface -------------------------------------------------------------
    def to(self, device) -> "BackboneLanes":
        self._bbs[0].to(device)
        import os
        share = self._bbs[0] if self._bbs[0]._precision in (PREC_BF16, PREC_FP16) else None      # one copy of the weights for all lanes
        if os.environ.get("R50_LANES_SHARE") == "0":                                              # A/B knob: a copy per lane
            share = None
        for bb in self._bbs[1:]:
            bb.to(device, share_from=share)
        self._device = self._bbs[0]._device
        if len(self._streams) != len(self._bbs):
            self._streams = self._pick_streams(len(self._bbs))
        if self._bbs[0]._precision == PREC_FP8:      # one calibration for all lanes: the features must not depend on the lane
            for bb in self._bbs[1:]:
                bb.set_fp8_scales(self._bbs[0].fp8_scales)
        return self

    # ---- streams that really run side by side ------------------------------------------------
    @staticmethod
    def _overlap(a: "torch.cuda.Stream", b: "torch.cuda.Stream", dev: torch.device, cycles: int = 1_500_000) -> bool:
        """Do kernels on ``a`` and ``b`` execute concurrently?  Two spin kernels, one per stream, against the same two on one stream."""
        import time
        torch.cuda._sleep(1000)                                  # (first use compiles / loads the spin kernel)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        with torch.cuda.stream(a):
            torch.cuda._sleep(cycles)
            torch.cuda._sleep(cycles)
        torch.cuda.synchronize(dev)
        serial = time.perf_counter() - t0
        t0 = time.perf_counter()
        with torch.cuda.stream(a):
            torch.cuda._sleep(cycles)
        with torch.cuda.stream(b):
            torch.cuda._sleep(cycles)
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) < 0.75 * serial

    def _pick_streams(self, n: int):
        """One HIP stream per lane, checked to overlap: the runtime multiplexes HIP streams onto a few hardware queues in creation order, and two
        streams that land on the SAME queue run their kernels strictly one after the other -- measured: one stream pair in eight of a process
        (profiles/r03_lanes_ab.txt), the whole two-lane gain gone.  A candidate that does not overlap with the lanes chosen so far is replaced by
        the next stream of PyTorch's pool (it never has to be: correctness does not depend on it, only the gain does)."""
        dev = self._device
        streams = [torch.cuda.Stream(dev)]
        self.stream_retries = 0
        for _ in range(1, n):
            cand = torch.cuda.Stream(dev)
            for _attempt in range(24):
                if all(self._overlap(s, cand, dev) for s in streams):
                    break
                self.stream_retries += 1
                cand = torch.cuda.Stream(dev)
            streams.append(cand)
        return streams

    @staticmethod
    def lane_plan(ratios, n_lanes: int, min_gain: float = 1.02, drop_below: float = 1.0):
        """What ``tune`` does with the ratios it measured (all lanes / one lane, one entry per try): ``(lanes to use, mode label)``.
        The last ratio decides: at or above ``min_gain`` the lanes run side by side; below ``drop_below`` they make the step SLOWER
        (big launches that leave no tail to fill: fp8 at batch 512 measured 0.97) and ``submit`` falls back to one lane; in between all
        lanes stay (no loss, the streams just did not overlap much)."""
        if n_lanes < 2:
            return 1, "one lane"
        if not ratios:
            return n_lanes, f"{n_lanes} lanes (not tuned)"
        r = ratios[-1]
        if r < drop_below:
            return 1, f"one lane (fallback: {n_lanes} lanes measured {r:.3f} x one lane)"
        return n_lanes, f"{n_lanes} lanes ({r:.3f} x one lane{'' if r >= min_gain else ', below the tuning target'})"

    @property
    def active_lanes(self) -> int:
        """Lanes ``submit`` deals batches over: all of them, or 1 after ``tune`` found that they do not pay for this workload."""
        return self._active

    def tune(self, x: torch.Tensor, steps: int = 8, tries: int = 4, min_gain: float = 1.02, drop_below: float = 1.0,
             min_frames: int = 32) -> float:
        """Check with the REAL workload that the lanes run side by side, and change streams if they do not.  The spin-kernel check of
        ``_pick_streams`` is necessary but was seen not to be sufficient (one bench.py process in ~10 still showed two lanes = one lane:
        how the firmware maps the runtime's queues onto the hardware is not ours to see).  ``steps`` forwards of ``x`` on lane 0 alone against
        ``steps`` forwards dealt over the lanes; below ``min_gain`` the lanes behind the first get new streams (again checked with the spin
        kernels) and the measurement is repeated, ``tries`` times at most.  If the last ratio is below ``drop_below`` the lanes cost more
        than they give for this workload and ``submit`` uses lane 0 only from here on (``active_lanes``, ``tune_mode``).  A batch of fewer than
        ``min_frames`` frames is too short to measure (the ratio would be launch noise): nothing is measured or changed.  Returns the last
        ratio measured (all lanes / one); the ratios are informational (``tune_log``) -- results never depend on them.  Costs ~2 x steps
        forwards per try."""
        import time
        if len(self._bbs) < 2:
            return 1.0
        if x.shape[0] < min_frames:
            self.tune_log = []
            self.tune_mode = f"{len(self._bbs)} lanes (not tuned: {x.shape[0]} frames are too few to measure)"
            return 1.0
        self._active = len(self._bbs)
        dev = self._device
        outs = [torch.empty((x.shape[0], FEATURE_DIM), dtype=torch.float32, device=dev) for _ in self._bbs]

        def run(lanes: bool) -> float:
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for k in range(steps):
                if lanes:
                    self.submit(x, out=outs[k % len(outs)])
                else:
                    self._bbs[0].features(x, outs[0])
            torch.cuda.synchronize(dev)
            return time.perf_counter() - t0

        run(True); run(False)                      # warm both paths
        ratio = 1.0
        self.tune_log = []
        for attempt in range(tries):
            t1 = min(run(False), run(False))
            t2 = min(run(True), run(True))
            ratio = t1 / t2
            self.tune_log.append(round(ratio, 4))
            if ratio >= min_gain:
                break
            if attempt + 1 < tries:
                keep = self._streams[0]
                new = [keep]
                for _ in self._streams[1:]:
                    cand = torch.cuda.Stream(dev)
                    for _a in range(24):
                        if all(self._overlap(s, cand, dev) for s in new):
                            break
                        cand = torch.cuda.Stream(dev)
                    new.append(cand)
                self._streams = new
        self._next = 0
        self._active, self.tune_mode = self.lane_plan(self.tune_log, len(self._bbs), min_gain, drop_below)
        return ratio

    def cuda(self, device=None) -> "BackboneLanes":
        return self.to(torch.device("cuda", device if device is not None else torch.cuda.current_device()))

    def eval(self) -> "BackboneLanes":
        return self

    def train(self, mode: bool = True):
        if mode:
            raise _lib.R50Error("the backbone is inference-only (the reference calls .eval(), :209)")
        return self

    @property
    def lanes(self) -> int:
        return len(self._bbs)

    @property
    def lane0(self) -> ResNet50Backbone:
        return self._bbs[0]

    @property
    def fp8_scales(self):
        return self._bbs[0].fp8_scales

    def set_option(self, key: str, value: int) -> None:
        for bb in self._bbs:
            bb.set_option(key, value)

    def get_option(self, key: str) -> int:
        return self._bbs[0].get_option(key)

    def set_fp8_scales(self, scales) -> None:
        for bb in self._bbs:
            bb.set_fp8_scales(scales)

    def calibrate_fp8(self, frames: Optional[torch.Tensor] = None, margin: float = FP8_MARGIN):
        scales = self._bbs[0].calibrate_fp8(frames=frames, margin=margin)
        for bb in self._bbs[1:]:
            bb.set_fp8_scales(scales)
        return scales

    # ---- batches in flight ------------------------------------------------------------------
    def submit(self, x: torch.Tensor, out: Optional[torch.Tensor] = None, after: Optional["torch.cuda.Event"] = None,
               u8: bool = False) -> LaneTicket:
        """Enqueue one batch on the next lane (round robin) and return without waiting.

        ``after``: event the lane waits for before it touches ``x`` / ``out`` (their producer / previous consumer); None = the
        caller guarantees both are ready (static inputs).  The lane's previous batch is ordered before this one by its stream.
        The caller keeps ``x`` and ``out`` alive until ``ticket.event`` has fired (``ticket.wait()``), and allocates ``out`` itself
        when it is given: a tensor made here is allocated on the CURRENT stream and first written on the lane's, so when ``out`` is
        None the lane additionally waits for an event recorded on the current stream right behind the allocation (whatever ``after`` says)."""
        if self._device is None:
            raise _lib.R50Error("call .to('cuda:N') before running the backbone")
        lane = self._next % self._active
        self._next = (lane + 1) % self._active
        bb, st = self._bbs[lane], self._streams[lane]
        if after is not None:
            st.wait_event(after)
        if out is None:
            # allocated on the CURRENT stream: the caching allocator may hand out a block whose last use on that stream was queued AFTER the
            # caller's `after` event was recorded, so the lane also waits for everything queued on the current stream up to this point
            out = torch.empty((x.shape[0], FEATURE_DIM), dtype=torch.float32, device=self._device)
            fresh = torch.cuda.Event()
            fresh.record(torch.cuda.current_stream(self._device))
            st.wait_event(fresh)
        # allocator bookkeeping: both tensors are used on the lane's stream, whichever stream they were allocated on (e.g. a prefetcher's copy
        # stream) -- their blocks must not be handed out again before the lane's kernels are done
        x.record_stream(st)
        out.record_stream(st)
        with torch.cuda.stream(st):
            (bb.features_u8 if u8 else bb.features)(x, out)
            done = torch.cuda.Event()
            done.record(st)
        return LaneTicket(out, done, lane)

    def features(self, x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One batch, ordered like ``ResNet50Backbone.features``: runs behind everything queued on the current stream, and the
        current stream waits for the result."""
        cur = torch.cuda.current_stream(self._device)
        ready = torch.cuda.Event()
        ready.record(cur)
        return self.submit(x, out, after=ready).wait(cur)

    def features_u8(self, x_u8: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        cur = torch.cuda.current_stream(self._device)
        ready = torch.cuda.Event()
        ready.record(cur)
        return self.submit(x_u8, out, after=ready, u8=True).wait(cur)

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        return self.features(x).view(x.shape[0], FEATURE_DIM, 1, 1)

    forward = __call__

    def layer(self, x: torch.Tensor, name: str) -> torch.Tensor:
        return self._bbs[0].layer(x, name)

    def drain(self) -> None:
        """The current stream waits for everything submitted so far."""
        cur = torch.cuda.current_stream(self._device)
        for st in self._streams:
            cur.wait_stream(st)

    def close(self) -> None:
        for bb in self._bbs:
            bb.close()
