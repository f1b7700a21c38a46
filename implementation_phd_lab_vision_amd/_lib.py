"""ctypes binding of libr50hip.so (C ABI: include/r50.h).

This is the stub a maintainer of the reference would add to call the MI355X path from
src/preprocess_resnet_features.py.  There is NO fallback: if the shared
library is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path
from typing import Optional

PKG_DIR = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("R50_LIB", str(PKG_DIR / "libr50hip.so")))   # R50_LIB: diagnostic builds only
CSRC = PKG_DIR / "csrc"

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off"]


class R50Error(RuntimeError):
    pass


class TensorDesc(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.POINTER(C.c_float)), ("numel", C.c_int64)]


def _source_digest() -> str:
    """sha256 over everything the shared library is built from: csrc/*, include/r50.h and the compiler flags."""
    import hashlib
    hsh = hashlib.sha256()
    files = sorted(p for p in CSRC.iterdir() if p.suffix in (".hip", ".h", ".hpp", ".cpp")) + [PKG_DIR.parent / "include" / "r50.h"]
    for f in files:
        hsh.update(f.name.encode() + b"\0")
        hsh.update(f.read_bytes())
    hsh.update(" ".join(HIPCC_FLAGS).encode())
    return hsh.hexdigest()


def build_library(force: bool = False, verbose: bool = False) -> Path:
    """Compile csrc/r50_abi.hip for gfx950 into libr50hip.so (in-tree).  The library is reused only if the digest
    stored beside it (libr50hip.so.sha256) equals the digest of the sources + flags it would be built from now:
    a stale binary is rebuilt whatever its mtime says."""
    if "R50_LIB" in os.environ:
        return LIB_PATH
    stamp = LIB_PATH.with_name(LIB_PATH.name + ".sha256")
    digest = _source_digest()
    if LIB_PATH.exists() and not force and stamp.exists() and stamp.read_text().strip() == digest:
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    tmp = LIB_PATH.with_name(LIB_PATH.name + f".tmp{os.getpid()}")
    cmd = [hipcc, *HIPCC_FLAGS, "-o", str(tmp), str(CSRC / "r50_abi.hip")]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        tmp.unlink(missing_ok=True)
        raise R50Error(f"hipcc failed ({res.returncode}):\n{res.stdout}\n{res.stderr}")
    os.replace(tmp, LIB_PATH)             # atomic: concurrent ranks never dlopen a half-written file
    stamp.write_text(digest + "\n")
    return LIB_PATH


_SIGNATURES = {
    "r50_version": (C.c_char_p, []),
    "r50_last_error": (C.c_char_p, [C.c_void_p]),
    "r50_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int]),
    "r50_destroy": (None, [C.c_void_p]),
    "r50_load_weights": (C.c_int, [C.c_void_p, C.POINTER(TensorDesc), C.c_int]),
    "r50_share_weights": (C.c_int, [C.c_void_p, C.c_void_p]),
    "r50_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "r50_forward_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "r50_forward_layer": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_char_p, C.c_void_p, C.c_int64,
                                    C.POINTER(C.c_int64), C.c_void_p]),
    "r50_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "r50_get_option": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64)]),
    "r50_profile_reset": (C.c_int, [C.c_void_p]),
    "r50_profile_collect": (C.c_int, [C.c_void_p]),
    "r50_profile_count": (C.c_int, [C.c_void_p]),
    "r50_profile_entry": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int64),
                                    C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "r50_set_fp8_scales": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_int]),
    "r50_get_packed": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]),
    "r50_op_conv2d": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "r50_op_conv2d_f16": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "r50_op_conv2d_fp8": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                    C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_void_p]),
    "r50_op_conv1x1_cat": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "r50_stem_scratch_bytes": (C.c_int64, [C.c_int]),
    "r50_op_stem": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "r50_op_maxpool": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "r50_op_bneck_tail": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "r50_op_bneck_block2": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "r50_op_bneck_block1_ds": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "r50_op_bneck_cat_chain": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p]),
    "r50_op_bneck_block1": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "r50_op_crop_resize_u8": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                        C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "r50_op_color_jitter_u8": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_float, C.c_float, C.c_float, C.c_float,
                                         C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "r50_op_cast_rows": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "r50_op_concat_pad": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "r50_op_add_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p]),
    "r50_op_gn_relu_causal3": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float,
                                         C.c_void_p, C.c_int, C.c_void_p]),
    "r50_op_transpose16": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "r50_op_mask_scale": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_int64, C.c_int, C.c_void_p]),
    "r50_op_relu_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_int64, C.c_int, C.c_void_p]),
    "r50_op_colsum": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "r50_op_colsum_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_void_p]),
    "r50_op_grad_accum": (C.c_int, [C.c_void_p, C.c_float, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "r50_op_mse_loss_grad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "r50_op_gn_relu_causal3_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                             C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "r50_op_check_finite": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "r50_op_check_overflow16": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p]),
    "r50_op_adamw": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_float,
                               C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "r50_op_avgpool": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib: Optional[C.CDLL] = None


def load_library() -> C.CDLL:
    """dlopen libr50hip.so and declare every prototype of include/r50.h.  Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch ships its own copy of the HIP runtime.  Import it first so that libr50hip.so binds to the runtime that is
    # already in the process: loaded the other way round (library, then torch) the process ends up with two runtimes and
    # this library's one sees no device.
    import torch  # noqa: F401
    if not LIB_PATH.exists():
        raise R50Error(f"{LIB_PATH} not found: build it first (python -c 'import __graft_entry__ as g; g.build()'). "
                       "There is no CPU/PyTorch fallback for this path.")
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, handle: Optional[int] = None, what: str = "") -> None:
    if rc != 0:
        lib = load_library()
        msg = lib.r50_last_error(C.c_void_p(handle) if handle else None)
        raise R50Error(f"{what or 'r50 call'} failed (status {rc}): {msg.decode() if msg else '?'}")
