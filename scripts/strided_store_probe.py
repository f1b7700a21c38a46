"""Does a 2-KB-strided stream of 256-B pieces (what bneck_tail3p_kernel reads as identity and writes as block output, chunk by chunk:
128 channels of a 1024-channel NHWC row) run slower than contiguous pieces of the same size?  torch copies, same bytes both ways.
usage: python scripts/strided_store_probe.py"""
import torch
d = torch.device("cuda:0")
M = 256 * 196                                   # pixels of a layer3 activation at batch 256
full = torch.empty((M, 8, 128), dtype=torch.bfloat16, device=d)        # NHWC rows of 1024 channels: chunk c of a pixel at c * 256 B
full2 = torch.empty_like(full)
chunked = torch.empty((8, M, 128), dtype=torch.bfloat16, device=d)     # chunk-major: chunk c contiguous over the pixels
chunked2 = torch.empty_like(chunked)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
nbytes = 2 * full.numel() * 2                   # read + write
def strided():
    for c in range(8): full2[:, c, :].copy_(full[:, c, :])
def contiguous():
    for c in range(8): chunked2[c].copy_(chunked[c])
def whole():
    full2.copy_(full)
for name, fn in (("8 chunk passes, 256-B pieces at a 2-KB stride (NHWC)", strided), ("8 chunk passes, chunk-major (contiguous)", contiguous), ("one pass over the whole tensor", whole)):
    us = timeit(fn)
    print(f"{name:58s} {us:7.1f} us  {nbytes / us / 1e6:6.2f} TB/s")
