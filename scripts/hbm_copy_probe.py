"""What does a plain device copy / read / write sustain on this box?  (calibration for the HBM-bound kernels' roofline fraction)"""
import torch
d = torch.device("cuda:0")
for mb in (256, 1024):
    n = mb * 1024 * 1024 // 4
    a = torch.randn(n, device=d); b = torch.empty_like(a)
    for name, fn, byts in (("copy (read + write)", lambda: b.copy_(a), 2 * n * 4), ("fill (write)", lambda: b.fill_(1.0), n * 4),
                           ("sum (read)", lambda: a.sum(), n * 4), ("add_ (read + write same)", lambda: a.add_(1.0), 2 * n * 4)):
        for _ in range(5): fn()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 50
        print(f"{mb:5d} MB  {name:26s} {byts / ms / 1e9:7.2f} TB/s", flush=True)
