"""Diagnostic: plain HBM write / copy bandwidth on this box via torch (for roofline context)."""
import torch, time
n = 1 << 30
a = torch.empty(n, dtype=torch.uint8, device="cuda"); b = torch.empty(n, dtype=torch.uint8, device="cuda")
def t(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it
ms = t(lambda: a.fill_(7)); print(f"fill 1 GiB: {ms:.3f} ms -> {n/ms/1e9*1e3:.0f} GB/s write")
ms = t(lambda: b.copy_(a)); print(f"copy 1 GiB: {ms:.3f} ms -> {2*n/ms/1e9*1e3:.0f} GB/s read+write")
a32 = a.view(torch.int32)
ms = t(lambda: a32.sum()); print(f"sum  1 GiB: {ms:.3f} ms -> {n/ms/1e9*1e3:.0f} GB/s read")
