#!/usr/bin/env python3
"""How fast can this GPU write a batch of frames at all?  (hipMemsetAsync / torch fill of 6.4 GB: the floor of the
visibility pass's black pixels)"""
import torch, time
n = 1024 * 1920 * 1080 * 3
x = torch.empty(n, dtype=torch.uint8, device="cuda")
for fn, name in ((lambda: x.zero_(), "zero_ (memset)"), (lambda: x.view(torch.int32).fill_(0), "fill_ int32")):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"{name:16s} {dt * 1e3:7.3f} ms  {n / dt / 1e12:5.2f} TB/s")
