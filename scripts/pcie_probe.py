#!/usr/bin/env python3
"""H2D / D2H rates of pinned buffers of the frame-loop sizes (dev aid for bench.py's end_to_end leg)."""
import time, torch
dev = torch.device("cuda:0")
for mb in (4, 12.6, 64, 256):
    n = int(mb * 1e6)
    h = torch.empty(n, dtype=torch.uint8).pin_memory()
    d = torch.empty(n, dtype=torch.uint8, device=dev)
    for name, fn in (("H2D", lambda: d.copy_(h, non_blocking=True)), ("D2H", lambda: h.copy_(d, non_blocking=True))):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print(f"{name} {mb:6.1f} MB pinned={h.is_pinned()}: {dt*1e3:7.3f} ms  {n/dt/1e9:6.2f} GB/s")
