#!/usr/bin/env python3
"""What this MI355X sustains on plain streaming kernels (torch elementwise ops on 1 GiB tensors): the practical
HBM ceiling the full-resolution conv layers and the upsample are compared with in DESIGN.md section 5."""
import time, torch
dev = torch.device("cuda:0")
n = 1 << 28                                    # 1 GiB of float32
a = torch.randn(n, device=dev); b = torch.empty_like(a)
def rate(label, fn, bytes_moved, reps=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    print(f"{label:34s} {dt*1e3:7.3f} ms  {bytes_moved/dt/1e12:5.2f} TB/s")
rate("copy (1 read + 1 write)", lambda: b.copy_(a), 2 * 4 * n)
rate("read only (sum)", lambda: a.sum(), 4 * n)
rate("write only (fill)", lambda: b.fill_(1.0), 4 * n)
rate("2 reads + 1 write (add)", lambda: torch.add(a, a, out=b) if False else torch.add(a, b, out=b), 3 * 4 * n)
