"""Practical streaming ceilings of the box (torch kernels only; a yardstick for DESIGN.md section 4, not part of the product):
read-only reduction and copy over working sets inside and beyond the 256 MB Infinity Cache."""
import json, sys, time
import torch
out = {}
for mb in (128, 256, 1024, 4096):
    n = mb * (1 << 20) // 8
    x = torch.ones(n, dtype=torch.float64, device="cuda")
    y = torch.empty_like(x)
    for name, fn, bytes_ in (("sum", lambda: x.sum(), n * 8), ("copy", lambda: y.copy_(x), 2 * n * 8)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        out["%s_%dMB" % (name, mb)] = {"us": round(us, 1), "TBps": round(bytes_ / us / 1e6, 3)}
    del x, y
print(json.dumps(out))
