"""Achievable HBM write / copy bandwidth with torch kernels (dev tool, GPU only)."""
import torch
n = 32400 * 32416
x = torch.empty(n, device="cuda"); y = torch.empty(n, device="cuda")
def t(f, reps=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms = t(lambda: x.zero_()); print(f"zero_  {n*4/1e9:.2f} GB  {ms*1e3:.0f} us  {n*4/ms/1e9:.2f} TB/s write")
ms = t(lambda: x.fill_(1.5)); print(f"fill_  {ms*1e3:.0f} us  {n*4/ms/1e9:.2f} TB/s write")
ms = t(lambda: y.copy_(x)); print(f"copy_  {ms*1e3:.0f} us  {2*n*4/ms/1e9:.2f} TB/s read+write")
ms = t(lambda: torch.mul(x, 2.0, out=y)); print(f"mul    {ms*1e3:.0f} us  {2*n*4/ms/1e9:.2f} TB/s read+write")
