import torch, time
x = torch.empty(1 << 28, dtype=torch.int32, device="cuda"); y = torch.empty_like(x)
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
ms = t(lambda: x.zero_()); print(f"memset 1 GiB: {ms:.3f} ms  {1.0737/ms:.2f} TB/s write")
ms = t(lambda: y.copy_(x)); print(f"copy 1 GiB->1 GiB: {ms:.3f} ms  {2*1.0737/ms:.2f} TB/s (r+w)")
ms = t(lambda: x.sum()); print(f"sum 1 GiB: {ms:.3f} ms  {1.0737/ms:.2f} TB/s read")
