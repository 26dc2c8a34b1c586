import torch, time
n = 800_000_000
a = torch.empty(n, dtype=torch.float32, device="cuda")
b = torch.empty(n, dtype=torch.float32, device="cuda")
def t(f, k=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k * 1e-3
s = t(lambda: a.fill_(1.0)); print(f"fill 3.2 GB: {s*1e6:.0f} us  {3.2e9/s/1e12:.2f} TB/s written")
s = t(lambda: b.copy_(a)); print(f"copy 3.2 GB: {s*1e6:.0f} us  {6.4e9/s/1e12:.2f} TB/s read+written")
s = t(lambda: a.sum()); print(f"sum 3.2 GB: {s*1e6:.0f} us  {3.2e9/s/1e12:.2f} TB/s read")
