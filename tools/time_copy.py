import torch
for n in (65536 * 39, 2097152 * 39, 16 * 2097152 * 39):
    x = torch.randn(n, device="cuda"); y = torch.empty_like(x)
    for _ in range(5): y.copy_(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200 if n < 1e8 else 20): y.copy_(x)
    e1.record(); torch.cuda.synchronize()
    k = 200 if n < 1e8 else 20
    us = e0.elapsed_time(e1) * 1e3 / k
    print(f"copy {n*4/1e6:.1f} MB: {us:.2f} us -> {2*n*4/us/1e6:.2f} TB/s (read+write)")
