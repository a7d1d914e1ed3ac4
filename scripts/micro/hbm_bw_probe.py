import torch, time
x = torch.empty(1 << 28, dtype=torch.float32, device='cuda')   # 1 GiB
y = torch.empty_like(x)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
w = t(lambda: x.fill_(1.0)); c = t(lambda: y.copy_(x)); r = t(lambda: x.sum())
print('fill (write) %.2f TB/s   copy (read + write) %.2f TB/s   sum (read) %.2f TB/s' % (x.numel() * 4 / w / 1e12, 2 * x.numel() * 4 / c / 1e12, x.numel() * 4 / r / 1e12))
z = torch.empty(33_554_432 // 4, dtype=torch.float32, device='cuda')   # 33.5 MB: one round of conv_wino4 output tiles
w2 = t(lambda: z.fill_(1.0), 200)
print('fill 33.5 MB: %.1f us  (%.2f TB/s)' % (w2 * 1e6, z.numel() * 4 / w2 / 1e12))
