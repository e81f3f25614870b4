import torch, time
n = 1 << 30
a = torch.randint(0, 128, (n,), dtype=torch.uint8, device="cuda")
b = torch.empty_like(a)
c = torch.empty((n * 7) // 8, dtype=torch.uint8, device="cuda")
for name, dst, src in (("copy 1GiB->1GiB", b, a), ("copy 0.875->0.875", c, a[: c.numel()])):
    for _ in range(3):
        dst.copy_(src)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("%s: %.3f ms  %.2f TB/s (read+write)" % (name, ms, 2 * dst.numel() / ms / 1e9))
# read-only: sum
for _ in range(3):
    s = a.view(torch.int64).sum()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    s = a.view(torch.int64).sum()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("read-only sum: %.3f ms  %.2f TB/s" % (ms, n / ms / 1e9))
