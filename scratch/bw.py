import torch, time
x = torch.empty(765_000_000, dtype=torch.float32, device="cuda").normal_()
y = torch.empty(765_000_000, dtype=torch.float32, device="cuda").normal_()
for fn, name, nb in ((lambda: x.sum(), "sum (read 3.06 GB)", 3.06e9), (lambda: x.max(), "max (read)", 3.06e9), (lambda: y.copy_(x), "copy (read+write 6.1 GB)", 6.12e9), (lambda: y.zero_(), "memset (write 3.06 GB)", 3.06e9)):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print("%-28s %.3f ms  %.2f TB/s" % (name, dt * 1e3, nb / dt / 1e12))
