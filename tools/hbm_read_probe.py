import torch, time
x = torch.empty(307_200_000 // 4, dtype=torch.float32, device="cuda").normal_()
for n in (1, 4):
    xs = [x] if n == 1 else [torch.empty_like(x).normal_() for _ in range(n)]
    for f, name in ((lambda t: t.sum(), "sum"), (lambda t: t.max(), "max")):
        for _ in range(3): f(xs[0])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        reps = 20
        for i in range(reps): f(xs[i % n])
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        print(name, "buffers", n, "%.1f us  %.2f TB/s" % (us, x.numel() * 4 / us / 1e6))
