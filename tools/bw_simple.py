import torch, time
def t(fn, n=50):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); s.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): fn()
    torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for mb in (57, 228, 285, 1024):
    n = mb * 1024 * 1024 // 4
    a = torch.empty(n, device="cuda"); b = torch.empty(n, device="cuda")
    us = t(lambda: a.fill_(1.5)); print("fill  %5d MB: %7.1f us  %.2f TB/s write" % (mb, us, mb * 1.048576 / us))
    us = t(lambda: b.copy_(a)); print("copy  %5d MB: %7.1f us  %.2f TB/s (read+write %.2f)" % (mb, us, mb * 1.048576 / us, 2 * mb * 1.048576 / us))
    us = t(lambda: a.sum()); print("sum   %5d MB: %7.1f us  %.2f TB/s read" % (mb, us, mb * 1.048576 / us))
