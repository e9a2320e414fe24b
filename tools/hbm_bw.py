import torch, time
n = 1 << 30
a = torch.empty(n, dtype=torch.uint8, device="cuda"); b = torch.empty(n, dtype=torch.uint8, device="cuda")
def t(f, reps=20):
    f(); torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
ms = t(lambda: a.fill_(7)); print("fill 1 GiB: %.3f ms = %.2f TB/s written" % (ms, n / ms / 1e9))
ms = t(lambda: b.copy_(a)); print("copy 1 GiB: %.3f ms = %.2f TB/s read+written" % (ms, 2 * n / ms / 1e9))
af = a.view(torch.float32)
ms = t(lambda: af.sum()); print("sum-reduce 1 GiB: %.3f ms = %.2f TB/s read" % (ms, n / ms / 1e9))
for mb in (32, 128):
    m = mb << 20
    x = torch.empty(m, dtype=torch.uint8, device="cuda")
    ms = t(lambda: x.fill_(3), 50); print("fill %d MiB: %.4f ms = %.2f TB/s" % (mb, ms, m / ms / 1e9))
