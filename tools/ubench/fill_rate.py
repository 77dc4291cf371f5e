import torch
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps / 1e3
x = torch.empty(1 << 28, device="cuda"); y = torch.randn(1 << 28, device="cuda")
for name, fn, bytes_ in (("fill 1 GiB", lambda: x.fill_(1.0), x.numel()*4), ("zero 1 GiB", lambda: x.zero_(), x.numel()*4),
                         ("copy 1 GiB", lambda: x.copy_(y), 2*x.numel()*4), ("read-reduce 1 GiB", lambda: y.sum(), x.numel()*4),
                         ("mul in place", lambda: y.mul_(1.0001), 2*x.numel()*4)):
    t = timeit(fn); print("%-20s %.3f ms  %.0f GB/s" % (name, t*1e3, bytes_/t/1e9))
