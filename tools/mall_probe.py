import torch, time
dev = torch.device("cuda:0")
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for mb in (16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 4096):
    n = mb * 1024 * 1024 // 4
    x = torch.empty(n, device=dev).normal_()
    y = torch.empty(n, device=dev)
    t_read = bench(lambda: torch.sum(x))            # read only
    t_copy = bench(lambda: y.copy_(x))              # read + write (2 buffers)
    t_inpl = bench(lambda: x.add_(1.0))             # read + write same buffer
    print("%5d MB  read %.1f GB/s   copy(r+w) %.1f GB/s  inplace(r+w) %.1f GB/s" % (
        mb, mb / 1024 / (t_read * 1e-3), 2 * mb / 1024 / (t_copy * 1e-3), 2 * mb / 1024 / (t_inpl * 1e-3)))
    del x, y
