"""What plain streaming kernels reach on this GPU for the traffic mixes of the SGM launches."""
import torch
dev = torch.device("cuda:0")
def bench(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
n = 1080 * 1920 * 256 * 2            # two 1080p volumes' worth of floats (4.25 GB)
a = torch.empty(n, device=dev).normal_(); b = torch.empty(n, device=dev).normal_(); c = torch.empty(n, device=dev)
gb = n * 4 / 1e9
t = bench(lambda: torch.add(a, b, out=c)); print("read 2 + write 1 (pair kernel mix): %.2f TB/s" % (3 * gb / t))
t = bench(lambda: torch.neg(a, out=c));    print("read 1 + write 1 (first pair mix):  %.2f TB/s" % (2 * gb / t))
t = bench(lambda: a.add_(1.0));            print("in-place read+write:               %.2f TB/s" % (2 * gb / t))
t = bench(lambda: torch.maximum(a, b));    print("read 2 + write 1 (fresh output):   %.2f TB/s" % (3 * gb / t))
t = bench(lambda: c.fill_(1.0));           print("write only:                        %.2f TB/s" % (gb / t))
t = bench(lambda: torch.dot(a, b));        print("read 2 (dot):                      %.2f TB/s" % (2 * gb / t))
