"""Does a GPU operation issued after the device sat idle sometimes take tens of ms?  Independent of this library: plain torch
copies / a trivial kernel, after host-side idle gaps of several lengths (sleep vs busy host).  python tools/diag/idle_stall.py"""
import sys, time
import torch
dev = torch.device('cuda:0')
x = torch.zeros(1002, 1002, device=dev)
h = torch.zeros(500, 2)
pin = torch.empty(1002, 1002, pin_memory=True)
torch.cuda.synchronize()


def busy(ms):
    t = time.perf_counter()
    while (time.perf_counter() - t) * 1e3 < ms:
        pass


ops = {'h2d small pageable': lambda: h.to(dev),
       'kernel + sync': lambda: (x.add_(1.0), torch.cuda.synchronize()),
       'd2h 4 MB pageable': lambda: x.cpu(),
       'd2h 4 MB pinned': lambda: (pin.copy_(x, non_blocking=True), torch.cuda.synchronize())}
for idle_kind, idle in (('sleep', time.sleep), ('busy host', lambda s: busy(s * 1e3))):
    for gap_ms in (0, 20, 100, 300):
        for name, op in ops.items():
            ts = []
            n = 60 if gap_ms >= 100 else 150
            for _ in range(n):
                if gap_ms:
                    idle(gap_ms / 1e3)
                t = time.perf_counter()
                op()
                ts.append((time.perf_counter() - t) * 1e3)
            ts.sort()
            print('%-9s gap %3d ms  %-20s n %3d  median %.3f  p90 %.3f  max %.3f  >5ms: %d' % (
                idle_kind, gap_ms, name, n, ts[n // 2], ts[int(n * .9)], ts[-1], sum(t > 5 for t in ts)), flush=True)
