"""When do the sporadic tens-of-ms stalls of a trivial GPU operation happen, relative to process start / first use of the
device?  Plain torch, no library code.  python tools/diag/stall_clock.py [seconds]"""
import sys, time
T0 = time.perf_counter()
import torch
dev = torch.device('cuda:0')
t_imp = time.perf_counter()
x = torch.zeros(1024, device=dev)
torch.cuda.synchronize()
t_first = time.perf_counter()
print('import torch %.2f s, first device use %.2f s' % (t_imp - T0, t_first - t_imp), flush=True)
dur = float(sys.argv[1]) if len(sys.argv) > 1 else 25.0
n = 0
while time.perf_counter() - t_first < dur:
    t = time.perf_counter()
    x.add_(1.0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    n += 1
    if dt > 2e-3:
        print('stall %.1f ms at %.3f s after first device use (op %d)' % (dt * 1e3, t - t_first, n), flush=True)
print('ops', n)
