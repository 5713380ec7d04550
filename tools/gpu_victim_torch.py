"""Pure-torch memory-bound workload (sorts, scans, gathers, scatters on large tensors) with result checks:
the control for "does ANY process fault when the GPU is shared on this box"."""
import sys, time, torch
seconds = float(sys.argv[1])
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
n = 200_000_000
t_end = time.time() + seconds
rounds = 0
while time.time() < t_end:
    x = torch.randint(0, 1 << 30, (n,), device=dev, dtype=torch.int64, generator=g)
    s, idx = torch.sort(x)
    assert bool((s[1:] >= s[:-1]).all())
    assert bool((x[idx] == s).all())
    c = torch.cumsum((s & 1), 0)
    assert int(c[-1]) == int((x & 1).sum())
    y = torch.zeros(1 << 20, device=dev, dtype=torch.int64)
    y.scatter_add_(0, x & ((1 << 20) - 1), torch.ones_like(x))
    assert int(y.sum()) == n
    del x, s, idx, c, y
    rounds += 1
torch.cuda.synchronize()
print("victim rounds", rounds, "OK", flush=True)
