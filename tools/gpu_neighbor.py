"""A second process on the same GPU, for reproducing faults that only show when the device is shared.
mode "compute": matmuls on preallocated buffers (no allocation after start-up);
mode "alloc": hipMalloc / hipFree churn through torch (empty_cache after every round), hardly any compute."""
import sys, time, torch
mode, seconds = sys.argv[1], float(sys.argv[2])
dev = torch.device("cuda", 0)
t_end = time.time() + seconds
if mode == "compute":
    a = torch.randn(8192, 8192, device=dev)
    b = torch.randn(8192, 8192, device=dev)
    n = 0
    while time.time() < t_end:
        c = a @ b
        a = c * (1.0 / 8192.0)
        n += 1
        if n % 20 == 0:
            torch.cuda.synchronize()
else:
    n = 0
    while time.time() < t_end:
        xs = [torch.empty((64 + 37 * (i % 7)) << 20, dtype=torch.uint8, device=dev) for i in range(24)]
        xs[0].fill_(1)
        torch.cuda.synchronize()
        del xs
        torch.cuda.empty_cache()
        n += 1
print("neighbor", mode, "rounds", n, flush=True)
