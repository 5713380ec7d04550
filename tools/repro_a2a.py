import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from outerspace_amd.distributed import all_to_all_v
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29535", RANK="0", WORLD_SIZE="1")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
for n in (100_000_000, 268_435_456, 300_000_000, 428_896_047, 600_000_000):
    for dt in (torch.int32, torch.float64):
        src = torch.arange(n, device=dev, dtype=torch.int64).to(dt)
        dst = torch.empty(n, device=dev, dtype=dt)
        dist.all_to_all_single(dst, src, [n], [n])
        torch.cuda.synchronize()
        neq = (dst != src).nonzero()
        print(n, dt, "bytes", n * src.element_size(), "mismatch count", neq.numel(), "first", int(neq[0]) if neq.numel() else -1, flush=True)
        dst.zero_()
        all_to_all_v(dst, src, [n], [n], dist, 1)
        torch.cuda.synchronize()
        print("   chunked all_to_all_v: equal", bool(torch.equal(dst, src)), flush=True)
        del src, dst, neq
dist.destroy_process_group()
