"""k-sharded multi-GPU SpGEMM: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL).

No reference counterpart -- the reference is one process, one thread (SURVEY.md sections 2, 5).
Shape of the computation (BASELINE.json north_star):

1. The shared dimension k is cut into ``world`` contiguous slabs of (nearly) equal partial-product
   count.  Rank g runs the single-GPU pipeline on its slab -> a partial CSR ``C_g`` over ALL rows.
2. One exchange step.  Output rows are cut into ``world`` contiguous ranges balanced by
   ``sum_g nnz(C_g[row])`` (one small all-reduce of per-row counts).  Because a CSR is row-major, the
   rows a rank owes to rank h are ONE contiguous slice of its colidx / vals arrays, so the exchange
   is a plain all-to-all-v with no packing: every GPU talks to all 7 peers at once over xGMI.
3. Rank h sums the ``world`` CSR pieces of its row range with ``osp_merge_csr_parts`` (the same
   LDS merge kernels).  The result stays row-sharded.

The communication-independent parts take the local product / local merge as callables, so the
world_size-2 ``gloo`` tests on CPU exercise exactly this code with CPU stand-ins.
"""
import time

import numpy as np
import torch


def plan_k_shards(a_colptr, b_rowptr, world):
    """Cut [0,K) into `world` slabs with ~equal partial products.  Returns a python list of world+1 bounds."""
    w = (a_colptr[1:] - a_colptr[:-1]) * (b_rowptr[1:] - b_rowptr[:-1])
    cum = torch.cumsum(w, 0)
    K = w.numel()
    total = int(cum[-1]) if K else 0
    bounds = [0]
    for g in range(1, world):
        target = total * g // world
        k = int(torch.searchsorted(cum, torch.tensor([target], device=cum.device, dtype=cum.dtype))[0]) + 1 if total else 0
        bounds.append(min(max(k, bounds[-1]), K))
    bounds.append(K)
    return bounds


def plan_row_ranges(row_weight, world):
    """Cut rows into `world` contiguous ranges of ~equal weight.  row_weight: 1-D integer tensor."""
    M = row_weight.numel()
    cum = torch.cumsum(row_weight.to(torch.int64), 0)
    total = int(cum[-1]) if M else 0
    bounds = [0]
    for h in range(1, world):
        target = total * h // world
        r = int(torch.searchsorted(cum, torch.tensor([target], device=cum.device, dtype=cum.dtype))[0]) + 1 if total else 0
        bounds.append(min(max(r, bounds[-1]), M))
    bounds.append(M)
    return bounds


# RCCL (ROCm 7.0.2 build shipped with torch 2.10) delivers only the first half of an all_to_all_single message
# larger than 2^30 bytes (tools/repro_a2a.py: 1.07 GB intact, 1.2 GB and up truncated, also with one rank).
# Large exchanges therefore go in rounds of at most this many bytes per (source, destination) pair.
A2A_MAX_BYTES = 1 << 29


def all_to_all_v(dst, src, recv_l, send_l, dist, world, group=None, max_bytes=None):
    """dst/src: 1-D tensors; rank h receives src's slice h of every rank (split sizes in elements)."""
    max_bytes = max_bytes or A2A_MAX_BYTES
    ch = max(1, max_bytes // src.element_size())
    biggest = torch.tensor([max(list(send_l) + list(recv_l) + [0])], dtype=torch.int64, device=src.device)
    dist.all_reduce(biggest, op=dist.ReduceOp.MAX, group=group)
    rounds = (int(biggest[0]) + ch - 1) // ch
    if rounds <= 1:
        dist.all_to_all_single(dst, src, list(recv_l), list(send_l), group=group)
        return
    soff = [0] * (world + 1)
    roff = [0] * (world + 1)
    for h in range(world):
        soff[h + 1] = soff[h] + send_l[h]
        roff[h + 1] = roff[h] + recv_l[h]
    lists_ok = dist.get_backend(group) == "nccl"  # gloo has no list all_to_all: pack the round's pieces instead
    for r in range(rounds):
        ins = [src[soff[h] + min(r * ch, send_l[h]): soff[h] + min((r + 1) * ch, send_l[h])] for h in range(world)]
        outs = [dst[roff[g] + min(r * ch, recv_l[g]): roff[g] + min((r + 1) * ch, recv_l[g])] for g in range(world)]
        if lists_ok:
            dist.all_to_all(outs, ins, group=group)  # views: no packing
        else:
            tmp = torch.empty(sum(o.numel() for o in outs), dtype=dst.dtype, device=dst.device)
            dist.all_to_all_single(tmp, torch.cat(ins), [o.numel() for o in outs], [i.numel() for i in ins], group=group)
            o0 = 0
            for o in outs:
                o.copy_(tmp[o0:o0 + o.numel()])
                o0 += o.numel()


def exchange_partial_csr(rowptr, colidx, vals, dist, world, group=None):
    """All-to-all-v of a partial CSR (all M rows) so that rank h ends up with every rank's rows of range h.

    rowptr int64 [M+1], colidx int32 [nnz], vals [nnz] -- torch tensors on the communication device.
    Returns (row_bounds, parts) where parts[g] = (rowptr_g int64 [nr+1], colidx_g, vals_g) for the rows
    [row_bounds[rank], row_bounds[rank+1]) as computed by rank g.
    """
    rank = dist.get_rank(group)
    M = rowptr.numel() - 1
    rownnz = (rowptr[1:] - rowptr[:-1]).to(torch.int32)
    weight = rownnz.clone()
    dist.all_reduce(weight, group=group)  # sum over ranks of per-row nnz (fits int32: asserted by caller sizes)
    rb = plan_row_ranges(weight, world)
    rb_t = torch.tensor(rb, device=rowptr.device, dtype=torch.int64)
    offs = rowptr[rb_t]                                   # element offset of each range in my arrays
    send_counts = (offs[1:] - offs[:-1]).to(torch.int64)
    recv_counts = torch.empty_like(send_counts)
    dist.all_to_all_single(recv_counts, send_counts, group=group)
    send_l = [int(x) for x in send_counts.tolist()]
    recv_l = [int(x) for x in recv_counts.tolist()]
    nr = rb[rank + 1] - rb[rank]
    row_send = [rb[h + 1] - rb[h] for h in range(world)]
    # per-row counts of my range as computed by every rank
    cnt_recv = torch.empty(nr * world, dtype=torch.int32, device=rowptr.device)
    all_to_all_v(cnt_recv, rownnz, [nr] * world, row_send, dist, world, group)
    col_recv = torch.empty(sum(recv_l), dtype=colidx.dtype, device=colidx.device)
    val_recv = torch.empty(sum(recv_l), dtype=vals.dtype, device=vals.device)
    all_to_all_v(col_recv, colidx, recv_l, send_l, dist, world, group)
    all_to_all_v(val_recv, vals, recv_l, send_l, dist, world, group)
    parts, o = [], 0
    for g in range(world):
        rp = torch.zeros(nr + 1, dtype=torch.int64, device=rowptr.device)
        rp[1:] = torch.cumsum(cnt_recv[g * nr:(g + 1) * nr].to(torch.int64), 0)
        parts.append((rp, col_recv[o:o + recv_l[g]], val_recv[o:o + recv_l[g]]))
        o += recv_l[g]
    return rb, parts


def k_sharded_product(local_product, merge_parts, k_bounds, dist, world, group=None, sync=None):
    """Generic driver: local slab product -> exchange -> local merge.

    local_product(k0, k1) -> (rowptr, colidx, vals) tensors (partial CSR over all rows)
    merge_parts(nrows, parts) -> (rowptr, colidx, vals) of the summed CSR
    Returns dict(row_bounds, rowptr, colidx, vals, seconds=(local, exchange, merge)).
    """
    rank = dist.get_rank(group)
    t0 = time.perf_counter()
    rowptr, colidx, vals = local_product(k_bounds[rank], k_bounds[rank + 1])
    if sync:
        sync()
    t1 = time.perf_counter()
    rb, parts = exchange_partial_csr(rowptr, colidx, vals, dist, world, group)
    if sync:
        sync()
    t2 = time.perf_counter()
    out = merge_parts(rb[rank + 1] - rb[rank], parts)
    if sync:
        sync()
    t3 = time.perf_counter()
    return dict(row_bounds=rb, rowptr=out[0], colidx=out[1], vals=out[2], seconds=(t1 - t0, t2 - t1, t3 - t2))


class _DevArray:
    """Zero-copy view of library-owned device memory for torch (``__cuda_array_interface__``)."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def _as_tensor(ptr, n, typestr, device, dtype):
    if n == 0:
        return torch.empty(0, dtype=dtype, device=device)
    return torch.as_tensor(_DevArray(ptr, n, typestr), device=device)


def spgemm_k_sharded(ctx, np_dtype, M, K, N, csc, csr, k_bounds, dist, rank, world, partial_capacity=0,
                     stage_through_host=False, checksum=False):
    """The GPU instantiation used by bench.py.  csc/csr: (ptr int64, idx int32, vals) CUDA tensors holding
    the FULL operands on every rank (each rank touches only its k slab).  Returns an info dict.

    stage_through_host=True moves the exchanged arrays through host memory so that the whole path can
    be rehearsed with the `gloo` backend (e.g. several ranks sharing one GPU); the product path over
    RCCL keeps everything in HBM."""
    device = csc[0].device
    ptrs = [t.data_ptr() for t in (*csc, *csr)]
    vt = "<f8" if np.dtype(np_dtype) == np.float64 else "<f4"
    tdt = torch.float64 if np.dtype(np_dtype) == np.float64 else torch.float32
    keep = {}

    def local_product(k0, k1):
        res = ctx.spgemm_csc_csr_device(np_dtype, M, K, N, ptrs, validate=False, partial_capacity=partial_capacity,
                                        k_range=(k0, k1))
        keep["local"] = res
        rp, ci, va = res.device_ptrs()
        out = (_as_tensor(rp, M + 1, "<i8", device, torch.int64), _as_tensor(ci, res.nnz, "<i4", device, torch.int32),
               _as_tensor(va, res.nnz, vt, device, tdt))
        return tuple(t.cpu() for t in out) if stage_through_host else out

    def merge_parts(nrows, parts):
        if stage_through_host:
            parts = [tuple(t.to(device) for t in p) for p in parts]
            keep["parts"] = parts
        torch.cuda.current_stream().synchronize()  # received data must have landed
        res = ctx.merge_csr_parts_device(np_dtype, nrows, N, [(r.data_ptr(), c.data_ptr(), v.data_ptr()) for r, c, v in parts],
                                         partial_capacity=partial_capacity)
        keep["final"] = res
        return res.device_ptrs()

    out = k_sharded_product(local_product, merge_parts, k_bounds, dist, world, sync=torch.cuda.synchronize)
    info = dict(keep["local"].info)
    fin = keep["final"].info
    tot = torch.tensor([fin["nnz_c"], info["partials"]], device="cpu" if stage_through_host else device, dtype=torch.int64)
    dist.all_reduce(tot)
    info.update(nnz_c_global=int(tot[0]), partials_global=int(tot[1]), nnz_c_final_local=fin["nnz_c"],
                ms_local=out["seconds"][0] * 1e3, ms_exchange=out["seconds"][1] * 1e3,
                ms_final_merge=out["seconds"][2] * 1e3, final_merge_partials=fin["partials"])
    if checksum:  # sum of all values of C (all ranks), for the 1^T C 1 = (1^T A)(B 1) sanity check
        _, _, va = keep["final"].device_ptrs()
        vs = _as_tensor(va, fin["nnz_c"], vt, device, tdt).sum(dtype=torch.float64).reshape(1)
        vs = vs.cpu() if stage_through_host else vs
        dist.all_reduce(vs)
        info["val_sum_global"] = float(vs[0])
    keep["local"].close()
    keep["final"].close()
    return info


def spgemm_row_sharded(ctx, np_dtype, M, K, N, ptrs, dist, rank, world, device, partial_capacity=0, host_collectives=False,
                       checksum=False):
    """Row-sharded product: rank i computes the i-th of `world` output-row ranges (balanced by partial products, derived
    by every rank from the replicated operands alone).  No data-path collective; only the counters of the report are
    all-reduced.  ptrs: the six device addresses of CSC(A) / CSR(B)."""
    res = ctx.spgemm_csc_csr_device(np_dtype, M, K, N, ptrs, validate=False, partial_capacity=partial_capacity,
                                    row_shard=(rank, world))
    info = dict(res.info)
    cdev = "cpu" if host_collectives else device
    tot = torch.tensor([info["nnz_c"], info["partials"]], device=cdev, dtype=torch.int64)
    dist.all_reduce(tot)
    info.update(nnz_c_global=int(tot[0]), partials_global=int(tot[1]))
    if checksum:
        vt = "<f8" if np.dtype(np_dtype) == np.float64 else "<f4"
        tdt = torch.float64 if np.dtype(np_dtype) == np.float64 else torch.float32
        _, _, va = res.device_ptrs()
        vs = _as_tensor(va, res.nnz, vt, device, tdt).sum(dtype=torch.float64).reshape(1)
        vs = vs.cpu() if host_collectives else vs
        dist.all_reduce(vs)
        info["val_sum_global"] = float(vs[0])
    res.close()
    return info
