"""k-sharded multi-GPU SpGEMM: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL).

No reference counterpart -- the reference is one process, one thread (SURVEY.md sections 2, 5).
Shape of the computation (BASELINE.json north_star):

1. The shared dimension k is cut into ``world`` contiguous slabs of (nearly) equal partial-product
   count.  Rank g holds ONLY its slab -- columns [k_g, k_g+1) of A, rows [k_g, k_g+1) of B (``slice_k_slab``) --
   and runs the single-GPU pipeline on it -> a partial CSR ``C_g`` over ALL rows.
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

from ._lib import OspError as _OspError


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


def exchange_partial_csr(rowptr, colidx, vals, dist, world, group=None, ncols=None, alloc=None, stats=None, widths=(1, 1)):
    """All-to-all-v of a partial CSR (all M rows) so that rank h ends up with every rank's rows of range h.

    rowptr int64 [M+1]; colidx [nnz * widths[0]], vals [nnz * widths[1]] -- the per-entry payload arrays, torch tensors on
    the communication device (`vals` may be None: one payload array only, e.g. packed records viewed as 32-bit words,
    widths = (3,) for {u32 col; f64 val}).
    ncols: number of columns of the matrix (bounds a row's entry count; None = unknown).
    alloc(numel, like) -> 1-D tensor for the large receive buffers (default torch.empty); the GPU path hands out
    memory of the library's pool, so that what the local product has just released is reused.
    stats (dict, optional) receives bytes_sent (payload this rank sends to OTHER ranks) and row_bounds.
    Returns (row_bounds, parts) where parts[g] = (rowptr_g int64 [nr+1], payload slices...) for the rows
    [row_bounds[rank], row_bounds[rank+1]) as computed by rank g.
    """
    rank = dist.get_rank(group)
    M = rowptr.numel() - 1
    payloads = [(colidx, int(widths[0]))] + ([(vals, int(widths[1]))] if vals is not None else [])
    # a row of one rank's partial CSR holds at most ncols entries (unmerged partial products: no such bound), the per-row
    # sum over the ranks at most world * ncols: 32-bit counts only where both are known to fit
    cnt_dtype = torch.int32 if (ncols is not None and int(ncols) < (1 << 31)) else torch.int64
    rownnz = (rowptr[1:] - rowptr[:-1]).to(cnt_dtype)
    wide = ncols is None or world * int(ncols) >= (1 << 31)
    weight = rownnz.to(torch.int64).clone() if wide else rownnz.clone()   # (all_reduce works in place: never on rownnz itself)
    dist.all_reduce(weight, group=group)  # sum over ranks of per-row entry counts
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
    cnt_recv = torch.empty(nr * world, dtype=cnt_dtype, device=rowptr.device)
    all_to_all_v(cnt_recv, rownnz, [nr] * world, row_send, dist, world, group)
    if alloc is None:
        alloc = lambda numel, like: torch.empty(numel, dtype=like.dtype, device=like.device)
    received = []
    for arr, wd in payloads:
        buf = alloc(sum(recv_l) * wd, arr)
        all_to_all_v(buf, arr, [x * wd for x in recv_l], [x * wd for x in send_l], dist, world, group)
        received.append((buf, wd))
    if stats is not None:
        away = sum(send_l) - send_l[rank]
        stats["bytes_sent"] = (away * sum(arr.element_size() * wd for arr, wd in payloads) + rownnz.element_size() * (M - row_send[rank])
                               + 8 * (world - 1))
        stats["row_bounds"] = list(rb)
    parts, o = [], 0
    for g in range(world):
        rp = torch.zeros(nr + 1, dtype=torch.int64, device=rowptr.device)
        rp[1:] = torch.cumsum(cnt_recv[g * nr:(g + 1) * nr].to(torch.int64), 0)
        parts.append((rp,) + tuple(buf[o * wd:(o + recv_l[g]) * wd] for buf, wd in received))
        o += recv_l[g]
    return rb, parts


class LocalDoesNotFit(Exception):
    """Raised by a ``local_product`` whose result does not fit this rank's device in the form agreed on."""


def k_sharded_product(local_product, merge_parts, k_bounds, dist, world, group=None, sync=None, ncols=None, alloc=None,
                      after_exchange=None, stats=None, widths=(1, 1), fallback=None, flag_device="cpu"):
    """Generic driver: local slab product -> exchange -> local merge.

    local_product(k0, k1) -> (rowptr, colidx, vals) tensors (partial CSR over all rows)
    merge_parts(nrows, parts) -> (rowptr, colidx, vals) of the summed CSR
    after_exchange() (optional) runs once the exchange has completed: the place to release the local partial CSR before
    the final merge allocates (at 2 GPUs and scale 22 both do not fit side by side).
    fallback (optional) = dict(local_product=, merge_parts=, ncols=, widths=): a second form of the same product that needs
    less memory.  When it is given, every rank reports after its local product whether that one raised
    ``LocalDoesNotFit`` (one all-reduce, before any other collective), and if ANY rank did, ALL ranks drop what they have
    and redo the local step in the fallback form -- the form of the exchange (payload widths, count types, merge function)
    is a property of the whole job, never of one rank.  Without it a ``LocalDoesNotFit`` simply propagates.
    Returns dict(row_bounds, rowptr, colidx, vals, seconds=(local, exchange, merge), fell_back).
    """
    rank = dist.get_rank(group)
    t0 = time.perf_counter()
    fell_back = False
    if fallback is None:
        local = local_product(k_bounds[rank], k_bounds[rank + 1])   # (rowptr, colidx, vals) or (rowptr, packed records)
    else:
        local, failed = None, 0
        try:
            local = local_product(k_bounds[rank], k_bounds[rank + 1])
        except LocalDoesNotFit:
            failed = 1
        flag = torch.tensor([failed], dtype=torch.int32, device=flag_device)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
        if int(flag[0]):
            del local
            if fallback.get("discard"):
                fallback["discard"]()   # whatever the first attempt left behind on this rank
            local = fallback["local_product"](k_bounds[rank], k_bounds[rank + 1])
            merge_parts, ncols, widths = fallback["merge_parts"], fallback.get("ncols"), fallback.get("widths", (1, 1))
            fell_back = True
    rowptr, colidx, vals = local if len(local) == 3 else (local[0], local[1], None)
    if sync:
        sync()
    t1 = time.perf_counter()
    rb, parts = exchange_partial_csr(rowptr, colidx, vals, dist, world, group, ncols=ncols, alloc=alloc, stats=stats, widths=widths)
    if sync:
        sync()
    del rowptr, colidx, vals, local
    if after_exchange:
        after_exchange()
    t2 = time.perf_counter()
    out = merge_parts(rb[rank + 1] - rb[rank], parts)
    if sync:
        sync()
    t3 = time.perf_counter()
    return dict(row_bounds=rb, rowptr=out[0], colidx=out[1], vals=out[2], seconds=(t1 - t0, t2 - t1, t3 - t2), fell_back=fell_back)


class _DevArray:
    """Zero-copy view of library-owned device memory for torch (``__cuda_array_interface__``)."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def _as_tensor(ptr, n, typestr, device, dtype):
    if n == 0:
        return torch.empty(0, dtype=dtype, device=device)
    return torch.as_tensor(_DevArray(ptr, n, typestr), device=device)


def _records_fit(ctx, csc, csr, np_dtype):
    """Can this rank hold all partial products of its slab at once, beside what it receives and merges?  P * E for the
    records it forms, the same again for the ones it receives, and the merge's staging on top: a quarter of the device."""
    w = (csc[0][1:] - csc[0][:-1]) * (csr[0][1:] - csr[0][:-1])
    P = int(w.sum())
    free_b, total_b = torch.cuda.mem_get_info(csc[0].device)
    return P * (4 + np.dtype(np_dtype).itemsize) < total_b // 4


def slice_k_slab(csc, csr, k0, k1):
    """The operands of one k shard: columns [k0,k1) of A (CSC) and rows [k0,k1) of B (CSR) as arrays of their own
    (pointers rebased to 0) -- all a rank of the k-sharded product holds ("only its columns of A and rows
    of B").  csc / csr: (ptr int64, idx int32, vals) tensors.  Returns (K', csc', csr')."""
    def cut(t):
        ptr, idx, val = t
        e0, e1 = int(ptr[k0]), int(ptr[k1])
        return ((ptr[k0:k1 + 1] - e0).contiguous(), idx[e0:e1].clone(), val[e0:e1].clone())
    return k1 - k0, cut(csc), cut(csr)


def agree_k_exchange(ctx, np_dtype, slab, dist, exchange="raw", stage_through_host=False):
    """The form of the exchange every rank will use, decided ONCE per slab (bench.py calls it beside ``slice_k_slab``,
    outside the timed steps): "raw" only if every rank expects its records to fit (``_records_fit``: a reduction over the
    slab, a host sync and a memory query -- not something to repeat in every step)."""
    if exchange != "raw":
        return exchange
    _, csc, csr = slab
    need = torch.tensor([1 if _records_fit(ctx, csc, csr, np_dtype) else 0], device="cpu" if stage_through_host else csc[0].device)
    dist.all_reduce(need, op=dist.ReduceOp.MIN)
    return "raw" if int(need[0]) else "merged"


def spgemm_k_sharded(ctx, np_dtype, M, N, slab, dist, rank, world, partial_capacity=0, stage_through_host=False, checksum=False,
                     fetch=False, exchange="raw", agreed=False, _fail_raw_on_rank=None):
    """The GPU instantiation used by bench.py.  slab = slice_k_slab(...) of THIS rank: (K', csc', csr') CUDA tensors.
    Returns an info dict (the local product's counters plus the exchange / final-merge figures); fetch=True adds
    info["final_csr"] = this rank's rows of C as host arrays (rowptr, colidx, vals) -- tests only.

    exchange="raw" (default): the rank runs the MULTIPLY phase only (``osp_spgemm_partials``) and sends its partial
    products as they are staged -- packed records grouped by output row, ascending k inside a row; the owner of a row range
    merges everything ONCE (``osp_merge_record_parts``), parts in rank order = ascending k, which is the single-GPU
    summation order: the result is bit-identical to the one-GPU product.  R-MAT products hardly compress (nnz(C)/P = 0.97),
    so a local merge would shrink the exchange by 3 % and cost a whole merge pass.
    exchange="merged": local product to a partial CSR first, the partial CSRs exchanged and merged (the round-1 form):
    less to send when the product compresses well (Graph500 parameters: x2.2).  Falls back to this when the records do not
    fit the device at once.

    Memory: the exchange receives into buffers of the library's pool (the staging memory the local product has just given
    back), and the local partial CSR is released as soon as the exchange has completed, before the final merge allocates.

    stage_through_host=True moves the exchanged arrays through host memory so that the whole path can
    be rehearsed with the `gloo` backend (e.g. several ranks sharing one GPU); the product path over
    RCCL keeps everything in HBM."""
    Ks, csc, csr = slab
    device = csc[0].device
    torch.cuda.synchronize(device)   # the operands may come from torch kernels still in flight; the library's stream does not wait
    ptrs = [t.data_ptr() for t in (*csc, *csr)]
    vt = "<f8" if np.dtype(np_dtype) == np.float64 else "<f4"
    tdt = torch.float64 if np.dtype(np_dtype) == np.float64 else torch.float32
    keep, pooled, stats = {}, [], {}

    words = 1 + np.dtype(np_dtype).itemsize // 4   # 32-bit words per packed record
    mode = {"exchange": exchange}

    def local_raw(_k0, _k1):
        try:
            if _fail_raw_on_rank is not None and rank == _fail_raw_on_rank:   # tests: the allocation failure of ONE rank
                raise _OspError(3, "forced allocation failure (test)")
            res = ctx.spgemm_partials_device(np_dtype, M, Ks, N, ptrs)
        except _OspError as e:
            if e.status != 3:   # anything but "does not fit": a real error
                raise
            raise LocalDoesNotFit(str(e))
        keep["local"] = res
        keep["local_info"] = dict(res.info)
        rp, rec = res.partials_ptrs()
        out = (_as_tensor(rp, M + 1, "<i8", device, torch.int64), _as_tensor(rec, res.nnz * words, "<i4", device, torch.int32))
        return tuple(t.cpu() for t in out) if stage_through_host else out

    def discard_raw():
        if "local" in keep:
            keep.pop("local").close()
        mode["exchange"] = "merged"

    def local_merged(_k0, _k1):
        res = ctx.spgemm_csc_csr_device(np_dtype, M, Ks, N, ptrs, validate=False, partial_capacity=partial_capacity)
        keep["local"] = res
        keep["local_info"] = dict(res.info)
        rp, ci, va = res.device_ptrs()
        out = (_as_tensor(rp, M + 1, "<i8", device, torch.int64), _as_tensor(ci, res.nnz, "<i4", device, torch.int32),
               _as_tensor(va, res.nnz, vt, device, tdt))
        return tuple(t.cpu() for t in out) if stage_through_host else out

    def pool_alloc(numel, like):
        if stage_through_host or numel == 0:
            return torch.empty(numel, dtype=like.dtype, device=like.device)
        p = ctx.alloc(numel * like.element_size())
        pooled.append(p)
        return _as_tensor(p, numel, {torch.int32: "<i4", torch.float32: "<f4", torch.float64: "<f8"}[like.dtype], device, like.dtype)

    def release_local():
        keep.pop("local").close()

    def landed(parts):
        if stage_through_host:
            parts = [tuple(t.to(device) for t in p) for p in parts]
            keep["parts"] = parts
        torch.cuda.current_stream().synchronize()  # received data must have landed
        return parts

    def merge_raw(nrows, parts):
        parts = landed(parts)
        res = ctx.merge_record_parts_device(np_dtype, nrows, N, [(r.data_ptr(), c.data_ptr()) for r, c in parts],
                                            partial_capacity=partial_capacity)
        keep["final"] = res
        return res.device_ptrs()

    def merge_merged(nrows, parts):
        parts = landed(parts)
        res = ctx.merge_csr_parts_device(np_dtype, nrows, N, [(r.data_ptr(), c.data_ptr(), v.data_ptr()) for r, c, v in parts],
                                         partial_capacity=partial_capacity)
        keep["final"] = res
        return res.device_ptrs()

    try:
        # Every rank must take the same form of exchange.  Two agreements: BEFORE the product, an estimate (a rank whose
        # records are not expected to fit turns all of them to the merged form; `agreed=True`: the caller has done that once
        # for this slab with agree_k_exchange); AFTER the local product, the fact -- a rank whose allocation failed all the
        # same makes every rank redo the step in the merged form (k_sharded_product's fallback), instead of leaving the
        # others in a collective it never joins.
        if not agreed:
            mode["exchange"] = agree_k_exchange(ctx, np_dtype, slab, dist, mode["exchange"], stage_through_host)
        k_bounds = [0] * (rank + 1) + [Ks] * (world - rank)
        common = dict(sync=torch.cuda.synchronize, alloc=pool_alloc, after_exchange=release_local, stats=stats,
                      flag_device="cpu" if stage_through_host else device)
        if mode["exchange"] == "raw":
            out = k_sharded_product(local_raw, merge_raw, k_bounds, dist, world, ncols=None, widths=(words,),
                                    fallback=dict(local_product=local_merged, merge_parts=merge_merged, ncols=N, widths=(1, 1),
                                                  discard=discard_raw), **common)
        else:
            out = k_sharded_product(local_merged, merge_merged, k_bounds, dist, world, ncols=N, widths=(1, 1), **common)
        info = keep["local_info"]
        fin = keep["final"].info
        tot = torch.tensor([fin["nnz_c"], info["partials"]], device="cpu" if stage_through_host else device, dtype=torch.int64)
        dist.all_reduce(tot)
        info.update(nnz_c_global=int(tot[0]), partials_global=int(tot[1]), nnz_c_final_local=fin["nnz_c"],
                    ms_local=out["seconds"][0] * 1e3, ms_exchange=out["seconds"][1] * 1e3,
                    ms_final_merge=out["seconds"][2] * 1e3, final_merge_partials=fin["partials"],
                    bytes_sent=stats.get("bytes_sent", 0), row_bounds=stats.get("row_bounds"), exchange=mode["exchange"],
                    final_info=dict(fin))
        if checksum:  # sum of all values of C (all ranks), for the 1^T C 1 = (1^T A)(B 1) sanity check
            _, _, va = keep["final"].device_ptrs()
            vs = _as_tensor(va, fin["nnz_c"], vt, device, tdt).sum(dtype=torch.float64).reshape(1)
            vs = vs.cpu() if stage_through_host else vs
            dist.all_reduce(vs)
            info["val_sum_global"] = float(vs[0])
        if fetch:
            info["final_csr"] = keep["final"].to_host()
        return info
    finally:
        for r in ("local", "final"):
            if r in keep:
                keep[r].close()
        keep.clear()
        for p in pooled:
            ctx.free(p)


def spgemm_row_sharded(ctx, np_dtype, M, K, N, ptrs, dist, rank, world, device, partial_capacity=0, host_collectives=False,
                       checksum=False, fetch=False):
    """Row-sharded product: rank i computes the i-th of `world` output-row ranges (balanced by partial products, derived
    by every rank from the replicated operands alone).  No data-path collective; only the counters of the report are
    all-reduced.  ptrs: the six device addresses of CSC(A) / CSR(B)."""
    torch.cuda.synchronize(device)   # (operands from torch kernels still in flight)
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
    if fetch:
        info["final_csr"] = res.to_host()
    res.close()
    return info
