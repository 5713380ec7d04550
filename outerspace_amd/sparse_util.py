"""NN-side glue: the reference's pruning helpers and `.mtx` hand-off, plus the GPU sparse layers they feed.

Mirrors, name for name, what a user of the reference imports today (all paths under
``/root/reference/NN_models``):

* ``get_sparsity`` / ``get_prune_threshold`` / ``get_sparse_mask`` / ``prune_to_sparsity``
  -- ``sparse_util.py:5-22`` (including the SIGNED mask ``mat > threshold`` of ``:12-15``, which drops every
  negative weight; ``main.py:208-211`` prunes with ``|w| > threshold`` instead -- ``prune_by_magnitude``).
  ``print_parameters_sparsity`` (``:24-30``) is training-side logging and is not mirrored.
* ``save_tensor_as_mtx`` -- ``util.py:61-62`` (``scipy.io.mmwrite`` of the CSR form; byte-identical files).
* ``sparse_linear`` / ``mlp_forward`` / ``mlp_forward_from_mtx`` -- the products the reference hands to its
  simulator one at a time (``get_mtx_files.py:76-96``: ``./simulator act_i.mtx fc{i+1}_weight.mtx`` computes
  ``act_i @ W.T``), here executed by the MI355X SpGEMM and chained with bias + ReLU like ``models.py:17-31``.
* ``weight_chain`` -- the sparse ``W_n ... W_2 W_1`` product of BASELINE.json's configs[4].

The helpers are plain torch / scipy plumbing (as in the reference); every matrix product goes through
``outerspace_amd.spgemm`` on the GPU.
"""
import os

import numpy as np
import scipy.io
import scipy.sparse as sp
import torch

from . import spgemm as _S


# ---- the pruning helpers of sparse_util.py:5-22, same names and results (pinned to captured values) -------------------
def get_sparsity(mat):
    """(non-zero count, element count, density) -- the triple ``sparse_util.py:5-7`` returns (count and density as tensors)."""
    nonzero = torch.count_nonzero(mat.abs() > 0)
    total = mat.numel()
    return nonzero, total, nonzero / total


def get_prune_threshold(mat, sparsity_level):
    """The |w| value below which all but a ``sparsity_level`` fraction of the entries lie (``sparse_util.py:9-10``)."""
    return torch.quantile(mat.abs(), 1 - sparsity_level)


def get_sparse_mask(mat, sparsity_level):
    """SIGNED comparison, as ``sparse_util.py:12-15`` does it: negative weights never pass."""
    return mat > get_prune_threshold(mat, sparsity_level)


def prune_to_sparsity(mat, sparsity_level):
    """``sparse_util.py:17-22``: a matrix already at or below the level is returned as it is."""
    _, _, density = get_sparsity(mat)
    return mat if density <= sparsity_level else mat * get_sparse_mask(mat, sparsity_level)


def prune_by_magnitude(mat, sparsity_level):
    """What ``main.py:208-211`` does per layer: keep ``|w| > quantile(|w|, 1 - s)``."""
    return mat * (mat.abs() > get_prune_threshold(mat, sparsity_level))


# ---- util.py:61-62 ------------------------------------------------------------------------------------
def save_tensor_as_mtx(a, save_file):
    scipy.io.mmwrite(save_file, sp.csr_matrix(a.numpy()))


# ---- the products -------------------------------------------------------------------------------------
def _coo_on_device(x, dtype, device):
    """dense tensor / ndarray / scipy sparse -> (nrow, ncol, rows i32, cols i32, vals) torch tensors on the GPU (the file
    order scipy's CSR gives, which is what ``save_tensor_as_mtx`` would have written)."""
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    if not sp.issparse(x):
        x = sp.csr_matrix(np.asarray(x))
    x = x.tocoo()
    tdt = torch.float32 if np.dtype(dtype) == np.float32 else torch.float64
    return (x.shape[0], x.shape[1], torch.from_numpy(x.row.astype(np.int32)).to(device), torch.from_numpy(x.col.astype(np.int32)).to(device),
            torch.from_numpy(x.data.astype(dtype)).to(device, tdt))


class _DeviceLayerInput:
    """An activation as the next product's A operand, resident on the GPU: COO arrays that either came from the host
    once (the network's input) or ARE the previous layer's result (its colidx / vals arrays plus a row array)."""

    def __init__(self, shape, rows, cols, vals, nnz, keep=()):
        self.shape, self.rows, self.cols, self.vals, self.nnz, self.keep = shape, rows, cols, vals, nnz, keep


def _layer_on_device(ctx, act, weight_coo, bias, relu, dtype, device):
    """relu(act @ W.T + bias) with everything on the GPU: osp_spgemm_coo on device arrays (COO -> CSC / CSR there), then
    osp_csr_bias_relu on the product.  Returns the CsrResult of the layer's output."""
    M, K = act.shape
    out_n, in_n, wr, wc, wv = weight_coo   # W is out x in; B = W^T: rows = in index, cols = out index
    if in_n != K:
        raise _S.OspError(1, f"inner dimensions differ: activation is {M}x{K}, weight is {out_n}x{in_n}")
    torch.cuda.synchronize(device)   # the library works on its own stream
    prod = ctx.spgemm_coo_device(dtype, M, K, out_n, act.nnz, (act.rows.data_ptr() if act.nnz else 0, act.cols.data_ptr() if act.nnz else 0,
                                                              act.vals.data_ptr() if act.nnz else 0),
                                 wv.numel(), (wc.data_ptr() if wv.numel() else 0, wr.data_ptr() if wv.numel() else 0,
                                              wv.data_ptr() if wv.numel() else 0))
    if bias is None and not relu:
        return prod
    b = None
    if bias is not None:
        b = bias.detach().cpu().numpy() if hasattr(bias, "detach") else np.asarray(bias)
        b = b.reshape(-1).astype(dtype)
    out = prod.bias_relu(b, relu)
    prod.close()
    return out


def _result_as_input(res, device):
    """The previous layer's CSR result as the next product's COO operand, without leaving the GPU: its column / value
    arrays are borrowed, the row array is written by the library (osp_result_coo_rows)."""
    from .distributed import _as_tensor
    _, ci, va = res.device_ptrs()
    nnz = res.nnz
    tdt = torch.float32 if res.dtype == np.float32 else torch.float64
    rows = torch.empty(max(nnz, 1), dtype=torch.int32, device=device)
    torch.cuda.synchronize(device)
    if nnz:
        res.coo_rows_into(rows.data_ptr())
    cols = _as_tensor(ci, nnz, "<i4", device, torch.int32)
    vals = _as_tensor(va, nnz, "<f4" if res.dtype == np.float32 else "<f8", device, tdt)
    return _DeviceLayerInput(res.shape, rows, cols, vals, nnz, keep=(res,))


def sparse_linear(act, weight, bias=None, relu=False, ctx=None, dtype=np.float32):
    """``relu(act @ weight.T + bias)``: product, bias and ReLU on the GPU (osp_spgemm_coo + osp_csr_bias_relu).
    act: (batch x in), weight: (out x in), both dense tensors / arrays or scipy sparse; returns scipy CSR."""
    out, _ = mlp_forward(act, [(weight, bias)], ctx=ctx, dtype=dtype, relu_last=relu)
    return out


def mlp_forward(x, layers, ctx=None, dtype=np.float32, relu_last=False):
    """layers = [(W1, b1), (W2, b2), ...]; ReLU after every layer but the last (``models.py:17-31``).
    Returns (logits CSR, [activation CSRs]) like ``MLP1.forward`` returns ``(x3, (x1, x2))``.
    The activations never leave the GPU between layers: a layer's result (CSR in HBM) is the next product's operand as it
    stands; only what is returned is copied to the host, at the end."""
    ctx = ctx or _S.default_context()
    device = torch.device("cuda", ctx.device)
    M, K, r, c, v = _coo_on_device(x, dtype, device)
    cur = _DeviceLayerInput((M, K), r, c, v, v.numel())
    results = []
    for li, (w, b) in enumerate(layers):
        last = li == len(layers) - 1
        res = _layer_on_device(ctx, cur, _coo_on_device(w, dtype, device), b, relu_last if last else True, dtype, device)
        results.append(res)
        if not last:
            cur = _result_as_input(res, device)
    outs = [r_.to_scipy() for r_ in results]
    for r_ in results:
        r_.close()
    return outs[-1], outs[:-1]


def mlp_forward_from_mtx(directory, nlayers=3, ctx=None, dtype=np.float32):
    """Run the chain on the files ``get_MLP1`` dumps: ``act_0.mtx``, ``fc{i}_weight.mtx``, ``fc{i}_bias.mtx``."""
    def load(name):
        nr, nc, r, c, v = _S.read_mtx(os.path.join(directory, name))
        return sp.csr_matrix((v.astype(dtype), (r, c)), shape=(nr, nc))
    layers = []
    for i in range(1, nlayers + 1):
        bias_path = os.path.join(directory, f"fc{i}_bias.mtx")
        bias = load(f"fc{i}_bias.mtx").toarray().reshape(-1) if os.path.exists(bias_path) else None
        layers.append((load(f"fc{i}_weight.mtx"), bias))
    return mlp_forward(load("act_0.mtx"), layers, ctx=ctx, dtype=dtype)


def weight_chain(weights, ctx=None, dtype=np.float32):
    """``W_n @ ... @ W_2 @ W_1`` for ``nn.Linear`` weights (each out x in), sparse x sparse on the GPU."""
    acc = weights[0]
    for w in weights[1:]:
        acc = _S.spgemm(w, acc, transpose_b=False, ctx=ctx, dtype=dtype)
    return acc if sp.issparse(acc) else sp.csr_matrix(np.asarray(acc))
