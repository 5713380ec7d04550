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
def _bias_relu_on_device(prod, bias, relu, dtype, device_index=0):
    """``relu(prod + bias)`` for a CSR product, on the GPU: the product is scattered into a dense block in HBM, bias and
    ReLU are applied there (a bias makes every entry of the row non-zero anyway), and what survives the ReLU comes back
    as CSR.  ``models.py:17-31`` does the same three steps on dense tensors."""
    dev = torch.device("cuda", device_index)
    m, n = prod.shape
    tdt = torch.float32 if np.dtype(dtype) == np.float32 else torch.float64
    dense = torch.zeros((m, n), dtype=tdt, device=dev)
    if prod.nnz:
        rows = torch.from_numpy(np.repeat(np.arange(m, dtype=np.int64), np.diff(prod.indptr))).to(dev)
        dense[rows, torch.from_numpy(prod.indices.astype(np.int64)).to(dev)] = torch.from_numpy(prod.data.astype(dtype)).to(dev)
    if bias is not None:
        b = bias.detach() if hasattr(bias, "detach") else torch.as_tensor(np.asarray(bias))
        dense += b.to(dev, tdt).reshape(1, -1)
    if relu:
        dense.clamp_(min=0)
    nz = dense != 0
    counts = nz.sum(dim=1)
    indptr = np.zeros(m + 1, np.int64)
    indptr[1:] = np.cumsum(counts.cpu().numpy())
    cols = nz.nonzero()[:, 1]
    return sp.csr_matrix((dense[nz].cpu().numpy(), cols.cpu().numpy(), indptr), shape=(m, n))


def sparse_linear(act, weight, bias=None, relu=False, ctx=None, dtype=np.float32):
    """``relu(act @ weight.T + bias)``: the product on the GPU through the SpGEMM library, bias and ReLU on the GPU as
    well.  act: (batch x in), weight: (out x in), both dense tensors / arrays or scipy sparse; returns scipy CSR."""
    prod = _S.spgemm(act, weight, transpose_b=True, ctx=ctx, dtype=dtype)
    if bias is None and not relu:
        return prod
    return _bias_relu_on_device(prod, bias, relu, dtype, (ctx or _S.default_context()).device)


def mlp_forward(x, layers, ctx=None, dtype=np.float32):
    """layers = [(W1, b1), (W2, b2), ...]; ReLU after every layer but the last (``models.py:17-31``).
    Returns (logits CSR, [activation CSRs]) like ``MLP1.forward`` returns ``(x3, (x1, x2))``."""
    acts = []
    cur = x
    for li, (w, b) in enumerate(layers):
        last = li == len(layers) - 1
        cur = sparse_linear(cur, w, b, relu=not last, ctx=ctx, dtype=dtype)
        if not last:
            acts.append(cur)
    return cur, acts


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
