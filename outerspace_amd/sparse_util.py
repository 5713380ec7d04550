"""NN-side glue: the reference's pruning helpers and `.mtx` hand-off, plus the GPU sparse layers they feed.

Mirrors, name for name, what a user of the reference imports today (all paths under
``/root/reference/NN_models``):

* ``get_sparsity`` / ``get_prune_threshold`` / ``get_sparse_mask`` / ``prune_to_sparsity``
  -- ``sparse_util.py:5-22`` (including the SIGNED mask ``mat > threshold`` of ``:12-15``, which drops every
  negative weight; ``main.py:208-211`` prunes with ``|w| > threshold`` instead -- ``prune_by_magnitude``).
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


# ---- sparse_util.py:5-30 ------------------------------------------------------------------------------
def get_sparsity(mat):
    non_zeros_count = abs(mat).gt(0).sum()
    return (non_zeros_count, torch.numel(mat), non_zeros_count / torch.numel(mat))


def get_prune_threshold(mat, sparsity_level):
    return torch.quantile(abs(mat), 1 - sparsity_level)


def get_sparse_mask(mat, sparsity_level):
    threshold = get_prune_threshold(mat, sparsity_level)
    return mat > threshold  # signed, as in the reference (sparse_util.py:14)


def prune_to_sparsity(mat, sparsity_level):
    if get_sparsity(mat)[2] <= sparsity_level:  # already at or below the desired level
        return mat
    return mat * get_sparse_mask(mat, sparsity_level)


def prune_by_magnitude(mat, sparsity_level):
    """What ``main.py:208-211`` does per layer: keep ``|w| > quantile(|w|, 1 - s)``."""
    return mat * (mat.abs() > get_prune_threshold(mat, sparsity_level))


def print_parameters_sparsity(model):
    print("parameters sparsity: ")
    for name, param in model.named_parameters():
        if param.requires_grad:
            print(name, get_sparsity(param))


# ---- util.py:61-62 ------------------------------------------------------------------------------------
def save_tensor_as_mtx(a, save_file):
    scipy.io.mmwrite(save_file, sp.csr_matrix(a.numpy()))


# ---- the products -------------------------------------------------------------------------------------
def sparse_linear(act, weight, bias=None, relu=False, ctx=None, dtype=np.float32):
    """``relu(act @ weight.T + bias)`` with the product on the GPU.  act: (batch x in), weight: (out x in), both
    dense tensors / arrays or scipy sparse; returns scipy CSR (bias and ReLU are applied to the product)."""
    prod = _S.spgemm(act, weight, transpose_b=True, ctx=ctx, dtype=dtype)
    if bias is None and not relu:
        return prod
    out = prod.toarray()
    if bias is not None:
        out = out + np.asarray(bias.detach().cpu() if hasattr(bias, "detach") else bias, dtype=out.dtype).reshape(1, -1)
    if relu:
        out = np.maximum(out, 0)
    return sp.csr_matrix(out)


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
