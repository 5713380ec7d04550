"""NN-side glue (SURVEY 8f.2 / BASELINE configs[4]): helpers pinned against values captured from the reference's
own ``sparse_util`` / ``util`` (tests/golden/make_golden.py), sparse MLP layers against dense torch."""
import os

import numpy as np
import pytest
import torch


def test_prune_helpers_match_captured_reference(golden_dir):
    from outerspace_amd import sparse_util as su
    g = np.load(os.path.join(golden_dir, "mlp_expected.npz"))
    w = np.load(os.path.join(golden_dir, "mlp_weight_dense.npz"))
    W, Wp = torch.from_numpy(w["W"]), torch.from_numpy(w["Wp"])
    thr = su.get_prune_threshold(W, 0.05)
    assert np.float32(thr) == g["prune_thr"]
    assert torch.equal(su.prune_by_magnitude(W, 0.05), Wp)
    cnt, numel, frac = su.get_sparsity(Wp)
    assert int(cnt) == int(g["w_nnz"]) and int(numel) == int(g["w_numel"]) and np.float32(frac) == g["w_frac"]
    # the reference's signed mask keeps only positive weights above the |w| threshold
    signed = su.prune_to_sparsity(W, 0.05)
    assert int((signed != 0).sum()) < int(cnt) and bool((signed >= 0).all())
    assert torch.equal(su.prune_to_sparsity(Wp, 0.5), Wp)  # already sparse enough -> unchanged


def test_mtx_writer_is_byte_identical(golden_dir, tmp_path):
    from outerspace_amd import sparse_util as su
    Wp = torch.from_numpy(np.load(os.path.join(golden_dir, "mlp_weight_dense.npz"))["Wp"])
    out = tmp_path / "w.mtx"
    su.save_tensor_as_mtx(Wp, str(out))
    assert out.read_bytes() == open(os.path.join(golden_dir, "mlp_fc1_weight.mtx"), "rb").read()


@pytest.mark.gpu
def test_sparse_mlp_chain_f32(golden_dir, tmp_path):
    """act(64x784) -> fc1(100) -> fc2(100) -> fc3(10) with pruned weights, f32, within 1e-5 of dense torch."""
    from outerspace_amd import sparse_util as su
    torch.manual_seed(1)
    x = torch.relu(torch.randn(64, 784) - 1.0)
    dims = [784, 100, 100, 10]
    layers = []
    for i in range(3):
        w = su.prune_by_magnitude(torch.randn(dims[i + 1], dims[i]) * 0.05, 0.10)
        b = torch.randn(dims[i + 1]) * 0.01
        layers.append((w, b))
    logits, acts = su.mlp_forward(x, layers)
    ref = x
    ref_acts = []
    for i, (w, b) in enumerate(layers):
        ref = ref @ w.T + b
        if i < 2:
            ref = torch.relu(ref)
            ref_acts.append(ref)
    assert np.allclose(logits.toarray(), ref.numpy(), rtol=1e-5, atol=1e-5)
    for a, r in zip(acts, ref_acts):
        assert np.allclose(a.toarray(), r.numpy(), rtol=1e-5, atol=1e-5)
    # the same through the .mtx hand-off the reference uses
    d = tmp_path / "mtx"
    d.mkdir()
    su.save_tensor_as_mtx(x, str(d / "act_0.mtx"))
    for i, (w, b) in enumerate(layers, 1):
        su.save_tensor_as_mtx(w, str(d / f"fc{i}_weight.mtx"))
        su.save_tensor_as_mtx(b.reshape(1, -1), str(d / f"fc{i}_bias.mtx"))
    logits2, _ = su.mlp_forward_from_mtx(str(d))
    assert np.allclose(logits2.toarray(), ref.numpy(), rtol=1e-4, atol=1e-5)  # values went through 8-digit text
    # sparse W3 W2 W1 chain
    chain = su.weight_chain([w for w, _ in layers])
    dense = layers[2][0] @ layers[1][0] @ layers[0][0]
    assert np.allclose(chain.toarray(), dense.numpy(), rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_bias_relu_epilogue_on_device(dt):
    """osp_csr_bias_relu against the dense definition (models.py:17-31: x = relu(fc(x)), the bias added to EVERY element):
    with and without bias, with and without ReLU; zeros are dropped, an entry that the bias lifts above zero appears,
    rows may end up empty; values bit-equal to the same arithmetic in numpy."""
    import scipy.sparse as sp
    from outerspace_amd import spgemm as S
    rng = np.random.default_rng(7)
    M, K, N = 37, 50, 130
    A = sp.random(M, K, 0.15, random_state=1, dtype=np.float64).astype(dt).tocsr()
    B = sp.random(K, N, 0.1, random_state=2, dtype=np.float64).astype(dt).tocsr()
    A.data -= dt(0.5)   # negative products too
    A[5, :] = 0          # an empty product row
    A.eliminate_zeros()
    bias = rng.standard_normal(N).astype(dt) * dt(0.1)
    bias[::7] = 0
    ctx = S.default_context()
    a, b = A.tocoo(), B.tocoo()
    res = ctx.spgemm_coo(M, K, N, (a.row, a.col, a.data), (b.row, b.col, b.data))
    C = res.to_scipy().toarray()
    for use_bias in (True, False):
        for relu in (True, False):
            out = res.bias_relu(bias if use_bias else None, relu)
            got = out.to_scipy()
            want = C + bias[None, :] if use_bias else C.copy()
            if not use_bias:
                want = np.where(res.to_scipy().toarray() != 0, want, 0)   # only stored entries take part
            if relu:
                want = np.maximum(want, 0)
            assert out.nnz == np.count_nonzero(want)
            assert np.array_equal(got.toarray(), want)
            assert np.all(np.diff(got.indptr) >= 0) and got.indptr[-1] == out.nnz
            for r in range(M):
                cols = got.indices[got.indptr[r]:got.indptr[r + 1]]
                assert np.all(np.diff(cols) > 0)
            out.close()
    res.close()
