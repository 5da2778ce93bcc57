"""CPU tests for the RMSNorm / LayerNorm rows: the oracle's pullback formulas (restating
src/rms_norm.jl:40-101 and src/layer_norm.jl:97-133) match finite differences of the naive forward formulas
(test/rmsnorm_tests.jl:7-9, test/layernorm_tests.jl:7-11); the C ABI validates descriptors and sizes the workspace."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle.naive_norms import (naive_layer_norm, naive_layer_norm_grads, naive_rms_norm, naive_rms_norm_grads,
                                norm_bytes)


def _fd(f, args, k, idx, eps=1e-6):
    a = [x.copy() for x in args]
    a[k][idx] += eps
    b = [x.copy() for x in args]
    b[k][idx] -= eps
    return (f(*a) - f(*b)) / (2 * eps)


@pytest.mark.parametrize("offset", [0.0, 1.0])
def test_rms_norm_grads_match_finite_differences(offset):
    rng = np.random.default_rng(0)
    x, w, dy = rng.random((5, 33)), rng.random(33), rng.standard_normal((5, 33))
    dx, dw = naive_rms_norm_grads(dy, x, w, offset=offset)
    loss = lambda x, w: (naive_rms_norm(x, w, offset=offset)[0] * dy).sum()
    for idx in [(0, 0), (2, 17), (4, 32)]:
        assert abs(_fd(loss, [x, w], 0, idx) - dx[idx]) < 1e-7
    for idx in [0, 16, 32]:
        assert abs(_fd(loss, [x, w], 1, idx) - dw[idx]) < 1e-7


def test_layer_norm_grads_match_finite_differences():
    rng = np.random.default_rng(1)
    x, w, b, dy = rng.random((4, 29)), rng.random(29), rng.random(29), rng.standard_normal((4, 29))
    dx, dw, db = naive_layer_norm_grads(dy, x, w)
    loss = lambda x, w, b: (naive_layer_norm(x, w, b)[0] * dy).sum()
    for idx in [(0, 0), (1, 13), (3, 28)]:
        assert abs(_fd(loss, [x, w, b], 0, idx) - dx[idx]) < 1e-6
    for idx in [0, 14, 28]:
        assert abs(_fd(loss, [x, w, b], 1, idx) - dw[idx]) < 1e-7
        assert abs(_fd(loss, [x, w, b], 2, idx) - db[idx]) < 1e-7


def test_forward_statistics():
    rng = np.random.default_rng(2)
    x, w, b = rng.standard_normal((6, 257)), rng.random(257), rng.random(257)
    y, mu, rstd = naive_layer_norm(x, w, b)
    xn = (y - b) / w
    np.testing.assert_allclose(xn.mean(-1), 0, atol=1e-12)
    np.testing.assert_allclose((xn ** 2).mean(-1), 1 - 1e-6 * rstd ** 2, rtol=1e-9)
    y2, r2 = naive_rms_norm(x, np.ones(257))
    np.testing.assert_allclose((y2 ** 2).mean(-1), 1 - 1e-6 * r2 ** 2, rtol=1e-9)
    assert norm_bytes(1024, 1024, 4) == 8 * 2 ** 20 and norm_bytes(1024, 1024, 2, bwd=True) == 6 * 2 ** 20


@pytest.mark.parametrize("kw,status", [
    (dict(dtype=9), "NNOP_ERR_DTYPE"),
    (dict(dtype=0, w_dtype=2), "NNOP_ERR_DTYPE"),       # w must be fp32 or T
    (dict(emb=0), "NNOP_ERR_SHAPE"),
    (dict(n=0), "NNOP_ERR_SHAPE"),
    (dict(reserved=1), "NNOP_ERR_SHAPE"),
    (dict(), "NNOP_ERR_NULL"),
])
def test_norm_descriptor_validation(pkg, kw, status):
    lib = pkg._lib.load()
    base = dict(dtype=2, w_dtype=0, emb=1024, reserved=0, n=64)
    base.update(kw)
    d = pkg._lib.NormDesc(**base)
    null = C.c_void_p(0)
    f = C.c_float(0.0)
    want = getattr(pkg._lib, status)
    assert lib.nnop_rms_norm(C.byref(d), null, null, null, null, f, f, null) == want
    assert lib.nnop_rms_norm_bwd(C.byref(d), null, null, null, null, null, null, f, null, 0, null) == want
    assert lib.nnop_layer_norm(C.byref(d), null, null, null, null, null, null, f, null) == want
    assert lib.nnop_layer_norm_bwd(C.byref(d), null, null, null, null, null, null, null, null, null, 0, null) == want
    if status != "NNOP_ERR_NULL":
        assert lib.nnop_norm_bwd_workspace_bytes(C.byref(d), 0) == 0


def test_norm_workspace_sizes(pkg):
    lib = pkg._lib.load()
    for n, emb in [(1, 15), (25, 1024), (4096, 4096), (10 ** 6, 512)]:
        d = pkg._lib.NormDesc(dtype=0, w_dtype=0, emb=emb, reserved=0, n=n)
        parts = min(1024, max(1, -(-n // 4)))
        assert lib.nnop_norm_bwd_workspace_bytes(C.byref(d), 0) == parts * emb * 4
        assert lib.nnop_norm_bwd_workspace_bytes(C.byref(d), 1) == 2 * parts * emb * 4


def test_norm_host_refuses_cpu_tensors(pkg):
    x, w = torch.ones(4, 32), torch.ones(32)
    with pytest.raises(pkg.NNopError, match="GPU-only"):
        pkg.rms_norm(x, w)
    with pytest.raises(pkg.NNopError, match="GPU-only"):
        pkg.layer_norm(x, w, w)


def test_oracle_reproduces_row_golden_fixtures():
    """tests/golden/rows_*.npz (make_golden.py) are what the oracles compute today, for softmax and both norms."""
    import glob
    import os
    from oracle.naive_softmax import naive_softmax, naive_softmax_grad
    files = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "rows_*.npz")))
    assert len(files) >= 3
    for f in files:
        g = np.load(f)
        off, eps = float(g["offset"]), float(g["eps"])
        np.testing.assert_allclose(naive_softmax(g["x"]), g["softmax_y"], rtol=1e-6, atol=1e-12)
        np.testing.assert_allclose(naive_softmax_grad(g["dy"], g["softmax_y"]), g["softmax_dx"], rtol=1e-5, atol=1e-9)
        np.testing.assert_allclose(naive_rms_norm(g["x"], g["w"], offset=off, eps=eps)[0], g["rms_y"], rtol=1e-6, atol=1e-6)
        dx, dw = naive_rms_norm_grads(g["dy"], g["x"], g["w"], offset=off, eps=eps)
        np.testing.assert_allclose(dx, g["rms_dx"], rtol=1e-6, atol=1e-6); np.testing.assert_allclose(dw, g["rms_dw"], rtol=1e-6, atol=1e-5)
        np.testing.assert_allclose(naive_layer_norm(g["x"], g["w"], g["b"], eps=eps)[0], g["ln_y"], rtol=1e-6, atol=1e-6)
        dx, dw, db = naive_layer_norm_grads(g["dy"], g["x"], g["w"], eps=eps)
        np.testing.assert_allclose(dx, g["ln_dx"], rtol=1e-6, atol=1e-6); np.testing.assert_allclose(db, g["ln_db"], rtol=1e-6, atol=1e-5)
