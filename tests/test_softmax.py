"""CPU tests for the online-softmax row: the oracle's naive formula (test/softmax_tests.jl:6-10)
agrees with a restatement of the reference kernel's own (m, d) evaluation order (src/softmax.jl:1-58,
src/groupreduce.jl:27-37), the pullback formula matches finite differences, and the C ABI validates descriptors."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle.naive_softmax import naive_softmax, naive_softmax_grad, online_softmax_md, softmax_bytes


@pytest.mark.parametrize("N", [32, 33, 63, 255, 256, 511, 512, 513, 1024])       # test/softmax_tests.jl:14-16
def test_md_order_equals_naive(N):
    x = np.random.default_rng(N).random((4, N)).astype(np.float32)
    np.testing.assert_allclose(online_softmax_md(x), naive_softmax(x), rtol=1e-13, atol=0)


def test_md_guard_for_minus_inf():
    x = np.full((2, 300), -np.inf); x[0, 7] = 1.0; x[0, 290] = 2.0
    y = online_softmax_md(x)
    assert np.isclose(y[0].sum(), 1.0) and y[0, 290] > y[0, 7] > 0 and (y[0, :7] == 0).all()
    assert np.isnan(y[1]).all()                        # an all -Inf column: 0/0, as the naive formula gives
    with np.errstate(invalid="ignore"):
        assert np.isnan(naive_softmax(x)[1]).all()


def test_grad_matches_finite_differences():
    rng = np.random.default_rng(0)
    x, dy = rng.standard_normal((3, 37)), rng.standard_normal((3, 37))
    y = naive_softmax(x)
    dx = naive_softmax_grad(dy, y)
    eps = 1e-6
    for (b, e) in [(0, 0), (1, 17), (2, 36)]:
        xp = x.copy(); xp[b, e] += eps
        xm = x.copy(); xm[b, e] -= eps
        fd = ((naive_softmax(xp) - naive_softmax(xm)) * dy).sum() / (2 * eps)
        assert abs(fd - dx[b, e]) < 1e-8
    # test/softmax_tests.jl:22-28: the gradient of sum(softmax(x)) is 0
    np.testing.assert_allclose(naive_softmax_grad(np.ones_like(y), y), 0, atol=1e-15)


def test_softmax_bytes():
    assert softmax_bytes(1024, 4, 4) == 2 * 4096 * 4 and softmax_bytes(1024, 4, 2, bwd=True) == 3 * 4096 * 2


@pytest.mark.parametrize("kw,status", [
    (dict(dtype=5), "NNOP_ERR_DTYPE"),
    (dict(n=0), "NNOP_ERR_SHAPE"),
    (dict(batch=0), "NNOP_ERR_SHAPE"),
    (dict(), "NNOP_ERR_NULL"),
])
def test_softmax_descriptor_validation(pkg, kw, status):
    lib = pkg._lib.load()
    base = dict(dtype=0, n=256, batch=4)
    base.update(kw)
    d = pkg._lib.SoftmaxDesc(**base)
    null = C.c_void_p(0)
    assert lib.nnop_online_softmax(C.byref(d), null, null, null) == getattr(pkg._lib, status)
    assert lib.nnop_online_softmax_bwd(C.byref(d), null, null, null, null) == getattr(pkg._lib, status)
    assert lib.nnop_online_softmax(None, null, null, null) == pkg._lib.NNOP_ERR_NULL


def test_softmax_host_refuses_cpu_and_non_matrix(pkg):
    with pytest.raises(pkg.NNopError, match="GPU-only"):
        pkg.online_softmax(torch.ones(4, 32))
    with pytest.raises(TypeError):
        pkg.online_softmax(torch.ones(4, 32, 2))
