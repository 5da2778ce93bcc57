"""GPU parity tests for online_softmax through the C ABI, against oracle/naive_softmax.py (fp64 on the rounded inputs).

Grid of the reference's test (test/softmax_tests.jl:12-29): Float32, seq_len in {32, 33, 63, 255, 256, 511, 512, 513,
1024}, 4 columns, rand inputs -- widened with 16-bit types, N(0, 3) inputs, every register shape / the generic path,
long rows, -Inf entries, full-size properties."""
import numpy as np
import pytest
import torch

from oracle.naive_softmax import naive_softmax, naive_softmax_grad
from util import TORCH_DT

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
# y in [0, 1]: fp32 a few ulp of exp; 16-bit one rounding of y (relative) -- plus an absolute floor of one ulp of the
# smallest normal-range outputs
TOL = {"f32": dict(rtol=2e-6, atol=1e-9), "f16": dict(rtol=1.5e-3, atol=1e-7), "bf16": dict(rtol=1e-2, atol=1e-9)}
GTOL = {"f32": dict(rtol=1e-5, atol=2e-7), "f16": dict(rtol=4e-3, atol=2e-4), "bf16": dict(rtol=3e-2, atol=2e-3)}


def _np(t):
    return t.detach().to(torch.float64).cpu().numpy()


def _x(seed, batch, N, dt, kind="rand"):
    rng = np.random.default_rng(seed)
    x = rng.random((batch, N)) if kind == "rand" else 3.0 * rng.standard_normal((batch, N))
    return torch.tensor(x.astype(np.float32)).to(TORCH_DT[dt]).to(DEV)


@pytest.mark.parametrize("N", [32, 33, 63, 255, 256, 511, 512, 513, 1024])
def test_reference_grid_f32(pkg, N):
    """test/softmax_tests.jl:12-29 (forward, and the gradient of sum(softmax(x)) with atol = rtol = 1e-6)."""
    x = _x(N, 4, N, "f32").requires_grad_(True)
    y = pkg.online_softmax(x)
    np.testing.assert_allclose(_np(y), naive_softmax(_np(x)), **TOL["f32"])
    y.sum().backward()
    np.testing.assert_allclose(_np(x.grad), naive_softmax_grad(np.ones(y.shape), naive_softmax(_np(x))),
                               rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("dt", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("N", [1, 2, 8, 31, 100, 256, 264, 520, 1000, 2048, 2056, 4096, 5000, 8192, 16384, 16392, 32768,
                               40000, 131072])
def test_shapes_and_dtypes(pkg, dt, N):
    batch = 7 if N <= 8192 else 3
    x = _x(N, batch, N, dt, "normal")
    y = pkg.online_softmax(x)
    ref = naive_softmax(_np(x))
    np.testing.assert_allclose(_np(y), ref, **TOL[dt])
    dy = _x(N + 1, batch, N, dt, "normal")
    dx = pkg.grad_online_softmax(dy, y)
    # pullback parity on the kernel's own (rounded) y, as ∇online_softmax receives it
    gref = naive_softmax_grad(_np(dy), _np(y))
    np.testing.assert_allclose(_np(dx), gref, rtol=GTOL[dt]["rtol"], atol=GTOL[dt]["atol"] * max(1.0, np.abs(gref).max() * 50))


def test_minus_inf_entries_and_large_magnitudes(pkg):
    x = _x(3, 5, 1024, "f32", "normal") * 30.0
    x[0, 100:900] = -float("inf")
    x[1, :1023] = -float("inf")                       # a single finite entry
    x[2] = 1e4 * torch.sign(x[2])                     # exp would overflow without the max subtraction
    y = pkg.online_softmax(x)
    ref = naive_softmax(_np(x))
    np.testing.assert_allclose(_np(y), ref, rtol=1e-5, atol=1e-12)
    assert (y[0, 100:900] == 0).all() and y[1, 1023] == 1.0
    x[3] = -float("inf")                              # all -Inf: NaN, as the reference's 0 * inv(0)
    assert torch.isnan(pkg.online_softmax(x)[3]).all() and not torch.isnan(pkg.online_softmax(x)[4]).any()


def test_in_place_rows_sum_to_one_and_shift_invariance_full_size(pkg):
    """Size-independent properties at 64 Mi elements: rows sum to 1, softmax(x + c) == softmax(x), argmax kept,
    bitwise reproducible, in-place == out-of-place."""
    g = torch.Generator(device=DEV).manual_seed(0)
    x = torch.randn(16384, 4096, device=DEV, generator=g)
    y = pkg.online_softmax(x)
    assert torch.allclose(y.sum(-1), torch.ones(16384, device=DEV), atol=1e-5)
    assert torch.equal(y.argmax(-1), x.argmax(-1))
    assert torch.equal(pkg.online_softmax(x), y)
    y2 = pkg.online_softmax(x + 3.0)
    assert torch.allclose(y2, y, rtol=1e-4, atol=1e-9)
    xc = x.clone()
    pkg.online_softmax_into(xc, xc)
    assert torch.equal(xc, y)
    # the pullback is orthogonal to constants: sum(dx) == 0 per row
    dx = pkg.grad_online_softmax(torch.randn_like(y), y)
    assert dx.sum(-1).abs().max() < 1e-5


def test_autograd_matches_torch_double(pkg):
    x = _x(8, 6, 777, "f32", "normal").requires_grad_(True)
    w = torch.randn(6, 777, device=DEV)
    (pkg.online_softmax(x) * w).sum().backward()
    xd = x.detach().double().requires_grad_(True)
    (torch.softmax(xd, -1) * w.double()).sum().backward()
    np.testing.assert_allclose(_np(x.grad), _np(xd.grad), rtol=2e-5, atol=1e-7)


def test_host_checks(pkg):
    x = _x(1, 4, 64, "f32")
    with pytest.raises(TypeError):
        pkg.grad_online_softmax(x.half(), x)
    with pytest.raises(TypeError):
        pkg.online_softmax(x.double())
    with pytest.raises(pkg.NNopError):
        pkg.online_softmax(x[:, :0])
