"""GPU parity tests for RMSNorm / LayerNorm through the C ABI, against oracle/naive_norms.py (fp64 on the rounded inputs).

Grids of the reference's tests (test/rmsnorm_tests.jl:11-33, test/layernorm_tests.jl:13-35): Float32, emb in {15, 255,
256, 257, 511, 512, 513, 1024}, n in {1, 2, 4, 15, 16, 17, 23, 25}, offset in {0, 1}, rand inputs, gradient of sum(y)
with atol = rtol = 1e-6 -- widened with 16-bit types, fp32 / T weights, random cotangents, every register shape and the
generic path, many rows (multi-partial dw/db), full-size properties."""
import numpy as np
import pytest
import torch

from oracle.naive_norms import naive_layer_norm, naive_layer_norm_grads, naive_rms_norm, naive_rms_norm_grads
from util import TORCH_DT, jl_isapprox

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
EMBS = [15, 255, 256, 257, 511, 512, 513, 1024]
NS = [1, 2, 4, 15, 16, 17, 23, 25]
# outputs are O(1): fp32 a few ulp; 16-bit one rounding of the result
YTOL = {"f32": dict(rtol=2e-6, atol=2e-6), "f16": dict(rtol=2e-3, atol=2e-3), "bf16": dict(rtol=1.6e-2, atol=1.6e-2)}


def _np(t):
    return t.detach().to(torch.float64).cpu().numpy()


def _t(a, dt):
    return torch.tensor(np.asarray(a, np.float32)).to(TORCH_DT[dt]).to(DEV)


def _close(got, ref, dt, scale=1.0, k=1.0):
    tol = YTOL[dt]
    np.testing.assert_allclose(_np(got), ref, rtol=tol["rtol"] * k, atol=tol["atol"] * k * scale)


@pytest.mark.parametrize("emb", EMBS)
@pytest.mark.parametrize("n", NS)
@pytest.mark.parametrize("offset", [0.0, 1.0])
def test_rms_norm_reference_grid_f32(pkg, emb, n, offset):
    rng = np.random.default_rng(emb * 100 + n)
    x, w = _t(rng.random((n, emb)), "f32").requires_grad_(True), _t(rng.random(emb), "f32").requires_grad_(True)
    y = pkg.rms_norm(x, w, offset=offset)
    ref, rstd = naive_rms_norm(_np(x), _np(w), offset=offset)
    np.testing.assert_allclose(_np(y), ref, rtol=2e-6, atol=2e-6)
    y.sum().backward()                                              # test/rmsnorm_tests.jl:25-32
    dx, dw = naive_rms_norm_grads(np.ones((n, emb)), _np(x), _np(w), offset=offset)
    # the reference's criterion (Julia isapprox on arrays is norm-wise), then element-wise at a few ulp of the O(4)
    # terms that cancel in dx / of the n-term sums in dw
    assert jl_isapprox(_np(y), ref) and jl_isapprox(_np(x.grad), dx, atol=1e-6, rtol=1e-6)
    assert jl_isapprox(_np(w.grad), dw, atol=1e-6, rtol=1e-6)
    np.testing.assert_allclose(_np(x.grad), dx, rtol=1e-6, atol=4e-6)
    np.testing.assert_allclose(_np(w.grad), dw, rtol=1e-6, atol=2e-6 * max(1.0, n / 4))
    _, rms = pkg._rms_norm(x.detach(), w.detach(), offset=offset)
    np.testing.assert_allclose(_np(rms), rstd, rtol=1e-6)


@pytest.mark.parametrize("emb", EMBS)
@pytest.mark.parametrize("n", NS)
def test_layer_norm_reference_grid_f32(pkg, emb, n):
    rng = np.random.default_rng(emb * 100 + n + 7)
    x = _t(rng.random((n, emb)), "f32").requires_grad_(True)
    w = _t(rng.random(emb), "f32").requires_grad_(True)
    b = _t(rng.random(emb), "f32").requires_grad_(True)
    y = pkg.layer_norm(x, w, b)
    ref, mu, rstd = naive_layer_norm(_np(x), _np(w), _np(b))
    np.testing.assert_allclose(_np(y), ref, rtol=5e-6, atol=5e-6)
    y.sum().backward()                                              # test/layernorm_tests.jl:26-34
    dx, dw, db = naive_layer_norm_grads(np.ones((n, emb)), _np(x), _np(w))
    assert jl_isapprox(_np(y), ref)
    assert jl_isapprox(_np(w.grad), dw, atol=1e-6, rtol=1e-6) and jl_isapprox(_np(b.grad), db, atol=1e-6, rtol=1e-6)
    # dx of sum(layer_norm(x)) is ~0 (a difference of O(1) terms scaled by rstd ~ 3.5): the reference's norm-wise
    # rtol is vacuous there, its atol = 1e-6 on the norm is not reachable in fp32 beyond ~1e3 elements by any
    # evaluation order; element-wise a few ulp of the cancelling terms
    assert np.linalg.norm(_np(x.grad) - dx) <= 1e-6 * np.sqrt(n * emb) * 2
    np.testing.assert_allclose(_np(x.grad), dx, rtol=1e-5, atol=5e-6)
    np.testing.assert_allclose(_np(w.grad), dw, rtol=1e-5, atol=2e-5 * max(1.0, n / 4))
    np.testing.assert_allclose(_np(b.grad), db, rtol=1e-6, atol=1e-6)
    _, m, s = pkg._layer_norm(x.detach(), w.detach(), b.detach())
    np.testing.assert_allclose(_np(m), mu, rtol=1e-6)
    np.testing.assert_allclose(_np(s), rstd, rtol=2e-6)


SHAPES = [(1, 8), (3, 40), (37, 264), (64, 520), (130, 1000), (130, 2048), (70, 2056), (64, 4096), (33, 5000),
          (40, 8192), (9, 16384), (5, 16392), (3, 32768), (2, 65536), (2, 70000), (5000, 128), (4100, 768)]


@pytest.mark.parametrize("dt,wdt", [("f32", "f32"), ("f16", "f16"), ("f16", "f32"), ("bf16", "bf16"), ("bf16", "f32")])
@pytest.mark.parametrize("n,emb", SHAPES)
def test_shapes_and_dtypes(pkg, dt, wdt, n, emb):
    rng = np.random.default_rng(n * 7 + emb)
    x = _t(rng.standard_normal((n, emb)) * 2.0 + 0.5, dt)
    w, b = _t(rng.standard_normal(emb), wdt), _t(rng.standard_normal(emb), wdt)
    dy = _t(rng.standard_normal((n, emb)), dt)
    # ---- RMSNorm
    y, rms = pkg._rms_norm(x, w, offset=0.5, eps=1e-5)
    ref, rstd = naive_rms_norm(_np(x), _np(w), offset=0.5, eps=1e-5)
    _close(y, ref, dt, scale=np.abs(ref).max())
    np.testing.assert_allclose(_np(rms), rstd, rtol=2e-6)
    dx, dw = pkg.grad_rms_norm(dy, rms, x, w, offset=0.5)
    rdx, rdw = naive_rms_norm_grads(_np(dy), _np(x), _np(w), offset=0.5, eps=1e-5)
    _close(dx, rdx, dt, scale=np.abs(rdx).max())
    assert dw.dtype == torch.float32                                # src/rms_norm.jl:146
    np.testing.assert_allclose(_np(dw), rdw, rtol=1e-4, atol=1e-5 * np.abs(rdw).max() + 1e-6 * n)
    # ---- LayerNorm
    y, mu, sg = pkg._layer_norm(x, w, b, eps=1e-5)
    ref, rmu, rsg = naive_layer_norm(_np(x), _np(w), _np(b), eps=1e-5)
    _close(y, ref, dt, scale=np.abs(ref).max())
    np.testing.assert_allclose(_np(mu), rmu, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(_np(sg), rsg, rtol=1e-5)
    dx, dw, db = pkg.grad_layer_norm(dy, mu, sg, x, w, b)
    rdx, rdw, rdb = naive_layer_norm_grads(_np(dy), _np(x), _np(w), eps=1e-5)
    _close(dx, rdx, dt, scale=np.abs(rdx).max(), k=2.0)
    assert dw.dtype == w.dtype and db.dtype == w.dtype              # src/layer_norm.jl:179-180
    wk = 1.0 if wdt == "f32" else 1.0
    tolw = dict(rtol=1e-4, atol=1e-5 * np.abs(rdw).max() + 1e-6 * n) if wdt == "f32" else \
        dict(rtol=YTOL[wdt]["rtol"], atol=YTOL[wdt]["atol"] * max(1.0, np.abs(rdw).max()))
    np.testing.assert_allclose(_np(dw), rdw, **tolw)
    np.testing.assert_allclose(_np(db), rdb, **tolw)


@pytest.mark.parametrize("op", ["rms", "ln"])
def test_many_rows_partials_and_determinism(pkg, op):
    """n large enough for the maximum number of partial rows; dw/db bitwise reproducible (no atomics)."""
    n, emb = 70001, 512
    rng = np.random.default_rng(5)
    x, dy = _t(rng.standard_normal((n, emb)), "bf16"), _t(rng.standard_normal((n, emb)), "bf16")
    w, b = _t(rng.standard_normal(emb), "f32"), _t(rng.standard_normal(emb), "f32")
    if op == "rms":
        y, rms = pkg._rms_norm(x, w)
        outs = [pkg.grad_rms_norm(dy, rms, x, w) for _ in range(2)]
        ref = naive_rms_norm_grads(_np(dy), _np(x), _np(w))
    else:
        y, mu, sg = pkg._layer_norm(x, w, b)
        outs = [pkg.grad_layer_norm(dy, mu, sg, x, w, b) for _ in range(2)]
        ref = naive_layer_norm_grads(_np(dy), _np(x), _np(w))
    for a, c in zip(*outs):
        assert torch.equal(a, c)
    _close(outs[0][0], ref[0], "bf16", scale=np.abs(ref[0]).max(), k=2.0)
    for got, r in zip(outs[0][1:], ref[1:]):
        np.testing.assert_allclose(_np(got), r, rtol=2e-4, atol=2e-4 * np.abs(r).max())


def test_properties_full_size(pkg):
    """Size-independent properties at 64 Mi elements: normalised rows have unit RMS / zero mean and unit variance;
    scale invariance of LayerNorm; the pullbacks are orthogonal to the directions the forward ignores."""
    n, emb = 16384, 4096
    g = torch.Generator(device=DEV).manual_seed(0)
    x = torch.randn(n, emb, device=DEV, generator=g) * 3 + 1
    ones, zeros = torch.ones(emb, device=DEV), torch.zeros(emb, device=DEV)
    y = pkg.rms_norm(x, ones, eps=0.0)
    assert torch.allclose((y * y).mean(-1), torch.ones(n, device=DEV), atol=1e-5)
    z, mu, sg = pkg._layer_norm(x, ones, zeros, eps=0.0)
    assert z.mean(-1).abs().max() < 1e-5 and torch.allclose((z * z).mean(-1), torch.ones(n, device=DEV), atol=1e-4)
    z2 = pkg.layer_norm(x * 7.0 + 3.0, ones, zeros, eps=0.0)
    assert torch.allclose(z2, z, atol=2e-5)
    dy = torch.randn(n, emb, device=DEV, generator=g)
    dx, dw, db = pkg.grad_layer_norm(dy, mu, sg, x, ones, zeros)
    # LayerNorm ignores shifts and scalings of a row: dx is orthogonal to 1 and to (x - mu)
    assert dx.sum(-1).abs().max() < 2e-3 and (dx * (x - mu[:, None])).sum(-1).abs().max() < 5e-2
    assert torch.allclose(db, dy.sum(0), rtol=1e-4, atol=1e-3)
    _, rms = pkg._rms_norm(x, ones, eps=0.0)
    dxr, dwr = pkg.grad_rms_norm(dy, rms, x, ones)
    assert (dxr * x).sum(-1).abs().max() < 5e-2                      # RMSNorm ignores scalings of a row


def test_autograd_matches_torch_double(pkg):
    rng = np.random.default_rng(3)
    x = _t(rng.standard_normal((33, 777)), "f32").requires_grad_(True)
    w = _t(rng.standard_normal(777), "f32").requires_grad_(True)
    b = _t(rng.standard_normal(777), "f32").requires_grad_(True)
    g = _t(rng.standard_normal((33, 777)), "f32")
    (pkg.layer_norm(x, w, b, eps=1e-5) * g).sum().backward()
    xd, wd, bd = (t.detach().double().requires_grad_(True) for t in (x, w, b))
    (torch.nn.functional.layer_norm(xd, (777,), wd, bd, eps=1e-5) * g.double()).sum().backward()
    for a, c in ((x.grad, xd.grad), (w.grad, wd.grad), (b.grad, bd.grad)):
        np.testing.assert_allclose(_np(a), _np(c), rtol=2e-4, atol=2e-5)
    x.grad = None; w.grad = None
    (pkg.rms_norm(x, w, eps=1e-5) * g).sum().backward()
    xd.grad = None; wd.grad = None
    (torch.nn.functional.rms_norm(xd, (777,), wd, eps=1e-5) * g.double()).sum().backward()
    for a, c in ((x.grad, xd.grad), (w.grad, wd.grad)):
        np.testing.assert_allclose(_np(a), _np(c), rtol=2e-4, atol=2e-5)


def test_host_checks(pkg):
    x, w = _t(np.ones((4, 64)), "f32"), _t(np.ones(64), "f32")
    with pytest.raises(pkg.NNopError, match="AssertionError"):
        pkg.rms_norm(x, w[:63])
    with pytest.raises(TypeError):
        pkg.rms_norm(x, w.half())                       # fp32 x with fp16 w
    with pytest.raises(TypeError):
        pkg.layer_norm(x.half(), w.half(), w)           # w, b of different dtypes
    with pytest.raises(TypeError):
        pkg.rms_norm(x[0], w)


import glob as _glob
import os as _os

_ROW_GOLDEN = sorted(_glob.glob(_os.path.join(_os.path.dirname(__file__), "golden", "rows_*.npz")))


@pytest.mark.parametrize("dt", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("path", _ROW_GOLDEN, ids=[_os.path.basename(p)[5:-4] for p in _ROW_GOLDEN])
def test_row_golden(pkg, path, dt):
    """Committed fixtures (bf16-exact inputs, fp64-oracle outputs): softmax, RMSNorm, LayerNorm, forward + pullback."""
    g = np.load(path)
    off, eps = float(g["offset"]), float(g["eps"])
    x, dy = _t(g["x"], dt), _t(g["dy"], dt)
    w, b = _t(g["w"], "f32"), _t(g["b"], "f32")
    f64 = lambda n: g[n].astype(np.float64)
    y = pkg.online_softmax(x)
    np.testing.assert_allclose(_np(y), f64("softmax_y"), rtol=YTOL[dt]["rtol"] * 1.5, atol=1e-7)
    if dt == "f32":       # the fixture's pullback is taken at the fp32 y; 16-bit y differs by its own rounding
        np.testing.assert_allclose(_np(pkg.grad_online_softmax(dy, _t(g["softmax_y"], dt))), f64("softmax_dx"),
                                   rtol=1e-5, atol=1e-7)
    yr, rms = pkg._rms_norm(x, w, offset=off, eps=eps)
    _close(yr, f64("rms_y"), dt, scale=np.abs(f64("rms_y")).max())
    np.testing.assert_allclose(_np(rms), f64("rms_rstd"), rtol=2e-6)
    dx, dw = pkg.grad_rms_norm(dy, rms, x, w, offset=off)
    _close(dx, f64("rms_dx"), dt, scale=np.abs(f64("rms_dx")).max())
    np.testing.assert_allclose(_np(dw), f64("rms_dw"), rtol=1e-4, atol=1e-5 * np.abs(f64("rms_dw")).max())
    yl, mu, sg = pkg._layer_norm(x, w, b, eps=eps)
    _close(yl, f64("ln_y"), dt, scale=np.abs(f64("ln_y")).max())
    dx, dw, db = pkg.grad_layer_norm(dy, mu, sg, x, w, b)
    _close(dx, f64("ln_dx"), dt, scale=np.abs(f64("ln_dx")).max(), k=2.0)
    np.testing.assert_allclose(_np(dw), f64("ln_dw"), rtol=1e-4, atol=1e-5 * np.abs(f64("ln_dw")).max())
    np.testing.assert_allclose(_np(db), f64("ln_db"), rtol=1e-4, atol=1e-5 * np.abs(f64("ln_db")).max())
