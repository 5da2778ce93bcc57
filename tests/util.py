"""Shared helpers for the parity tests (test infrastructure)."""
import json
import os

import numpy as np
import torch

from oracle.naive_attention import naive_attention, naive_attention_grads

TORCH_DT = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}
# Tolerances: BASELINE.json north_star states the ceiling (fp32 rtol <= 1e-4, fp16/bf16 rtol <= 1e-2), element-wise
# against the fp64 oracle evaluated on the ROUNDED inputs; atol is scaled to the tensor's max magnitude (SURVEY.md
# section 8(c)).  The 16-bit values are CALIBRATED: at most ~2x the worst error the whole GPU suite measured (9063
# comparisons, profiles/r02/parity_errors.json; worst error / tolerance there: bf16 0.49, f16 0.45, f32 0.46), so a
# regression that doubles the rounding error of any kernel fails.  Round 3 (new backward kernels, exact-scale forward; 10202
# comparisons, profiles/r03/parity_errors.json): worst error / tolerance bf16 0.65 (dq), f16 0.44, f32 0.75 (dk) -- unchanged tolerances.
RTOL = {"f32": 1e-4, "f16": 4e-3, "bf16": 8e-3}
ATOL_FRAC = {"f32": 1e-5, "f16": 8e-4, "bf16": 8e-3}
# gradients: the residuals ms, ls they are computed from are stored in T (src/attention.jl:128-129).  Round 4: the 16-bit gates are INSIDE
# north_star's rtol <= 1e-2 (bf16: 8e-3 x 1.25 = 1.0e-2, fp16: 4e-3 x 2 = 8e-3; round 3 enforced 1.28e-2 for bf16 gradients)
GRAD_SCALE = {"f32": 1.0, "f16": 2.0, "bf16": 1.25}


def make_inputs(seed, B, QH, KH, QL, KL, E, dt, dev, *, pair=False, pad=None, need_do=True):
    """N(0,1) inputs generated in fp32 with numpy default_rng(seed), cast to the test dtype."""
    rng = np.random.default_rng(seed)
    tdt = TORCH_DT[dt]
    mk = lambda *s: torch.tensor(rng.standard_normal(s).astype(np.float32)).to(tdt).to(dev)
    d = dict(q=mk(B, QH, QL, E), k=mk(B, KH, KL, E), v=mk(B, KH, KL, E))
    d["do"] = mk(B, QH, QL, E) if need_do else None
    d["pair"] = mk(B, KL, QL, QH) if pair else None
    d["mask"] = None
    if pad is not None:
        m = np.ones((B, KL), dtype=bool)
        if pad == "ref":                    # test/attention_tests.jl:27-28: last 11 keys of last batch
            m[-1, -11:] = False
        elif pad == "lens":                 # variable sequence lengths (prefix masks)
            lens = rng.integers(1, KL + 1, size=B)
            m = np.arange(KL)[None, :] < lens[:, None]
        elif pad == "random":
            m = rng.random((B, KL)) < 0.7
            m[:, 0] = True
        d["mask"] = torch.tensor(m).to(dev)
    return d


def to64(t):
    return None if t is None else t.detach().to(torch.float64).cpu().numpy()


def oracle_fwd(d, causal):
    mask = None if d["mask"] is None else d["mask"].cpu().numpy()
    return naive_attention(to64(d["q"]), to64(d["k"]), to64(d["v"]), to64(d["pair"]), causal=causal,
                           kpad_mask=mask, return_stats=True)


def oracle_bwd(d, causal):
    mask = None if d["mask"] is None else d["mask"].cpu().numpy()
    return naive_attention_grads(to64(d["q"]), to64(d["k"]), to64(d["v"]), to64(d["do"]), to64(d["pair"]),
                                 causal=causal, kpad_mask=mask)


# absolute floor for results dominated by cancellation of O(1) terms (e.g. dS = P (dP - delta) == 0 exactly when a
# row sees a single key): a few rounding units of the O(1) operands, summed over up to a few hundred rows
ABS_FLOOR = {"f32": 1e-5, "f16": 5e-3, "bf16": 4e-2}


def record_error(kind, rec):
    """Calibration aid: with NNOP_TEST_ERRLOG=<file> every comparison appends its measured error (one JSON line), which
    is where the 16-bit tolerances above come from (profiles/r02/parity_errors.json)."""
    path = os.environ.get("NNOP_TEST_ERRLOG")
    if path:
        with open(path, "a") as f:
            f.write(json.dumps(dict(kind=kind, **rec)) + "\n")


def assert_close(name, got, ref, dt, scale=1.0, floor=False, kind="fwd"):
    """|got - ref| <= atol + rtol*|ref| element-wise, atol = ATOL_FRAC * max|ref|; plus the
    reference's own norm-wise check (isapprox atol=rtol=1e-3, test/attention_tests.jl:42-48) for f32.
    kind="grad": gradient tensors (16-bit: the residuals ms, ls they are computed from are stored in T)."""
    if kind == "grad" and scale == 1.0:
        scale = GRAD_SCALE[dt]
    g = to64(got) if isinstance(got, torch.Tensor) else np.asarray(got, np.float64)
    assert g.shape == ref.shape, f"{name}: shape {g.shape} vs {ref.shape}"
    both_nan = np.isnan(g) & np.isnan(ref)
    assert (np.isnan(g) == np.isnan(ref)).all(), f"{name}: NaN pattern differs"
    gz, rz = np.where(both_nan, 0.0, g), np.where(both_nan, 0.0, ref)
    mag = np.abs(rz).max() if rz.size else 0.0
    tol = scale * (ATOL_FRAC[dt] * mag + RTOL[dt] * np.abs(rz)) + (ABS_FLOOR[dt] if floor else 0.0)
    err = np.abs(gz - rz)
    bad = err > tol
    record_error("assert_close", dict(name=name.split(" ")[-1], dt=dt, rel_to_max=float(err.max() / max(mag, 1e-30)) if err.size else 0.0,
                                      worst_ratio=float((err / np.maximum(tol, 1e-300)).max()) if err.size else 0.0,
                                      scale=scale, floor=bool(floor)))
    assert not bad.any(), (f"{name} [{dt}]: {bad.sum()} / {bad.size} outside tolerance; "
                           f"max err {err.max():.3e} (max|ref| {mag:.3e})")
    if dt == "f32" and not floor:
        nrm = np.linalg.norm(gz - rz)
        assert nrm <= max(1e-3, 1e-3 * max(np.linalg.norm(gz), np.linalg.norm(rz))), f"{name}: norm-wise 1e-3"
    return float(err.max() / max(mag, 1e-30))


def jl_isapprox(got, ref, atol=0.0, rtol=None):
    """Julia's isapprox for ARRAYS, as the reference's tests use it: norm(x - y) <= max(atol, rtol*max(norm(x), norm(y)))
    -- a norm-wise criterion (Frobenius), not element-wise; default rtol = sqrt(eps(Float32)) when atol == 0."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    if rtol is None:
        rtol = float(np.sqrt(np.finfo(np.float32).eps)) if atol == 0.0 else 0.0
    return np.linalg.norm(got - ref) <= max(atol, rtol * max(np.linalg.norm(got), np.linalg.norm(ref)))
