"""dev: localise errors of the one-wave-per-SIMD backward (csrc/fa_bwd_w64.hpp) with probe cotangents.
dO = one-hot at (q0, e0)  ->  dV[key][e0] = P[q0][key]: the kernel's own P, read out through its dV product.
usage: bw64_debug.py [E] [QL] [KL] [causal]"""
import os, sys
os.environ.setdefault("NNOP_DEBUG_HOOKS", "1")          # unlock the kernel-form hook nnop_debug_set (csrc/nnop_debug.h)
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
from util import make_inputs, oracle_bwd

pkg = ge.load_package()
dev = torch.device("cuda:0")
E = int(sys.argv[1]) if len(sys.argv) > 1 else 64
QL = int(sys.argv[2]) if len(sys.argv) > 2 else 256
KL = int(sys.argv[3]) if len(sys.argv) > 3 else 256
causal = bool(int(sys.argv[4])) if len(sys.argv) > 4 else False
dt = "bf16"
d = make_inputs(5, 1, 1, 1, QL, KL, E, dt, dev)
o, ms, ls = pkg._flash_attention(d["q"], d["k"], d["v"], None, causal=causal, kpad_mask=None)


def bwd(which, do):
    pkg._lib.debug_set("bwd_w64", which)
    g = pkg.grad_flash_attention(do, o, ms, ls, d["q"], d["k"], d["v"], None, causal=causal, kpad_mask=None)
    torch.cuda.synchronize()
    return [x.float().cpu().numpy()[0, 0] for x in g[:3]]


def blocks(name, got, ref, rb, cb):
    err = np.abs(got - ref)
    R, Cn = err.shape
    print(f"{name}: max err {err.max():.3e} max|ref| {np.abs(ref).max():.3e}; per [{rb}-row block] x [{cb}-col block] max err:")
    for r0 in range(0, R, rb):
        print("   rows %4d.. " % r0 + " ".join(f"{err[r0:r0 + rb, c0:c0 + cb].max():9.2e}" for c0 in range(0, Cn, cb)))


# 1. full gradients, new form vs old form
for which in (2, 3):
    new = bwd(which, d["do"])
    old = bwd(0, d["do"])
    for n, a, b in zip(("dq", "dk", "dv"), new, old):
        if (which == 2 and n == "dq") or (which == 3 and n != "dq"):
            continue
        blocks(f"[bwd_w64={which}] {n} vs 32-row form", a, b, 32, 32)

# 2. P read out through dV
q64, k64 = d["q"].double().cpu().numpy()[0, 0], d["k"].double().cpu().numpy()[0, 0]
s = q64 @ k64.T / np.sqrt(E)
if causal:
    s = np.where(np.arange(QL)[:, None] >= np.arange(KL)[None, :], s, -np.inf)
P = np.exp(s - s.max(1, keepdims=True))
P /= P.sum(1, keepdims=True)
for q0 in (0, 5, 37, QL - 1):
    do = torch.zeros_like(d["do"])
    do[0, 0, q0, 3] = 1.0
    dv = bwd(2, do)[2]
    got = dv[:, 3]
    ref = P[q0]
    err = np.abs(got - ref)
    other = np.abs(np.delete(dv, 3, axis=1)).max()
    print(f"P[q0={q0}] via dV: max err {err.max():.3e} (max P {ref.max():.3e}); leak into other columns {other:.3e}; "
          f"got/ref at argmax: {got[ref.argmax()]:.4f} / {ref.max():.4f}; sum got {got.sum():.4f}")
    if err.max() > 2e-2 * ref.max():
        print("    got[:8]", np.round(got[:8], 4), " ref[:8]", np.round(ref[:8], 4))
        nz = np.nonzero(np.abs(dv) > 1e-6)
        print("    nonzero dv entries: keys", np.unique(nz[0])[:16], "cols", np.unique(nz[1])[:16])
pkg._lib.debug_set("bwd_w64", -1)
