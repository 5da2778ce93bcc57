"""dev: randomized cross-check of the one-wave-per-SIMD backward (csrc/fa_bwd_w64.hpp, `bwd_w64=1`) against the 32-row kernels of
csrc/fa_bwd.hpp (`bwd_w64=0`) on the same residuals -- no oracle, so it is cheap enough for hundreds of shapes, including the corners
the structured tests may miss (tiny / ragged lengths, GQA ratios, QL != KL under the causal mask, every padding kind, E = 256).
Also repeats every new-form launch once and requires bitwise equality.   usage: fuzz_bw64.py [n_cases] [seed]"""
import os, sys
os.environ.setdefault("NNOP_DEBUG_HOOKS", "1")          # unlock the kernel-form hook nnop_debug_set (csrc/nnop_debug.h)
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
from util import make_inputs

pkg = ge.load_package()
dev = torch.device("cuda:0")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
bad = 0
for i in range(n_cases):
    E = int(rng.choice([64, 64, 128, 128, 256]))
    dt = str(rng.choice(["bf16", "f16"]))
    KH = int(rng.choice([1, 2, 3]))
    QH = KH * int(rng.choice([1, 1, 2, 4]))
    B = int(rng.integers(1, 4))
    lmax = 900 if E == 256 else 2600
    QL = int(rng.choice([int(rng.integers(1, 70)), int(rng.integers(1, lmax)), 32 * int(rng.integers(1, 40))]))
    KL = QL if rng.random() < 0.4 else int(rng.choice([int(rng.integers(1, 70)), int(rng.integers(1, lmax)), 64 * int(rng.integers(1, 20))]))
    causal = bool(rng.random() < 0.5)
    pad = [None, None, "lens", "random", "ref"][int(rng.integers(0, 5))]
    if pad == "ref" and KL < 12:
        pad = None
    d = make_inputs(7000 + i, B, QH, KH, QL, KL, E, dt, dev, pad=pad)
    o, ms, ls = pkg._flash_attention(d["q"], d["k"], d["v"], None, causal=causal, kpad_mask=d["mask"])

    def bwd(which):
        pkg._lib.debug_set("bwd_w64", which)
        g = pkg.grad_flash_attention(d["do"], o, ms, ls, d["q"], d["k"], d["v"], None, causal=causal, kpad_mask=d["mask"])
        torch.cuda.synchronize()
        return g[:3]

    new, new2, old = bwd(1), bwd(1), bwd(0)
    tol = 1.6e-2 if dt == "bf16" else 2e-3
    msg = []
    vmax = float(torch.nan_to_num(old[2].float()).abs().max())
    for name, a, a2, b in zip(("dq", "dk", "dv"), new, new2, old):
        if not torch.equal(a, a2):
            msg.append(f"{name}: not reproducible")
        if not bool(torch.isfinite(a.float()).all()) and bool(torch.isfinite(b.float()).all()):
            msg.append(f"{name}: non-finite")
        scale = float(torch.nan_to_num(b.float()).abs().max())
        err = float(torch.nan_to_num(a.float() - b.float()).abs().max())
        # (a single visible key: dS cancels, and dq / dk are rounding noise ~1e-6 in either form -- hence the floor taken from dv)
        if err > tol * max(scale, 1e-2 * vmax):
            msg.append(f"{name}: max diff {err:.3e} vs max {scale:.3e}")
    tag = f"{i}: {dt} E{E} B{B} H{QH}/{KH} L{QL}x{KL} causal={int(causal)} pad={pad}"
    if msg:
        bad += 1
        print("FAIL", tag, "; ".join(msg), flush=True)
    elif i % 25 == 0:
        print("ok  ", tag, flush=True)
pkg._lib.debug_set("bwd_w64", -1)
print(f"{n_cases} cases, {bad} failures", flush=True)
sys.exit(1 if bad else 0)
