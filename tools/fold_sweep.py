"""dev / evidence: where does the 64-row forward's DEFAULT scale handling (scale * log2(e) folded into Q, rounded to T once) leave the
standard parity tolerance?  Plants, for every query row, TWO near-tied dominant keys whose logits |s * scale| = M (natural units)
are built either DENSE (q parallel to k: the logit is spread over all E channels) or SPARSE (one outlier channel carries it --
the "massive activation" pattern of trained transformers), sweeps M, and prints max |o - oracle| / tolerance for the folded
default and for the exact-scale variant (tests/util.py tolerances; > 1 = outside).
usage: fold_sweep.py [dt] [E]"""
import os, sys
os.environ.setdefault("NNOP_DEBUG_HOOKS", "1")          # unlock the kernel-form hook nnop_debug_set (csrc/nnop_debug.h)
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
from util import ATOL_FRAC, RTOL, TORCH_DT
from oracle.naive_attention import naive_attention


def planted(M, E, L, structure, dt, seed=0):
    """q, k, v [1, 1, L, E] in T: background N(0, 0.3); for every query two keys (2i, 2i+1 mod L) with logit * scale ~ M, M - 0.5"""
    rng = np.random.default_rng(seed)
    q = 0.3 * rng.standard_normal((L, E)).astype(np.float32)
    k = 0.3 * rng.standard_normal((L, E)).astype(np.float32)
    v = rng.standard_normal((L, E)).astype(np.float32)
    s = M * np.sqrt(E)                                   # target raw logit q.k
    # The rounding of the folded Q is per QUERY element: keys that share their direction see the same error (it cancels in the
    # softmax).  The two tied keys therefore get ORTHOGONAL supports: the query carries both directions.
    a = np.sqrt(s)
    if structure == "dense":
        d1 = rng.standard_normal(E).astype(np.float32)
        d2 = rng.standard_normal(E).astype(np.float32)
        d1 /= np.linalg.norm(d1)
        d2 -= d1 * (d1 @ d2)
        d2 /= np.linalg.norm(d2)
    else:
        d1 = np.zeros(E, np.float32); d1[3] = 1.0
        d2 = np.zeros(E, np.float32); d2[E // 2 + 5] = 1.0
    q += a * (d1 + d2)                                   # every query carries both directions
    for j in range(0, L, 64):                            # one strong key pair per 64-key tile
        k[j] += a * d1
        k[j + 1] += (a - 0.5 * np.sqrt(E) / a) * d2
    tdt = TORCH_DT[dt]
    t = lambda x: torch.tensor(x).to(tdt)
    return t(q)[None, None], t(k)[None, None], t(v)[None, None]


def run(dt="bf16", E=64, L=1024, Ms=(2, 5, 8, 12, 16, 20, 30, 45, 60)):
    pkg = ge.load_package()
    dev = torch.device("cuda:0")
    rows = []
    for structure in ("dense", "sparse"):
        for M in Ms:
            q, k, v = (x.to(dev) for x in planted(M, E, L, structure, dt))
            ref = naive_attention(q.double().cpu().numpy(), k.double().cpu().numpy(), v.double().cpu().numpy(), None, causal=False)
            mag = np.abs(ref).max()
            smax = float((q.double() @ k.double().transpose(-1, -2)).abs().max()) / np.sqrt(E)
            out = {}
            for exact in (0, 1):
                pkg._lib.debug_set("fwd_w64", 1)
                pkg._lib.debug_set("fwd_exact_scale", exact)
                o, ms, ls = pkg._flash_attention(q, k, v, causal=False)
                torch.cuda.synchronize()
                g = o.double().cpu().numpy()
                tol = ATOL_FRAC[dt] * mag + RTOL[dt] * np.abs(ref)
                out[exact] = float((np.abs(g - ref) / tol).max())
            rows.append(dict(structure=structure, M=M, max_logit=round(smax, 2), folded=round(out[0], 3), exact=round(out[1], 3)))
            pkg._lib.debug_set("fwd_exact_scale", -1)
            pkg._lib.debug_set("fwd_w64", -1)
    return rows


if __name__ == "__main__":
    dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    E = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    for r in run(dt, E):
        print(r, flush=True)
