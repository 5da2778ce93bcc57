import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
from util import make_inputs, oracle_fwd, oracle_bwd, to64
pkg = ge.load_package(); dev = torch.device("cuda:0")
for dt in ("f32", "f16", "bf16"):
    B, H, L, E = 1, 2, 640, 64
    d = make_inputs(43, B, H, H, L, L, E, dt, dev)
    q, k = d["q"].float(), d["k"].float()
    for row, key, gain in [(5, 70, 9.0), (5, 400, 20.0), (100, 639, 30.0), (333, 200, 14.0), (600, 3, 25.0)]:
        k[0, :, key] = q[0, :, row] * gain / q[0, :, row].norm(dim=-1, keepdim=True) * (E ** 0.5) / 3
    q[0, 1, 50:60] *= 40.0
    k[0, 1, :5] = -k[0, 1, :5].abs() * 3
    d["q"], d["k"] = q.to(d["v"].dtype), k.to(d["v"].dtype)
    for causal in (False, True):
        for split in ("1", "0"):
            os.environ["NNOP_FWD_SPLIT"] = split
            o, ms, ls = pkg._flash_attention(d["q"], d["k"], d["v"], causal=causal)
            g = pkg.grad_flash_attention(d["do"], o, ms, ls, d["q"], d["k"], d["v"], causal=causal)
            torch.cuda.synchronize()
            o_ref, ms_ref, ls_ref = oracle_fwd(d, causal)
            rq, rk, rv, _ = oracle_bwd(d, causal)
            nrm = lambda a, b: np.linalg.norm(to64(a) - b) / np.linalg.norm(b)
            mx = lambda a, b: np.abs(to64(a) - b).max() / np.abs(b).max()
            print(f"{dt} causal={causal} split={split}: o {nrm(o, o_ref):.2e}/{mx(o, o_ref):.2e}  dq {nrm(g[0], rq):.2e}/{mx(g[0], rq):.2e}  dk {nrm(g[1], rk):.2e}/{mx(g[1], rk):.2e}  dv {nrm(g[2], rv):.2e}/{mx(g[2], rv):.2e}  max|s|~{float((d['q'].float() @ d['k'].float().transpose(-1,-2)).abs().max() / 8):.0f}")
