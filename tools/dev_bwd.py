import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
from util import make_inputs, oracle_fwd, oracle_bwd, to64
pkg = ge.load_package()
dev = torch.device("cuda:0")
def rel(a, b):
    return np.nanmax(np.abs(to64(a) - b)) / max(np.nanmax(np.abs(b)), 1e-30)
cases = [
    (1, 1, 1, 256, 256, 64, False, None, False),
    (2, 2, 2, 512, 512, 64, False, None, False),
    (2, 2, 2, 512, 512, 128, False, None, False),
    (2, 2, 2, 256, 256, 16, False, None, False),
    (2, 2, 2, 256, 256, 32, False, None, False),
    (2, 2, 2, 255, 300, 64, False, None, False),
    (2, 2, 2, 512, 512, 64, True, None, False),
    (2, 4, 2, 511, 511, 64, True, "ref", False),
    (2, 6, 2, 257, 257, 32, True, None, False),
    (2, 2, 2, 300, 300, 32, False, "lens", True),
    (2, 2, 2, 300, 300, 32, True, "ref", True),
    (2, 2, 2, 300, 300, 128, True, "random", False),
]
for dt in ["bf16", "f16", "f32"]:
    for (B, QH, KH, QL, KL, E, causal, pad, pair) in cases:
        d = make_inputs(0, B, QH, KH, QL, KL, E, dt, dev, pair=pair, pad=pad)
        o, ms, ls = pkg._flash_attention(d["q"], d["k"], d["v"], d["pair"], causal=causal, kpad_mask=d["mask"])
        dq, dk, dv, dp = pkg.grad_flash_attention(d["do"], o, ms, ls, d["q"], d["k"], d["v"], d["pair"], causal=causal, kpad_mask=d["mask"])
        torch.cuda.synchronize()
        rq, rk, rv, rp = oracle_bwd(d, causal)
        msg = f"{dt} B{B} QH{QH} KH{KH} QL{QL} KL{KL} E{E} c{int(causal)} pad={pad} pair={pair}: dq {rel(dq,rq):.2e} dk {rel(dk,rk):.2e} dv {rel(dv,rv):.2e}"
        if pair: msg += f" dpair {rel(dp,rp):.2e}"
        print(msg, flush=True)
for dt in ["bf16", "f32"]:
    d = make_inputs(0, 4, 4, 4, 4096, 4096, 64, dt, dev)
    o, ms, ls = pkg._flash_attention(d["q"], d["k"], d["v"], causal=False)
    f = lambda: pkg.grad_flash_attention(d["do"], o, ms, ls, d["q"], d["k"], d["v"], causal=False)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10
    print(f"C2 bwd {dt}: {t*1e3:.1f} us  {68.719476736*2.5/t:.1f} TFLOP/s (algorithmic)", flush=True)
